"""Deterministic, build-owned synthetic weights.

No checkpoint of the reference is available offline, so parity and benchmarks run on weights regenerated identically
on both sides (the reference modules in tools/make_golden.py, the oracle, and the HIP modules) from the parameter NAME
and SHAPE alone: numpy PCG64 seeded by crc32(name) ^ seed.  Scales are chosen per parameter role so activations stay
O(1) through 36+ layers, and BatchNorm statistics / PReLU slopes / weight-norm gains are randomised (defaults would
hide folding bugs).
"""
import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF))


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int = 0, dtype=torch.float32) -> torch.Tensor:
    r = _rng(name, seed)
    shape = tuple(int(s) for s in shape)
    leaf = name.split(".")[-1]
    n = int(np.prod(shape)) if shape else 1

    def normal(std):
        return r.standard_normal(shape).astype(np.float32) * np.float32(std)

    def uniform(lo, hi):
        return r.uniform(lo, hi, size=shape).astype(np.float32)

    if leaf == "num_batches_tracked":
        return torch.tensor(100, dtype=torch.long)
    if leaf == "running_var":
        v = uniform(0.5, 1.5)
    elif leaf == "running_mean":
        v = normal(0.1)
    elif leaf in ("pos_bias_u", "pos_bias_v"):
        v = normal(0.1)
    elif leaf == "weight_g":
        v = uniform(0.8, 1.2)  # rescaled against ||v|| in finalize_weight_norm
    elif leaf == "weight_v":
        fan_in = n // shape[0] if len(shape) > 1 else n
        v = normal(1.0 / np.sqrt(max(fan_in, 1)))
    elif leaf == "bias":
        v = normal(0.05)
    elif leaf == "weight" and len(shape) == 1:
        # norm gains / PReLU slopes: PReLU modules are named relu*/frontend3D.2 in the reference
        mod = name.split(".")[-2] if "." in name else ""
        if mod.startswith("relu") or name.endswith("frontend3D.2.weight"):
            v = uniform(0.1, 0.4)
        else:
            v = uniform(0.7, 1.3)
    elif leaf == "weight" and len(shape) == 2 and ("dict" in name.split(".")[-2:] or name.startswith("dict.")):
        v = normal(1.0)  # nn.Embedding
    elif len(shape) >= 2:
        fan_in = n // shape[0]
        if "ups." in name or name.startswith("layer.0"):
            # ConvTranspose1d weight is [Cin, Cout, k]: fan-in of an output sample is Cin*k/stride ~ Cin*k/2
            fan_in = shape[0] * shape[2] / 2.0
        v = normal(1.0 / np.sqrt(max(fan_in, 1)))
    else:
        v = normal(0.05)
    return torch.from_numpy(np.ascontiguousarray(v)).to(dtype)


def synth_state_dict(spec: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0) -> Dict[str, torch.Tensor]:
    """spec: (name, shape) pairs, e.g. [(k, v.shape) for k, v in module.state_dict().items()]."""
    sd = {}
    for name, shape in spec:
        sd[name] = synth_tensor(name, tuple(shape), seed)
    finalize_weight_norm(sd)
    return sd


def finalize_weight_norm(sd: Dict[str, torch.Tensor]) -> None:
    """Scale every weight_g so the effective weight g*v/||v|| has the norm of v times the drawn U(0.8,1.2) factor."""
    for k in list(sd.keys()):
        if not k.endswith("weight_g"):
            continue
        v = sd[k[:-1] + "v"]
        g = sd[k]
        # norm taken over every dim where g has extent 1 (torch weight_norm `dim` semantics)
        dims = [i for i in range(v.dim()) if g.shape[i] == 1]
        nrm = v.pow(2).sum(dim=dims, keepdim=True).sqrt()
        sd[k] = (g * nrm).contiguous()


def spec_of(module: torch.nn.Module):
    return [(k, tuple(v.shape)) for k, v in module.state_dict().items()]


# ---- "decisive" regime: synthetic weights / inputs whose unit logits are as peaked as a trained model's -----------------
# With the plain generator above the network's output barely depends on its input: every residual branch is as strong as
# the stream it is added to, so 36 blocks of random mixing dilute the frame-to-frame variation of the head input to 0.3-0.4 %
# of its norm (measured on the fp32 oracle) - the same size as fp16 rounding noise.  An arg-max over 200 flat logits then
# tests the noise, not the kernels.  Trained checkpoints are not like that: their units are k-means ids of well-separated
# features.  The three pieces below restore that property without training anything:
#   * `structured_frames_u8`: frames with large-scale content that changes every frame (iid pixel noise averages out in the
#     frontend's pooling);
#   * `branch_scale` < 1 on the output projection of every transformer / conformer residual branch, so the skip path carries
#     the input's variation through all blocks (frame-to-frame step of the head input: 18 % of its norm at 0.25);
#   * `nearest_centroid_head`: the unit head as the nearest-centroid classifier over k-means centroids of the head input -
#     how HuBERT units are defined - so every frame sits inside its cluster with a margin far above the rounding noise.
BRANCH_OUT_LEAVES = ("self_attn.out_proj", "fc2", "self_attn.linear_out", "feed_forward.w_2", "feed_forward_macaron.w_2",
                     "conv_module.pointwise_cov2")


def scale_residual_branches(sd: Dict[str, torch.Tensor], scale: float) -> Dict[str, torch.Tensor]:
    """Multiply weight and bias of every residual branch's output projection by `scale` (a new dict; tensors not listed in
    BRANCH_OUT_LEAVES are shared)."""
    out = dict(sd)
    for k, v in sd.items():
        mod = k.rsplit(".", 1)[0]
        if any(mod.endswith(leaf) for leaf in BRANCH_OUT_LEAVES):
            out[k] = v * scale
    return out


def structured_frames_u8(B: int, T: int, seed: int, size: int = 96, ncomp: int = 6) -> torch.Tensor:
    """uint8 [B,T,size,size] frames: per frame a random mix of `ncomp` low-frequency plane waves (independent amplitudes and
    phases per frame) around mid-grey plus +-20 levels of pixel texture."""
    import math
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(size).float() / size, torch.arange(size).float() / size, indexing="ij")
    fx = torch.randint(-3, 4, (ncomp,), generator=g).float()
    fy = torch.randint(-3, 4, (ncomp,), generator=g).float()
    amp = torch.randn(B, T, ncomp, generator=g)
    ph = torch.rand(B, T, ncomp, generator=g) * 2 * math.pi
    arg = 2 * math.pi * (fx[:, None, None] * xx + fy[:, None, None] * yy)
    out = torch.empty(B, T, size, size, dtype=torch.uint8)
    for b in range(B):      # per clip: the [T,ncomp,size,size] intermediate is 22 MB at T = 100
        f = (amp[b, :, :, None, None] * torch.cos(arg[None] + ph[b, :, :, None, None])).sum(1) / math.sqrt(ncomp / 2)
        tex = torch.randint(-20, 21, (T, size, size), generator=g).float()
        out[b] = (128 + 70 * f + tex).clamp(0, 255).to(torch.uint8)
    return out


def nearest_centroid_head(rows: torch.Tensor, n_units: int = 200, n_special: int = 4, iters: int = 30,
                          median_margin: float = 10.0, min_margin: float = 8.0, margin_passes: int = 400):
    """Unit head (weight [n_special+n_units, d], bias) = nearest-centroid classifier over k-means centroids of `rows`
    ([N, d] head-input rows from the fp32 oracle): logit_k = (2 (c_k - mu) . h - |c_k|^2 + |mu|^2) / tau, whose arg-max is the
    nearest centroid (the class-independent 2 mu . h is dropped, so the weights only carry the varying part of the
    features).  tau puts the median top-2 margin at `median_margin`; frames left closer than `min_margin` to a neighbouring
    cluster (k-means boundaries) are then pushed inside by margin-perceptron passes on the head alone.  Deterministic: Lloyd's iterations in fp64 from evenly
    spaced seeds; k = min(n_units, N // 2); unused unit rows and the special symbols get a -1e4 bias."""
    H = rows.detach().double().reshape(-1, rows.shape[-1])
    N, d = H.shape
    k = max(1, min(n_units, N // 2))
    C = H[torch.linspace(0, N - 1, k).round().long()].clone()
    for _ in range(iters):
        d2 = (H * H).sum(-1, keepdim=True) - 2 * H @ C.T + (C * C).sum(-1)[None]
        a = d2.argmin(-1)
        newC = C.clone()
        for j in range(k):
            sel = a == j
            if bool(sel.any()):
                newC[j] = H[sel].mean(0)
        if torch.equal(newC, C):
            break
        C = newC
    mu = H.mean(0)
    V = H - mu
    W = 2 * (C - mu)
    b = -((C * C).sum(-1) - (mu * mu).sum())
    lg = V @ W.T + (b + 2 * (C - mu) @ mu)          # == H @ W.T + b
    b = b + W @ mu                                   # work on centred rows from here on; folded back at the end
    if k > 1:
        top2 = lg.topk(2, -1).values
        tau = float((top2[:, 0] - top2[:, 1]).median()) / median_margin
        tau = tau if tau > 0 else 1.0
        W, b = W / tau, b / tau
        # margin passes: a frame whose top-2 gap is below `min_margin` pulls its own class towards it and pushes the runner-up
        # away, by exactly the missing gap (a perceptron with margin on frozen features - fine-tuning the head alone)
        label = (V @ W.T + b).argmax(-1)
        vn = (V * V).sum(-1).clamp_min(1e-12)
        for _ in range(margin_passes):
            lg = V @ W.T + b
            own = lg.gather(1, label[:, None])[:, 0]
            lg.scatter_(1, label[:, None], float("-inf"))
            todo = (own - lg.max(-1).values < min_margin).nonzero()[:, 0]
            if todo.numel() == 0:
                break
            for i in todo.tolist():                  # Gauss-Seidel: every frame sees the updates made before it
                row = W @ V[i] + b
                a = int(label[i])
                own_i = float(row[a])
                row[a] = float("-inf")
                rv, r = row.max(0)
                short = min_margin - (own_i - float(rv))
                if short > 0:
                    step = (0.5 * short / float(vn[i])) * V[i]
                    W[a] += step
                    W[int(r)] -= step
    b = b - W @ mu
    weight = torch.zeros(n_special + n_units, d, dtype=torch.float64)
    bias = torch.full((n_special + n_units,), -1e4, dtype=torch.float64)
    weight[n_special:n_special + k] = W
    bias[n_special:n_special + k] = b
    return weight.float(), bias.float()
