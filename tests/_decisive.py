"""Test helper: the "decisive" synthetic regime of lip2speech_unit_amd/weights.py (structured frames, weak residual branches,
nearest-centroid unit head) wired to the CPU oracle.  The head is fitted on the ORACLE's fp32 head input for the clips the
test checks, then the oracle is run again with the fitted head: both sides (oracle, HIP path) load the same state dict."""
import math
from typing import Dict

import torch

BRANCH_SCALE = 0.25       # default attenuation of the residual branches (test_full_depth_decisive_full_strength runs 1.0 as well)
# absolute bounds on the logit error of the decisive runs (logits there have |.| ~ 10-40: peaked like a trained model's); measured
# at full depth 6.5e-2 ... 8.9e-2 fp16 / 5.6e-1 ... 6.5e-1 bf16, 2-layer models 1e-2 / 1e-1
MAX_LOGIT_ERR = {"fp16": 0.25, "bf16": 1.0}



def frames_from_u8(u8, crop=88):
    """hubert_dataset.py:242-245 on uint8 [B,T,96,96] -> fp32 [B,1,T,88,88]."""
    d = (u8.shape[-1] - crop) // 2
    return ((u8[:, :, d:d + crop, d:d + crop].float() / 255.0 - 0.421) / 0.165).unsqueeze(1).contiguous()


def fit_decisive_head(sd, run_oracle, clip_ids, head="conformer.proj_out", fit_ids=None):
    """run_oracle(sd, clip, taps) -> the oracle's result dict for that clip run ALONE (taps receives "head_in").
    The head is fitted on the oracle's head input of `fit_ids` (default: the checked clips themselves - then every checked frame
    has a margin >= 8 by construction and the test is an arg-max PLUMBING check; with other clips in `fit_ids` the checked clips
    are held out and their margins are whatever the classifier gives them).
    Returns (state dict with the fitted unit head, {clip: oracle result under that state dict})."""
    rows = []
    with torch.no_grad():
        for b in (clip_ids if fit_ids is None else fit_ids):
            taps = {}
            run_oracle(sd, b, taps)
            rows.append(taps["head_in"].reshape(-1, taps["head_in"].shape[-1]))
        w, bias = nearest_centroid_head(torch.cat(rows), n_units=sd[head + ".weight"].shape[0] - 4)
        sd = dict(sd)
        sd[head + ".weight"], sd[head + ".bias"] = w, bias
        refs = {b: run_oracle(sd, b, None) for b in clip_ids}
    return sd, refs


def margins(logits_2d):
    """Top-2 gap over the unit ids (columns 4..) of oracle logits [L, V]."""
    top2 = logits_2d[:, 4:].topk(2, -1).values
    return top2[:, 0] - top2[:, 1]


# ---- the regime's ingredients (test infrastructure: moved here from lip2speech_unit_amd/weights.py in round 4) ----


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
