"""Host-side weight packing for the tap-GEMM layouts (pure tensor re-layout, done once at load time)."""
from typing import Dict, List

import torch


def weight_norm_weight(sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """Effective weight of a (possibly weight-normed) module: `prefix.weight`, or g*v/||v|| from weight_g/weight_v.

    The norm runs over every dim where g has extent 1 (torch.nn.utils.weight_norm `dim` semantics): dim=0 for the
    vocoder convs (speech-resynthesis/models.py:19-31,78-96), dim=2 for fairseq's pos_conv.
    """
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"].float()
    g, v = sd[prefix + ".weight_g"].float(), sd[prefix + ".weight_v"].float()
    dims = [i for i in range(v.dim()) if g.shape[i] == 1]
    return v * (g / v.pow(2).sum(dim=dims, keepdim=True).sqrt())


def pack_conv1d(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, k] -> [Cout, k*Cin] with K index = tap*Cin + c."""
    return w.permute(0, 2, 1).reshape(w.shape[0], -1).contiguous()


def convtranspose_phases(w: torch.Tensor, stride: int, padding: int) -> List[dict]:
    """Split ConvTranspose1d(weight [Cin, Cout, k], stride s, padding p) into s stride-1 correlations.

    Output sample t = s*q + r only receives taps k = k0 + s*j with k0 = (r+p) % s, from input q + (r+p-k0)/s - j, so
    phase r is a CONV1D tap-GEMM with dil=-1, off=(r+p-k0)/s whose rows land on output rows q*s + r.
    """
    cin, cout, k = w.shape
    out = []
    for r in range(stride):
        k0 = (r + padding) % stride
        taps = list(range(k0, k, stride))
        wp = torch.stack([w[:, :, kk] for kk in taps], dim=0)  # [J, Cin, Cout]
        wp = wp.permute(2, 0, 1).reshape(cout, len(taps) * cin).contiguous()
        out.append({"r": r, "off": (r + padding - k0) // stride, "ntaps": len(taps), "w": wp})
    return out


def convtranspose_fused(w: torch.Tensor, stride: int, padding: int, fold: int = 1) -> dict:
    """The same ConvTranspose1d as ONE stride-1 correlation with N = stride*Cout output channels: output sample s*q + r,
    channel c is column r*Cout + c of GEMM row q, i.e. the [M_in, s*Cout] output matrix IS the interleaved [M_in*s, Cout]
    activation.  The phases share the union of their input shifts (tap i reads input q + off - i, dil = -1); where a phase
    does not use a shift its weights are zero.  One launch reads the input once instead of once per phase - the
    narrow late stages of the vocoder are HBM-bound, so the 1.4-1.5x MFMA work of the zero taps is free there.
    fold = f > 1 additionally views f consecutive time steps as one row ([M_in/f, f*Cin] in, [M_in/f, f*s*Cout] out: free
    views of the same memory), which widens a 32-channel layer to the 64-channel kernels' shape."""
    cin, cout, k = w.shape
    ph = convtranspose_phases(w, stride, padding)
    d_max = max(p["off"] for p in ph)
    d_min = min(p["off"] - (p["ntaps"] - 1) for p in ph)
    S = d_max - d_min + 1
    wf = torch.zeros(stride * cout, S, cin, dtype=w.dtype)
    for p in ph:
        wp = p["w"].view(cout, p["ntaps"], cin)
        for j in range(p["ntaps"]):
            i = d_max - (p["off"] - j)
            wf[p["r"] * cout:(p["r"] + 1) * cout, i] = wp[:, j]
    if fold == 1:
        return {"w": wf.reshape(stride * cout, S * cin).contiguous(), "ntaps": S, "off": d_max, "n": stride * cout, "fold": 1}
    f, N = fold, stride * cout
    D_max, D_min = (f - 1 + d_max) // f, (d_max - (S - 1)) // f            # floor division: folded row shifts
    S2 = D_max - D_min + 1
    w2 = torch.zeros(f * N, S2, f * cin, dtype=w.dtype)
    for b in range(f):                   # output sub-step inside the folded row
        for i in range(S):               # un-folded tap: input step f*q' + b + d_max - i
            m = b + d_max - i
            D, a = m // f, m % f         # folded row q' + D, sub-step a
            w2[b * N:(b + 1) * N, D_max - D, a * cin:(a + 1) * cin] = wf[:, i]
    return {"w": w2.reshape(f * N, S2 * f * cin).contiguous(), "ntaps": S2, "off": D_max, "n": f * N, "fold": f}
