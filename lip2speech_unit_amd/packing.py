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
