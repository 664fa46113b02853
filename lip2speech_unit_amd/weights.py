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
