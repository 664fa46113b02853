#!/usr/bin/env python3
"""Checks the phase-staggered 256x256 kernel (csrc/phasegemm.hip, forced with L2S_PHASEGEMM=2) against torch fp32:
every epilogue family it is built for, ragged M / N tails, one to many tiles per block, repeated runs (the kernel's
LDS hand-over is ordered by counted vmcnt + barriers: a race would show as rare wrong tiles).
Used by tests/test_tiles_gpu.py."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from lip2speech_unit_amd import _lib, ops


def rel_err(got, ref):
    return (got.float().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)


def main():
    worst = 0.0
    lib = _lib.load()
    shapes = [(256, 256, 64), (1000, 384, 128), (2304, 512, 192), (16000, 1024, 512), (8200, 768, 1024)]
    for dt, tol in ((ops.F16, 2e-3), (ops.BF16, 1.5e-2)):
        t16 = ops.torch_dtype(dt)
        for si, (M, N, K) in enumerate(shapes):
            g = torch.Generator().manual_seed(100 + si)
            a = torch.randn(M, K, generator=g).to(t16)
            w = (torch.randn(N, K, generator=g) / K ** 0.5).to(t16)
            b = torch.randn(N, generator=g)
            sl = torch.rand(N, generator=g) * 0.3
            r32 = torch.randn(M, N, generator=g)
            r16 = torch.randn(M, N, generator=g).to(t16)
            T = 125 if M % 125 == 0 else M                      # rows per "clip" for the length mask
            lens = torch.randint(1, T + 1, (M // T,), generator=g).int()
            keep = (torch.arange(T)[None, :] < lens[:, None]).reshape(M, 1).float()
            lin = a.float() @ w.float().t() + b
            ag, wg, bg = a.cuda(), w.cuda(), b.cuda()
            d = _lib.GemmDesc(M=M, N=N, Cin=K, ntaps=1, mode=ops.MODE_LINEAR, lda=K, ldc=N, groups=1, flags=0, act=0)
            assert lib.l2s_tapgemm_variant(ctypes.byref(d)) == 256256, "phase kernel not selected (L2S_PHASEGEMM=2?)"
            cases = [
                ("none", dict(), lin, None),
                ("relu", dict(act=ops.ACT_RELU), F.relu(lin), None),
                ("lrelu+alpha", dict(act=ops.ACT_LRELU, act_slope=0.1, alpha=0.5), F.leaky_relu(lin * 0.5, 0.1), None),
                ("prelu", dict(act=ops.ACT_PRELU, slope=sl.cuda()), torch.where(lin >= 0, lin, lin * sl), None),
                ("gelu", dict(act=ops.ACT_GELU), F.gelu(lin), None),
                ("gelu+mask", dict(act=ops.ACT_GELU, lens=lens.cuda(), mask_T=T, mask_mul=1, flags=ops.F_MASK), F.gelu(lin) * keep, None),
                ("none+mask", dict(lens=lens.cuda(), mask_T=T, mask_mul=1, flags=ops.F_MASK), lin * keep, None),
                ("res32 post", dict(R=r32.cuda(), flags=ops.F_RES_POST, alpha=0.5), lin * 0.5 + r32, torch.float32),
                ("res32 pre relu", dict(R=r32.cuda(), flags=ops.F_RES_PRE, act=ops.ACT_RELU), F.relu(lin + r32), torch.float32),
                ("res16 post f32out", dict(R=r16.cuda(), flags=ops.F_RES_POST), lin + r16.float(), torch.float32),
            ]
            for name, kw, ref, odt in cases:
                for rep in range(3 if M >= 8000 else 1):
                    C = torch.full((M, N), float("nan"), device="cuda", dtype=odt or t16)
                    ops.tapgemm(ag, wg, C, M=M, N=N, Cin=K, bias=bg, dtype=dt, **kw)
                    torch.cuda.synchronize()
                    e = rel_err(C, ref)
                    if not (e <= tol):
                        print(f"FAIL {name} dt={dt} shape={M}x{N}x{K} rep={rep}: rel err {e:.3e} (tol {tol})")
                    worst = max(worst, (e if e == e else 1e9) / tol)
    print(f"phasegemm worst err/tol = {worst:.3f}")
    sys.exit(0 if worst <= 1.0 else 1)


if __name__ == "__main__":
    main()
