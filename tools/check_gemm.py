#!/usr/bin/env python3
"""Checks one forced tap-GEMM tile shape (env L2S_FORCE_TILE, read once per process) against torch fp32 on CPU.
Used by tests/test_tiles_gpu.py: every kernel instantiation gets exercised regardless of the tile heuristic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from lip2speech_unit_amd import ops


def main():
    worst = 0.0
    for dt, tol in ((ops.F16, 2e-3), (ops.BF16, 1.5e-2)):
        t16 = ops.torch_dtype(dt)
        g = torch.Generator().manual_seed(3)
        # linear, ragged M / N / K tails, fp32 out + residual, and 16-bit out + dual
        M, N, K = 777, 208, 200
        a = torch.randn(M, K, generator=g).to(t16).float()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(t16).float()
        b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g)
        ref = r + F.gelu(a @ w.t() + b)
        C = torch.empty(M, N, device="cuda")
        ops.tapgemm(a.to(t16).cuda(), w.to(t16).cuda(), C, M=M, N=N, Cin=K, bias=b.cuda(), act=ops.ACT_GELU, R=r.cuda(),
                    flags=ops.F_RES_POST, dtype=dt)
        C16 = torch.empty(M, N, device="cuda", dtype=t16)
        C2 = torch.empty(M, N, device="cuda", dtype=t16)
        ops.tapgemm(a.to(t16).cuda(), w.to(t16).cuda(), C16, M=M, N=N, Cin=K, bias=b.cuda(), C2=C2, flags=ops.F_DUAL,
                    slope2=0.1, dtype=dt)
        torch.cuda.synchronize()
        ref16 = a @ w.t() + b
        for got, rf in ((C, ref), (C16, ref16), (C2, F.leaky_relu(ref16, 0.1))):
            e = (got.float().cpu() - rf).abs().max().item() / (rf.abs().max().item() + 1e-6)
            worst = max(worst, e / tol)
        # dilated conv1d with per-lane taps (Cin = 24) and a uniform-tap one (Cin = 64)
        for Cin in (24, 64):
            B, T, Co, k, dil = 2, 301, 48, 5, 3
            x = torch.randn(B, Cin, T, generator=g).to(t16).float()
            wc = (torch.randn(Co, Cin, k, generator=g) / (Cin * k) ** 0.5).to(t16).float()
            pad = (k * dil - dil) // 2
            rf = F.conv1d(x, wc, None, 1, pad, dil).transpose(1, 2).reshape(B * T, Co)
            A = x.transpose(1, 2).contiguous().reshape(B * T, Cin).to(t16).cuda()
            Wp = wc.permute(0, 2, 1).reshape(Co, k * Cin).contiguous().to(t16).cuda()
            out = torch.empty(B * T, Co, device="cuda", dtype=t16)
            ops.tapgemm(A, Wp, out, M=B * T, N=Co, Cin=Cin, ntaps=k, mode=ops.MODE_CONV1D, T_out=T, T_in=T, stride=1,
                        dil=dil, off=-pad, dtype=dt)
            torch.cuda.synchronize()
            e = (out.float().cpu() - rf).abs().max().item() / (rf.abs().max().item() + 1e-6)
            worst = max(worst, e / tol)
    print(f"tile {os.environ.get('L2S_FORCE_TILE', 'auto')} worst err/tol = {worst:.3f}")
    sys.exit(0 if worst <= 1.0 else 1)


if __name__ == "__main__":
    main()
