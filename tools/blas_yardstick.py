#!/usr/bin/env python3
"""Yardstick only (never on the product path): torch.mm (hipBLASLt/rocBLAS) on the GEMM shapes of the path,
to see how far the hand-written tap-GEMM is from the vendor library on the same box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.gemm_bench import SHAPES

def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        b = torch.randn(N, device="cuda").half()
        for _ in range(3):
            torch.addmm(b, a, w.t())
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            torch.addmm(b, a, w.t())
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(reps):
                c = torch.addmm(b, a, w.t())
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"{name:10s} M={M:5d} N={N:5d} K={K:5d}  {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s (torch.addmm)", flush=True)

if __name__ == "__main__":
    main()
