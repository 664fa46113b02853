#!/usr/bin/env python3
"""Isolated tap-GEMM timings (HIP events around N back-to-back launches) for the shapes the path uses."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_unit_amd import ops

SHAPES = [  # (name, M, N, K)
    ("4096^3", 4096, 4096, 4096), ("8192^3", 8192, 8192, 8192),
    ("enc qkv", 3200, 3072, 1024), ("enc out", 3200, 1024, 1024), ("enc fc1", 3200, 4096, 1024), ("enc fc2", 3200, 1024, 4096),
    ("conf ffn1", 6400, 2048, 512), ("conf ffn2", 6400, 512, 2048), ("conf qkv", 6400, 1536, 512), ("conf out", 6400, 512, 512),
    ("conf pw1", 6400, 1024, 512),
    ("b80 qkv", 8000, 3072, 1024), ("b80 out", 8000, 1024, 1024), ("b80 fc1", 8000, 4096, 1024), ("b80 fc2", 8000, 1024, 4096),
    ("b160 qkv", 16000, 3072, 1024), ("b160 out", 16000, 1024, 1024), ("b160 fc1", 16000, 4096, 1024), ("b160 fc2", 16000, 1024, 4096),
    ("b160 cffn1", 32000, 2048, 512), ("b160 cqkv", 32000, 1536, 512), ("b160 cpw1", 32000, 1024, 512),
    ("b80 cffn1", 16000, 2048, 512), ("b80 cffn2", 16000, 512, 2048), ("b80 cqkv", 16000, 1536, 512), ("b80 cpw1", 16000, 1024, 512), ("tiny", 256, 128, 64), ("epi k64", 4096, 4096, 64), ("epi k128", 4096, 4096, 128), ("epi k512", 4096, 4096, 512),
]

def main():
    dt = ops.F16
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    only = sys.argv[2] if len(sys.argv) > 2 else None
    for name, M, N, K in SHAPES:
        if only and only not in name:
            continue
        a = (torch.randn(M, K, device="cuda") ).half()
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        c = torch.empty(M, N, device="cuda", dtype=torch.float16)
        b = None if os.environ.get("NOBIAS") else torch.randn(N, device="cuda")
        for _ in range(3):
            ops.tapgemm(a, w, c, M=M, N=N, Cin=K, bias=b, dtype=dt)
        torch.cuda.synchronize()
        # replay from a hipGraph: python/ctypes launch cost (~20 us) would otherwise hide kernels shorter than that
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            ops.tapgemm(a, w, c, M=M, N=N, Cin=K, bias=b, dtype=dt)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                ops.tapgemm(a, w, c, M=M, N=N, Cin=K, bias=b, dtype=dt)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"{name:10s} M={M:5d} N={N:5d} K={K:5d}  {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    main()
