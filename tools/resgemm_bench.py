#!/usr/bin/env python3
"""The fp32-residual-stream GEMMs of the encoder / conformer (x += W h + b, fp32 in place) in isolation, hipGraph replay.
L2S_PHASEGEMM=0 forces the 256x128 two-blocks-per-CU kernel, =2 the phase-staggered 256x256 kernel (A/B).
usage: python tools/resgemm_bench.py [M multiplier]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_unit_amd import ops
from lip2speech_unit_amd.ops import F_RES_POST

MUL = int(sys.argv[1]) if len(sys.argv) > 1 else 1      # 1 = 160 clips, 4 = the bench default's 640
SHAPES = [("enc out", 16000 * MUL, 1024, 1024), ("enc fc2", 16000 * MUL, 1024, 4096), ("conf ffn2", 32000 * MUL, 512, 2048),
          ("conf out", 32000 * MUL, 512, 512)]
reps = 20 if MUL == 1 else 8
for name, M, N, K in SHAPES:
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda")
    x = torch.randn(M, N, device="cuda")

    def run():
        ops.tapgemm(a, w, x, M=M, N=N, Cin=K, bias=b, R=x, ldr=N, flags=F_RES_POST, dtype=ops.F16)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    mb = (M * K * 2 + N * K * 2 + 2 * M * N * 4) / 1e6
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  {mb / us:5.2f} TB/s algorithmic", flush=True)
