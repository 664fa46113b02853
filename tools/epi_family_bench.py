#!/usr/bin/env python3
"""One C = 256 vocoder ResBlock convolution (Conv1d 256 -> 256, k taps, 640 clips x 1000 samples) under the epilogue families it
could use: lean + LeakyReLU + mask (the first convs), 16-bit residual (G16A), residual + dual + mask (G16B: the second convs
today).  Same FLOPs; the spread is what the epilogue costs.  usage: python tools/epi_family_bench.py [k] [clips]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_unit_amd import ops
from lip2speech_unit_amd.ops import F_RES_POST, F_RES_PRE, F_DUAL, F_MASK, MODE_CONV1D, ACT_LRELU

k = int(sys.argv[1]) if len(sys.argv) > 1 else 7
B = int(sys.argv[2]) if len(sys.argv) > 2 else 640
T, C = 1000, 256
M = B * T
dt = ops.F16
a = torch.randn(M, C, device="cuda").half()
w = (torch.randn(C, k * C, device="cuda") / (k * C) ** 0.5).half()
b = torch.randn(C, device="cuda")
r = torch.randn(M, C, device="cuda").half()
y = torch.empty(M, C, device="cuda", dtype=torch.float16)
y2 = torch.empty(M, C, device="cuda", dtype=torch.float16)
lens = torch.full((B,), 200, dtype=torch.int32, device="cuda")
base = dict(M=M, N=C, Cin=C, ntaps=k, mode=MODE_CONV1D, T_out=T, T_in=T, stride=1, dil=1, off=-(k - 1) // 2, bias=b, dtype=dt)
cases = {
    "lean + lrelu + mask (e3)": dict(act=ACT_LRELU, act_slope=0.1, lens=lens, mask_T=T, mask_mul=5, flags=F_MASK),
    "residual (G16A, e6)": dict(R=r, ldr=C, flags=F_RES_POST),
    "residual + lrelu (G16A, e6)": dict(R=r, ldr=C, flags=F_RES_PRE, act=ACT_LRELU, act_slope=0.1),
    "residual + dual + mask (G16B, e7)": dict(R=r, ldr=C, C2=y2, ldc2=C, lens=lens, mask_T=T, mask_mul=5, slope2=0.1,
                                              flags=F_RES_POST | F_DUAL | F_MASK),
}
for name, kw in cases.items():
    def run():
        ops.tapgemm(a, w, y, **base, **kw)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(8):
            run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 8
    print(f"k={k} {name:36s} {us:8.1f} us  {2.0 * M * C * C * k / us / 1e6:7.1f} TFLOP/s", flush=True)
