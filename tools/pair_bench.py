#!/usr/bin/env python3
"""Micro-benchmark of the fused conv pairs (csrc/respair.hip C = 64 / 128, csrc/respair256.hip C = 256) at the vocoder's
shapes, launched as the pipeline launches them: per (C, k) the two mid pairs (dil 1, 3) and the last pair (dil 5, accumulate
into the fp32 ResBlock sum).  With `unfused` the C = 256 stage is also timed as its tap-GEMM launches (the path until round 4).
usage: python tools/pair_bench.py [B] [C ...] [unfused]      (L2S_LIB_PATH selects an A/B build of the library)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_unit_amd import ops
from lip2speech_unit_amd.ops import ACT_LRELU, F_ACCUM, F_DUAL, F_MASK, F_RES_POST, MODE_CONV1D

args = [a for a in sys.argv[1:] if a != "unfused"]
unfused = "unfused" in sys.argv[1:]
B = int(args[0]) if args else 160
Cs = [int(c) for c in args[1:]] or [256, 128, 64]
TS = {256: 2000, 128: 8000, 64: 16000}


def timeit(run, n=10):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for C in Cs:
    T = TS[C]
    M = B * T
    xl = torch.randn(M, C, device="cuda").half()
    y = torch.empty(M, C, device="cuda", dtype=torch.float16)
    xs = torch.zeros(M, C, device="cuda")
    lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
    tot = 0.0
    tot_fl = 0.0
    for k in (3, 7, 11):
        w1 = (torch.randn(C, k * C, device="cuda") / (k * C) ** 0.5).half()
        w2 = (torch.randn(C, k * C, device="cuda") / (k * C) ** 0.5).half()
        b1, b2 = torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
        fl = 2 * 2.0 * M * C * C * k
        row = []
        for dil, kind in ((1, "mid"), (3, "mid"), (5, "last")):
            if kind == "mid":
                run = lambda: ops.respair(xl, w1, b1, w2, b2, B=B, T=T, C=C, k=k, dil=dil, slope=0.1, y=y, lens=lens, len_mul=1)
            else:
                run = lambda: ops.respair(xl, w1, b1, w2, b2, B=B, T=T, C=C, k=k, dil=dil, slope=0.1, xs=xs, accumulate=True,
                                          lens=lens, len_mul=1)
            us = timeit(run)
            tot += us
            tot_fl += fl
            row.append(f"{kind} d{dil} {us:8.1f} us {fl / us * 1e-6:6.0f} TF")
        print(f"C{C} k{k:2d}: " + "   ".join(row), flush=True)
    print(f"C{C} stage (9 pairs): {tot / 1e3:8.3f} ms per {B} clips   {tot_fl / tot * 1e-6:6.0f} TFLOP/s", flush=True)
    if unfused and C == 256:
        t1 = torch.empty(M, C, device="cuda", dtype=torch.float16)
        o = torch.empty(M, C, device="cuda", dtype=torch.float16)
        ol = torch.empty(M, C, device="cuda", dtype=torch.float16)
        tot_u = 0.0
        for k in (3, 7, 11):
            w1 = (torch.randn(C, k * C, device="cuda") / (k * C) ** 0.5).half()
            b1 = torch.randn(C, device="cuda") * 0.1
            for dil, kind in ((1, "mid"), (3, "mid"), (5, "last")):
                def run():
                    ops.tapgemm(xl, w1, t1, M=M, N=C, Cin=C, ntaps=k, mode=MODE_CONV1D, T_out=T, T_in=T, stride=1, dil=dil,
                                off=-(k - 1) // 2 * dil, bias=b1, act=ACT_LRELU, act_slope=0.1, lens=lens, mask_T=T, mask_mul=1,
                                flags=F_MASK)
                    if kind == "mid":
                        ops.tapgemm(t1, w1, o, M=M, N=C, Cin=C, ntaps=k, mode=MODE_CONV1D, T_out=T, T_in=T, stride=1, dil=1,
                                    off=-(k - 1) // 2, bias=b1, R=xl, ldr=C, C2=ol, ldc2=C, lens=lens, mask_T=T, mask_mul=1,
                                    flags=F_RES_POST | F_DUAL | F_MASK, slope2=0.1)
                    else:
                        ops.tapgemm(t1, w1, xs, M=M, N=C, Cin=C, ntaps=k, mode=MODE_CONV1D, T_out=T, T_in=T, stride=1, dil=1,
                                    off=-(k - 1) // 2, bias=b1, R=xl, ldr=C, lens=lens, mask_T=T, mask_mul=1,
                                    flags=F_RES_POST | F_MASK | F_ACCUM)
                us = timeit(run)
                tot_u += us
                print(f"  unfused C256 k{k:2d} {kind} d{dil}: {us:8.1f} us {2 * 2.0 * M * C * C * k / us * 1e-6:6.0f} TF", flush=True)
        print(f"C256 stage unfused (18 launches): {tot_u / 1e3:8.3f} ms per {B} clips", flush=True)
