"""Micro-benchmark of l2s_resblock_fused on the vocoder's two narrow stages, launched as the pipeline launches them
(k = 3 writes xs, k = 7 accumulates, k = 11 accumulates and writes the next stage's leaky_relu copy).
usage: python tools/resblock_bench.py [B]      (L2S_LIB_PATH selects an A/B build of the library)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_unit_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 160
tot = 0.0
for C, T in ((32, 32000), (16, 64000)):
    xl = torch.randn(B * T, C, device="cuda").half()
    xs = torch.zeros(B * T, C, device="cuda")
    nxt = torch.empty(B * T, C, device="cuda", dtype=torch.float16)
    lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
    for k in (3, 7, 11):
        w = (torch.randn(6, C, ((k * C + 31) // 32) * 32, device="cuda") / (k * C) ** 0.5).half()
        b = torch.randn(6, C, device="cuda")

        def run():
            ops.resblock_fused(xl, w, b, xs, nxt if k == 11 else None, B=B, T=T, C=C, k=k, dil=(1, 3, 5),
                               accumulate=k != 3, slope=0.1, lens=lens, len_mul=1, dtype=ops.F16)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        tot += us
        fl = 6 * 2.0 * B * T * C * C * k
        print(f"C{C} k{k:2d}: {us:8.1f} us   {fl / us * 1e-6:6.0f} TFLOP/s", flush=True)
print(f"total {tot:8.1f} us")

# the same two stages as ONE launch each (l2s_resstage_fused)
tot2 = 0.0
for C, T in ((32, 32000), (16, 64000)):
    xl = torch.randn(B * T, C, device="cuda").half()
    xs = torch.zeros(B * T, C, device="cuda")
    nxt = torch.empty(B * T, C, device="cuda", dtype=torch.float16)
    lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
    ws = [(torch.randn(6, C, ((k * C + 31) // 32) * 32, device="cuda") / (k * C) ** 0.5).half() for k in (3, 7, 11)]
    bs = [torch.randn(6, C, device="cuda") for _ in range(3)]

    def run():
        ops.resstage_fused(xl, ws, bs, xs, nxt, B=B, T=T, C=C, ks=(3, 7, 11), dils=((1, 3, 5),) * 3, slope=0.1, lens=lens,
                           len_mul=1, dtype=ops.F16)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    tot2 += us
    fl = sum(6 * 2.0 * B * T * C * C * k for k in (3, 7, 11))
    print(f"stage C{C}: {us:8.1f} us   {fl / us * 1e-6:6.0f} TFLOP/s", flush=True)
print(f"total {tot2:8.1f} us")
