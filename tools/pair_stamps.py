#!/usr/bin/env python3
"""Where a block of the phase-staggered conv-pair kernel (csrc/respair_phase.hip) spends its time (diagnostic build):
  tools/build_variant.sh pstamps -DL2S_PAIR_STAMPS respair_phase.hip
  L2S_LIB_PATH=build_ab/pstamps/liblip2speech_hip.so python tools/pair_stamps.py [clips=640] [C=256|128]
Per (k, dil, kind): tile-start wait / conv1 / patch -> t1 hand-over / conv2 / epilogue per tile for wave 0 (lower wave row) and
wave 7 (upper row), averaged over blocks; s_memtime ticks are converted with the launch's HIP-event time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_unit_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 640
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = {256: 2000, 128: 8000, 64: 16000}[C]
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.l2s_debug_pair_stamps.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(256 * 2 * 16, dtype=torch.int64, device="cuda")
assert raw.l2s_debug_pair_stamps(stamps.data_ptr()) == 0
M = B * T
xl = torch.randn(M, C, device="cuda").half()
y = torch.empty(M, C, device="cuda", dtype=torch.float16)
xs = torch.zeros(M, C, device="cuda")
lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
for k in (3, 7, 11):
    w1 = (torch.randn(C, k * C, device="cuda") / (k * C) ** 0.5).half()
    w2 = (torch.randn(C, k * C, device="cuda") / (k * C) ** 0.5).half()
    b1, b2 = torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
    for dil, kind in ((1, "mid"), (5, "last")):
        if kind == "mid":
            run = lambda: ops.respair(xl, w1, b1, w2, b2, B=B, T=T, C=C, k=k, dil=dil, slope=0.1, y=y, lens=lens, len_mul=1)
        else:
            run = lambda: ops.respair(xl, w1, b1, w2, b2, B=B, T=T, C=C, k=k, dil=dil, slope=0.1, xs=xs, accumulate=True, lens=lens, len_mul=1)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        stamps.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        st = stamps.cpu().view(-1, 2, 16).double()
        st = st[st[:, 0, 8] > 0]
        tick_us = us / st[:, :, 9].max().item()
        nph = 2 * k * (C // 64) * 2
        print(f"k{k:2d} d{dil} {kind:4s}: launch {us:7.1f} us, tiles/block {st[:,0,8].min():.0f}-{st[:,0,8].max():.0f}, tick {tick_us*1e3:.3f} ns ({1/tick_us/1e3:.2f} GHz)")
        for wv, label in ((0, "wave 0"), (1, "wave 7")):
            n = st[:, wv, 8]
            seg = [(st[:, wv, i] / n * tick_us).mean().item() for i in range(8)]
            cyc = [(st[:, wv, i] / n).mean().item() for i in range(8)]
            print(f"   {label}: start wait {seg[0]:5.2f}  conv1 {seg[1]:6.2f}  level {seg[7]:4.2f} + hand-over {seg[2]:5.2f}  conv2 {seg[3]:6.2f}  "
                  f"level {seg[5]:4.2f} + patch issue {seg[6]:4.2f} + epilogue {seg[4]:5.2f} us per tile"
                  f"   | cycles per phase: conv1 {cyc[1] / (nph / 2):6.0f} conv2 {cyc[3] / (nph / 2):6.0f} (512 = MFMA-bound)")
