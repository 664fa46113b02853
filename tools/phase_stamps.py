#!/usr/bin/env python3
"""Where a block of the phase-staggered GEMM spends its time (diagnostic build, -DL2S_PHASE_STAMPS):
  tools/build_variant.sh stamps -DL2S_PHASE_STAMPS phasegemm_inst:e0_m0
  L2S_LIB_PATH=build_ab/stamps/liblip2speech_hip.so L2S_PHASEGEMM=2 python tools/phase_stamps.py [clips=640]
Per shape: K loop / epilogue / first-K-tile-after-epilogue (includes the wait for the block's slowest wave) per tile, for wave 0
(lower wave row) and wave 7 (upper row, one barrier behind), averaged over blocks; s_memtime ticks are converted with the
launch's HIP-event time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_unit_amd import _lib, ops
from lip2speech_unit_amd.ops import ACT_GELU, F_RES_POST

CLIPS = int(sys.argv[1]) if len(sys.argv) > 1 else 640
M1, M2 = CLIPS * 100, CLIPS * 200
SHAPES = [("enc qkv", M1, 3072, 1024, "b"), ("enc fc1", M1, 4096, 1024, "g"), ("enc fc2", M1, 1024, 4096, "r"),
          ("enc out", M1, 1024, 1024, "r"), ("conf ffn1", M2, 2048, 512, "b"), ("conf ffn2", M2, 512, 2048, "r"),
          ("conf out", M2, 512, 512, "r")]
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.l2s_debug_phase_stamps.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(256 * 2 * 8, dtype=torch.int64, device="cuda")
assert raw.l2s_debug_phase_stamps(stamps.data_ptr()) == 0
for name, M, N, K, kind in SHAPES:
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda")
    if kind == "r":
        x = torch.randn(M, N, device="cuda")
        run = lambda: ops.tapgemm(a, w, x, M=M, N=N, Cin=K, bias=b, R=x, ldr=N, flags=F_RES_POST, dtype=ops.F16)
    else:
        c = torch.empty(M, N, device="cuda", dtype=torch.float16)
        run = lambda: ops.tapgemm(a, w, c, M=M, N=N, Cin=K, bias=b, act=ACT_GELU if kind == "g" else 0, dtype=ops.F16)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    stamps.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    st = stamps.cpu().view(-1, 2, 8).double()
    live = st[:, 0, 3] > 0
    st = st[live]
    tick_us = us / st[:, :, 4].max().item()          # the longest block ~ the launch (minus launch overhead)
    print(f"{name:10s} M={M} N={N} K={K}: launch {us:7.1f} us, {int(live.sum())} blocks, tiles/block {st[:,0,3].min():.0f}-{st[:,0,3].max():.0f}, "
          f"K-tiles {int(st[0,0,5])}, tick {tick_us*1e3:.2f} ns")
    for wv, label in ((0, "wave 0"), (1, "wave 7")):
        n = st[:, wv, 3]
        kl, ep, fw = (st[:, wv, i] / n * tick_us for i in range(3))
        tot = st[:, wv, 4] * tick_us
        print(f"   {label}: per tile  K loop {kl.mean():6.2f} us (+ first K-tile after an epilogue {fw.mean():5.2f})  epilogue {ep.mean():6.2f} us "
              f"[min {ep.min():.2f} max {ep.max():.2f}]  block total {tot.mean():7.1f} us (max {tot.max():.1f})")
    del a, w
