#!/usr/bin/env python3
"""Isolated timings of the Cin = N = 64 convolutions (ResNet layer1 3x3, vocoder C=64 Conv1d) through l2s_tapgemm.
usage: conv_bench.py [reps] [filter] ; L2S_NO_STREAMCONV=1 / L2S_NO_PATCHCONV=1 select the older kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_unit_amd import ops

CASES = [  # name, mode, units, T or H, taps, dil, residual, dual
    ("l1 conv1", "2d", 8000, 22, 9, 1, False, False),
    ("l1 conv2+res", "2d", 8000, 22, 9, 1, True, False),
    ("s3 k3 c1", "1d", 80, 16000, 3, 1, False, False),
    ("s3 k3 c2+res+dual", "1d", 80, 16000, 3, 1, True, True),
    ("s3 k7 c1 d3", "1d", 80, 16000, 7, 3, False, False),
    ("s3 k11 c1 d5", "1d", 80, 16000, 11, 5, False, False),
    ("s3 k11 c2+res+dual", "1d", 80, 16000, 11, 1, True, True),
]


STAMPS = None


def main():
    global STAMPS
    if os.environ.get("L2S_PATCH_STAMPS"):
        import ctypes
        raw = ctypes.CDLL(os.environ["L2S_LIB_PATH"])
        raw.l2s_debug_patch_stamps.argtypes = [ctypes.c_void_p]
        STAMPS = torch.zeros(512 * 8, dtype=torch.int64, device="cuda")
        assert raw.l2s_debug_patch_stamps(STAMPS.data_ptr()) == 0
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    only = sys.argv[2] if len(sys.argv) > 2 else None
    dt, C = ops.F16, 64
    for name, mode, units, T, k, dil, res, dual in CASES:
        if only and only not in name:
            continue
        M = units * (T * T if mode == "2d" else T)
        a = torch.randn(M, C, device="cuda").half()
        w = (torch.randn(C, k * C, device="cuda") / (k * C) ** 0.5).half()
        b = torch.randn(C, device="cuda")
        c = torch.empty(M, C, device="cuda", dtype=torch.float16)
        c2 = torch.empty(M, C, device="cuda", dtype=torch.float16) if dual else None
        r = torch.randn(M, C, device="cuda").half() if res else None
        kw = dict(M=M, N=C, Cin=C, ntaps=k, bias=b, dtype=dt, act=ops.ACT_LRELU, act_slope=0.1)
        if mode == "2d":
            kw.update(mode=ops.MODE_CONV2D, Ho=T, Wo=T, Hi=T, Wi=T, KW=3, pad=1, stride=1)
        else:
            kw.update(mode=ops.MODE_CONV1D, T_out=T, T_in=T, stride=1, dil=dil, off=-(k * dil - dil) // 2)
        fl = 0
        if res:
            kw.update(R=r, ldr=C)
            fl |= ops.F_RES_POST
        if dual:
            kw.update(C2=c2, ldc2=C, slope2=0.1)
            fl |= ops.F_DUAL
        kw["flags"] = fl

        def run():
            ops.tapgemm(a, w, c, **kw)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        if STAMPS is not None:   # debug build (-DL2S_PATCH_STAMPS): per-tile cycle shares of the last launch, wave 0 of every block
            torch.cuda.synchronize()
            st = STAMPS.cpu().view(-1, 8).double()
            st = st[st[:, 5] > 0]
            per_tile = st[:, :5].sum(0) / st[:, 5].sum()
            print("      ticks/tile: start-wait %.0f  taps %.0f  pre-epilogue barrier %.0f  epilogue %.0f  patch issue %.0f  (sum %.0f)"
                  % (*per_tile.tolist(), per_tile.sum().item()))
            STAMPS.zero_()
        nbytes = M * C * 2 * (2 + int(res) + int(dual))
        print(f"{name:20s} M={M:8d} k={k:2d}  {us:8.1f} us  {2.0 * M * C * C * k / us / 1e6:7.1f} TFLOP/s  "
              f"{nbytes / us / 1e3:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
