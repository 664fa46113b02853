#!/usr/bin/env python3
"""Per-phase cycle counts (s_memtime) of l2s_resblock_fused / l2s_resstage_fused.  Needs a diagnostic build of the library:
  cd lip2speech_unit_amd/csrc && make && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DRB_STAMPS -c resblock.hip -o /tmp/rb_st.o \
    && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/lib_rbstamps.so $(ls build/*.o | grep -v build/resblock.o) /tmp/rb_st.o
usage: python tools/resblock_stamps.py [B] [stage]"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
os.environ["L2S_LIB_PATH"] = os.path.join(os.getcwd(), "build_ab", "lib_rbstamps.so")
from lip2speech_unit_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 160
stage = len(sys.argv) > 2 and sys.argv[2] == "stage"
lib = _lib.load()
raw = ctypes.CDLL(os.environ["L2S_LIB_PATH"])
raw.l2s_debug_rb_stamps.argtypes = [ctypes.c_void_p]
CFG = {(32, 3): (384, 4), (32, 7): (368, 4), (32, 11): (512, 8), (16, 3): (512, 4), (16, 7): (512, 4), (16, 11): (512, 4)}
SCFG = {32: (512, 8), 16: (512, 4)}
names = ["tile load", "weights->regs + barrier", "conv loops (c1, c2)", "last conv loop (global)", "commit + end barrier", "TOTAL"]


def report(s, NW, ki):
    clk = (s[:, 0, ki, 5] / s[:, 0, ki, 6]).median().item() * 100.0
    print(f"    in-kernel clock {clk:6.0f} MHz")
    for i, n in enumerate(names):
        print(f"    {n:28s} wave0 {s[:, 0, ki, i].mean():9.0f}   wave{NW-1} {s[:, NW - 1, ki, i].mean():9.0f} cycles")


for C, T in ((32, 32000), (16, 64000)):
    xl = torch.randn(B * T, C, device="cuda").half()
    xs = torch.zeros(B * T, C, device="cuda")
    nxt = torch.empty(B * T, C, device="cuda", dtype=torch.float16)
    lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
    ws = {k: (torch.randn(6, C, ((k * C + 31) // 32) * 32, device="cuda") / (k * C) ** 0.5).half() for k in (3, 7, 11)}
    bs = {k: torch.randn(6, C, device="cuda") for k in (3, 7, 11)}
    if stage:
        TT, NW = SCFG[C]
        nblk = ((T + TT - 1) // TT) * B
        buf = torch.zeros(nblk * NW * 3 * 8, dtype=torch.int64, device="cuda")
        assert raw.l2s_debug_rb_stamps(buf.data_ptr()) == 0
        for _ in range(3):
            ops.resstage_fused(xl, [ws[k] for k in (3, 7, 11)], [bs[k] for k in (3, 7, 11)], xs, nxt, B=B, T=T, C=C,
                               ks=(3, 7, 11), dils=((1, 3, 5),) * 3, slope=0.1, lens=lens, len_mul=1, dtype=ops.F16)
        torch.cuda.synchronize()
        s = buf.cpu().view(nblk, NW, 3, 8).double()
        for ki, k in enumerate((3, 7, 11)):
            print(f"stage C{C}, ResBlock k{k}: {nblk} blocks")
            report(s, NW, ki)
        continue
    for ki, k in enumerate((3, 7, 11)):
        TT, NW = CFG[(C, k)]
        nblk = ((T + TT - 1) // TT) * B
        buf = torch.zeros(nblk * NW * 3 * 8, dtype=torch.int64, device="cuda")
        assert raw.l2s_debug_rb_stamps(buf.data_ptr()) == 0
        for _ in range(3):
            ops.resblock_fused(xl, ws[k], bs[k], xs, nxt if k == 11 else None, B=B, T=T, C=C, k=k, dil=(1, 3, 5),
                               accumulate=k != 3, slope=0.1, lens=lens, len_mul=1, dtype=ops.F16)
        torch.cuda.synchronize()
        s = buf.cpu().view(nblk, NW, 3, 8).double()
        print(f"C{C} k{k}: {nblk} blocks")
        report(s, NW, ki)
