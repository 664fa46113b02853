#!/usr/bin/env python3
"""Per-phase cycle counts (s_memtime) of the fused stem+pool kernel.  Needs a debug build of the library:
  cd lip2speech_unit_amd/csrc && make && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DL2S_STEM_STAMPS -c frontend.hip -o /tmp/fe_st.o \
    && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libstamps.so $(ls build/*.o | grep -v frontend.o) /tmp/fe_st.o
The stamps serialise the LDS pipeline a little (s_memtime returns through lgkmcnt): read them as shares, not times."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
os.environ.setdefault("L2S_LIB_PATH", os.path.join(os.getcwd(), "build_ab", "stemst", "liblip2speech_hip.so"))   # tools/build_variant.sh stemst -DL2S_STEM_STAMPS frontend.hip
from lip2speech_unit_amd import ops, _lib
B, T = 160, 100
x = torch.randn(B, T, 88, 88, device="cuda")
w = (torch.randn(64, 288) * 0.05).half().cuda()
bias = torch.randn(64).cuda(); slope = torch.rand(64).cuda()
y = torch.empty(B * T * 22 * 22, 64, device="cuda", dtype=torch.float16)
lib = _lib.load()
raw = ctypes.CDLL(os.environ["L2S_LIB_PATH"])
nblk = 6 * 10 * B
buf = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda")
raw.l2s_debug_stem_stamps.argtypes = [ctypes.c_void_p]
print("set", raw.l2s_debug_stem_stamps(buf.data_ptr()))
def run():
    rc = lib.l2s_stem_pool_fused(x.data_ptr(), 1, w.data_ptr(), bias.data_ptr(), slope.data_ptr(), y.data_ptr(), B, T, 88, 88, ops.F16, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
for _ in range(3): run()
torch.cuda.synchronize()
s = buf.cpu().view(nblk, 4, 4).double()   # [block, wave, stamp]
names = ["sync0 (slab/pool wait)", "conv loop", "  of which epilogues", "pool"]
for i, n in enumerate(names):
    print(f"{n:28s} mean per frame: wave0 {s[:, 0, i].mean() / 10:9.0f}  wave3 {s[:, 3, i].mean() / 10:9.0f} ticks")
print("sum per frame wave0:", s[:, 0].sum(-1).mean().item() / 10)
