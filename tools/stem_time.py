#!/usr/bin/env python3
"""Times l2s_stem_pool_fused alone at the bench shape (160 clips x 100 frames); L2S_LIB_PATH selects a second build."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from lip2speech_unit_amd import ops, _lib
B, T = 160, 100
x = torch.randn(B, T, 88, 88, device="cuda")
w = (torch.randn(64, 288) * 0.05).half().cuda()
bias = torch.randn(64).cuda(); slope = torch.rand(64).cuda()
y = torch.empty(B * T * 22 * 22, 64, device="cuda", dtype=torch.float16)
lib = _lib.load()
import ctypes
def run():
    rc = lib.l2s_stem_pool_fused(x.data_ptr(), 1, w.data_ptr(), bias.data_ptr(), slope.data_ptr(), y.data_ptr(), B, T, 88, 88, ops.F16, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
u8 = torch.randint(0, 256, (B, T, 96, 96), dtype=torch.uint8, device="cuda")
def run_u8():
    rc = lib.l2s_stem_pool_fused_u8(u8.data_ptr(), 96, 96, 88, ctypes.c_float(0.421), ctypes.c_float(0.165), w.data_ptr(), bias.data_ptr(), slope.data_ptr(), y.data_ptr(), B, T, ops.F16, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
if os.environ.get("STEM_NEG"):          # one negative slope: the kernel takes the activation-before-pool order
    slope[3] = -0.2
if os.environ.get("STEM_U8", "1") != "0":
    run = run_u8
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(os.environ.get("L2S_LIB_PATH", "default"), "stem_pool ms:", ms, "per 640 clips:", ms * 4, "TFLOP/s (44x44x64x245 MACs):", B * T * 44 * 44 * 64 * 245 * 2 / ms / 1e9)
