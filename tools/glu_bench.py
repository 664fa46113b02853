import os, sys
sys.path.insert(0, os.getcwd())
import torch
from lip2speech_unit_amd import ops
B, T, C, k = (int(sys.argv[1]) if len(sys.argv) > 1 else 160), 200, 512, 31
x = torch.randn(B * T, 2 * C, device="cuda").half()
w = torch.randn(k, C, device="cuda"); b = torch.randn(C, device="cuda")
y = torch.empty(B * T, C, device="cuda", dtype=torch.float16)
lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
def run(): ops.glu_dwconv_swish(x, w, b, y, B=B, T=T, C=C, k=k, lens=lens, len_mul=1, dtype=ops.F16)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"glu_dwconv B={B} T=200 C=512: {e0.elapsed_time(e1)/20*1e3:.1f} us")
