import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_unit_amd import ops
B = 80
for C, T in ((32, 32000), (16, 64000)):
    for k in (3, 7, 11):
        xl = torch.randn(B * T, C, device="cuda").half()
        w = (torch.randn(6, C, ((k * C + 31) // 32) * 32, device="cuda") / (k * C) ** 0.5).half()
        b = torch.randn(6, C, device="cuda")
        xs = torch.zeros(B * T, C, device="cuda")
        nxt = torch.empty(B * T, C, device="cuda", dtype=torch.float16)
        lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
        def run():
            ops.resblock_fused(xl, w, b, xs, nxt, B=B, T=T, C=C, k=k, dil=(1, 3, 5), accumulate=True, slope=0.1, lens=lens, len_mul=1, dtype=ops.F16)
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        print(f"C{C} k{k}: {e0.elapsed_time(e1)/5*1e3:8.1f} us", flush=True)
