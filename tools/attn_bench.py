#!/usr/bin/env python3
"""Times l2s_attention at the bench shapes (HIP events on the launch stream): plain T=100 H=16 and rel-pos T=200 H=8 at B
clips; run once per L2S_ATTN_RESIDENT setting (the switch is read once per process):  python tools/attn_bench.py [B]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[2] == "child":
    import torch
    from lip2speech_unit_amd import ops
    B = int(sys.argv[1])
    g = torch.Generator().manual_seed(0)
    for (T, H, rel) in ((100, 16, False), (200, 8, True)):
        qkv = (torch.randn(B * T, 3 * H * 64, generator=g) * 0.5).half().cuda()
        out = torch.empty(B * T, H * 64, device="cuda", dtype=torch.float16)
        lens = torch.full((B,), T, dtype=torch.int32).cuda()
        kw = {}
        if rel:
            kw = dict(pos=(torch.randn(2 * T - 1, 12 * H * 64, generator=g) * 0.5).half().cuda()[:, : H * 64], ldp=12 * H * 64,
                      bias_u=torch.randn(H, 64).cuda() * 0.1, bias_v=torch.randn(H, 64).cuda() * 0.1)
        for _ in range(5):
            ops.attention(qkv, out, B=B, T=T, H=H, lens=lens, dtype=ops.F16, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            ops.attention(qkv, out, B=B, T=T, H=H, lens=lens, dtype=ops.F16, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / n
        fl = 2.0 * B * H * T * T * 64 * (3 if rel else 2)
        print(f"  T={T} H={H} relpos={rel}: {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s useful  "
              f"{(B * T * 4 * H * 64 * 2) / us / 1e3:6.0f} GB/s (qkv in + out)")
else:
    B = sys.argv[1] if len(sys.argv) > 1 else "160"
    for res in ("0", "1"):
        print(f"L2S_ATTN_RESIDENT={res}", flush=True)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), B, "child"], env=dict(os.environ, L2S_ATTN_RESIDENT=res))
