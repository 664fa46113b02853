#!/usr/bin/env python3
"""LayerNorm at the bench shapes (fp32 in, 16-bit out), HIP events over a hipGraph of back-to-back launches; run per
L2S_LN_ROWS setting:  python tools/ln_bench.py"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from lip2speech_unit_amd import ops
    for (M, C) in ((16000, 1024), (32000, 512)):
        xs = [torch.randn(M, C, device="cuda") for _ in range(6)]     # rotate buffers: 6 x 65 MB > the 256 MB Infinity Cache
        y = torch.empty(M, C, device="cuda", dtype=torch.float16)
        g, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        for x in xs:
            ops.layernorm(x, g, b, 1e-5, y, M=M, C=C, dtype=ops.F16)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 60
        e0.record()
        for i in range(n):
            ops.layernorm(xs[i % 6], g, b, 1e-5, y, M=M, C=C, dtype=ops.F16)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / n
        print(f"  M={M} C={C}: {us:6.1f} us  {M * C * 6 / us / 1e6:5.2f} TB/s (4 B in + 2 B out)")
else:
    for v in ("0", "1"):
        print(f"L2S_LN_ROWS={v}", flush=True)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, L2S_LN_ROWS=v))
