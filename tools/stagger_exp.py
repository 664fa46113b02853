#!/usr/bin/env python3
"""Encoder / conformer GEMMs at the bench's 640-clip shapes on the phase-staggered kernel (L2S_PHASEGEMM=2 forces it), hipGraph
replay.  Used to A/B the start stagger of the persistent blocks: run once per L2S_PHASE_STAGGER / L2S_PHASE_GROUPS setting
(the library reads them once per process) inside ONE gpurun call.
usage: python tools/stagger_exp.py [clips=640]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_unit_amd import ops
from lip2speech_unit_amd.ops import ACT_GELU, F_RES_POST

CLIPS = int(sys.argv[1]) if len(sys.argv) > 1 else 640
M1, M2 = CLIPS * 100, CLIPS * 200
SHAPES = [("enc qkv", M1, 3072, 1024, "b"), ("enc fc1", M1, 4096, 1024, "g"), ("enc fc2", M1, 1024, 4096, "r"),
          ("enc out", M1, 1024, 1024, "r"), ("conf ffn1", M2, 2048, 512, "b"), ("conf ffn2", M2, 512, 2048, "r"),
          ("conf qkv", M2, 1536, 512, "b"), ("conf out", M2, 512, 512, "r")]
reps = 6
tot = 0.0
for name, M, N, K, kind in SHAPES:
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda")
    if kind == "r":
        x = torch.randn(M, N, device="cuda")
        run = lambda: ops.tapgemm(a, w, x, M=M, N=N, Cin=K, bias=b, R=x, ldr=N, flags=F_RES_POST, dtype=ops.F16)
    else:
        c = torch.empty(M, N, device="cuda", dtype=torch.float16)
        act = ACT_GELU if kind == "g" else 0
        run = lambda: ops.tapgemm(a, w, c, M=M, N=N, Cin=K, bias=b, act=act, dtype=ops.F16)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            run()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    tot += best
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d}  {best:8.1f} us  {2.0 * M * N * K / best / 1e6:7.1f} TFLOP/s", flush=True)
    del a, w
print(f"sum {tot:8.1f} us  stagger={os.environ.get('L2S_PHASE_STAGGER', '0')} groups={os.environ.get('L2S_PHASE_GROUPS', '2')} "
      f"phasegemm={os.environ.get('L2S_PHASE_STAGGER') and os.environ.get('L2S_PHASEGEMM')}")
