#!/usr/bin/env python3
"""Yardstick only (never on the product path): the encoder / conformer GEMM shapes at the bench's clip count, ours (the
phase-staggered tap-GEMM with the layer's real epilogue: bias / GELU / fp32 residual stream) against torch.addmm (hipBLASLt,
fp16 out, bias) on the same box, interleaved in one process, hipGraph replay of 6 launches, best of 3.  Third column ("lib layer"):
what a library-based layer pays for the SAME epilogue - the library's fused activation epilogue where torch exposes one
(torch._addmm_activation: ReLU, tanh-form GELU) and otherwise the elementwise pass behind the GEMM (exact-erf GELU as fairseq's
F.gelu; the fp32 residual stream as `x += out`).
usage: python tools/gemm_yardstick.py [clips=640]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_unit_amd import ops
from lip2speech_unit_amd.ops import ACT_GELU, ACT_RELU, F_RES_POST

CLIPS = int(sys.argv[1]) if len(sys.argv) > 1 else 640
M1, M2 = CLIPS * 100, CLIPS * 200
SHAPES = [("enc qkv", M1, 3072, 1024, "b"), ("enc fc1", M1, 4096, 1024, "g"), ("enc fc2", M1, 1024, 4096, "r"),
          ("enc out", M1, 1024, 1024, "r"), ("conf ffn1", M2, 2048, 512, "u"), ("conf ffn2", M2, 512, 2048, "r"),
          ("conf qkv", M2, 1536, 512, "b"), ("conf out", M2, 512, 512, "r")]
REPS = 6


def graph_time(run):
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
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
        best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
    return best


print(f"{CLIPS} clips; ours = lip2speech tap-GEMM with the layer's epilogue, lib = torch.addmm (hipBLASLt) fp16 out + bias")
tot_o = tot_l = tot_y = 0.0
for name, M, N, K, kind in SHAPES:
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda")
    b16 = b.half()
    if kind == "r":
        x = torch.randn(M, N, device="cuda")
        ours = lambda: ops.tapgemm(a, w, x, M=M, N=N, Cin=K, bias=b, R=x, ldr=N, flags=F_RES_POST, dtype=ops.F16)
        epi = "fp32 residual stream"
    else:
        c = torch.empty(M, N, device="cuda", dtype=torch.float16)
        act = {"g": ACT_GELU, "u": ACT_RELU}.get(kind, 0)
        ours = lambda: ops.tapgemm(a, w, c, M=M, N=N, Cin=K, bias=b, act=act, dtype=ops.F16)
        epi = {"g": "GELU, 16-bit out", "u": "ReLU, 16-bit out"}.get(kind, "bias, 16-bit out")
    wt = w.t()
    lib = lambda: torch.addmm(b16, a, wt)
    layer = []          # (label, callable): the library path with the layer's real epilogue
    if kind == "g":
        layer.append(("addmm + F.gelu (erf)", lambda: torch.nn.functional.gelu(torch.addmm(b16, a, wt))))
        layer.append(("_addmm_activation gelu(tanh)", lambda: torch._addmm_activation(b16, a, wt, use_gelu=True)))
    elif kind == "u":
        layer.append(("_addmm_activation relu", lambda: torch._addmm_activation(b16, a, wt, use_gelu=False)))
    elif kind == "r":
        layer.append(("addmm + x.add_(out)", lambda: x.add_(torch.addmm(b16, a, wt))))
    t_o = graph_time(ours)
    t_l = graph_time(lib)
    t_o = min(t_o, graph_time(ours))
    t_l = min(t_l, graph_time(lib))
    fl = 2.0 * M * N * K
    tot_o += t_o
    tot_l += t_l
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d}  ours {t_o:7.1f} us {fl / t_o / 1e6:7.1f} TF ({epi:22s})   lib {t_l:7.1f} us {fl / t_l / 1e6:7.1f} TF"
          f"   ours/lib time {t_o / t_l:5.2f}", flush=True)
    best_layer = t_l
    for label, fn in layer:
        try:
            t_y = graph_time(fn)
        except Exception as e:                      # an epilogue this torch / hipBLASLt build does not offer
            print(f"{'':10s}   lib layer [{label}]: not available ({type(e).__name__})", flush=True)
            continue
        print(f"{'':10s}   lib layer [{label}] {t_y:7.1f} us {fl / t_y / 1e6:7.1f} TF   ours/lib-layer time {t_o / t_y:5.2f}", flush=True)
        best_layer = t_y if best_layer == t_l else min(best_layer, t_y)
    tot_y += best_layer
    del a, w
print(f"sum ours {tot_o:8.1f} us   lib (bias only) {tot_l:8.1f} us   lib with the layers' epilogues (best form each) {tot_y:8.1f} us")
