#!/usr/bin/env python3
"""Experiment: one 160-clip batch on one stream vs two 80-clip half batches on two streams inside one hipGraph (the
half batches are independent, so the tail of every kernel of one half can be filled by the other half's kernels).
usage: python tools/dual_stream_exp.py [B] [streams]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build, synth_inputs
from lip2speech_unit_amd import ops
from lip2speech_unit_amd.pipeline import LipToSpeechPipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 160
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda")
model, voc, _, _ = build(ops.F16, dev, 24, 12)
pipe = LipToSpeechPipeline(model, voc)
_, spk_cpu, u8_cpu = synth_inputs(B, 100, seed=1234, with_u8=True)
spk, frames = spk_cpu.to(dev), u8_cpu.to(dev)
h = B // NS
parts = [(frames[i * h:(i + 1) * h].contiguous(), spk[i * h:(i + 1) * h].contiguous()) for i in range(NS)]


def single():
    return pipe.forward_device_u8(frames, None, spk)


streams = [torch.cuda.Stream() for _ in range(NS)]


def dual():
    cur = torch.cuda.current_stream()
    outs = []
    for st, (f, s_) in zip(streams, parts):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(pipe.forward_device_u8(f, None, s_))
    for st in streams:
        cur.wait_stream(st)
    return outs


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    g.replay()
    torch.cuda.synchronize()
    return g, out


def timeit(g, n=10):
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


g1, o1 = capture(single)
g2, outs = capture(dual)
for _ in range(2):
    t1, t2 = timeit(g1), timeit(g2)
    print(f"single stream, {B} clips: {t1:7.3f} ms ({t1 / B * 160:6.2f} per 160)    {NS} streams x {h} clips: {t2:7.3f} ms ({t2 / B * 160:6.2f} per 160)", flush=True)
tok = torch.cat([o["tokens"] for o in outs])
print("unit ids equal:", bool(torch.equal(tok, o1["tokens"])))
