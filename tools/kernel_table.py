#!/usr/bin/env python3
"""Full per-kernel table of one bench step (HIP-event timed eager pass): python tools/kernel_table.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build, synth_inputs
from lip2speech_unit_amd import ops
from lip2speech_unit_amd.pipeline import LipToSpeechPipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
DETAIL = "--detail" in sys.argv
model, voc, _, _ = build(ops.F16, torch.device("cuda"))
pipe = LipToSpeechPipeline(model, voc)
video, spk = synth_inputs(B, 100)
video, spk = video.cuda(), spk.cuda()
for _ in range(2):
    pipe.forward_device(video, None, spk)
prof = ops.KernelProfiler(detail=DETAIL)
ops.set_profiler(prof)
R = 3
for _ in range(R):
    pipe.forward_device(video, None, spk)
ops.set_profiler(None)
agg = prof.summary()
tot = sum(a["ms"] for a in agg.values())
print(f"B={B}  sum of kernels {tot / R:.3f} ms/step")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
    tf = f"{a['flops'] / a['ms'] / 1e9:7.1f} TF" if a["flops"] else "          "
    gb = f"{a['bytes'] / a['ms'] / 1e6:7.0f} GB/s(alg)" if a["bytes"] else ""
    print(f"{k:{64 if DETAIL else 34}s} {a['calls'] // R:4d} calls {a['ms'] / R:7.3f} ms {100 * a['ms'] / tot:5.1f}%  {tf} {gb}")
