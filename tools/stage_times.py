#!/usr/bin/env python3
"""Per-stage GPU time of one bench step (HIP events, eager launches, averaged over a few repeats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build, synth_inputs
from lip2speech_unit_amd import ops
from lip2speech_unit_amd.pipeline import LipToSpeechPipeline

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    T = 100
    dt = ops.F16
    model, voc, _, _ = build(dt, torch.device("cuda"))
    pipe = LipToSpeechPipeline(model, voc)
    video, spk = synth_inputs(B, T)
    video, spk = video.cuda(), spk.cuda()
    w2v = model.encoder.w2v_model
    res = w2v.feature_extractor_video.resnet
    def ev():
        e = torch.cuda.Event(enable_timing=True); e.record(); return e
    for it in range(3):
        marks = [ev()]
        feat, _, _ = res.forward_rows(video); marks.append(ev())
        enc, lens, _, _ = w2v.extract_rows(video, None); marks.append(ev())   # includes the frontend again
        src16 = torch.empty(B * 2 * T, enc.shape[1], device="cuda", dtype=torch.float16)
        ops.repeat2_cast(enc, src16, B, T, enc.shape[1], dt)
        logits, mel, _ = model.conformer.forward_rows(src16, lens, B, 2 * T, spk, len_mul=2); marks.append(ev())
        s1 = pipe.stage1_device(video, None, spk); marks.append(ev())
        wav, pcm = pipe.stage2_device(s1, spk); marks.append(ev())
        torch.cuda.synchronize()
    t = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)]
    fl = {"frontend": 63.23, "encoder": 63.6, "conformer+heads": 34.98, "vocoder": 99.10}
    print(f"B={B}")
    print(f"frontend           {t[0]:7.2f} ms  {fl['frontend']*B/t[0]:7.1f} TFLOP/s")
    print(f"encoder (no front) {t[1]-t[0]:7.2f} ms  {fl['encoder']*B/(t[1]-t[0]):7.1f} TFLOP/s")
    print(f"conformer+heads    {t[2]:7.2f} ms  {fl['conformer+heads']*B/t[2]:7.1f} TFLOP/s")
    print(f"stage1 total       {t[3]:7.2f} ms")
    print(f"vocoder            {t[4]:7.2f} ms  {fl['vocoder']*B/t[4]:7.1f} TFLOP/s")

if __name__ == "__main__":
    main()
