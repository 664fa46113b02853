#!/usr/bin/env python3
"""A/B of the vocoder stage on one box: L2S_RESPAIR=0 (unfused tap-GEMM conv pairs) vs 1 (csrc/respair.hip) - or, with a second
argument, another switch (L2S_FUSED_UPS) - each setting in its own process (the switches are read at import), HIP events over
eager launches:  python tools/vocoder_ab.py [B] [SWITCH]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[2] == "child":
    import torch
    from bench import VOC_H
    from lip2speech_unit_amd import ops, weights
    from lip2speech_unit_amd.vocoder import AttrDict, MelCodeGenerator
    B = int(sys.argv[1])
    voc = MelCodeGenerator(AttrDict(VOC_H), dtype=ops.F16)
    voc.load_state_dict(weights.synth_state_dict(weights.spec_of(voc), seed=1))
    voc.remove_weight_norm()
    voc.cuda().eval()
    g = torch.Generator().manual_seed(0)
    code = torch.randint(0, 200, (B, 200), generator=g).cuda()
    mel = (-11.5 + 11.6 * torch.rand(B, 80, 400, generator=g)).cuda()
    spk = torch.rand(B, 256, generator=g).cuda()
    with torch.no_grad():
        for _ in range(2):
            wav, _ = voc.forward_rows(code, mel, spk)
        prof = ops.KernelProfiler()
        ops.set_profiler(prof)
        R = 3
        for _ in range(R):
            voc.forward_rows(code, mel, spk)
        ops.set_profiler(None)
    agg = prof.summary()
    tot = sum(a["ms"] for a in agg.values())
    print(f"  vocoder kernels {tot / R:.3f} ms per {B} clips   wav checksum {float(wav.double().abs().sum()):.6e}")
    groups = {}
    for k, a in agg.items():
        key = k if k.startswith("l2s_res") else k.split(",")[1] + k[k.index(",mode"):] if k.startswith("tapgemm") else k
        gk = groups.setdefault(key, [0.0, 0.0, 0])
        gk[0] += a["ms"]; gk[1] += a["flops"]; gk[2] += a["calls"]
    for k, (ms, fl, n) in sorted(groups.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("L2S_AB_ROWS", "14"))]:
        print(f"    {k:42s} {n // R:3d} calls {ms / R:7.3f} ms  {fl / ms / 1e9 if fl else 0:7.1f} TF")
else:
    B = sys.argv[1] if len(sys.argv) > 1 else "160"
    var = sys.argv[2] if len(sys.argv) > 2 else "L2S_RESPAIR"      # or L2S_FUSED_UPS
    for v in ("0", "1", "0", "1"):
        print(f"{var}={v}", flush=True)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), B, "child"], env=dict(os.environ, **{var: v}))
