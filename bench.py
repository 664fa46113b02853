#!/usr/bin/env python3
"""Headline benchmark: end-to-end lip -> units -> 16 kHz waveform on synthetic 4-s 25-fps 88x88 clips.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the whole hot path over one batch resident in HBM: fp32 frames [B,1,100,88,88] + speaker
embeddings -> ResNet-18 frontend -> AV-HuBERT large encoder (24 layers) -> conformer (12 blocks) -> unit/mel heads ->
greedy unit decode -> multi-input HiFi-GAN vocoder -> int16 PCM (all on device; weights random-init of the reference
architecture, data synthetic).  Prints ONE JSON line (rank 0) with the real-time factor (audio-seconds per wall-second,
whole job), the roofline of the dominant kernel (HIP-event timed on the launch stream) and a CPU baseline (the oracle,
on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from lip2speech_unit_amd import distributed as l2s_dist  # noqa: E402
from lip2speech_unit_amd import ops, weights  # noqa: E402
from lip2speech_unit_amd.conformer import ConformerConfig  # noqa: E402
from lip2speech_unit_amd.hubert import AVHubertConfig  # noqa: E402
from lip2speech_unit_amd.model_avhubert import MultiTargetAVHubertEncoderModel  # noqa: E402
from lip2speech_unit_amd.pipeline import LipToSpeechPipeline  # noqa: E402
from lip2speech_unit_amd.vocoder import AttrDict, MelCodeGenerator  # noqa: E402

VOC_H = dict(resblock="1", upsample_rates=[5, 4, 2, 2, 2], upsample_kernel_sizes=[11, 8, 4, 4, 4],
             upsample_initial_channel=512, resblock_kernel_sizes=[3, 7, 11],
             resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], num_embeddings=200, embedding_dim=128,
             model_in_dim=336, embedder_dim=256, multispkr="_", num_mels=80, text_supervision=False)
PEAK_MFMA_TFLOPS = 2500.0   # dense bf16/fp16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0


def synth_inputs(B, T, seed=1234):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (B, T, 96, 96), generator=g, dtype=torch.uint8)
    x = (u8[:, :, 4:92, 4:92].float() / 255.0 - 0.421) / 0.165          # hubert_dataset.py:242-245
    spk = torch.rand(B, 256, generator=g).relu()
    spk = spk / spk.norm(dim=-1, keepdim=True)
    return x.unsqueeze(1).contiguous(), spk


def build(dtype, device, enc_layers=24, conf_layers=12, seed=0):
    model = MultiTargetAVHubertEncoderModel.build_model(
        dtype=dtype, w2v_cfg=AVHubertConfig(encoder_layers=enc_layers),
        conformer_cfg=ConformerConfig(conformer_layers=conf_layers))
    sd = weights.synth_state_dict(weights.spec_of(model), seed=seed)
    model.load_state_dict(sd)
    voc = MelCodeGenerator(AttrDict(VOC_H), dtype=dtype)
    vsd = weights.synth_state_dict(weights.spec_of(voc), seed=seed + 1)
    voc.load_state_dict(vsd)
    voc.remove_weight_norm()
    model.to(device).eval()
    voc.to(device).eval()
    return model, voc, sd, vsd


def cpu_baseline(sd, vsd, video, spk, gpu_out, n_clips, enc_layers=24, conf_layers=12):
    """The oracle (CPU restatement, kind 'port') on the first n_clips clips of the GPU batch, one clip per forward (the
    reference's own batch size); also the full-size parity check of the GPU step against it."""
    from oracle import stage1 as os1
    from oracle import vocoder as ov
    T = video.shape[2]
    n_safe = n_ok = 0
    mel_err = wav_err = 0.0
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(n_clips):
            r = os1.generate(sd, video[i:i + 1], torch.zeros(1, T, dtype=torch.bool), spk[i:i + 1], enc_layers=enc_layers,
                             conf_layers=conf_layers)
            code = (r["tokens"][0][:-1] - 4).clamp(min=0).unsqueeze(0)
            mel = r["mels"][0].t().unsqueeze(0)
            wav = ov.mel_code_generator(vsd_removed(vsd), VOC_H, code, mel, spk[i:i + 1])
            ov.to_int16(wav)
            # parity of the GPU step (same clip, batched with the others) against the oracle run alone
            lr = r["logits"][:, 0, 4:]
            top2 = lr.topk(2, -1).values
            safe = (top2[:, 0] - top2[:, 1]) > 2e-2
            gt = gpu_out["tokens"][i, : 2 * T].long()
            n_safe += int(safe.sum())
            n_ok += int((gt[safe] == r["tokens"][0][: 2 * T][safe]).sum())
            mel_err = max(mel_err, float((gpu_out["mel"][i] - r["mels"][0]).abs().max()))
            if bool((gt == r["tokens"][0][: 2 * T]).all()):
                wav_err = max(wav_err, float((gpu_out["wav"][i] - wav[0, 0]).abs().max()))
    dt = time.perf_counter() - t0
    parity = {"clips": n_clips, "unit_ids_equal": n_ok, "unit_ids_compared": n_safe, "unit_frames": n_clips * 2 * T,
              "mel_max_abs_err": round(mel_err, 5), "wav_max_abs_err": round(wav_err, 5)}
    return n_clips * T / 25.0 / dt, dt, parity


_VSD_CACHE = {}


def vsd_removed(vsd):
    if "x" not in _VSD_CACHE:
        from lip2speech_unit_amd.packing import weight_norm_weight
        out = {}
        for k, v in vsd.items():
            if k.endswith("weight_g"):
                p = k[: -len(".weight_g")]
                out[p + ".weight"] = weight_norm_weight(vsd, p)
            elif not k.endswith("weight_v"):
                out[k] = v
        _VSD_CACHE["x"] = out
    return _VSD_CACHE["x"]


def bench_mixed(args, pipe, rank, world, dev):
    """BASELINE configs[4]: mixed 1-10 s clips.  The global clip list is dealt to ranks by sorted length
    (distributed.shard_by_length), every rank pads its clips into length buckets and replays one hipGraph per bucket;
    a step = all buckets of the rank once + one padded all_gather of the unit ids per bucket."""
    import numpy as np
    rng = np.random.default_rng(1234)
    lengths_all = rng.integers(25, 251, size=args.clips * world)
    mine = l2s_dist.shard_by_length(lengths_all.tolist(), world, rank)
    my_lens = sorted((int(lengths_all[i]) for i in mine), reverse=True)
    buckets = [my_lens[i:i + args.bucket] for i in range(0, len(my_lens), args.bucket)]
    work = []
    for bi, lens in enumerate(buckets):
        Tb, Bb = max(lens), len(lens)
        video, spk = synth_inputs(Bb, Tb, seed=4321 + 97 * rank + bi)
        pad = torch.ones(Bb, Tb, dtype=torch.bool)
        for j, n in enumerate(lens):
            pad[j, :n] = False
            video[j, :, n:] = 0
        work.append({"video": video.to(dev), "spk": spk.to(dev), "pad": pad.to(dev), "lens": lens})

    for w in work:                                   # warm-up + graph capture per bucket shape
        for _ in range(max(args.warmup, 1)):
            w["out"] = pipe.forward_device(w["video"], w["pad"], w["spk"])
    torch.cuda.synchronize()
    if not args.no_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for w in work:
                pipe.forward_device(w["video"], w["pad"], w["spk"])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for w in work:
            w["graph"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(w["graph"]):
                w["out"] = pipe.forward_device(w["video"], w["pad"], w["spk"])
        torch.cuda.synchronize()

    def run_step():
        for w in work:
            if "graph" in w:
                w["graph"].replay()
            else:
                w["out"] = pipe.forward_device(w["video"], w["pad"], w["spk"])
            if world > 1:
                l2s_dist.gather_padded(w["out"]["tokens"], w["out"]["lens"] * 2)

    run_step()
    torch.cuda.synchronize()
    l2s_dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    torch.cuda.synchronize()
    l2s_dist.barrier()
    torch.cuda.synchronize()
    elapsed = l2s_dist.max_over_ranks(time.perf_counter() - t0, dev)
    audio_s = float(lengths_all.sum()) / 25.0 * args.steps
    padded = sum(max(w["lens"]) * len(w["lens"]) for w in work)
    if rank == 0:
        print(json.dumps({
            "metric": "real-time factor (audio-sec/wall-sec), end-to-end lip->16kHz audio, mixed 1-10 s clips",
            "value": round(audio_s / elapsed, 2), "unit": "audio-sec/wall-sec",
            "clips_per_sec": round(args.clips * world * args.steps / elapsed, 2), "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "fp16" if args.dtype == "f16" else "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: mixed 1-10 s clips (25..250 frames, seed 1234), %d clips per GPU, "
                                   "dealt by sorted length, buckets of %d" % (args.clips, args.bucket),
                       "clips_per_gpu": args.clips, "bucket": args.bucket, "hipgraph": not args.no_graph,
                       "padding_overhead_rank0": round(padded / float(sum(my_lens)), 4),
                       "parallelism": f"clip-parallel dp{world}"},
            "roofline": None, "cpu_baseline": None}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=160,
                    help="clips per GPU per step (160 x 100 frames = 16000 rows = 62.5 row tiles of 256: whole waves of "
                         "tiles on 256 CUs; 32 = BASELINE configs[2] batch)")
    ap.add_argument("--frames", type=int, default=100, help="video frames per clip (100 = 4 s @ 25 fps)")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=2)
    ap.add_argument("--u8", action="store_true",
                    help="feed uint8 96x96 frames and run the crop/normalise kernel inside the step (SURVEY 8f row 1)")
    ap.add_argument("--mixed", action="store_true",
                    help="BASELINE configs[4]: --clips clips per GPU of 1-10 s (25..250 frames, seed 1234), dealt to ranks by "
                         "sorted length and run as length buckets of --bucket clips (one hipGraph per bucket shape)")
    ap.add_argument("--clips", type=int, default=256)
    ap.add_argument("--bucket", type=int, default=32)
    ap.add_argument("--enc-layers", type=int, default=24)
    ap.add_argument("--conf-layers", type=int, default=12)
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    rank, world, local = l2s_dist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dt = ops.F16 if args.dtype == "f16" else ops.BF16
    B, T = args.batch, args.frames

    model, voc, sd, vsd = build(dt, dev, args.enc_layers, args.conf_layers)
    pipe = LipToSpeechPipeline(model, voc)
    if args.mixed:
        return bench_mixed(args, pipe, rank, world, dev)
    video, spk = synth_inputs(B, T, seed=1234 + rank)
    video, spk = video.to(dev), spk.to(dev)
    frames_u8 = None
    if args.u8:   # the same pixels as uint8 [B,T,96,96]: synth_inputs' generator draws them first
        g = torch.Generator().manual_seed(1234 + rank)
        frames_u8 = torch.randint(0, 256, (B, T, 96, 96), generator=g, dtype=torch.uint8).to(dev)

    def step():
        if frames_u8 is not None:
            return pipe.forward_device_u8(frames_u8, None, spk)
        return pipe.forward_device(video, None, spk)

    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    graph = None
    if not args.no_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = step()
        graph.replay()
        torch.cuda.synchronize()

    def run_step():
        if graph is not None:
            graph.replay()
        else:
            step()
        if world > 1:  # batch collation: one padded all_gather of the unit ids per batch (RCCL over xGMI)
            l2s_dist.gather_padded(out["tokens"], out["lens"] * 2)

    run_step()
    torch.cuda.synchronize()
    l2s_dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    torch.cuda.synchronize()
    l2s_dist.barrier()
    torch.cuda.synchronize()
    elapsed = l2s_dist.max_over_ranks(time.perf_counter() - t0, dev)

    audio_s = world * B * (T / 25.0) * args.steps
    rtf = audio_s / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernel: per-launch HIP events on the launch stream, eager pass ----
    roofline = None
    top = []
    if rank == 0:
        prof = ops.KernelProfiler()
        ops.set_profiler(prof)
        for _ in range(2):
            step()
        ops.set_profiler(None)
        agg = prof.summary()
        tot_ms = sum(a["ms"] for a in agg.values())
        ranked = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
        for k, a in ranked[:8]:
            top.append({"kernel": k, "calls_per_step": a["calls"] // 2, "ms_per_step": round(a["ms"] / 2, 3),
                        "share": round(a["ms"] / tot_ms, 3),
                        "tflops": round(a["flops"] / a["ms"] / 1e9, 1) if a["flops"] else None})
        dom_k, dom = next(((k, a) for k, a in ranked if a["flops"] > 0), ranked[0])
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (tools/collect_traffic.sh:
        # separate FETCH_SIZE / WRITE_SIZE runs, gfx950 FETCH_SIZE x2 correction); only valid for the batch it was
        # collected at
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("batch") == B and tj.get("frames", 100) == T:
                tk = tj["kernels"].get(dom_k)
                traffic = tk["hbm_bytes_per_launch"] if tk else None
        roofline = {"kernel": dom_k, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_TFLOPS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["calls"]),
                    "avg_launch_us": round(1e3 * dom["ms"] / dom["calls"], 2),
                    "flop_per_launch": round(dom["flops"] / dom["calls"]), "share_of_step": round(dom["ms"] / tot_ms, 3)}

    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        torch.cuda.synchronize()
        gpu_out = {k: out[k].float().cpu() if k != "tokens" else out[k].cpu() for k in ("tokens", "mel", "wav")}
        val, secs, parity = cpu_baseline({k: v.float() for k, v in sd.items()}, vsd, video.cpu(), spk.cpu(), gpu_out,
                                         args.cpu_clips, args.enc_layers, args.conf_layers)
        cpu = {"value": round(val, 4), "unit": "audio-sec/wall-sec", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{args.cpu_clips} x 4-s clips, batch 1, full path (oracle fp32), {secs:.1f} s wall"}

    if rank == 0:
        line = {
            "metric": "real-time factor (audio-sec/wall-sec), end-to-end lip->16kHz audio, 4s@25fps clips",
            "value": round(rtf, 2), "unit": "audio-sec/wall-sec", "clips_per_sec": round(world * B * args.steps / elapsed, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp16" if dt == ops.F16 else "bf16", "data": "synthetic",
            "config": {"workload": "e2e lip->units->wav (BASELINE configs[3]: AV-HuBERT large 24L + conformer 12x512 + "
                                   "multi_input HiFi-GAN), 4-s 100-frame 88x88 clips, batch %d per GPU" % B,
                       "clips_per_gpu": B, "frames_per_clip": T, "hipgraph": graph is not None,
                       "input": "uint8 96x96 frames, crop+normalise on device" if args.u8 else "fp32 88x88 normalised frames",
                       "enc_layers": args.enc_layers, "conf_layers": args.conf_layers,
                       "parallelism": f"clip-parallel dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu, "parity_vs_oracle": parity, "top_kernels": top,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
