#!/usr/bin/env python3
"""Headline benchmark: end-to-end lip -> units -> 16 kHz waveform on synthetic 4-s 25-fps 88x88 clips.

  python bench.py --gpus N --steps K --warmup W
      N > 1 without a torch.distributed.run environment: this process spawns the N ranks itself (python -m
      torch.distributed.run --nproc-per-node N ... bench.py, one rank per GPU over RCCL) BEFORE it touches the GPU and
      exits with their return code; under torch.distributed.run (RANK/WORLD_SIZE set) it is one of the ranks.

A step = one pass of the whole hot path over one batch resident in HBM (640 clips by default): uint8 96x96 frames [B,100,96,96] + speaker
embeddings -> crop/normalise -> ResNet-18 frontend -> AV-HuBERT large encoder (24 layers) -> conformer (12 blocks) ->
unit/mel heads -> greedy unit decode -> multi-input HiFi-GAN vocoder -> int16 PCM (all on device; weights random-init of
the reference architecture, data synthetic).  Prints ONE JSON line (rank 0): the real-time factor (audio-seconds per
wall-second, whole job, inputs resident in HBM), the same with the PCIe transfers inside the timed region
(`transfer_inclusive`: double-buffered pinned-host -> device frames on a side stream, PCM back to the host), the
roofline of the dominant kernel (HIP-event timed on the launch stream), whole-step MFMA utilisation and a CPU baseline
(the oracle, on a bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
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
GFLOP_PER_4S_CLIP = 260.9   # frontend 63.23 + encoder 63.6 + conformer/heads 34.98 + vocoder 99.10


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no rendezvous environment: start the N ranks as children of this process, which has
    not initialised the GPU (no torch.cuda call so far; a process that has must never exec), relay their output (rank 0
    prints the JSON line on the inherited stdout) and return their exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    # The pool's host driver supports dmabuf IPC only; without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL and tensor sharing across the
    # ranks fail with `hipIpcGetMemHandle: invalid argument` (stated by the pool's operators for this image, exported on its
    # boxes).  setdefault: an exported value - the caller's choice, whatever it is - passes through untouched; the variable is
    # added only where the environment lacks it.  L2S_BENCH_KEEP_ENV=1 forwards the environment exactly as it is.
    if os.environ.get("L2S_BENCH_KEEP_ENV", "0") != "1":
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd, env=env)


def timed_region(run_step, steps, sync, dev):
    """The contract's timing: barrier + device sync on both sides of EXACTLY `steps` steps, max over ranks."""
    sync()
    l2s_dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_step()
    sync()
    own = time.perf_counter() - t0             # this rank's K steps alone (before the closing barrier): skew between ranks
    l2s_dist.barrier()
    sync()
    PER_RANK["ms_per_step"] = [round(1e3 * x / steps, 3) for x in l2s_dist.all_ranks(own, dev)]
    return l2s_dist.max_over_ranks(time.perf_counter() - t0, dev)


PER_RANK = {"ms_per_step": None}   # filled by timed_region: every rank's own ms per step, rank order (printed beside the max)


def bench_stub(args):
    """CPU rehearsal of the launcher + timing harness + collation (tests/test_bench_launcher_cpu.py): gloo ranks, a stub step."""
    rank, world, _ = l2s_dist.init_from_env("gloo")
    assert world == args.gpus, (world, args.gpus)
    if os.environ.get("L2S_BENCH_STUB_FAIL_RANK") == str(rank):   # test hook: a failing child must fail the launcher
        sys.exit(3)
    dev = torch.device("cpu")
    if args.mixed:
        return bench_stub_mixed(args, rank, world, dev)
    B, T2 = 4, 40
    a = torch.randn(64, 64)
    toks = torch.full((B, T2 + 1), 4 + rank, dtype=torch.int32)
    lens = torch.full((B,), T2, dtype=torch.int32)
    seen = {}

    def run_step():
        torch.mm(a, a)
        if world > 1:
            seen["t"], seen["l"] = l2s_dist.gather_padded(toks, lens, static_shape=True)

    run_step()
    elapsed = timed_region(run_step, args.steps, lambda: None, dev)
    if world > 1:
        assert seen["t"].shape == (world * B, T2 + 1) and all(int(seen["t"][r * B, 0]) == 4 + r for r in range(world))
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": round(world * B * args.steps / elapsed, 2), "unit": "stub-clips/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * elapsed / args.steps, 3), "per_rank_ms_per_step": PER_RANK["ms_per_step"],
                          "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "synthetic", "config": {"workload": "stub step on CPU (gloo)"},
                          "roofline": None, "cpu_baseline": None}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def bench_stub_mixed(args, rank, world, dev):
    """CPU rehearsal of `--mixed` at N > 1: the real length list / dealing / buckets, a stub step per bucket and the RAGGED
    padded all_gather (ranks hold buckets of different clip counts and lengths, so the shapes are agreed on first)."""
    lengths_all, my_lens, buckets = mixed_buckets(args.clips, args.bucket, rank, world, args.bucket_frames)
    a = torch.randn(32, 32)
    seen = []

    def run_step():
        seen.clear()
        for lens in buckets:
            torch.mm(a, a)
            Tb = max(lens)
            toks = torch.full((len(lens), 2 * Tb + 1), 1, dtype=torch.int32)
            for j, n in enumerate(lens):
                toks[j, : 2 * n] = 4 + (n + j + rank) % 200
                toks[j, 2 * n] = 2
            seen.append(l2s_dist.gather_padded(toks, torch.tensor([2 * n for n in lens], dtype=torch.int32)))

    run_step()
    elapsed = timed_region(run_step, args.steps, lambda: None, dev)
    # every rank must hold every rank's clips of each bucket round: check rows, lengths and payload against the dealing
    per_rank = [mixed_buckets(args.clips, args.bucket, r, world, args.bucket_frames)[2] for r in range(world)]
    for bi, (all_t, all_l) in enumerate(seen):
        bmax = max(len(per_rank[r][bi]) for r in range(world))
        lmax = max(2 * max(per_rank[r][bi]) + 1 for r in range(world))
        assert all_t.shape == (world * bmax, lmax), (all_t.shape, world, bmax, lmax)
        for r in range(world):
            for j, n in enumerate(per_rank[r][bi]):
                row = all_t[r * bmax + j]
                assert int(all_l[r * bmax + j]) == 2 * n and int(row[0]) == 4 + (n + j + r) % 200 and int(row[2 * n]) == 2
            assert all(int(x) == 0 for x in all_l[r * bmax + len(per_rank[r][bi]): (r + 1) * bmax])
    if rank == 0:
        audio_s = float(lengths_all.sum()) / 25.0 * args.steps
        print(json.dumps({"metric": "stub", "value": round(audio_s / elapsed, 2), "unit": "stub-audio-sec/wall-sec",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "synthetic",
                          "config": {"workload": "stub mixed-length step on CPU (gloo)", "buckets_rank0": len(buckets),
                                     "clips_total": int(len(lengths_all))},
                          "roofline": None, "cpu_baseline": None}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def synth_inputs(B, T, seed=1234, with_u8=False):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (B, T, 96, 96), generator=g, dtype=torch.uint8)
    x = (u8[:, :, 4:92, 4:92].float() / 255.0 - 0.421) / 0.165          # hubert_dataset.py:242-245
    spk = torch.rand(B, 256, generator=g).relu()
    spk = spk / spk.norm(dim=-1, keepdim=True)
    if with_u8:
        return x.unsqueeze(1).contiguous(), spk, u8
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


def _median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def cpu_baseline(sd, vsd, clips, spk, gpu_out=None, warm=3, timed=10, enc_layers=24, conf_layers=12, margin_eps=2e-2,
                 frontend_only=False):
    """SURVEY 8(d)'s CPU protocol on the oracle (CPU restatement, kind 'port'): batch 1 (the reference's own batch size,
    inference.py:161), `warm` un-timed clips, then `timed` clips timed per stage - frontend only / stage 1 end to end
    incl. greedy decode (frontend inside) / vocoder - and the MEDIAN per stage reported.  `clips` is a list of
    [1,1,T,88,88] tensors (the first warm + timed are used; fewer clips shrink `warm` first).  With `gpu_out` the timed
    clips double as the full-size parity sample of the GPU step (same clips, batched with the others on the GPU, against
    the oracle run on each clip alone): a list with one dict per clip (tokens [>=2T], mel [>=4T,80], wav [>=640T], optional
    logits [>=2T,V]).  frontend_only=True times the frontend alone (BASELINE configs[1])."""
    from oracle import frontend as ofe
    from oracle import stage1 as os1
    from oracle import vocoder as ov
    n = min(len(clips), warm + timed)
    warm = max(0, n - timed)
    fsd = {k[len("encoder.w2v_model.feature_extractor_video.resnet."):]: v for k, v in sd.items()
           if k.startswith("encoder.w2v_model.feature_extractor_video.resnet.")}
    vr = None if frontend_only else vsd_removed(vsd)
    t_fe, t_s1, t_voc, audio = [], [], [], []
    n_safe = n_ok = n_flip = n_frames = 0
    mel_err = wav_err = logit_err = 0.0
    t_all = time.perf_counter()
    with torch.no_grad():
        for i in range(n):
            video = clips[i]
            T = video.shape[2]
            t0 = time.perf_counter()
            ofe.res_encoder(fsd, video)
            t1 = time.perf_counter()
            if frontend_only:
                if i >= warm:
                    t_fe.append(t1 - t0)
                    audio.append(T / 25.0)
                continue
            r = os1.generate(sd, video, torch.zeros(1, T, dtype=torch.bool), spk[i:i + 1], enc_layers=enc_layers,
                             conf_layers=conf_layers)
            t2 = time.perf_counter()
            code = (r["tokens"][0][:-1] - 4).clamp(min=0).unsqueeze(0)
            wav = ov.mel_code_generator(vr, VOC_H, code, r["mels"][0].t().unsqueeze(0), spk[i:i + 1])
            ov.to_int16(wav)
            t3 = time.perf_counter()
            if i < warm:
                continue
            t_fe.append(t1 - t0)
            t_s1.append(t2 - t1)
            t_voc.append(t3 - t2)
            audio.append(T / 25.0)
            if gpu_out is None:
                continue
            # parity of the GPU step (same clip, batched with the others) against the oracle run alone
            lr = r["logits"][:, 0, 4:]
            top2 = lr.topk(2, -1).values
            safe = (top2[:, 0] - top2[:, 1]) > margin_eps
            g = gpu_out[i]
            gt = g["tokens"][: 2 * T].long()
            same = gt == r["tokens"][0][: 2 * T]
            n_frames += 2 * T
            n_safe += int(safe.sum())
            n_ok += int(same[safe].sum())
            n_flip += int((~same).sum())
            if "logits" in g:
                logit_err = max(logit_err, float((g["logits"][: 2 * T, 4:] - lr).abs().max()))
            mel_err = max(mel_err, float((g["mel"][: 4 * T] - r["mels"][0]).abs().max()))
            if bool(same.all()):
                wav_err = max(wav_err, float((g["wav"][: 640 * T] - wav[0, 0]).abs().max()))
    wall = time.perf_counter() - t_all
    span = "4-s" if len(set(audio)) == 1 and audio[0] == 4.0 else "%.1f-%.1f s" % (min(audio), max(audio))
    if frontend_only:
        return {"value": round(_median([a / x for a, x in zip(audio, t_fe)]), 4), "unit": "audio-sec/wall-sec",
                "cores": torch.get_num_threads(), "kind": "port",
                "sample": "batch 1, %d warm-up + %d timed clips (%s), oracle fp32 ResNet-18 frontend alone on the host CPU; "
                          "median per clip; %.1f s wall" % (warm, len(audio), span, wall),
                "median_s_per_clip": {"frontend_s": round(_median(t_fe), 4)},
                "protocol": "SURVEY 8(d) (B=1, 3 warm-up + 10 timed clips, median); this run: %d warm-up + %d timed" % (warm, len(audio)),
                "nproc": os.cpu_count()}, None
    med = {"frontend_s": _median(t_fe), "stage1_s": _median(t_s1), "vocoder_s": _median(t_voc)}
    per_clip = [a / (x + y) for a, x, y in zip(audio, t_s1, t_voc)]
    rtf = _median(per_clip)
    parity = None
    if gpu_out is not None:
        parity = {"clips": len(audio), "unit_frames": n_frames, "unit_ids_differ_all_frames": n_flip,
                  "near_tie_margin": margin_eps, "unit_ids_compared": n_safe, "unit_ids_equal": n_ok,
                  "logit_max_abs_err": round(logit_err, 5), "mel_max_abs_err": round(mel_err, 5),
                  "wav_max_abs_err": round(wav_err, 5)}
    info = {"value": round(rtf, 4), "unit": "audio-sec/wall-sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "batch 1, %d warm-up + %d timed clips (%s), oracle fp32 on the host CPU; median per clip of audio / "
                      "(stage 1 + vocoder); %.1f s wall" % (warm, len(audio), span, wall),
            "median_s_per_clip": {k: round(v, 4) for k, v in med.items()},
            "stage_split": "frontend_s = ResNet-18 frontend alone; stage1_s = frontend + AV-HuBERT + conformer + heads + greedy "
                           "decode; vocoder_s = unit/mel/speaker -> int16 PCM",
            "protocol": "SURVEY 8(d) (B=1, 3 warm-up + 10 timed clips, median); this run: %d warm-up + %d timed" % (warm, len(audio)),
            "nproc": os.cpu_count()}
    return info, parity


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


def mixed_buckets(clips, bucket, rank, world, bucket_frames=0):
    """BASELINE configs[4] workload: clips * world clip lengths of 25..250 frames (1-10 s, seed 1234), dealt to ranks by
    sorted length; this rank's clips in descending length, cut into buckets.  bucket (clips per bucket) set: fixed-size buckets.
    Otherwise buckets of at most bucket_frames PADDED frames (clips x longest clip): short clips share a launch with many more
    of their kind, so every bucket's GEMMs see about the same number of rows (64 one-second clips are 1 600 rows: a 256-row-tile
    GEMM with N = 1024 then has 28 tiles for 256 CUs).  The cut is made on the GLOBAL length list in whole serpentine rounds (one
    clip per rank), so every rank holds the same number of buckets of the same clip counts - the per-bucket all_gather needs that.
    Returns (all lengths, this rank's lengths, list of buckets)."""
    import numpy as np
    rng = np.random.default_rng(1234)
    lengths_all = rng.integers(25, 251, size=clips * world)
    mine = l2s_dist.shard_by_length(lengths_all.tolist(), world, rank)
    my_lens = sorted((int(lengths_all[i]) for i in mine), reverse=True)
    if bucket:
        return lengths_all, my_lens, [my_lens[i:i + bucket] for i in range(0, len(my_lens), bucket)]
    glob = sorted((int(x) for x in lengths_all), reverse=True)
    sizes, tops, cur, top = [], [], 0, 0
    for r in range(clips):                      # round r = global clips r*world .. r*world + world - 1, one per rank
        if cur and (cur + 1) * top > bucket_frames:
            sizes.append(cur); tops.append(top); cur = 0
        if cur == 0:
            top = glob[r * world]
        cur += 1
    sizes.append(cur); tops.append(top)
    if len(sizes) > 1 and sizes[-1] * tops[-1] * 4 < bucket_frames:     # a sliver of a last bucket rides with the one before
        sliver = sizes.pop()
        sizes[-1] += sliver
    buckets, i = [], 0
    for n in sizes:
        buckets.append(my_lens[i:i + n]); i += n
    return lengths_all, my_lens, buckets


def bench_mixed(args, pipe, rank, world, dev, sd=None, vsd=None):
    """BASELINE configs[4]: mixed 1-10 s clips.  The global clip list is dealt to ranks by sorted length
    (distributed.shard_by_length), every rank pads its clips into length buckets and replays one hipGraph per bucket;
    a step = all buckets of the rank once + one padded all_gather of the unit ids per bucket."""
    lengths_all, my_lens, buckets = mixed_buckets(args.clips, args.bucket, rank, world, args.bucket_frames)
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

    # --mixed-streams S > 1 (default 8: every bucket on a stream of its own): the buckets' hipGraphs are dealt to S HIP streams
    # (serpentine over the length-sorted list: equal frames per stream) and replayed side by side - the short clips' buckets are
    # launches of a few thousand rows that leave most CUs idle on their own.  Measured in one call: 268.0 ms per step on one stream,
    # 248.6 on four, 245.3 on eight (two: no gain).  The roofline pass below is one eager pass on ONE stream.
    ns = max(1, min(args.mixed_streams, len(work))) if not args.no_graph else 1
    lanes = [torch.cuda.Stream() for _ in range(ns)] if ns > 1 else []
    for i, w in enumerate(work):
        rnd, slot = divmod(i, ns)
        w["lane"] = slot if rnd % 2 == 0 else ns - 1 - slot

    def run_step():
        if ns > 1:
            main = torch.cuda.current_stream()
            for st in lanes:
                st.wait_stream(main)
            for w in work:
                with torch.cuda.stream(lanes[w["lane"]]):
                    w["graph"].replay()
            for st in lanes:
                main.wait_stream(st)
            if world > 1:
                for w in work:
                    l2s_dist.gather_padded(w["out"]["tokens"], w["out"]["lens"] * 2)
            return
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
    own = time.perf_counter() - t0
    l2s_dist.barrier()
    torch.cuda.synchronize()
    per_rank = [round(1e3 * x / args.steps, 3) for x in l2s_dist.all_ranks(own, dev)]
    elapsed = l2s_dist.max_over_ranks(time.perf_counter() - t0, dev)
    audio_s = float(lengths_all.sum()) / 25.0 * args.steps
    padded = sum(max(w["lens"]) * len(w["lens"]) for w in work)
    roofline, top, cpu, parity = None, [], None, None
    if rank == 0:
        # dominant kernel over one eager pass of every bucket (per-launch HIP events on the launch stream).  Buckets have
        # different shapes, so per-launch figures are averages over the buckets' launches; no PMC traffic at these shapes
        prof = ops.KernelProfiler()
        ops.set_profiler(prof)
        for w in work:
            pipe.forward_device(w["video"], w["pad"], w["spk"])
        ops.set_profiler(None)
        roofline, top = dominant_roofline(prof.summary(), passes=1)
        roofline["note"] = "launch shapes differ per length bucket: per-launch figures are means over the buckets"
    if rank == 0 and world == 1 and not args.no_cpu_baseline and sd is not None:
        # CPU baseline + parity on a bounded sample of THIS workload: warm + timed clips spread evenly over the rank's
        # length-sorted clip list, each run ALONE through the oracle and compared with its row of the batched GPU bucket
        torch.cuda.synchronize()
        flat = [(bi, j) for bi, w in enumerate(work) for j in range(len(w["lens"]))]
        n = min(len(flat), args.cpu_warm + args.cpu_clips)
        picks = [flat[round(i * (len(flat) - 1) / max(n - 1, 1))] for i in range(n)]
        picks = picks[-args.cpu_warm:] + picks[:-args.cpu_warm] if n > args.cpu_clips else picks  # shortest clips warm up
        clips, spks, gouts = [], [], []
        for bi, j in picks:
            w = work[bi]
            nfr = w["lens"][j]
            clips.append(w["video"][j:j + 1, :, :nfr].float().cpu())
            spks.append(w["spk"][j].float().cpu())
            gouts.append({"tokens": w["out"]["tokens"][j].cpu(), "mel": w["out"]["mel"][j].float().cpu(),
                          "wav": w["out"]["wav"][j].float().cpu(), "logits": w["out"]["logits"][j].float().cpu()})
        cpu, parity = cpu_baseline({k: v.float() for k, v in sd.items()}, vsd, clips, torch.stack(spks), gouts, args.cpu_warm,
                                   args.cpu_clips, args.enc_layers, args.conf_layers)
    if rank == 0:
        print(json.dumps({
            "metric": "real-time factor (audio-sec/wall-sec), end-to-end lip->16kHz audio, mixed 1-10 s clips",
            "value": round(audio_s / elapsed, 2), "unit": "audio-sec/wall-sec",
            "clips_per_sec": round(args.clips * world * args.steps / elapsed, 2), "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "per_rank_ms_per_step": per_rank,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "fp16" if args.dtype == "f16" else "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: mixed 1-10 s clips (25..250 frames, seed 1234), %d clips per GPU, "
                                   "dealt by sorted length, %s" % (args.clips, ("buckets of %d clips" % args.bucket) if args.bucket else
                                                                  ("%d buckets of <= %d padded frames" % (len(buckets), args.bucket_frames))),
                       "clips_per_gpu": args.clips, "bucket": args.bucket, "bucket_frames": None if args.bucket else args.bucket_frames,
                       "bucket_shapes_rank0": [[len(b), max(b)] for b in buckets], "hipgraph": not args.no_graph, "streams": ns,
                       "padding_overhead_rank0": round(padded / float(sum(my_lens)), 4),
                       "parallelism": f"clip-parallel dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu, "parity_vs_oracle": parity, "top_kernels": top}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def bench_frontend(args, model, sd, rank, world, dev):
    """BASELINE configs[1]: the ResNet-18 3D/2D lip frontend kernels only (avhubert/resnet.py:131-169) on synthetic 25-fps
    88x88x1 batches, fp16 frames resident in HBM.  A step = stem+pool, 16 trunk convs (+ 3 downsample convs), avg-pool over
    one batch of clips.  SURVEY 8(d): report GB/s (compulsory bytes: 88*88*2 in + 512*2 out per frame) AND TFLOP/s (0.6323
    GFLOP per frame) and name the binding roof."""
    B, T = args.batch, args.frames
    res = model.encoder.w2v_model.feature_extractor_video.resnet
    t16 = ops.torch_dtype(res.dtype)
    video_cpu, spk_cpu = synth_inputs(B, T, seed=1234 + rank)
    x16 = video_cpu[:, 0].to(dev).to(t16).contiguous()           # [B,T,88,88] normalised frames, 16-bit, resident in HBM

    def step():
        return res.forward_rows(x16)[0]

    for _ in range(max(args.warmup, 1)):
        feat = step()
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
            feat = step()
        graph.replay()
        torch.cuda.synchronize()

    def step_core():
        if graph is not None:
            graph.replay()
        else:
            step()

    step_core()
    elapsed = timed_region(step_core, args.steps, torch.cuda.synchronize, dev)
    frames_total = world * B * T * args.steps
    ms_per_step = 1e3 * elapsed / args.steps
    gflop_frame, bytes_frame = 0.6323, 88 * 88 * 2 + 512 * 2
    tflops = frames_total * gflop_frame / elapsed / 1e3
    gbs = frames_total * bytes_frame / elapsed / 1e9
    ai = gflop_frame * 1e9 / bytes_frame
    ridge = PEAK_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
    roofline = top = cpu = None
    if rank == 0:
        prof = ops.KernelProfiler()
        ops.set_profiler(prof)
        for _ in range(2):
            step()
        ops.set_profiler(None)
        dom, top = dominant_roofline(prof.summary(), passes=2, clips_per_launch=B, frames=T)
        # stage-level roofline (the whole frontend as one unit of work) with BOTH roofs; the dominant kernel's own line rides along
        roofline = {"scope": "stage: whole frontend per step", "bound": "mfma" if ai >= ridge else "hbm",
                    "achieved": round(tflops / world, 2), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tflops / world / PEAK_MFMA_TFLOPS, 4), "traffic": None,
                    "hbm": {"achieved": round(gbs / world, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": round(gbs / world / PEAK_HBM_GBS, 5),
                            "algorithmic_bytes_per_frame": bytes_frame},
                    "algorithmic_gflop_per_frame": gflop_frame, "arithmetic_intensity_flop_per_byte": round(ai, 1),
                    "ridge_flop_per_byte": round(ridge, 1),
                    "binding": "MFMA: %.0f FLOP per compulsory byte is %.0fx the ridge of the two peaks, so the HBM figure can "
                               "only ever be a small fraction of 8 TB/s - it is reported because BASELINE configs[1] asks for it" % (
                                   ai, ai / ridge),
                    "dominant_kernel": dom}
        if world == 1 and not args.no_cpu_baseline:
            n_cpu = min(B, args.cpu_warm + args.cpu_clips)
            cpu, _ = cpu_baseline({k: v.float() for k, v in sd.items()}, None, [video_cpu[i:i + 1] for i in range(n_cpu)], spk_cpu,
                                  None, args.cpu_warm, args.cpu_clips, frontend_only=True)
            # parity of the timed step's own output against the oracle frontend on the first clip
            from oracle import frontend as ofe
            fsd = {k[len("encoder.w2v_model.feature_extractor_video.resnet."):]: v.float() for k, v in sd.items()
                   if k.startswith("encoder.w2v_model.feature_extractor_video.resnet.")}
            with torch.no_grad():
                ref = ofe.res_encoder(fsd, x16[:1].float().cpu().unsqueeze(1))[0].t()      # [T,512]
            err = float((feat[:T].float().cpu() - ref).abs().max() / ref.abs().max())
            cpu["parity_rel_max_err_clip0"] = round(err, 5)
        print(json.dumps({
            "metric": "real-time factor (audio-sec/wall-sec), ResNet-18 3D/2D lip frontend only, 4s@25fps clips",
            "value": round(world * B * (T / 25.0) * args.steps / elapsed, 2), "unit": "audio-sec/wall-sec",
            "frames_per_sec": round(frames_total / elapsed, 1), "clips_per_sec": round(world * B * args.steps / elapsed, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "per_rank_ms_per_step": PER_RANK["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp16" if res.dtype == ops.F16 else "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: ResNet-18 3D/2D lip frontend HIP kernels only, synthetic 25 fps 88x88x1 "
                                   "batches, %d clips x %d frames per GPU, 16-bit frames resident in HBM" % (B, T),
                       "clips_per_gpu": B, "frames_per_clip": T, "hipgraph": graph is not None,
                       "parallelism": f"clip-parallel dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu, "top_kernels": top}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def bench_latency(args, pipe, dev):
    """The reference's own operating point: ONE clip per forward (multi_target_lip2speech/inference.py:161 forces batch_size = 1;
    the micro-servers answer one request at a time, inference_server.py:250).  Per clip length (4-s = --frames, and 10 s = 250
    frames) the request path is replayed --requests times: pinned-host uint8 frames + speaker embedding -> device, the hipGraph
    of the whole lip -> units -> mel -> int16 PCM path, PCM + unit ids -> pinned host, synchronise.  Reports p50 / p95 / mean
    wall ms per request (transfers included), the graph's device time alone (HIP events around the replay), the number of kernel
    launches inside the graph, and the sum of the kernels' own times (per-launch HIP events over eager passes: an upper bound,
    each event pair adds ~2 us) - wall minus that sum is what launch gaps / dependencies cost at batch 1."""
    B = args.batch
    lines = []
    for T in sorted({args.frames, 250}):
        _, spk_cpu, u8_cpu = synth_inputs(B, T, seed=777 + T, with_u8=True)
        host_in, host_spk = u8_cpu.pin_memory(), spk_cpu.pin_memory()
        frames_dev, spk = torch.empty_like(u8_cpu, device=dev), torch.empty_like(spk_cpu, device=dev)
        frames_dev.copy_(host_in)
        spk.copy_(host_spk)

        def step():
            return pipe.forward_device_u8(frames_dev, None, spk)

        for _ in range(3):
            out = step()
        torch.cuda.synchronize()
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
        host_pcm = torch.empty(out["pcm"].shape, dtype=out["pcm"].dtype).pin_memory()
        host_tok = torch.empty(out["tokens"].shape, dtype=out["tokens"].dtype).pin_memory()

        def request():
            frames_dev.copy_(host_in, non_blocking=True)
            spk.copy_(host_spk, non_blocking=True)
            graph.replay()
            host_pcm.copy_(out["pcm"], non_blocking=True)
            host_tok.copy_(out["tokens"], non_blocking=True)
            torch.cuda.synchronize()

        for _ in range(max(args.warmup, 5)):
            request()
        wall = []
        for _ in range(args.requests):
            t0 = time.perf_counter()
            request()
            wall.append(1e3 * (time.perf_counter() - t0))
        wall.sort()
        dev_ms = []
        for _ in range(50):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            dev_ms.append(e0.elapsed_time(e1))
        dev_ms.sort()
        prof = ops.KernelProfiler()
        ops.set_profiler(prof)
        passes = 5
        for _ in range(passes):
            step()
        ops.set_profiler(None)
        agg = prof.summary()
        launches = sum(a["calls"] for a in agg.values()) // passes
        ksum = sum(a["ms"] for a in agg.values()) / passes
        roofline, top = dominant_roofline(agg, passes=passes)
        pct = lambda q: wall[min(len(wall) - 1, int(q * len(wall)))]
        audio_s = T / 25.0
        lines.append({
            "clip_seconds": audio_s, "frames": T, "clips_per_request": B,
            "wall_ms": {"p50": round(pct(0.50), 3), "p95": round(pct(0.95), 3), "mean": round(sum(wall) / len(wall), 3),
                        "min": round(wall[0], 3), "requests": len(wall)},
            "rtf_p50": round(B * audio_s / (pct(0.50) * 1e-3), 1),
            "graph_device_ms_p50": round(dev_ms[len(dev_ms) // 2], 3),
            "kernel_launches_in_graph": launches, "sum_of_kernel_ms": round(ksum, 3),
            "launch_bound_gap_ms": round(dev_ms[len(dev_ms) // 2] - ksum, 3),
            "h2d_bytes": u8_cpu.numel() + spk_cpu.numel() * 4, "d2h_bytes": out["pcm"].numel() * 2 + out["tokens"].numel() * 4,
            "top_kernels": top[:5], "dominant_kernel": roofline})
    first = lines[0]
    print(json.dumps({
        "metric": "latency per request, batch %d, end-to-end lip->16kHz int16 PCM incl. PCIe (wall ms, p50)" % B,
        "value": first["wall_ms"]["p50"], "unit": "ms", "n_gpus": 1, "steps": args.requests, "warmup": max(args.warmup, 5),
        "ms_per_step": first["wall_ms"]["mean"], "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp16" if args.dtype == "f16" else "bf16", "data": "synthetic",
        "config": {"workload": "the reference's operating point: ONE request at a time (multi_target_lip2speech/inference.py:161, "
                               "inference_server.py:250), batch %d, hipGraph replay, uint8 frames in -> int16 PCM + unit ids out, "
                               "transfers inside the timed region" % B,
                   "clips_per_request": B, "hipgraph": True, "enc_layers": args.enc_layers, "conf_layers": args.conf_layers},
        "roofline": first["dominant_kernel"], "cpu_baseline": None, "latency": lines}), flush=True)


def dominant_roofline(agg, passes, launches_scale=1, clips_per_launch=None, frames=100):
    """`roofline` of the dominant kernel (largest share of the profiled time among kernels with algorithmic FLOPs) + the
    top-8 table, from an ops.KernelProfiler summary of `passes` eager passes (per-launch HIP events on the launch stream).
    `traffic` (HBM bytes per launch) is NOT measured in the run: it is looked up in the committed rocprofv3 PMC passes
    (tools/collect_traffic.sh: separate FETCH_SIZE / WRITE_SIZE runs, gfx950 FETCH_SIZE x2 correction) and only when that
    file was collected at this launch shape and carries this kernel key; the source is named in the line."""
    tot_ms = sum(a["ms"] for a in agg.values())
    ranked = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
    top = []
    for k, a in ranked[:8]:
        top.append({"kernel": k, "calls_per_step": a["calls"] // passes * launches_scale,
                    "ms_per_step": round(a["ms"] / passes * launches_scale, 3), "share": round(a["ms"] / tot_ms, 3),
                    "tflops": round(a["flops"] / a["ms"] / 1e9, 1) if a["flops"] else None})
    dom_k, dom = next(((k, a) for k, a in ranked if a["flops"] > 0), ranked[0])
    secs = dom["ms"] * 1e-3
    # Which roof: SURVEY 8(d) assigns the dense contractions (every tap-GEMM / phase-GEMM instantiation, the fused conv kernels)
    # to the MFMA roof - a kernel with algorithmic FLOPs on the matrix pipe is priced against 2.5 PFLOP/s whatever the pooled
    # intensity of the launches that share its instantiation (round 3 let that intensity pick the roof: when the short-K residual
    # GEMMs joined the dominant key it slipped under the ridge and the line flipped to "hbm" without any kernel changing).
    # The HBM-side figure is always printed beside it, and `by_shape` prices each problem shape of the instantiation on its own.
    ai = dom["flops"] / max(dom["bytes"], 1.0)
    ridge = PEAK_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
    on_mfma = dom["flops"] > 0
    tf, gbs = dom["flops"] / secs / 1e12, dom["bytes"] / secs / 1e9
    if on_mfma:
        bound, ach, peak, unit = "mfma", tf, PEAK_MFMA_TFLOPS, "TFLOP/s"
    else:
        bound, ach, peak, unit = "hbm", gbs, PEAK_HBM_GBS, "GB/s"
    traffic = traffic_source = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if clips_per_launch is not None and os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("batch") == clips_per_launch and tj.get("frames", 100) == frames and dom_k in tj["kernels"]:
            traffic = tj["kernels"][dom_k]["hbm_bytes_per_launch"]
            traffic_source = "profiles/traffic_latest.json" + (("@" + tj["commit"]) if tj.get("commit") else "")
    # matrix-pipe busy cycles of the same kernel from the PMC pass of tools/collect_mfma_util.sh (profiles/mfma_util_latest.json):
    # the FLOPs the hardware issued per launch and the fraction of shader cycles its matrix pipes were busy
    mfma_pmc = None
    mpath = os.path.join(ROOT, "profiles", "mfma_util_latest.json")
    if clips_per_launch is not None and frames == 100 and os.path.exists(mpath):
        mj = json.load(open(mpath))
        if mj.get("clips") == clips_per_launch and dom_k in mj["kernels"]:
            mk = mj["kernels"][dom_k]
            mfma_pmc = {"busy_frac_of_cycles": mk["mfma_util"], "issued_tflop_per_launch": mk["mfma_tflop_per_launch"],
                        "source": "profiles/mfma_util_latest.json" + (("@" + mj["commit"]) if mj.get("commit") else "")}
    by_shape = []
    for shp, a in sorted(dom.get("shapes", {}).items(), key=lambda kv: -kv[1]["ms"]):
        if shp is None or not a["flops"]:
            continue
        ssec = a["ms"] * 1e-3
        by_shape.append({"shape": shp, "calls_per_step": a["calls"] // passes * launches_scale,
                         "avg_launch_us": round(1e3 * a["ms"] / a["calls"], 2),
                         "tflops": round(a["flops"] / ssec / 1e12, 1), "frac_mfma": round(a["flops"] / ssec / 1e12 / PEAK_MFMA_TFLOPS, 4),
                         "alg_gbs": round(a["bytes"] / ssec / 1e9, 1), "frac_hbm": round(a["bytes"] / ssec / 1e9 / PEAK_HBM_GBS, 4),
                         "flop_per_byte": round(a["flops"] / max(a["bytes"], 1.0), 1)})
    roofline = {"kernel": dom_k, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_source, "mfma_pmc": mfma_pmc,
                "arithmetic_intensity_flop_per_byte": round(ai, 1), "ridge_flop_per_byte": round(ridge, 1),
                "hbm_side": {"achieved": round(gbs, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)},
                "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["calls"]),
                "avg_launch_us": round(1e3 * dom["ms"] / dom["calls"], 2),
                "flop_per_launch": round(dom["flops"] / dom["calls"]), "share_of_step": round(dom["ms"] / tot_ms, 3),
                "by_shape": by_shape}
    return roofline, top


def transfer_inclusive(args, step_core, frames_dev, out, u8_host, steps, dev):
    """The same K steps with the PCIe legs inside the timed region: the next batch's uint8 frames go pinned host -> device
    staging on a side stream while the current batch computes (double-buffered), a device-to-device copy hands them to the
    step's (graph-captured) input, and the int16 PCM + unit ids return to pinned host memory on a second side stream."""
    h2d, d2h = torch.cuda.Stream(), torch.cuda.Stream()
    main = torch.cuda.current_stream()
    # two DIFFERENT batches alternate through the staging buffers (the second = the first with its clips rotated by one), so a
    # stale or mis-ordered staging copy, or a wrong event dependency, changes the result checked after the region
    host_in = [u8_host.pin_memory(), torch.roll(u8_host, 1, 0).contiguous().pin_memory()]
    stage_in = [torch.empty_like(frames_dev) for _ in range(2)]
    stage_pcm = [torch.empty_like(out["pcm"]) for _ in range(2)]
    stage_tok = [torch.empty_like(out["tokens"]) for _ in range(2)]
    host_pcm = [torch.empty(out["pcm"].shape, dtype=out["pcm"].dtype).pin_memory() for _ in range(2)]
    host_tok = [torch.empty(out["tokens"].shape, dtype=out["tokens"].dtype).pin_memory() for _ in range(2)]
    ev_in = [torch.cuda.Event() for _ in range(2)]        # staging i filled
    ev_free = [torch.cuda.Event() for _ in range(2)]      # staging i consumed by the step
    ev_out = [torch.cuda.Event() for _ in range(2)]       # output staging i written
    ev_back = [torch.cuda.Event() for _ in range(2)]      # output staging i copied to the host
    k = {"n": 0}

    def prefetch(i):
        with torch.cuda.stream(h2d):
            h2d.wait_event(ev_free[i])
            stage_in[i].copy_(host_in[i], non_blocking=True)
            ev_in[i].record(h2d)

    for i in range(2):
        ev_free[i].record(main)
        ev_back[i].record(main)
    prefetch(0)

    def run_step():
        i = k["n"] & 1
        k["n"] += 1
        prefetch(i ^ 1)                                    # next batch travels while this one computes
        main.wait_event(ev_in[i])
        frames_dev.copy_(stage_in[i], non_blocking=True)
        ev_free[i].record(main)
        step_core()
        main.wait_event(ev_back[i])
        stage_pcm[i].copy_(out["pcm"], non_blocking=True)
        stage_tok[i].copy_(out["tokens"], non_blocking=True)
        ev_out[i].record(main)
        with torch.cuda.stream(d2h):
            d2h.wait_event(ev_out[i])
            host_pcm[i].copy_(stage_pcm[i], non_blocking=True)
            host_tok[i].copy_(stage_tok[i], non_blocking=True)
            ev_back[i].record(d2h)

    # what each of the two batches must produce (computed one at a time, outside the region)
    want = []
    for i in range(2):
        frames_dev.copy_(host_in[i], non_blocking=True)
        step_core()
        torch.cuda.synchronize()
        want.append((out["tokens"].cpu().clone(), out["pcm"].cpu().clone()))
    # (on a tiny random model the outputs can coincide - its units barely depend on the frames; the line says which case it was)
    distinct = not (torch.equal(want[0][0], want[1][0]) and torch.equal(want[0][1], want[1][1]))
    run_step()
    elapsed = timed_region(run_step, steps, torch.cuda.synchronize, dev)
    torch.cuda.synchronize()
    for back in (1, 2):   # the last two steps went through different staging buffers: each carried ITS batch, end to end
        i = (k["n"] - back) & 1
        assert torch.equal(host_tok[i], want[i][0]) and torch.equal(host_pcm[i], want[i][1]), f"staging buffer {i} carried the wrong batch"
    frames_dev.copy_(host_in[0], non_blocking=True)     # leave the caller's batch and its outputs in place
    step_core()
    torch.cuda.synchronize()
    return elapsed, u8_host.numel(), out["pcm"].numel() * 2 + out["tokens"].numel() * 4, distinct


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=640,
                    help="clips per GPU per step (640 x 100 frames = 64000 rows = 250 row tiles of 256: the per-launch tails "
                         "and the weight traffic amortise over 4x the work of the 160-clip step of round 1: -3.8 %% per clip; "
                         "32 = BASELINE configs[2] batch)")
    ap.add_argument("--streams", type=int, default=1,
                    help="run the step's batch as this many independent sub-batches on their own HIP streams inside the one "
                         "hipGraph (LipToSpeechPipeline.forward_device_u8_streams; 2 x 320 clips: another -3 %% per clip). "
                         "Not the default: kernels of the two streams then wait for each other's CUs inside their measured "
                         "durations, so neither HIP events nor rocprofv3 give a clean per-kernel time for the roofline")
    ap.add_argument("--frames", type=int, default=100, help="video frames per clip (100 = 4 s @ 25 fps)")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-transfers", action="store_true", help="skip the second timed region with the PCIe legs inside")
    ap.add_argument("--cpu-clips", type=int, default=10, help="timed clips of the CPU baseline / parity sample (SURVEY 8d: 10)")
    ap.add_argument("--cpu-warm", type=int, default=3, help="un-timed warm-up clips of the CPU baseline (SURVEY 8d: 3)")
    ap.add_argument("--stage", default="e2e", choices=["e2e", "frontend"],
                    help="frontend = BASELINE configs[1]: the ResNet-18 3D/2D lip frontend kernels only, fp16 88x88 frames "
                         "resident in HBM; reports GB/s and TFLOP/s against both roofs and names the binding one")
    ap.add_argument("--fp32-input", action="store_true",
                    help="feed CPU-normalised fp32 88x88 frames (the reference's collater output) instead of uint8 96x96 frames "
                         "with the crop/normalise kernel inside the step")
    ap.add_argument("--u8", action="store_true", help="(default since round 2; accepted for compatibility)")
    ap.add_argument("--mixed", action="store_true",
                    help="BASELINE configs[4]: --clips clips per GPU of 1-10 s (25..250 frames, seed 1234), dealt to ranks by "
                         "sorted length and run as length buckets of --bucket clips (one hipGraph per bucket shape)")
    # 512 clips x 5.5 s on average = 2 816 audio-seconds per step, the volume of the default line's 640 x 4 s; sweep on one box
    # (DESIGN.md section 5): 256 / 32 RTF 8 494, 512 / 64 9 402, 1 024 / 128 9 875 (padding 9.5 / 9.8 / 10.4 %)
    ap.add_argument("--clips", type=int, default=512)
    ap.add_argument("--mixed-streams", type=int, default=8,
                    help="--mixed: HIP streams the length buckets' hipGraphs are replayed on side by side (1 = one after the other)")
    ap.add_argument("--bucket", type=int, default=64, help="--mixed: buckets of this many clips (0: use --bucket-frames)")
    ap.add_argument("--bucket-frames", type=int, default=16000,
                    help="--mixed with --bucket 0: buckets of at most this many PADDED frames (clips x longest clip of the bucket).  "
                         "Measured SLOWER than 64-clip buckets (DESIGN.md section 5): 8 000 / 12 000 / 16 000 / 20 000 frames 279 / 281 / "
                         "281 / 297 ms against 270 ms - the padding of the merged short clips costs more than their small launches")
    ap.add_argument("--enc-layers", type=int, default=24)
    ap.add_argument("--conf-layers", type=int, default=12)
    ap.add_argument("--latency", action="store_true",
                    help="per-request latency at --batch clips per request (use --batch 1: the reference decodes one clip per forward): "
                         "p50 / p95 wall ms incl. PCIe, graph device time, launches, sum of kernel times; 4-s and 10-s clips")
    ap.add_argument("--requests", type=int, default=200, help="timed requests of --latency")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)   # CPU rehearsal of launcher + harness (tests)
    args = ap.parse_args()

    # N > 1 and no rendezvous environment: become the launcher.  Nothing above or in this branch touches the GPU.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.stub:
        return bench_stub(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    rank, world, local = l2s_dist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dt = ops.F16 if args.dtype == "f16" else ops.BF16
    B, T = args.batch, args.frames

    model, voc, sd, vsd = build(dt, dev, args.enc_layers, args.conf_layers)
    pipe = LipToSpeechPipeline(model, voc)
    if args.latency:
        if world != 1:
            raise SystemExit("--latency is a single-GPU, single-request measurement")
        return bench_latency(args, pipe, dev)
    if args.mixed:
        return bench_mixed(args, pipe, rank, world, dev, sd, vsd)
    if args.stage == "frontend":
        return bench_frontend(args, model, sd, rank, world, dev)
    video_cpu, spk_cpu, u8_cpu = synth_inputs(B, T, seed=1234 + rank, with_u8=True)
    spk = spk_cpu.to(dev)
    use_u8 = not args.fp32_input
    frames_dev = u8_cpu.to(dev) if use_u8 else video_cpu.to(dev)

    nstreams = args.streams if use_u8 else 1
    if nstreams < 1 or B % nstreams:
        raise SystemExit(f"--batch {B} is not a multiple of --streams {nstreams}")

    def step():
        if use_u8:
            return pipe.forward_device_u8_streams(frames_dev, None, spk, nstreams)
        return pipe.forward_device(frames_dev, None, spk)

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

    def step_core():
        if graph is not None:
            graph.replay()
        else:
            step()
        if world > 1:  # batch collation: ONE padded all_gather of the unit ids per batch (RCCL over xGMI), static shapes
            l2s_dist.gather_padded(out["tokens"], out["lens"] * 2, static_shape=True)

    step_core()
    elapsed = timed_region(step_core, args.steps, torch.cuda.synchronize, dev)
    per_rank = PER_RANK["ms_per_step"]

    audio_s = world * B * (T / 25.0) * args.steps
    rtf = audio_s / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    xfer = None
    if use_u8 and not args.no_transfers:
        el2, h2d_bytes, d2h_bytes, distinct = transfer_inclusive(args, step_core, frames_dev, out, u8_cpu, args.steps, dev)
        xfer = {"value": round(audio_s / el2, 2), "unit": "audio-sec/wall-sec", "ms_per_step": round(1e3 * el2 / args.steps, 3),
                "vs_device_only": round(el2 / elapsed, 4), "h2d_bytes_per_step": h2d_bytes, "d2h_bytes_per_step": d2h_bytes,
                "staging_check": "both staging buffers carried their own batch end to end (unit ids + PCM of the last two steps); the "
                                 "two batches' results " + ("differ" if distinct else "coincide on this model: the check is degenerate"),
                "how": "uint8 frames pinned host -> device on a side stream, double-buffered under the previous step; int16 PCM + "
                       "unit ids -> pinned host on a second side stream"}

    # ---- roofline of the dominant kernel: per-launch HIP events on the launch stream, eager pass ----
    roofline = None
    top = []
    if rank == 0:
        prof = ops.KernelProfiler()
        ops.set_profiler(prof)
        bl = B // nstreams                                   # clips per launch
        for _ in range(2):
            if nstreams > 1:   # one sub-batch alone on the launch stream: the shapes of the step's launches, nothing beside them
                pipe.forward_device_u8(frames_dev[:bl], None, spk[:bl])
            else:
                step()
        ops.set_profiler(None)
        roofline, top = dominant_roofline(prof.summary(), passes=2, launches_scale=nstreams, clips_per_launch=B // nstreams,
                                          frames=T)

    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        torch.cuda.synchronize()
        n_cpu = min(B, args.cpu_warm + args.cpu_clips)
        host = {k: out[k][:n_cpu].float().cpu() if k != "tokens" else out[k][:n_cpu].cpu()
                for k in ("tokens", "mel", "wav", "logits") if k in out}
        gpu_out = [{k: v[i] for k, v in host.items()} for i in range(n_cpu)]
        cpu, parity = cpu_baseline({k: v.float() for k, v in sd.items()}, vsd, [video_cpu[i:i + 1] for i in range(n_cpu)],
                                   spk_cpu, gpu_out, args.cpu_warm, args.cpu_clips, args.enc_layers, args.conf_layers)

    if rank == 0:
        full = args.enc_layers == 24 and args.conf_layers == 12
        step_tflops = GFLOP_PER_4S_CLIP * (T / 100.0) * B * 1e9 / (ms_per_step * 1e-3) / 1e12 if full else None
        line = {
            "metric": "real-time factor (audio-sec/wall-sec), end-to-end lip->16kHz audio, 4s@25fps clips",
            "value": round(rtf, 2), "unit": "audio-sec/wall-sec", "clips_per_sec": round(world * B * args.steps / elapsed, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "per_rank_ms_per_step": per_rank,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp16" if dt == ops.F16 else "bf16", "data": "synthetic",
            "config": {"workload": "e2e lip->units->wav (BASELINE configs[3]: AV-HuBERT large 24L + conformer 12x512 + "
                                   "multi_input HiFi-GAN), 4-s 100-frame 88x88 clips, batch %d per GPU%s" % (
                                       B, "" if nstreams == 1 else " as %d independent sub-batches of %d clips on %d HIP "
                                       "streams inside one hipGraph" % (nstreams, B // nstreams, nstreams)),
                       "clips_per_gpu": B, "streams": nstreams, "clips_per_launch": B // nstreams,
                       "frames_per_clip": T, "hipgraph": graph is not None,
                       "input": "uint8 96x96 frames resident in HBM, crop+normalise on device" if use_u8
                                else "fp32 88x88 normalised frames resident in HBM",
                       "quality": "random-init weights: STOI is undefined; parity_vs_oracle (unit ids, mel / waveform max abs "
                                  "error against the fp32 CPU oracle) stands in for configs[3]'s STOI tolerance",
                       "enc_layers": args.enc_layers, "conf_layers": args.conf_layers,
                       "parallelism": f"clip-parallel dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu,
            "whole_step": {"tflops": round(step_tflops, 1), "frac_of_mfma_peak": round(step_tflops / PEAK_MFMA_TFLOPS, 4),
                           "gflop_per_clip": GFLOP_PER_4S_CLIP} if step_tflops else None,
            "transfer_inclusive": xfer, "parity_vs_oracle": parity, "top_kernels": top,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
