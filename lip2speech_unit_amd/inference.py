#!/usr/bin/env python3
"""Stage-1 CLI — mirrors multi_target_lip2speech/inference.py (`--config-name decode` hydra surface, :46-71,:97-317).

hydra/omegaconf/fairseq are not dependencies here; the same `key=value` overrides are parsed directly:
  python -m lip2speech_unit_amd.inference common_eval.path=<ckpt.pt> common_eval.results_path=<dir> \
      override.data=<label_dir> override.label_dir=<label_dir> [fp16=true] [dataset.gen_subset=test] \
      [generation.beam=50] [dataset.batch_size=N]
Reads <label_dir>/{test.tsv,test.unt,dict.unt.txt} + video/, mel/, spk_emb/ siblings; writes decode.log, pred_mel/,
pred_unit/, hypo-<fid>.json, wer.<fid> like :250-315.  Unlike the reference (batch_size forced to 1, :161) clips are
batched; results equal the one-clip-at-a-time results by construction (row masking).
Launch under torch.distributed.run for clip-parallel multi-GPU: one rank per GPU, clips dealt by sorted length, no
data-path collective; the per-clip records are gathered once at the end and rank 0 writes ONE hypo-<fid>.json / wer.<fid>
in dataset order.
"""
import hashlib
import json
import logging
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

from . import distributed as l2s_dist
from . import ops, weights
from .model_avhubert import MultiTargetAVHubertEncoderModel
from .task import Lip2SpeechConfig, Lip2SpeechTask

DEFAULTS = {  # conf/decode.yaml
    "common_eval.path": None, "common_eval.results_path": None, "override.data": None, "override.label_dir": None,
    "dataset.gen_subset": "test", "dataset.batch_size": 8, "generation.beam": 50, "generation.temperature": 1.0,
    "generation.lenpen": 1.0, "generation.nbest": 1, "fp16": False, "common.fp16": False, "dtype": "f16", "synthetic_weights": False,
    "common.user_dir": None, "model.encoder_layers": 24, "model.conformer_layers": 12, "model.check_resnet_checksum": True,
}


def parse_overrides(argv):
    cfg = dict(DEFAULTS)
    for a in argv:
        if a.startswith("--") or "=" not in a:
            continue  # --config-dir / --config-name decode are accepted and ignored
        k, v = a.split("=", 1)
        if v.lower() in ("true", "false"):
            v = v.lower() == "true"
        elif v.lower() in ("null", "none"):
            v = None
        else:
            try:
                v = int(v)
            except ValueError:
                try:
                    v = float(v)
                except ValueError:
                    pass
        cfg[k] = v
    return cfg


def edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def main(argv=None):
    cfg = parse_overrides(sys.argv[1:] if argv is None else argv)
    results_path = cfg["common_eval.results_path"]
    assert results_path, "common_eval.results_path is required"
    os.makedirs(results_path, exist_ok=True)
    rank, world, local = l2s_dist.init_from_env()
    logging.basicConfig(format="%(asctime)s | %(levelname)s | %(name)s | %(message)s", level=logging.INFO,
                        handlers=[logging.FileHandler(os.path.join(results_path, "decode.log")),
                                  logging.StreamHandler(sys.stdout)])
    logger = logging.getLogger("hybrid.speech_recognize")
    if not torch.cuda.is_available():
        raise SystemExit("this build runs on MI355X only: no CPU path")
    torch.cuda.set_device(local)

    tcfg = Lip2SpeechConfig(data=cfg["override.data"], label_dir=cfg["override.label_dir"], fp16=bool(cfg["fp16"]))
    task = Lip2SpeechTask(tcfg)
    dtype = ops.BF16 if cfg["dtype"] == "bf16" else ops.F16
    from .conformer import ConformerConfig
    from .hubert import AVHubertConfig
    model = MultiTargetAVHubertEncoderModel.build_model(
        task=task, dtype=dtype, w2v_cfg=AVHubertConfig(encoder_layers=int(cfg["model.encoder_layers"])),
        conformer_cfg=ConformerConfig(conformer_layers=int(cfg["model.conformer_layers"])))
    if cfg["synthetic_weights"]:
        model.load_state_dict(weights.synth_state_dict(weights.spec_of(model), seed=0))
    else:
        state = torch.load(cfg["common_eval.path"], map_location="cpu")
        # strict apart from the documented allow-lists + the reference's resnet known-answer check (model_avhubert.py:119-123)
        r = model.load_checkpoint_state(state["model"], check_resnet_sum=bool(cfg["model.check_resnet_checksum"]))
        logger.info(f"checkpoint loaded; tolerated missing={len(r.missing_keys)} unexpected={len(r.unexpected_keys)}")
    model.cuda().eval()
    ds = task.load_dataset(cfg["dataset.gen_subset"])
    gen_args = SimpleNamespace(beam=cfg["generation.beam"], temperature=cfg["generation.temperature"],
                               lenpen=cfg["generation.lenpen"], nbest=cfg["generation.nbest"])
    generator = task.build_generator([model], gen_args, extra_gen_cls_kwargs={})
    generator.results_path = results_path
    dictionary = task.target_dictionary
    ignore = {dictionary.pad(), dictionary.bos(), dictionary.unk(), dictionary.eos()}

    mine = l2s_dist.shard_by_length(ds.sizes, world, rank)
    bs = int(cfg["dataset.batch_size"])
    records = []          # (dataset index, utt_id, ref, hypo) of this rank's clips
    n_tok, t_gen = 0, 0.0
    for s in range(0, len(mine), bs):
        batch = ds.collater([ds[i] for i in mine[s:s + bs]])
        ni = batch["net_input"]
        ni["source"]["video"] = ni["source"]["video"].cuda()
        ni["padding_mask"], ni["spk_emb"] = ni["padding_mask"].cuda(), ni["spk_emb"].cuda()
        if batch["target"] is not None:
            batch["target"] = batch["target"].cuda()
        t0 = time.perf_counter()
        hypos, batch = task.inference_step(generator, [model], batch)
        torch.cuda.synchronize()
        t_gen += time.perf_counter() - t0
        for i, utt in enumerate(batch["utt_id"]):
            n = int(batch["target_lengths"][i])
            hyp = hypos[i][0]["tokens"].int().cpu()[:n]
            hypo_str = dictionary.string(hyp, ignore)
            ref_str = dictionary.string(batch["target"][i].int().cpu()[:n], ignore) if batch["target"] is not None else ""
            records.append((int(batch["id"][i]), utt, ref_str, hypo_str))
            logger.info(f"\nREF:{ref_str}\nHYP:{hypo_str}\n")
            mel_path = os.path.join(results_path, "pred_mel", utt + ".npy")
            os.makedirs(os.path.dirname(mel_path), exist_ok=True)
            np.save(mel_path, batch["mels"][i])
            unit_path = os.path.join(results_path, "pred_unit", utt + ".txt")
            os.makedirs(os.path.dirname(unit_path), exist_ok=True)
            with open(unit_path, "w") as f:
                f.write(hypo_str)
            n_tok += n + 1
    logger.info("Recognized {:,} utterances ({} tokens) in {:.1f}s ({:.2f} sentences/s, {:.2f} tokens/s)".format(
        len(records), n_tok, t_gen, len(records) / max(t_gen, 1e-9), n_tok / max(t_gen, 1e-9)))
    gen_yaml = "".join(f"{k}: {v}\n" for k, v in sorted(vars(gen_args).items()))
    fid = int(hashlib.md5(gen_yaml.encode("utf-8")).hexdigest(), 16) % 1000000
    # run-level collation: every rank's records, in dataset order; rank 0 writes the ONE hypo / wer pair (the reference
    # lets the ranks overwrite each other's files, :297-311)
    records = l2s_dist.gather_results(records)
    result = {"utt_id": [r[1] for r in records], "ref": [r[2] for r in records], "hypo": [r[3] for r in records]}
    if rank != 0:
        l2s_dist.barrier()
        return result
    json.dump(result, open(f"{results_path}/hypo-{fid}.json", "w"), indent=4)
    n_err = n_total = n_equal = 0
    for hypo, ref in zip(result["hypo"], result["ref"]):
        h, r = hypo.strip().split(), ref.strip().split()
        n_err += edit_distance(h, r)
        n_equal += sum(a == b for a, b in zip(h, r))
        n_total += len(r)
    if n_total:
        wer, acc = 100 * n_err / n_total, 100 * n_equal / n_total
        with open(f"{results_path}/wer.{fid}", "w") as fo:
            fo.write(f"WER: {wer}\nAccuracy: {acc}\nerr / num_ref_words = {n_err} / {n_total}\n\n{gen_yaml}")
        logger.info(f"WER: {wer}%")
        logger.info(f"Accuracy: {acc}%")
    l2s_dist.barrier()
    return result


if __name__ == "__main__":
    main()
