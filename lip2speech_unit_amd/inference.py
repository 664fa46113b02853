#!/usr/bin/env python3
"""Stage-1 CLI — mirrors multi_target_lip2speech/inference.py (`--config-name decode` hydra surface, :46-71,:97-317).

hydra/omegaconf/fairseq are not dependencies here; the same command line is parsed directly: `--config-dir D --config-name N`
reads D/N.yaml (default directory: this package's conf/, which holds the reference's decode.yaml values), then `key=value`
overrides apply on top:
  python -m lip2speech_unit_amd.inference --config-name decode common_eval.path=<ckpt.pt> common_eval.results_path=<dir> \
      override.data=<label_dir> override.label_dir=<label_dir> [fp16=true] [dataset.gen_subset=test] \
      [generation.beam=50] [generation.nbest=1] [dataset.batch_size=N] \
      [vocoder.config=<multi_input.json> vocoder.checkpoint=<g_xxx>]      (fused run: pred_wav/ as well, no file round trip)
      [hipgraph=false]      (default true: the device part of a batch replays from a hipGraph captured per batch shape)
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
from .task import Lip2SpeechTask, decode_config

CONF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conf")
# Values that hold when no config file is named: fairseq's dataclass defaults for the keys the path reads
# (GenerationConfig: beam 5, nbest 1, temperature 1, lenpen 1; DatasetConfig: gen_subset "test") + this build's own switches.
DEFAULTS = {
    "common_eval.path": None, "common_eval.results_path": None, "override.data": None, "override.label_dir": None,
    "dataset.gen_subset": "test", "dataset.batch_size": 8, "generation.beam": 5, "generation.temperature": 1.0,
    "generation.lenpen": 1.0, "generation.nbest": 1, "generation.max_len_a": 0, "generation.max_len_b": 200,
    "generation.lm_weight": 0.0, "hipgraph": True, "fp16": False, "common.fp16": False, "dtype": "f16",
    "synthetic_weights": False, "common.user_dir": None, "vocoder.config": None, "vocoder.checkpoint": None,
    "model.encoder_layers": 24, "model.conformer_layers": 12, "model.check_resnet_checksum": True,
}
# top-level groups of the reference's InferConfig (inference.py:55-71) + this build's own (model, vocoder)
CONFIG_GROUPS = ("task", "generation", "common", "common_eval", "checkpoint", "distributed_training", "dataset", "override",
                 "model", "vocoder")
CONFIG_SCALARS = ("is_ax", "fp16", "hipgraph", "dtype", "synthetic_weights", "port", "host")


def _scalar(v):
    if not isinstance(v, str):
        return v
    if v.lower() in ("true", "false"):
        return v.lower() == "true"
    if v.lower() in ("null", "none"):
        return None
    for conv in (int, float):
        try:
            return conv(v)
        except ValueError:
            pass
    return v


def load_config_file(config_dir, config_name):
    """The hydra part of inference.py:46-71,:319-337: `<config_dir>/<config_name>.yaml` (config_dir defaults to this
    package's conf/, as the reference's `config_path`) flattened to dotted keys.  `???` (omegaconf's mandatory-missing
    marker) becomes None: the CLI asserts the values it needs.  A top-level key outside the InferConfig groups raises, as
    hydra's struct mode would."""
    import yaml
    path = os.path.join(config_dir or CONF_DIR, config_name if config_name.endswith((".yaml", ".yml")) else config_name + ".yaml")
    if not os.path.exists(path):
        raise FileNotFoundError(f"config file {path} not found (--config-dir / --config-name)")
    with open(path) as f:
        tree = yaml.safe_load(f) or {}
    flat = {}

    def walk(prefix, node):
        for k, v in node.items():
            key = f"{prefix}.{k}" if prefix else str(k)
            if isinstance(v, dict):
                walk(key, v)
            else:
                flat[key] = None if v == "???" else v
    for top, node in tree.items():
        if isinstance(node, dict):
            if top not in CONFIG_GROUPS:
                raise KeyError(f"{path}: unknown config group '{top}' (known: {', '.join(CONFIG_GROUPS)})")
            walk(top, node)
        else:
            if top not in CONFIG_SCALARS:
                raise KeyError(f"{path}: unknown config key '{top}'")
            flat[top] = None if node == "???" else node
    return flat


def parse_overrides(argv):
    """hydra-style command line: `--config-dir D --config-name N` (also `--config-dir=D`) select a YAML file whose values
    replace the defaults, then `group.key=value` overrides replace those."""
    cfg = dict(DEFAULTS)
    config_dir = config_name = None
    overrides, it = [], iter(argv)
    for a in it:
        if a.startswith("--config-dir") or a.startswith("--config-path") or a.startswith("--config-name"):
            flag, _, val = a.partition("=")
            if not val:
                val = next(it, None)
                if val is None:
                    raise SystemExit(f"{flag} needs a value")
            if flag == "--config-name":
                config_name = val
            else:
                config_dir = val
        elif a.startswith("--"):
            raise SystemExit(f"unknown option {a} (hydra-style `key=value` overrides expected)")
        elif "=" in a:
            overrides.append(a)
        else:
            raise SystemExit(f"cannot parse argument '{a}' (expected key=value)")
    if config_dir is not None and config_name is None:
        raise SystemExit("--config-dir needs --config-name")
    if config_name is not None:
        cfg.update(load_config_file(config_dir, config_name))
    else:
        # The reference's entry point is `@hydra.main(config_name="infer")` (inference.py:320): without --config-name it runs on
        # fairseq's dataclass defaults (beam 5, max_len_b 200), NOT on conf/decode.yaml (beam 50) which its scripts always name
        # (scripts/lrs3/inference_avhubert.sh:6-15).  Same here - and said loudly, because the two differ in beam / n-best width.
        logging.getLogger("lip2speech.inference").warning(
            "no --config-name: fairseq dataclass defaults in force (generation.beam=%s, max_len_b=%s); the reference's scripts "
            "run `--config-name decode` (beam 50) - pass it to get conf/decode.yaml", cfg["generation.beam"], cfg["generation.max_len_b"])
    cfg["_config_name"] = config_name
    for a in overrides:
        k, v = a.split("=", 1)
        cfg[k.lstrip("+")] = _scalar(v)
    return cfg


def edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def build_model(cfg, task, logger=None, checkpoint_path=None):
    """inference.py:108-117,155-156: the `multi_target_avhubert` model for `task`, weights from `checkpoint_path`
    (default common_eval.path; "synthetic[:seed]" = the build-owned seeded generator), on the GPU in eval mode."""
    from .conformer import ConformerConfig
    from .hubert import AVHubertConfig
    dtype = ops.BF16 if cfg["dtype"] == "bf16" else ops.F16
    model = MultiTargetAVHubertEncoderModel.build_model(
        task=task, dtype=dtype, w2v_cfg=AVHubertConfig(encoder_layers=int(cfg["model.encoder_layers"])),
        conformer_cfg=ConformerConfig(conformer_layers=int(cfg["model.conformer_layers"])))
    path = checkpoint_path if checkpoint_path is not None else cfg["common_eval.path"]
    if cfg["synthetic_weights"] or (isinstance(path, str) and path.startswith("synthetic")):
        seed = int(path.split(":", 1)[1]) if isinstance(path, str) and ":" in path else 0
        model.load_state_dict(weights.synth_state_dict(weights.spec_of(model), seed=seed))
    else:
        state = torch.load(path, map_location="cpu")
        # strict apart from the documented allow-lists + the reference's resnet known-answer check (model_avhubert.py:119-123)
        r = model.load_checkpoint_state(state["model"], check_resnet_sum=bool(cfg["model.check_resnet_checksum"]))
        if logger:
            logger.info(f"checkpoint loaded; tolerated missing={len(r.missing_keys)} unexpected={len(r.unexpected_keys)}")
    return model.cuda().eval()


def build_vocoder(cfg):
    """The stage-2 generator for the fused path (multi_input_vocoder/inference.py:85-149): config json + checkpoint
    (`vocoder.checkpoint`; "synthetic[:seed]" accepted), weight norm removed, on the GPU."""
    from .vocoder import AttrDict, MelCodeGenerator
    h = AttrDict(json.load(open(cfg["vocoder.config"])))
    h.text_supervision = False
    voc = MelCodeGenerator(h, dtype=ops.BF16 if cfg["dtype"] == "bf16" else ops.F16)
    path = cfg["vocoder.checkpoint"]
    if path is None or str(path).startswith("synthetic"):
        seed = int(str(path).split(":", 1)[1]) if path and ":" in str(path) else 1
        voc.load_state_dict(weights.synth_state_dict(weights.spec_of(voc), seed=seed))
    else:
        voc.load_state_dict(torch.load(path, map_location="cpu")["generator"])
    voc.cuda().eval()
    voc.remove_weight_norm()
    return voc, h


def build_generator(cfg, task, model, results_path):
    """task.build_generator with the decode.yaml generation values (inference.py:199-202); returns (generator, gen_args)."""
    gen_args = SimpleNamespace(**{k.split(".", 1)[1]: v for k, v in sorted(cfg.items()) if k.startswith("generation.")})
    generator = task.build_generator([model], gen_args, extra_gen_cls_kwargs={})
    generator.use_hipgraph = bool(cfg["hipgraph"])   # not part of the generation config the result-file id is hashed from
    generator.results_path = results_path
    return generator, gen_args


def decode_dataset(cfg, task, model, ds, results_path, logger, rank=0, world=1, vocoder=None, sampling_rate=16000,
                   generator=None):
    """The body of the reference's decode loop (inference.py:199-317): units + mel per clip, hypo / wer summary.  With
    `vocoder` the stage-1 outputs are handed to stage 2 in device memory (no pred_unit/pred_mel ->
    create_dataset.py -> MelCodeDataset round trip) and pred_wav/<spk>/<utt>.wav is written as well, with the file
    names of multi_input_vocoder/inference.py:157-165.  `generator` = a (generator, gen_args) pair to reuse across calls (the
    servers keep one per loaded checkpoint so its captured hipGraphs survive between requests)."""
    if generator is None:
        generator, gen_args = build_generator(cfg, task, model, results_path)
    else:
        generator, gen_args = generator
    dictionary = task.target_dictionary
    ignore = {dictionary.pad(), dictionary.bos(), dictionary.unk(), dictionary.eos()}

    mine = l2s_dist.shard_by_length(ds.sizes, world, rank)
    bs = int(cfg["dataset.batch_size"])
    records = []          # (dataset index, utt_id, ref, hypo) of this rank's clips
    n_tok, t_gen = 0, 0.0
    # the batch goes where the model lives: the GPU (main() refuses to start without one).  A parameter-less test double of the
    # model keeps the host loop on the CPU (tests/test_cli_summary_cpu.py: world-2 gloo run of the sharding + summary collation)
    dev = next((p.device for p in model.parameters()), torch.device("cpu")) if hasattr(model, "parameters") else torch.device("cpu")
    for s in range(0, len(mine), bs):
        batch = ds.collater([ds[i] for i in mine[s:s + bs]])
        ni = batch["net_input"]
        ni["source"]["video"] = ni["source"]["video"].to(dev)
        ni["padding_mask"], ni["spk_emb"] = ni["padding_mask"].to(dev), ni["spk_emb"].to(dev)
        if batch["target"] is not None:
            batch["target"] = batch["target"].to(dev)
        t0 = time.perf_counter()
        hypos, batch = task.inference_step(generator, [model], batch)
        pcm = None
        if vocoder is not None:
            # in-memory hand-off: token t -> unit t-4, mel [B,4T,80] -> [B,80,4T]; rows past a clip's length are masked
            T2 = generator.last_logits.shape[1]
            toks = torch.stack([torch.nn.functional.pad(h[0]["tokens"][:-1], (0, T2 - (h[0]["tokens"].shape[0] - 1)), value=4)
                                for h in hypos])
            code = (toks - 4).clamp_(min=0)
            mel = generator.last_mel.transpose(1, 2).contiguous()
            _, pcm = vocoder.forward_rows(code, mel, ni["spk_emb"], batch["target_lengths"].to(torch.int32))
            pcm = pcm.cpu().numpy()
        if dev.type == "cuda":
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
            if pcm is not None:
                from scipy.io.wavfile import write as write_wav
                wav_path = os.path.join(results_path, "pred_wav", *utt.split("/")[-2:]) + ".wav"
                os.makedirs(os.path.dirname(wav_path), exist_ok=True)
                write_wav(wav_path, sampling_rate, pcm[i, : 320 * n].astype(np.int16))
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
    return result


def setup_logging(results_path):
    os.makedirs(results_path, exist_ok=True)
    logging.basicConfig(format="%(asctime)s | %(levelname)s | %(name)s | %(message)s", level=logging.INFO, force=True,
                        handlers=[logging.FileHandler(os.path.join(results_path, "decode.log")),
                                  logging.StreamHandler(sys.stdout)])
    return logging.getLogger("hybrid.speech_recognize")


def main(argv=None):
    cfg = parse_overrides(sys.argv[1:] if argv is None else argv)
    results_path = cfg["common_eval.results_path"]
    assert results_path, "common_eval.results_path is required"
    rank, world, local = l2s_dist.init_from_env()
    logger = setup_logging(results_path)
    if not torch.cuda.is_available():
        raise SystemExit("this build runs on MI355X only: no CPU path")
    torch.cuda.set_device(local)
    tcfg = decode_config(data=cfg["override.data"], label_dir=cfg["override.label_dir"], fp16=bool(cfg["fp16"]))
    task = Lip2SpeechTask(tcfg)
    model = build_model(cfg, task, logger)
    ds = task.load_dataset(cfg["dataset.gen_subset"])
    vocoder = sr = None
    if cfg["vocoder.config"]:      # fused lip -> units -> waveform run: also writes pred_wav
        vocoder, h = build_vocoder(cfg)
        sr = h.get("sampling_rate", 16000)
    result = decode_dataset(cfg, task, model, ds, results_path, logger, rank, world, vocoder=vocoder,
                            sampling_rate=sr or 16000)
    l2s_dist.barrier()
    return result


if __name__ == "__main__":
    main()
