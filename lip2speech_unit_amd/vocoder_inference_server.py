#!/usr/bin/env python3
"""Stage-2 micro-server — mirrors multi_input_vocoder/inference_server.py:163-215 on the HIP path.

  python -m lip2speech_unit_amd.vocoder_inference_server <config.json> <label/test.tsv> <dict.unt.txt> \
      --output_dir D --checkpoint_file C [--pad N] [--port 5005]
  POST /vocoder -> 204: re-reads the manifest (+ sibling .unt) and synthesises ITEM 0 to D/pred_wav/<spk>/<utt>.wav (:207-213).
"""
import argparse
import json
import os
from http import HTTPStatus

import numpy as np
import torch
from scipy.io.wavfile import write

from . import ops, weights
from .data import MelCodeDataset, parse_manifest
from .vocoder import AttrDict, MelCodeGenerator


def build(a):
    if not torch.cuda.is_available():
        raise SystemExit("this build runs on MI355X only: no CPU path")
    h = AttrDict(json.load(open(a.config_file)))
    h.code_dict_path = a.code_dict_path
    h.text_supervision = bool(int(os.environ.get("TEXT_SUPERVISION", 0)))
    gen = MelCodeGenerator(h, dtype=ops.BF16 if a.dtype == "bf16" else ops.F16)
    if a.checkpoint_file is None or str(a.checkpoint_file).startswith("synthetic"):
        gen.load_state_dict(weights.synth_state_dict(weights.spec_of(gen), seed=1))
    else:
        gen.load_state_dict(torch.load(a.checkpoint_file, map_location="cpu")["generator"])   # :117-120
    gen.cuda().eval()
    gen.remove_weight_norm()                                                                     # :124-125
    os.makedirs(a.output_dir, exist_ok=True)
    return gen, h


def create_app(a):
    from flask import Flask
    gen, h = build(a)
    app = Flask(__name__)

    @app.post("/vocoder")
    def vocoder():
        ds = MelCodeDataset(parse_manifest(a.input_code_file), h.code_hop_size, h.mel_hop_size,
                            code_dict_path=a.code_dict_path, pad=a.pad)                         # dataset.reset(...) :209
        feats, _, filename, _ = ds[0]                                                            # inference(item_index=0) :210
        code = {k: torch.from_numpy(v).cuda().unsqueeze(0) for k, v in feats.items()}
        with torch.no_grad():
            _, pcm = gen.forward_rows(code["code"], code["mel"], code["spkr"])
        out = os.path.join(a.output_dir, os.path.join("pred_wav", *(filename.split("/")[-2:]))[:-4] + ".wav")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        write(out, h.sampling_rate, pcm[0].cpu().numpy().astype(np.int16))
        return "", HTTPStatus.NO_CONTENT

    return app


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("config_file")
    p.add_argument("input_code_file")
    p.add_argument("code_dict_path")
    p.add_argument("--output_dir", default="generated_files")
    p.add_argument("--checkpoint_file", default=None)
    p.add_argument("--pad", default=None, type=int)
    p.add_argument("--debug", action="store_true")
    p.add_argument("-n", type=int, default=10)
    p.add_argument("--port", type=int, default=5005)
    p.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    return p.parse_args(argv)


def main(argv=None):
    a = parse_args(argv)
    create_app(a).run(port=a.port)


if __name__ == "__main__":
    main()
