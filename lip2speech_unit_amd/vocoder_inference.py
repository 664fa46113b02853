#!/usr/bin/env python3
"""Stage-2 CLI — mirrors multi_input_vocoder/inference.py:167-259 (argparse surface :170-180, worker :85-165).

  python -m lip2speech_unit_amd.vocoder_inference <config.json> <label/test.tsv> <dict.unt.txt> \
      --output_dir D --checkpoint_file C -n -1 [--pad N] [--synthetic_weights]
Writes D/pred_wav/<spk>/<utt>.wav (int16, 16 kHz) like :157-165.
"""
import argparse
import json
import os
import time

import numpy as np
import torch
from scipy.io.wavfile import write

from . import weights
from .data import MelCodeDataset, parse_manifest
from .vocoder import AttrDict, MelCodeGenerator


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("config_file")
    p.add_argument("input_code_file")
    p.add_argument("code_dict_path")
    p.add_argument("--code_file", default=None)
    p.add_argument("--output_dir", default="generated_files")
    p.add_argument("--checkpoint_file", required=False, default=None)
    p.add_argument("--pad", default=None, type=int)
    p.add_argument("--debug", action="store_true")
    p.add_argument("-n", type=int, default=10)
    p.add_argument("--synthetic_weights", action="store_true")
    p.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    a = p.parse_args(argv)
    if a.code_file is not None:
        raise NotImplementedError("--code_file (units without mel/speaker) is not the multi-input path")
    if not torch.cuda.is_available():
        raise SystemExit("this build runs on MI355X only: no CPU path")
    from . import ops
    h = AttrDict(json.load(open(a.config_file)))
    h.code_dict_path = a.code_dict_path
    h.text_supervision = bool(int(os.environ.get("TEXT_SUPERVISION", 0)))
    gen = MelCodeGenerator(h, dtype=ops.BF16 if a.dtype == "bf16" else ops.F16)
    if a.synthetic_weights:
        gen.load_state_dict(weights.synth_state_dict(weights.spec_of(gen), seed=1))
    else:
        gen.load_state_dict(torch.load(a.checkpoint_file, map_location="cpu")["generator"])   # :119-120
    gen.cuda().eval()
    gen.remove_weight_norm()                                                                   # :142-143
    ds = MelCodeDataset(parse_manifest(a.input_code_file), h.code_hop_size, h.mel_hop_size, code_dict_path=a.code_dict_path,
                        pad=a.pad)
    os.makedirs(a.output_dir, exist_ok=True)
    n = len(ds) if a.n == -1 else min(a.n, len(ds))
    audio_s, wall = 0.0, 0.0
    for i in range(n):
        feats, _, filename, _ = ds[i]
        code = {k: torch.from_numpy(v).cuda().unsqueeze(0) for k, v in feats.items()}        # :155
        t0 = time.perf_counter()
        with torch.no_grad():
            _, pcm = gen.forward_rows(code["code"], code["mel"], code["spkr"])
        audio = pcm[0].cpu().numpy()                                                          # :79-81
        wall += time.perf_counter() - t0
        audio_s += audio.shape[0] / h.sampling_rate
        out = os.path.join(a.output_dir, os.path.join("pred_wav", *(filename.split("/")[-2:]))[:-4] + ".wav")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        write(out, h.sampling_rate, audio.astype(np.int16))
    print(f"synthesised {n} clips, {audio_s:.1f} s of audio in {wall:.2f} s (RTF {audio_s / max(wall, 1e-9):.1f}x)")


if __name__ == "__main__":
    main()
