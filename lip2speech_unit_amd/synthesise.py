#!/usr/bin/env python3
"""End-to-end CLI: lips -> units + mel -> waveform in ONE process, the two stages joined in device memory.

The reference chains two CLIs through files (synthesise.sh:10 -> multi_target_lip2speech/inference.py:267-274 writes
pred_unit/pred_mel -> create_dataset.py:366-428 repacks them -> vocoder.sh:9 -> multi_input_vocoder/inference.py).  This
entry point produces the same three artefact trees for CLI parity - pred_unit/<utt>.txt, pred_mel/<utt>.npy,
pred_wav/<spk>/<utt>.wav - plus hypo-<fid>.json / wer.<fid>, without the round trip:
  python -m lip2speech_unit_amd.synthesise common_eval.path=<stage1.pt> common_eval.results_path=<dir> \
      override.data=<label_dir> override.label_dir=<label_dir> vocoder.config=<multi_input.json> vocoder.checkpoint=<g_xxx>
It is `lip2speech_unit_amd.inference` with the vocoder arguments required.
"""
import sys

from . import inference


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if not any(a.startswith("vocoder.config=") for a in argv):
        raise SystemExit("vocoder.config=<multi_input.json> is required (use lip2speech_unit_amd.inference for stage 1 alone)")
    return inference.main(argv)


if __name__ == "__main__":
    main()
