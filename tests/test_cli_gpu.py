"""Both CLIs end to end on a synthetic dataset in the reference's directory layout (synthetic weights)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests._synth_dataset import make  # noqa: E402


def test_stage1_and_stage2_cli(tmp_path):
    from lip2speech_unit_amd import inference as s1
    from lip2speech_unit_amd import vocoder_inference as s2
    root = str(tmp_path / "ds")
    lab = make(root, frames=(12, 9, 5))
    out1 = str(tmp_path / "out1")
    res = s1.main([f"common_eval.results_path={out1}", f"override.data={lab}", f"override.label_dir={lab}",
                   "synthetic_weights=true", "dataset.batch_size=2", "model.encoder_layers=2",
                   "model.conformer_layers=2"])
    assert len(res["utt_id"]) == 3
    for utt, T in zip(("test/spk0/00000", "test/spk1/00001", "test/spk0/00002"), (12, 9, 5)):
        units = open(os.path.join(out1, "pred_unit", utt + ".txt")).read().split()
        assert len(units) == 2 * T and all(0 <= int(u) < 200 for u in units)
        mel = np.load(os.path.join(out1, "pred_mel", utt + ".npy"))
        assert mel.shape == (4 * T, 80) and mel.dtype == np.float32
    assert any(f.startswith("hypo-") for f in os.listdir(out1)) and any(f.startswith("wer.") for f in os.listdir(out1))
    assert os.path.exists(os.path.join(out1, "decode.log"))
    # batch of 2 + batch of 1 must equal one-clip-at-a-time decoding (the reference's batch_size=1)
    out1b = str(tmp_path / "out1b")
    res1 = s1.main([f"common_eval.results_path={out1b}", f"override.data={lab}", f"override.label_dir={lab}",
                    "synthetic_weights=true", "dataset.batch_size=1", "model.encoder_layers=2",
                    "model.conformer_layers=2"])
    assert dict(zip(res["utt_id"], res["hypo"])) == dict(zip(res1["utt_id"], res1["hypo"]))
    # stage 2 on the dataset's own labels (multi_input_vocoder/scripts/lrs3/inference_aug.sh)
    cfg = str(tmp_path / "cfg.json")
    from tests.test_models_gpu import VOC_H
    json.dump(dict(VOC_H, code_hop_size=320, mel_hop_size=160, sampling_rate=16000), open(cfg, "w"))
    out2 = str(tmp_path / "out2")
    s2.main([cfg, os.path.join(lab, "test.tsv"), os.path.join(lab, "dict.unt.txt"), "--output_dir", out2, "-n", "-1",
             "--synthetic_weights"])
    from scipy.io import wavfile
    sr, wav = wavfile.read(os.path.join(out2, "pred_wav", "spk1", "00001.wav"))
    assert sr == 16000 and wav.dtype == np.int16 and wav.shape[0] % 320 == 0 and wav.shape[0] >= 17 * 320
    assert np.abs(wav).max() > 0
