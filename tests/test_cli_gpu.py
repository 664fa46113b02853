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


def test_generator_hipgraph_replay_equals_eager(tmp_path):
    """The CLIs replay the device part of a batch from a hipGraph captured per (batch, frames-bucket) shape: same units /
    mel as the eager launches, one capture per shape, frames padded to the bucket do not change a clip's result."""
    import torch
    from lip2speech_unit_amd import ops, weights
    from lip2speech_unit_amd.conformer import ConformerConfig
    from lip2speech_unit_amd.hubert import AVHubertConfig
    from lip2speech_unit_amd.model_avhubert import MultiTargetAVHubertEncoderModel
    from lip2speech_unit_amd.sequence_generator import MultiTargetSequenceGenerator
    from lip2speech_unit_amd.task import UnitDictionary
    from tests.test_models_gpu import _frames
    m = MultiTargetAVHubertEncoderModel.build_model(dtype=ops.F16, w2v_cfg=AVHubertConfig(encoder_layers=2),
                                                    conformer_cfg=ConformerConfig(conformer_layers=2))
    m.load_state_dict(weights.synth_state_dict(weights.spec_of(m), seed=3))
    m = m.cuda().eval()
    d = UnitDictionary([str(i) for i in range(200)])
    eager = MultiTargetSequenceGenerator([m], d, beam_size=5)
    graphed = MultiTargetSequenceGenerator([m], d, beam_size=5, use_hipgraph=True, frame_bucket=8)

    def batch(seed, T, lens):
        B = len(lens)
        video = _frames(B, T, seed)
        pad = torch.zeros(B, T, dtype=torch.bool)
        for b, n in enumerate(lens):
            pad[b, n:] = True
            video[b, :, n:] = 0
        spk = torch.rand(B, 256, generator=torch.Generator().manual_seed(seed))
        return {"net_input": {"source": {"audio": None, "video": video.cuda()}, "padding_mask": pad.cuda(), "spk_emb": spk.cuda()},
                "target": None}

    for seed, T, lens in ((1, 13, [13, 9]), (2, 11, [11, 4]), (3, 16, [16, 16]), (4, 13, [10, 13])):   # T 13, 11, 16 -> bucket 16
        fe, se = eager.generate([m], batch(seed, T, lens))
        fg, sg = graphed.generate([m], batch(seed, T, lens))
        assert se["target_lengths"].tolist() == sg["target_lengths"].tolist()
        for b in range(len(lens)):
            assert torch.equal(fe[b][0]["tokens"], fg[b][0]["tokens"])
            assert np.abs(se["mels"][b] - sg["mels"][b]).max() < 2e-3       # other tile shapes at another M: fp32 order only
    assert graphed._graphs.captures == 1                                    # one shape after bucketing: (2, 16 frames)
