"""Serving wrappers through Flask's test client, and the single fused CLI (8f row 2): same routes, status
codes and artefact trees as multi_target_lip2speech/inference_server.py:229-384 and multi_input_vocoder/inference_server.py:207-215."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests._synth_dataset import make  # noqa: E402
from tests.test_models_gpu import VOC_H  # noqa: E402


def _voc_cfg(tmp_path):
    cfg = str(tmp_path / "multi_input.json")
    json.dump(dict(VOC_H, code_hop_size=320, mel_hop_size=160, sampling_rate=16000), open(cfg, "w"))
    return cfg


def test_stage1_server_routes_and_hot_swap(tmp_path):
    from lip2speech_unit_amd import inference as s1
    from lip2speech_unit_amd import inference_server as srv
    lab = make(str(tmp_path / "ds"), frames=(9, 6))
    out = str(tmp_path / "out")
    ck = str(tmp_path / "checkpoints.json")
    json.dump({"default_checkpoint_id": "base", "checkpoints": {"base": "synthetic:0", "other": "synthetic:5"}}, open(ck, "w"))
    cfg = s1.parse_overrides([f"common_eval.results_path={out}", f"override.data={lab}", f"override.label_dir={lab}",
                              f"override.checkpoints_data_path={ck}", "dataset.batch_size=2", "model.encoder_layers=2",
                              "model.conformer_layers=1", f"vocoder.config={_voc_cfg(tmp_path)}"])
    app = srv.create_app(cfg)
    c = app.test_client()
    r = c.get("/checkpoints")
    assert r.status_code == 200 and r.get_json() == ["base", "other"]
    assert c.post("/synthesise").status_code == 204
    u0 = open(os.path.join(out, "pred_unit", "test/spk0/00000.txt")).read()
    assert len(u0.split()) == 18 and np.load(os.path.join(out, "pred_mel", "test/spk1/00001.npy")).shape == (24, 80)
    from scipy.io import wavfile
    sr, wav = wavfile.read(os.path.join(out, "pred_wav", "spk0", "00000.wav"))     # fused stage 2 from the same request
    assert sr == 16000 and wav.dtype == np.int16 and wav.shape == (9 * 640,) and np.abs(wav).max() > 0
    assert c.post("/load_checkpoint", json={"checkpoint_id": "base"}).status_code == 204      # already loaded
    r = c.post("/load_checkpoint", json={"checkpoint_id": "nope"})
    assert r.status_code == 400 and "does not exist" in r.get_json()["message"]
    assert c.post("/load_checkpoint", json={"checkpoint_id": "other"}).status_code == 204
    assert srv.state["loaded_checkpoint_id"] == "other"
    assert c.post("/synthesise").status_code == 204
    assert open(os.path.join(out, "pred_unit", "test/spk0/00000.txt")).read() != u0           # another model answered
    assert any(f.startswith("hypo-") for f in os.listdir(out)) and any(f.startswith("wer.") for f in os.listdir(out))


def test_vocoder_server_route(tmp_path):
    from lip2speech_unit_amd import vocoder_inference_server as vs
    lab = make(str(tmp_path / "ds"), frames=(7, 5))
    out = str(tmp_path / "out")
    a = vs.parse_args([_voc_cfg(tmp_path), os.path.join(lab, "test.tsv"), os.path.join(lab, "dict.unt.txt"), "--output_dir", out,
                       "--checkpoint_file", "synthetic"])
    c = vs.create_app(a).test_client()
    assert c.post("/vocoder").status_code == 204
    from scipy.io import wavfile
    sr, wav = wavfile.read(os.path.join(out, "pred_wav", "spk0", "00000.wav"))                 # item 0 only (:210)
    assert sr == 16000 and wav.dtype == np.int16 and wav.shape[0] % 320 == 0 and wav.shape[0] >= 13 * 320
    assert not os.path.exists(os.path.join(out, "pred_wav", "spk1"))


def test_fused_cli_matches_two_stage_cli(tmp_path):
    """One process writing pred_unit + pred_mel + pred_wav == the reference's two CLIs chained through files."""
    from lip2speech_unit_amd import inference as s1
    from lip2speech_unit_amd import synthesise, vocoder_inference as s2
    root = str(tmp_path / "ds")
    lab = make(root, frames=(11, 6, 4))
    vcfg = _voc_cfg(tmp_path)
    common = [f"override.data={lab}", f"override.label_dir={lab}", "synthetic_weights=true", "dataset.batch_size=2",
              "model.encoder_layers=2", "model.conformer_layers=2"]
    fused = str(tmp_path / "fused")
    synthesise.main([f"common_eval.results_path={fused}", f"vocoder.config={vcfg}", "vocoder.checkpoint=synthetic:1", *common])
    with pytest.raises(SystemExit):
        synthesise.main([f"common_eval.results_path={fused}", *common])
    # two-stage route: stage 1 files -> a vocoder dataset (what create_dataset.py:366-428 builds) -> stage 2 CLI
    st1 = str(tmp_path / "st1")
    s1.main([f"common_eval.results_path={st1}", *common])
    import shutil
    import wave
    vroot = str(tmp_path / "vds")
    vlab = os.path.join(vroot, "label")
    os.makedirs(vlab)
    rows = open(os.path.join(lab, "test.tsv")).read().splitlines()[1:]
    units = []
    for r in rows:
        utt, T = r.split("\t")[0], int(r.split("\t")[3])
        for kind in ("audio", "mel", "spk_emb"):
            os.makedirs(os.path.join(vroot, kind, os.path.dirname(utt)), exist_ok=True)
        assert open(os.path.join(st1, "pred_unit", utt + ".txt")).read() == open(os.path.join(fused, "pred_unit", utt + ".txt")).read()
        units.append(open(os.path.join(st1, "pred_unit", utt + ".txt")).read())
        shutil.copyfile(os.path.join(st1, "pred_mel", utt + ".npy"), os.path.join(vroot, "mel", utt + ".npy"))
        shutil.copyfile(os.path.join(root, "spk_emb", utt + ".npy"), os.path.join(vroot, "spk_emb", utt + ".npy"))
        with wave.open(os.path.join(vroot, "audio", utt + ".wav"), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes(np.zeros(T * 640, np.int16).tobytes())
    open(os.path.join(vlab, "test.tsv"), "w").write(vroot + "\n" + "\n".join(rows) + "\n")
    open(os.path.join(vlab, "test.unt"), "w").write("\n".join(units) + "\n")
    shutil.copyfile(os.path.join(lab, "dict.unt.txt"), os.path.join(vlab, "dict.unt.txt"))
    st2 = str(tmp_path / "st2")
    s2.main([vcfg, os.path.join(vlab, "test.tsv"), os.path.join(vlab, "dict.unt.txt"), "--output_dir", st2, "-n", "-1",
             "--synthetic_weights"])
    from scipy.io import wavfile
    for r in rows:
        utt = r.split("\t")[0]
        a = wavfile.read(os.path.join(fused, "pred_wav", *utt.split("/")[-2:]) + ".wav")[1].astype(np.int32)
        b = wavfile.read(os.path.join(st2, "pred_wav", *utt.split("/")[-2:]) + ".wav")[1].astype(np.int32)
        assert a.shape == b.shape and np.abs(a - b).max() <= 2      # the npy/txt round trip is lossless; same kernels
