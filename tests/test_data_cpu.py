"""Input-boundary host logic (a1, a14 of SURVEY section 8) on CPU: manifests, normalisation, collation, trimming rule."""
import os

import numpy as np
import torch

from lip2speech_unit_amd import data
from lip2speech_unit_amd.task import LabelEncoderUnit, UnitDictionary
from tests._synth_dataset import make


def test_stage1_dataset_and_collater(tmp_path):
    lab = make(str(tmp_path), frames=(12, 9, 5))
    d = UnitDictionary.load(os.path.join(lab, "dict.unt.txt"))
    ds = data.MultiTargetDataset(os.path.join(lab, "test.tsv"), os.path.join(lab, "test.unt"), LabelEncoderUnit(d))
    assert len(ds) == 3 and ds.sizes == [12, 9, 5]
    s = ds[1]
    raw = np.load(os.path.join(str(tmp_path), "video", "test/spk1/00001.npy"))
    ref = (raw[:, 4:92, 4:92].astype(np.float32) / 255.0 - 0.421) / 0.165        # hubert_dataset.py:242-245
    assert s["video_source"].shape == (9, 88, 88) and np.allclose(s["video_source"].numpy(), ref, atol=1e-6)
    assert s["spk_emb"].shape == (256,) and s["mel"].shape[1] == 80
    b = ds.collater([ds[0], ds[1], ds[2]])
    v, pm = b["net_input"]["source"]["video"], b["net_input"]["padding_mask"]
    assert v.shape == (3, 1, 12, 88, 88) and b["net_input"]["source"]["audio"] is None
    assert pm.sum(-1).tolist() == [0, 3, 7] and bool(pm[2, 5:].all()) and not bool(pm[2, :5].any())
    assert v[2, 0, 5:].abs().max() == 0                                             # zero-filled padded frames
    assert b["target"].shape[0] == 3 and int(b["target"][1, int(b["target_lengths"][1]) - 1]) == 2   # eos appended
    assert b["utt_id"][0] == "test/spk0/00000"


def test_center_crop_matches_reference_rounding():
    f = np.arange(2 * 97 * 99, dtype=np.float32).reshape(2, 97, 99)
    c = data.center_crop(f, 88)
    assert c.shape == (2, 88, 88)
    assert c[0, 0, 0] == f[0, int(round(97 - 88) / 2.), int(round(99 - 88) / 2.)]


def test_stage2_manifest_and_trimming(tmp_path):
    lab = make(str(tmp_path), frames=(12, 9, 5))
    files = data.parse_manifest(os.path.join(lab, "test.tsv"))
    ds = data.MelCodeDataset(files, 320, 160, code_dict_path=os.path.join(lab, "dict.unt.txt"))
    for i, T in enumerate((12, 9, 5)):
        feats, _, fn, _ = ds[i]
        n_audio = data.audio_num_samples(fn)
        L = min(n_audio // 320, 2 * T + (i % 2))
        Lm = min(n_audio // 160, 4 * T + 2)
        cut = min(Lm * 160, L * 320)                                              # dataset_multi_input.py:235
        assert feats["code"].shape == (cut // 320,) and feats["mel"].shape == (80, cut // 160)
        assert feats["mel"].shape[1] == 2 * feats["code"].shape[0]
        assert feats["code"].dtype == np.int64 and feats["code"].min() >= 0 and feats["code"].max() < 200
        assert feats["spkr"].shape == (256,)
    cd = data.load_code_dict(os.path.join(lab, "dict.unt.txt"))
    assert data.code_to_sequence(["3", "3", "zz", "7"], cd) == [3, 3, 7]
    assert data.code_to_sequence(["3", "3", "7"], cd, collapse_code=True) == [3, 7]


def test_traffic_summary_kernel_keys():
    """tools/traffic_summary.py must name rocprofv3's kernels exactly like ops.KernelProfiler does, or bench.py's
    `roofline.traffic` lookup (profiles/traffic_latest.json) silently misses the dominant kernel."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("traffic_summary", os.path.join(root, "tools", "traffic_summary.py"))
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    ns = "void (anonymous namespace)::"
    cases = {
        ns + "tapgemm_kernel<ElemF16, 256, 128, 4, 2, 1, 3, 7, true>(l2s_gemm_desc, int, int, int, int)": "tapgemm<f16,256x128,mode1,e7>",
        ns + "phasegemm_kernel<ElemF16, 0, 8>(l2s_gemm_desc, int, int, int, int)": "tapgemm<f16,256x256,mode0,e8>",
        ns + "patchconv64_kernel<ElemF16, 1, 3, 128>(l2s_gemm_desc, int, int, int)": "tapgemm<f16,999x128,mode1,e3>",
        ns + "patchconv64_kernel<ElemBF16, 2, 6, 64>(l2s_gemm_desc, int, int, int)": "tapgemm<bf16,999x64,mode2,e6>",
        ns + "resblock_kernel<ElemF16, 32, 11>(unsigned short const*)": "l2s_resblock_fused<C32,k11>",
        ns + "layernorm_kernel<ElemF16, true, false>(void const*)": "l2s_layernorm",
    }
    for name, key in cases.items():
        assert ts.norm(name) == key, (name, ts.norm(name))
    # the committed traffic file carries the key of the kernel the committed bench line calls dominant
    traffic = json.load(open(os.path.join(root, "profiles", "traffic_latest.json")))["kernels"]
    line = json.loads(open(os.path.join(root, "profiles", "r01_final_bench.json")).read().strip().splitlines()[-1])
    assert line["roofline"]["kernel"] in traffic
    assert abs(traffic[line["roofline"]["kernel"]]["hbm_bytes_per_launch"] - line["roofline"]["traffic"]) < 1e-3 * line["roofline"]["traffic"]
