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
        ns + "layernorm_rows_kernel<ElemF16, 4, 2, false>(float const*)": "l2s_layernorm",
        ns + "respair_kernel<ElemF16, 128, 0>((anonymous namespace)::RpArgs)": "l2s_respair<C128,mid>",
        ns + "respair_kernel<ElemBF16, 64, 1>((anonymous namespace)::RpArgs)": "l2s_respair<C64,last>",
        ns + "attention_resident_kernel<ElemF16, true>(unsigned short const*)": "l2s_attention",
        ns + "resstage_kernel<ElemF16, 32>((anonymous namespace)::RbArgs, (anonymous namespace)::RsW, int)": "l2s_resstage_fused<C32>",
        ns + "stem_pool_kernel<ElemF16, 2, false>(void const*)": "l2s_stem_pool_fused",
        ns + "basicblock_kernel<ElemF16>((anonymous namespace)::BbArgs)": "l2s_basiclayer_fused",
        ns + "phasegemm_kernel<ElemF16, 1, 10>(l2s_gemm_desc, int, int, int, int)": "tapgemm<f16,256x256,mode1,e10>",
    }
    for name, key in cases.items():
        assert ts.norm(name) == key, (name, ts.norm(name))
    # the committed traffic file carries the key of the kernel the committed bench line calls dominant, collected at that line's
    # clips per launch; the PMC bytes cannot be below the algorithmic bytes the line was computed from, nor far above them
    tj = json.load(open(os.path.join(root, "profiles", "traffic_latest.json")))
    line = json.loads(open(os.path.join(root, "profiles", "r03_bench.json")).read().strip().splitlines()[-1])
    assert tj["batch"] == line["config"]["clips_per_launch"]
    k = line["roofline"]["kernel"]
    assert k in tj["kernels"]
    alg = line["roofline"]["algorithmic_bytes_per_launch"]
    assert 0.95 * alg <= tj["kernels"][k]["hbm_bytes_per_launch"] <= 1.6 * alg


def test_lrs3_sample_label_files_through_both_loaders(tmp_path, golden_dir):
    """BASELINE configs[0] data: the reference's own sample manifests (datasets/lrs3/label/{test.tsv,test.unt,dict.unt.txt},
    committed verbatim under tests/golden/lrs3_sample) through the stage-1 and stage-2 loaders, and the trimming rule of
    dataset_multi_input.py:222-239 on the two clips whose mel / spk_emb the vocoder_lrs3 fixture carries."""
    from tests._lrs3_sample import materialise
    lab, names, g = materialise(str(tmp_path), golden_dir)
    d = UnitDictionary.load(os.path.join(lab, "dict.unt.txt"))
    assert len(d) == 204 and d.index("0") == 4 and d.index("199") == 203        # unit k <-> token k+4
    ds = data.MultiTargetDataset(os.path.join(lab, "test.tsv"), os.path.join(lab, "test.unt"), LabelEncoderUnit(d))
    assert ds.sizes == [107, 62, 31, 89, 37] and ds.ids[4] == "test/62cNtvx6P8E/00001"
    assert [len(ds.label_processor(l)) for l in ds.labels] == [215, 125, 64, 179, 77]     # units + eos
    assert ds.label_processor(ds.labels[0])[:4].tolist() == [18, 18, 18, 135]    # "14 14 14 131"
    files = data.parse_manifest(os.path.join(lab, "test.tsv"))                   # asserts |len(code) - 2*frames| <= 2
    assert [len(c.split()) for c in files[2]] == [214, 124, 63, 178, 76]
    mds = data.MelCodeDataset(files, 320, 160, code_dict_path=os.path.join(lab, "dict.unt.txt"))
    for ci, clip in enumerate(g["clips"]):
        feats, _, fn, _ = mds[names.index(str(clip))]
        assert feats["code"].shape == (int(g[f"c{ci}_code_len"]),) and feats["mel"].shape == (80, int(g[f"c{ci}_mel_len"]))
        assert np.array_equal(feats["mel"], g[f"c{ci}_mel_raw"][: int(g[f"c{ci}_mel_len"])].T)
        assert np.array_equal(feats["spkr"], g[f"c{ci}_spk"])
    assert mds[4][0]["code"].shape == (76,) and mds[4][0]["mel"].shape == (80, 152)


def test_checkpoint_state_dict_round_trip_is_strict():
    """A checkpoint in the reference's key layout loads; a dropped / renamed parameter or a wrong frontend checksum raises
    (inference.py used to load with strict=False and only count the misses)."""
    import pytest
    from lip2speech_unit_amd import weights
    from lip2speech_unit_amd.conformer import ConformerConfig
    from lip2speech_unit_amd.hubert import AVHubertConfig
    from lip2speech_unit_amd.model_avhubert import CheckpointMismatch, MultiTargetAVHubertEncoderModel

    def fresh():
        return MultiTargetAVHubertEncoderModel.build_model(w2v_cfg=AVHubertConfig(encoder_layers=1),
                                                           conformer_cfg=ConformerConfig(conformer_layers=1))
    src = fresh()
    sd = weights.synth_state_dict(weights.spec_of(src), seed=3)
    for k in ("encoder.w2v_model.encoder.layers.0.self_attn.q_proj.weight", "conformer.encoder.encoders.0.self_attn.pos_bias_u",
              "encoder.w2v_model.feature_extractor_video.resnet.frontend3D.0.weight", "conformer.mel_conv.6.bias",
              "encoder.w2v_model.encoder.pos_conv.0.weight_g", "conformer.encoder.encoders.0.conv_module.norm.running_var"):
        assert k in sd, k                                   # the reference's names
    src.load_state_dict(sd)
    want = src.resnet_weight_checksum()
    ck = dict(sd)
    del ck["encoder.w2v_model.mask_emb"]                    # model_avhubert.py:105
    ck["encoder.w2v_model.final_proj.weight"] = torch.zeros(3, 3)
    ck["encoder.w2v_model.label_embs_concat"] = torch.zeros(3)
    ck["conformer.encoder.frontend.frontend3D.0.weight"] = torch.zeros(2)     # `multi_target` checkpoints
    m = fresh()
    m.load_checkpoint_state(ck, expected_resnet_sum=want)
    for (k, a), b in zip(m.state_dict().items(), src.state_dict().values()):
        assert torch.equal(a, b) or k.endswith("mask_emb"), k
    with pytest.raises(CheckpointMismatch, match="checksum"):
        fresh().load_checkpoint_state(ck)                   # synthetic frontend != large_vox_iter5.pt's -13260.4916
    bad = dict(ck)
    bad["encoder.w2v_model.encoder.layers.0.fc1.weight_renamed"] = bad.pop("encoder.w2v_model.encoder.layers.0.fc1.weight")
    with pytest.raises(CheckpointMismatch, match="missing"):
        fresh().load_checkpoint_state(bad, expected_resnet_sum=want)


def test_mfma_util_file_agrees_with_the_committed_bench_line():
    """profiles/mfma_util_latest.json (tools/collect_mfma_util.sh: SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE per kernel) against the
    committed round-4 bench line: the matrix FLOPs the hardware issued for the dominant kernel are the algorithmic FLOPs the
    roofline was computed from (16 busy cycles per 16x16x32 MFMA), and its busy fraction of cycles cannot be below the
    fraction of the 2.4 GHz data-sheet peak the line reports (the kernel runs below 2.4 GHz), nor above 1."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mj = json.load(open(os.path.join(root, "profiles", "mfma_util_latest.json")))
    line = json.loads(open(os.path.join(root, "profiles", "r04_bench.json")).read().strip().splitlines()[-1])
    assert mj["clips"] == line["config"]["clips_per_launch"]
    r = line["roofline"]
    mk = mj["kernels"][r["kernel"]]
    assert abs(mk["mfma_tflop_per_launch"] * 1e12 - r["flop_per_launch"]) <= 0.01 * r["flop_per_launch"]
    assert r["frac"] <= mk["mfma_util"] <= 1.0
    for k, v in mj["kernels"].items():
        assert 0.0 < v["mfma_util"] <= 1.0, (k, v)
