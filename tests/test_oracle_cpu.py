"""CPU suite (-m "not gpu"): the oracle against the committed golden fixtures (outputs of the reference's own modules),
host-side logic, and the C-ABI library's load/export contract.  No compute call reaches the GPU here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from lip2speech_unit_amd import _lib, weights
from lip2speech_unit_amd.packing import convtranspose_phases, pack_conv1d, weight_norm_weight
from lip2speech_unit_amd.task import LabelEncoderUnit, UnitDictionary
from oracle import avhubert as oa
from oracle import conformer as oc
from oracle import decode as od
from oracle import frontend as ofe
from oracle import vocoder as ov

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VOC_H = dict(resblock="1", upsample_rates=[5, 4, 2, 2, 2], upsample_kernel_sizes=[11, 8, 4, 4, 4],
             upsample_initial_channel=512, resblock_kernel_sizes=[3, 7, 11],
             resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], num_embeddings=200, embedding_dim=128,
             model_in_dim=336, embedder_dim=256, multispkr="_", num_mels=80, text_supervision=False)


def _spec_frontend():
    from lip2speech_unit_amd.resnet import ResEncoder
    return weights.spec_of(ResEncoder("prelu", None))


def test_oracle_frontend_matches_reference_fixture(golden_dir):
    d = np.load(os.path.join(golden_dir, "frontend.npz"))
    sd = weights.synth_state_dict(_spec_frontend(), seed=int(d["seed"]))
    x = ((torch.from_numpy(d["frames_u8"]).float() / 255.0 - 0.421) / 0.165).unsqueeze(1)
    taps = {}
    with torch.no_grad():
        y = ofe.res_encoder(sd, x, taps)
    assert np.abs(y.numpy() - d["out"]).max() < 1e-4
    assert np.abs(taps["stem"][0, :, 2].numpy() - d["stem_t2"].astype(np.float32)).max() < 5e-3  # fp16-stored tap


def test_oracle_conformer_matches_reference_fixture(golden_dir):
    d = np.load(os.path.join(golden_dir, "conformer.npz"))
    from lip2speech_unit_amd.conformer import Conformer
    spec = [("e." + k, s) for k, s in weights.spec_of(Conformer().encoder)]
    sd = weights.synth_state_dict([(k[2:], s) for k, s in spec], seed=int(d["seed"]))
    sd = {"e." + k: v for k, v in sd.items()}
    x = torch.from_numpy(d["x"])
    lens = torch.from_numpy(d["lens"])
    masks = (torch.arange(x.shape[1])[None, :] < lens[:, None]).unsqueeze(1)
    with torch.no_grad():
        y, _ = oc.espnet_encoder_after_frontend(sd, "e", x, masks)
        n = int(lens[1])
        y1, _ = oc.espnet_encoder_after_frontend(sd, "e", x[1:2, :n], masks[1:2, :, :n])
    assert np.abs(y.numpy() - d["out"]).max() < 2e-4
    assert np.abs(y1.numpy() - d["out_clip1_alone"]).max() < 2e-4


def test_oracle_vocoder_matches_reference_fixture(golden_dir):
    d = np.load(os.path.join(golden_dir, "vocoder.npz"))
    from lip2speech_unit_amd.vocoder import AttrDict, MelCodeGenerator
    sd = weights.synth_state_dict(weights.spec_of(MelCodeGenerator(AttrDict(VOC_H))), seed=int(d["seed"]))
    with torch.no_grad():
        y = ov.mel_code_generator(sd, VOC_H, torch.from_numpy(d["code"]), torch.from_numpy(d["mel"]),
                                  torch.from_numpy(d["spkr"]))
    assert np.abs(y.numpy() - d["wav"]).max() < 1e-5
    pcm = ov.to_int16(y)
    assert np.abs(pcm.astype(np.int32) - d["pcm"].astype(np.int32)).max() <= 1


def test_oracle_transformer_encoder_matches_hf_standin_fixture(golden_dir):
    """NOT a reference fixture: HuggingFace's independent port of the un-vendored fairseq TransformerEncoder."""
    d = np.load(os.path.join(golden_dir, "hubert_standin.npz"))
    L = int(d["layers"])
    from lip2speech_unit_amd.hubert import AVHubertConfig, TransformerEncoder
    spec = weights.spec_of(TransformerEncoder(AVHubertConfig(encoder_layers=L)))
    sd = weights.synth_state_dict([("enc." + k, s) for k, s in spec], seed=int(d["seed"]))
    x = torch.from_numpy(d["x"])
    lens = torch.from_numpy(d["lens"])
    pad = torch.arange(x.shape[1])[None, :] >= lens[:, None]
    with torch.no_grad():
        y = oa.transformer_encoder(sd, "enc", x, pad, layers=L)
    assert np.abs(y.numpy() - d["out"])[~pad.numpy()].max() < 5e-5


@pytest.mark.parametrize("beam", [1, 5, 50])
def test_beam_search_hypothesis0_is_masked_argmax(beam):
    g = torch.Generator().manual_seed(beam)
    logits = torch.randn(16, 3, 204, generator=g) * 2
    logits[3, 0, 2] = 50.0   # a huge EOS / pad / unk logit must never be selected (:276-282)
    logits[5, 1, 1] = 50.0
    logits[6, 2, 3] = 50.0
    tl = [16, 10, 2]
    fin = od.beam_search_decode(logits, tl, beam_size=beam)
    gr = od.greedy_decode(logits, tl)
    for b in range(3):
        assert torch.equal(fin[b][0]["tokens"], gr[b]["tokens"])
        assert fin[b][0]["tokens"][-1].item() == 2 and (fin[b][0]["tokens"][:-1] >= 4).all()
        assert abs(float(fin[b][0]["score"]) - float(gr[b]["score"])) < 1e-5
        assert len(fin[b]) == min(beam, 203)


def test_rel_pos_table_matches_espnet_layout():
    pe = oc.rel_pos_table(5, 8)[0]
    assert pe.shape == (9, 8)
    assert torch.allclose(pe[4], torch.tensor([0., 1.] * 4))          # relative position 0
    assert torch.allclose(pe[0, 0], torch.sin(torch.tensor(4.0)))       # row 0 <-> +T-1
    assert torch.allclose(pe[8, 0], torch.sin(torch.tensor(-4.0)))


def test_convtranspose_phase_decomposition_cpu():
    g = torch.Generator().manual_seed(0)
    for (cin, cout, k, s) in [(6, 4, 11, 5), (5, 3, 8, 4), (4, 4, 4, 2)]:
        x = torch.randn(2, cin, 9, generator=g)
        w = torch.randn(cin, cout, k, generator=g)
        p = (k - s) // 2
        ref = F.conv_transpose1d(x, w, None, s, p)
        out = torch.zeros_like(ref)
        for ph in convtranspose_phases(w, s, p):
            wp = ph["w"].view(cout, ph["ntaps"], cin)
            for q in range(9):
                for j in range(ph["ntaps"]):
                    src = q + ph["off"] - j
                    if 0 <= src < 9:
                        out[:, :, q * s + ph["r"]] += x[:, :, src] @ wp[:, j, :].t()
        assert (out - ref).abs().max() < 1e-4


def test_convtranspose_fused_single_gemm_cpu():
    """packing.convtranspose_fused: ConvTranspose1d as ONE correlation with N = stride*Cout (the [M, s*Cout] GEMM output is the
    interleaved activation); checked against F.conv_transpose1d for the vocoder's three (k, stride) pairs."""
    from lip2speech_unit_amd.packing import convtranspose_fused
    g = torch.Generator().manual_seed(2)
    for (cin, cout, k, s) in [(6, 4, 11, 5), (5, 3, 8, 4), (4, 4, 4, 2)]:
        T = 9
        x = torch.randn(2, cin, T, generator=g)
        w = torch.randn(cin, cout, k, generator=g)
        p = (k - s) // 2
        ref = F.conv_transpose1d(x, w, None, s, p)                     # [2, cout, T*s]
        f = convtranspose_fused(w, s, p)
        wf = f["w"].view(s * cout, f["ntaps"], cin)
        out = torch.zeros(2, T, s * cout)
        for q in range(T):
            for i in range(f["ntaps"]):
                src = q + f["off"] - i
                if 0 <= src < T:
                    out[:, q] += x[:, :, src] @ wf[:, i, :].t()
        got = out.view(2, T * s, cout).transpose(1, 2)                # row q, column r*cout + c  ==  sample q*s + r, channel c
        assert (got - ref).abs().max() < 1e-4 and f["n"] == s * cout and f["ntaps"] == 3
        # folded: 2 time steps per row in and out (T even)
        T2 = 10
        x = torch.randn(2, cin, T2, generator=g)
        ref = F.conv_transpose1d(x, w, None, s, p)
        f2 = convtranspose_fused(w, s, p, fold=2)
        xr = x.transpose(1, 2).reshape(2, T2 // 2, 2 * cin)              # row q' = [x[2q'], x[2q'+1]]
        w2 = f2["w"].view(f2["n"], f2["ntaps"], 2 * cin)
        out = torch.zeros(2, T2 // 2, f2["n"])
        for q in range(T2 // 2):
            for i in range(f2["ntaps"]):
                src = q + f2["off"] - i
                if 0 <= src < T2 // 2:
                    out[:, q] += xr[:, src] @ w2[:, i, :].t()
        got = out.view(2, T2 * s, cout).transpose(1, 2)
        assert (got - ref).abs().max() < 1e-4 and f2["n"] == 2 * s * cout


def test_weight_norm_and_conv_packing():
    g = torch.Generator().manual_seed(1)
    v = torch.randn(8, 4, 3, generator=g)
    gg = torch.rand(8, 1, 1, generator=g) + 0.5
    w = weight_norm_weight({"c.weight_g": gg, "c.weight_v": v}, "c")
    assert torch.allclose(w.flatten(1).norm(dim=1), gg.flatten(), atol=1e-5)
    g2 = torch.rand(1, 1, 3, generator=g) + 0.5      # fairseq pos_conv: dim=2
    w2 = weight_norm_weight({"c.weight_g": g2, "c.weight_v": v}, "c")
    assert torch.allclose(w2.pow(2).sum(dim=(0, 1)).sqrt(), g2.flatten(), atol=1e-5)
    pk = pack_conv1d(v)
    assert pk.shape == (8, 12) and torch.equal(pk[:, 4:8], v[:, :, 1])


def test_unit_dictionary_and_label_codec(tmp_path):
    p = tmp_path / "dict.unt.txt"
    p.write_text("".join(f"{i} 1\n" for i in range(200)))
    d = UnitDictionary.load(str(p))
    assert len(d) == 204 and (d.bos(), d.pad(), d.eos(), d.unk()) == (0, 1, 2, 3)
    assert d.index("0") == 4 and d.index("199") == 203 and d.index("zzz") == 3
    enc = LabelEncoderUnit(d)
    t = enc("14 14 131")
    assert t.tolist() == [18, 18, 135, 2]
    assert enc.decode(torch.tensor([18, 18, 135, 2, 1]), {1, 0, 3}) == "14 14 131"


def test_synthetic_weights_are_deterministic_and_sane():
    a = weights.synth_tensor("conformer.encoder.encoders.3.feed_forward.w_1.weight", (2048, 512), 0)
    b = weights.synth_tensor("conformer.encoder.encoders.3.feed_forward.w_1.weight", (2048, 512), 0)
    c = weights.synth_tensor("conformer.encoder.encoders.4.feed_forward.w_1.weight", (2048, 512), 0)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(a.std().item() - 512 ** -0.5) < 2e-3
    assert weights.synth_tensor("x.running_var", (64,), 0).min() > 0


def test_library_loads_and_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "lip2speech_hip.h")).read()
    declared = set(re.findall(r"\b(l2s_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("l2s_gemm_desc")
    assert declared, "no declarations parsed"
    lib = _lib.load()
    assert set(_lib.SIGNATURES) == declared, (set(_lib.SIGNATURES) ^ declared)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.l2s_abi_version() == _lib.ABI_VERSION
    assert b"gfx950" in lib.l2s_build_info()
    assert ctypes.sizeof(_lib.GemmDesc) % 8 == 0
    # argument validation happens on the host before anything is launched
    assert lib.l2s_tapgemm(None, None) == -1
    d = _lib.GemmDesc()
    assert lib.l2s_tapgemm(ctypes.byref(d), None) == -1
    assert lib.l2s_tapgemm_variant(ctypes.byref(_lib.GemmDesc(M=3200, N=1024, groups=1))) == 256064
    assert lib.l2s_layernorm(None, 1, 0, None, None, 1e-5, None, 0, 0, None, 0, 1, 4, 0, None, 1, 0, 0, None) == -1


def test_ops_refuse_cpu_tensors():
    from lip2speech_unit_amd import ops
    with pytest.raises(_lib.L2SError):
        ops.tapgemm(torch.zeros(8, 8, dtype=torch.float16), torch.zeros(8, 8, dtype=torch.float16),
                    torch.zeros(8, 8, dtype=torch.float16), M=8, N=8, Cin=8)


def test_oracle_vocoder_matches_reference_on_lrs3_sample(golden_dir):
    """BASELINE configs[0] data: the reference MelCodeGenerator's output on the reference's own LRS3 sample clips (real units /
    mel / speaker embedding, trimmed by dataset_multi_input.py:222-239); fixture made by tools/make_golden.py vocoder_lrs3."""
    d = np.load(os.path.join(golden_dir, "vocoder_lrs3.npz"))
    from lip2speech_unit_amd import data
    from lip2speech_unit_amd.vocoder import AttrDict, MelCodeGenerator
    sd = weights.synth_state_dict(weights.spec_of(MelCodeGenerator(AttrDict(VOC_H))), seed=int(d["seed"]))
    cd = data.load_code_dict(os.path.join(golden_dir, "lrs3_sample", "dict.unt.txt"))
    for ci in range(len(d["clips"])):
        code = np.array(data.code_to_sequence(str(d[f"c{ci}_unt_line"]).split(), cd))[: int(d[f"c{ci}_code_len"])]
        mel = d[f"c{ci}_mel_raw"][: int(d[f"c{ci}_mel_len"])].T.copy()
        with torch.no_grad():
            y = ov.mel_code_generator(sd, VOC_H, torch.from_numpy(code)[None], torch.from_numpy(mel)[None],
                                      torch.from_numpy(d[f"c{ci}_spk"])[None])
        assert y.shape[-1] == d[f"c{ci}_wav"].shape[0] == 320 * code.shape[0]
        assert np.abs(y[0, 0].numpy() - d[f"c{ci}_wav"]).max() < 1e-5
        assert np.abs(ov.to_int16(y).astype(np.int32).ravel() - d[f"c{ci}_pcm"].astype(np.int32)).max() <= 1


def _rand_layer_sd(d, ffn, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
        sd[f"l.self_attn.{n}.weight"] = torch.randn(d, d, generator=g) * d ** -0.5
        sd[f"l.self_attn.{n}.bias"] = torch.randn(d, generator=g) * 0.1
    for n in ("self_attn_layer_norm", "final_layer_norm"):
        sd[f"l.{n}.weight"] = 1 + 0.1 * torch.randn(d, generator=g)
        sd[f"l.{n}.bias"] = 0.1 * torch.randn(d, generator=g)
    sd["l.fc1.weight"], sd["l.fc1.bias"] = torch.randn(ffn, d, generator=g) * d ** -0.5, torch.randn(ffn, generator=g) * 0.1
    sd["l.fc2.weight"], sd["l.fc2.bias"] = torch.randn(d, ffn, generator=g) * ffn ** -0.5, torch.randn(d, generator=g) * 0.1
    return sd


def test_oracle_mha_matches_torch_multi_head_attention_forward():
    """Second witness for the parity-unpinned a7 oracle: fairseq's MultiheadAttention dispatches to
    torch.nn.functional.multi_head_attention_forward(use_separate_proj_weight=True) on this path, and its encoder layer with
    layer_norm_first is torch.nn.TransformerEncoderLayer(norm_first=True, activation='gelu')."""
    d, heads, ffn, B, T = 128, 8, 256, 3, 17
    sd = _rand_layer_sd(d, ffn, 5)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, T, d, generator=g)
    pad = torch.zeros(B, T, dtype=torch.bool)
    pad[1, 11:] = True
    pad[2, 5:] = True
    p = "l.self_attn"
    with torch.no_grad():
        got = oa.mha(sd, p, x, pad, heads)
        ref, _ = F.multi_head_attention_forward(
            x.transpose(0, 1), x.transpose(0, 1), x.transpose(0, 1), d, heads, None,
            torch.cat([sd[f"{p}.q_proj.bias"], sd[f"{p}.k_proj.bias"], sd[f"{p}.v_proj.bias"]]), None, None, False, 0.0,
            sd[f"{p}.out_proj.weight"], sd[f"{p}.out_proj.bias"], training=False, key_padding_mask=pad,
            need_weights=False, use_separate_proj_weight=True, q_proj_weight=sd[f"{p}.q_proj.weight"],
            k_proj_weight=sd[f"{p}.k_proj.weight"], v_proj_weight=sd[f"{p}.v_proj.weight"])
        assert (got - ref.transpose(0, 1))[~pad].abs().max() < 2e-5
        # one whole pre-LN layer against torch.nn.TransformerEncoderLayer
        layer = torch.nn.TransformerEncoderLayer(d, heads, ffn, dropout=0.0, activation="gelu", batch_first=True,
                                                 norm_first=True).eval()
        layer.load_state_dict({
            "self_attn.in_proj_weight": torch.cat([sd[f"{p}.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")]),
            "self_attn.in_proj_bias": torch.cat([sd[f"{p}.{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")]),
            "self_attn.out_proj.weight": sd[f"{p}.out_proj.weight"], "self_attn.out_proj.bias": sd[f"{p}.out_proj.bias"],
            "linear1.weight": sd["l.fc1.weight"], "linear1.bias": sd["l.fc1.bias"],
            "linear2.weight": sd["l.fc2.weight"], "linear2.bias": sd["l.fc2.bias"],
            "norm1.weight": sd["l.self_attn_layer_norm.weight"], "norm1.bias": sd["l.self_attn_layer_norm.bias"],
            "norm2.weight": sd["l.final_layer_norm.weight"], "norm2.bias": sd["l.final_layer_norm.bias"]})
        ref_l = layer(x, src_key_padding_mask=pad)
        h = oa._ln(sd, "l.self_attn_layer_norm", x)
        y = x + oa.mha(sd, p, h, pad, heads)
        y = y + oa._lin(sd, "l.fc2", F.gelu(oa._lin(sd, "l.fc1", oa._ln(sd, "l.final_layer_norm", y))))
        assert (y - ref_l)[~pad].abs().max() < 5e-5


def test_oracle_swish_frontend_matches_reference_fixture(golden_dir):
    """ESPnet Conv3dResNet(relu_type='swish') - fixture = output of the reference's own module."""
    d = np.load(os.path.join(golden_dir, "frontend_swish.npz"))
    from lip2speech_unit_amd.conv3d_extractor import Conv3dResNet
    sd = weights.synth_state_dict(weights.spec_of(Conv3dResNet()), seed=int(d["seed"]))
    assert not any(k.endswith("relu1.weight") or k.endswith("frontend3D.2.weight") for k in sd)   # Swish has no parameters
    x = (torch.from_numpy(d["frames_u8"]).float() / 255.0 - 0.421) / 0.165
    with torch.no_grad():
        y = ofe.conv3d_resnet(sd, x)
    assert y.shape == d["out"].shape and np.abs(y.numpy() - d["out"]).max() < 1e-4


def test_oracle_raven_encoder_matches_reference_fixture(golden_dir):
    """RAVEn visual encoder (multi_target_lip2speech/model_raven.py:107-132): fixture = outputs of the reference's
    raven/_espnet Encoder - full path (Conv3dResNet frontend + transformer) and the transformer alone on a padded batch."""
    d = np.load(os.path.join(golden_dir, "raven.npz"))
    L = int(d["layers"])
    from lip2speech_unit_amd.model_raven import RAVENConfig, RAVENEncoder
    sd = weights.synth_state_dict(weights.spec_of(RAVENEncoder(RAVENConfig(encoder_num_blocks=L)).encoder), seed=int(d["seed"]))
    sd = {"e." + k: v for k, v in sd.items()}
    x = torch.from_numpy(d["x"])
    lens = torch.from_numpy(d["lens"])
    masks = (torch.arange(x.shape[1])[None, :] < lens[:, None]).unsqueeze(1)
    with torch.no_grad():
        y, _ = oc.raven_encoder_after_frontend(sd, "e", x, masks, layers=L)
        n = int(lens[1])
        y1, _ = oc.raven_encoder_after_frontend(sd, "e", x[1:2, :n], masks[1:2, :, :n], layers=L)
        fsd = {k[len("e.frontend."):]: v for k, v in sd.items() if k.startswith("e.frontend.")}
        frames = (torch.from_numpy(d["frames_u8"]).float() / 255.0 - 0.421) / 0.165
        feats = ofe.conv3d_resnet(fsd, frames)
        yf, _ = oc.raven_encoder_after_frontend(sd, "e", feats, torch.ones(1, 1, feats.shape[1], dtype=torch.bool), layers=L)
    v = masks[:, 0]
    assert np.abs(y.numpy() - d["out"])[v.numpy()].max() < 2e-4
    assert np.abs(y1.numpy() - d["out_clip1_alone"]).max() < 2e-4
    assert np.abs(yf.numpy() - d["out_full"]).max() < 5e-4
    # this block has no temporal op besides attention (key-masked): the padded-batch output of the reference equals clip-alone
    assert np.abs(d["out"][1, :n] - d["out_clip1_alone"][0]).max() < 1e-4
