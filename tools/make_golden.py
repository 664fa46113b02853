#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own modules (CPU, fp32) on build-owned synthetic weights.

Runs only in the build container (needs /root/reference); the fixtures it writes are data (seed, shapes, inputs,
outputs) and travel with the repo - the reference itself never does.  Weights are not stored: both sides regenerate
them from parameter names/shapes with lip2speech_unit_amd.weights.synth_state_dict(seed).

  frontend.npz   avhubert/resnet.py ResEncoder('prelu')                 (loaded as a single file: the package imports fairseq)
  frontend_swish.npz  espnet/nets/pytorch_backend/backbones/conv3d_extractor.py Conv3dResNet('resnet', 'swish') (the `multi_target` frontend)
  raven.npz      raven/_espnet/nets/pytorch_backend/transformer/encoder.py Encoder as model_raven.py:107-132 configures it (3 blocks)
  conformer.npz  espnet/nets/pytorch_backend/transformer/encoder.py Encoder.forward_after_frontend (12 x 512, rel_mha, macaron, cnn k=31)
  vocoder.npz    multi_input_vocoder/models_multi_input.py MelCodeGenerator (configs/lrs3/multi_input.json, weight norm removed)
  vocoder_lrs3.npz  the same MelCodeGenerator fed the reference's OWN sample data (datasets/lrs3: units of label/test.unt,
                 mel/*.npy, spk_emb/*.npy of two test clips, trimmed by the rule of dataset_multi_input.py:222-239) - BASELINE configs[0]
  lrs3_sample/   the sample's label files (test.tsv, test.unt, dict.unt.txt): data files, copied verbatim
  hubert_standin.npz  NOT reference code: HuggingFace transformers HubertEncoderStableLayerNorm, an independent port of the
                 fairseq TransformerEncoder that the reference imports but does not vendor.
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from lip2speech_unit_amd import weights  # noqa: E402


def spec(m):
    return [(k, tuple(v.shape)) for k, v in m.state_dict().items()]


def frames(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (B, T, 88, 88), generator=g)
    return ((u8.float() / 255.0 - 0.421) / 0.165).unsqueeze(1), u8


def make_frontend():
    s = importlib.util.spec_from_file_location("ref_resnet", f"{REF}/avhubert/resnet.py")
    m = importlib.util.module_from_spec(s)
    s.loader.exec_module(m)
    enc = m.ResEncoder("prelu", None).eval()
    sd = weights.synth_state_dict(spec(enc), seed=11)
    enc.load_state_dict(sd, strict=True)
    x, u8 = frames(1, 6, 101)
    with torch.no_grad():
        y = enc(x)
        stem = enc.frontend3D[2](enc.frontend3D[1](enc.frontend3D[0](x)))
    np.savez_compressed(os.path.join(OUT, "frontend.npz"), seed=11, frames_u8=u8.numpy().astype(np.uint8),
                        out=y.numpy(), stem_t2=stem[0, :, 2].numpy().astype(np.float16))


def make_frontend_swish():
    """ESPnet Conv3dResNet(relu_type='swish'), the `multi_target` model's frontend."""
    sys.path.insert(0, REF)
    from espnet.nets.pytorch_backend.backbones.conv3d_extractor import Conv3dResNet
    enc = Conv3dResNet(backbone_type="resnet", relu_type="swish").eval()
    sd = weights.synth_state_dict(spec(enc), seed=15)
    enc.load_state_dict(sd, strict=True)
    x, u8 = frames(1, 6, 505)
    with torch.no_grad():
        y = enc(x[:, 0])                                           # [B,T,512]
    np.savez_compressed(os.path.join(OUT, "frontend_swish.npz"), seed=15, frames_u8=u8.numpy().astype(np.uint8), out=y.numpy())
    sys.path.pop(0)


def make_raven():
    """RAVEn visual encoder as model_raven.py:107-132 builds it, from the reference's second vendored ESPnet copy
    (raven/_espnet): full path on one clip (frontend included) and the transformer alone on a padded batch."""
    sys.path.insert(0, f"{REF}/raven")
    from _espnet.nets.pytorch_backend.transformer.encoder import Encoder
    L = 3
    e = Encoder(idim=512, attention_dim=1024, attention_heads=16, linear_units=4096, num_blocks=L, dropout_rate=0.1,
                attention_dropout_rate=0.1, frontend="conv3d", input_layer="vanilla_linear", macaron_style=False,
                encoder_attn_layer_type="rel_mha", use_cnn_module=False, zero_triu=False, cnn_module_kernel=31,
                relu_type="swish", a_upsample_ratio=1, layerscale=True, init_values=0.1, ff_bn_pre=True, post_norm=False,
                gamma_zero=False, gamma_init=0.1, mask_init_type=None, drop_path=0.1).eval()
    sd = weights.synth_state_dict(spec(e), seed=16)
    e.load_state_dict(sd, strict=True)
    x, u8 = frames(1, 6, 606)
    g = torch.Generator().manual_seed(607)
    feats = torch.randn(2, 40, 512, generator=g)
    masks = torch.ones(2, 1, 40, dtype=torch.bool)
    masks[1, :, 27:] = False
    with torch.no_grad():
        y_full, _ = e(x[:, 0], torch.ones(1, 1, 6, dtype=torch.bool))
        e.frontend = None
        y, _ = e(feats, masks)
        y1, _ = e(feats[1:2, :27], masks[1:2, :, :27])
    np.savez_compressed(os.path.join(OUT, "raven.npz"), seed=16, layers=L, frames_u8=u8.numpy().astype(np.uint8),
                        out_full=y_full.numpy(), x=feats.numpy(), lens=np.array([40, 27]), out=y.numpy(),
                        out_clip1_alone=y1.numpy())
    sys.path.pop(0)


def make_conformer():
    sys.path.insert(0, REF)
    from espnet.nets.pytorch_backend.transformer.encoder import Encoder
    e = Encoder(idim=-1, attention_dim=512, attention_heads=8, linear_units=2048, num_blocks=12, dropout_rate=0.1,
                positional_dropout_rate=0.1, attention_dropout_rate=0.0, input_layer="conv3d", normalize_before=True,
                macaron_style=1, encoder_attn_layer_type="rel_mha", use_cnn_module=1, zero_triu=False,
                cnn_module_kernel=31, relu_type="swish", a_upsample_ratio=1)
    e.frontend = None
    e.eval()
    sd = weights.synth_state_dict(spec(e), seed=12)
    e.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(202)
    x = torch.randn(2, 70, 512, generator=g)
    masks = torch.ones(2, 1, 70, dtype=torch.bool)
    masks[1, :, 44:] = False
    with torch.no_grad():
        y, _ = e.forward_after_frontend(x, masks)
        y_single, _ = e.forward_after_frontend(x[1:2, :44], masks[1:2, :, :44])
    np.savez_compressed(os.path.join(OUT, "conformer.npz"), seed=12, x=x.numpy(), lens=np.array([70, 44]),
                        out=y.numpy(), out_clip1_alone=y_single.numpy())
    sys.path.pop(0)


def make_vocoder():
    sys.path.insert(0, f"{REF}/speech-resynthesis")
    sys.path.insert(0, f"{REF}/multi_input_vocoder")
    from models_multi_input import MelCodeGenerator
    from utils import AttrDict
    h = AttrDict(json.load(open(f"{REF}/multi_input_vocoder/configs/lrs3/multi_input.json")))
    h.text_supervision = False
    g = MelCodeGenerator(h).eval()
    sd = weights.synth_state_dict(spec(g), seed=13)
    g.load_state_dict(sd, strict=True)
    g.remove_weight_norm()
    gen = torch.Generator().manual_seed(303)
    code = torch.randint(0, 200, (1, 25), generator=gen)
    mel = -11.5 + 11.6 * torch.rand(1, 80, 50, generator=gen)
    spk = torch.rand(1, 256, generator=gen).relu()
    spk = spk / spk.norm()
    with torch.no_grad():
        y = g(code=code, mel=mel, spkr=spk)
    pcm = (y.squeeze() * 32768.0).numpy().astype("int16")
    np.savez_compressed(os.path.join(OUT, "vocoder.npz"), seed=13, code=code.numpy(), mel=mel.numpy(), spkr=spk.numpy(),
                        wav=y.numpy(), pcm=pcm)
    sys.path.pop(0)
    sys.path.pop(0)


LRS3_CLIPS = ["test/UmvOgW6iV2s/00007", "test/62cNtvx6P8E/00001"]   # line 1 of test.tsv/.unt; the shortest clip


def make_vocoder_lrs3():
    """configs[0] data: real LRS3 sample units / mel / speaker embedding through the reference MelCodeGenerator (synthetic
    weights, seed 13 like vocoder.npz).  The mp4s cannot be decoded here, so stage 1 cannot see this sample; stage 2 can."""
    import shutil
    import wave
    sys.path.insert(0, f"{REF}/speech-resynthesis")
    sys.path.insert(0, f"{REF}/multi_input_vocoder")
    from models_multi_input import MelCodeGenerator
    from utils import AttrDict
    h = AttrDict(json.load(open(f"{REF}/multi_input_vocoder/configs/lrs3/multi_input.json")))
    h.text_supervision = False
    g = MelCodeGenerator(h).eval()
    g.load_state_dict(weights.synth_state_dict(spec(g), seed=13), strict=True)
    g.remove_weight_norm()
    ds = f"{REF}/datasets/lrs3"
    lab = os.path.join(OUT, "lrs3_sample")
    os.makedirs(lab, exist_ok=True)
    for f in ("test.tsv", "test.unt", "dict.unt.txt"):
        shutil.copyfile(f"{ds}/label/{f}", os.path.join(lab, f))
    names = [l.split("\t")[0] for l in open(f"{ds}/label/test.tsv").read().splitlines()[1:]]
    units = open(f"{ds}/label/test.unt").read().splitlines()
    syms = [l.rstrip().rsplit(" ", 1)[0] for l in open(f"{ds}/label/dict.unt.txt")]
    code_dict = {c: i for i, c in enumerate(syms)}
    out = {"seed": 13, "clips": np.array(LRS3_CLIPS)}
    for ci, clip in enumerate(LRS3_CLIPS):
        line = units[names.index(clip)]
        code = np.array([code_dict[c] for c in line.split() if c in code_dict])        # dataset_multi_input.py:128-141
        mel = np.load(f"{ds}/mel/{clip}.npy")
        spk = np.load(f"{ds}/spk_emb/{clip}.npy")
        with wave.open(f"{ds}/audio/{clip}.wav") as w:
            n_audio = w.getnframes()
        code_length = min(n_audio // 320, code.shape[0])                                # :222
        mel_length = min(n_audio // 160, mel.shape[0])                                  # :232
        cut = min(mel_length * 160, code_length * 320)                                  # :235
        code_t, mel_t = code[: cut // 320], mel[: cut // 160]
        with torch.no_grad():
            y = g(code=torch.from_numpy(code_t)[None].long(), mel=torch.from_numpy(mel_t.T.copy())[None],
                  spkr=torch.from_numpy(spk)[None])
        pcm = (y.squeeze() * 32768.0).numpy().astype("int16")                           # inference.py:79-81
        out.update({f"c{ci}_unt_line": np.array(line), f"c{ci}_mel_raw": mel, f"c{ci}_spk": spk, f"c{ci}_n_audio": n_audio,
                    f"c{ci}_code_len": cut // 320, f"c{ci}_mel_len": cut // 160, f"c{ci}_wav": y[0, 0].numpy(),
                    f"c{ci}_pcm": pcm})
        print(clip, "units", len(code), "->", cut // 320, "mel", mel.shape, "->", cut // 160, "samples", y.shape[-1])
    # ground-truth side of the sample: 24 576 samples // 320 = 76 units, mel 154 -> 153 -> cut 24 320 samples -> 152 frames
    assert (out["c1_code_len"], out["c1_mel_len"], out["c1_wav"].shape[0]) == (76, 152, 24320)
    assert (out["c0_code_len"], out["c0_mel_len"], out["c0_wav"].shape[0]) == (214, 428, 68480)
    np.savez_compressed(os.path.join(OUT, "vocoder_lrs3.npz"), **out)
    sys.path.pop(0)
    sys.path.pop(0)


def make_hubert_standin():
    from transformers import HubertConfig
    from transformers.models.hubert.modeling_hubert import HubertEncoderStableLayerNorm
    L = 24   # full AV-HuBERT large depth (conf/pretrain/large_vox_iter5.yaml:96)
    cfg = HubertConfig(hidden_size=1024, num_hidden_layers=L, intermediate_size=4096, num_attention_heads=16,
                       num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16, do_stable_layer_norm=True,
                       layer_norm_eps=1e-5, hidden_act="gelu", hidden_dropout=0.0, attention_dropout=0.0,
                       activation_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0)
    cfg._attn_implementation = "eager"
    enc = HubertEncoderStableLayerNorm(cfg).eval()
    # fairseq-named synthetic weights -> HF names (mapping of HF's own conversion script)
    fs_spec = [("pos_conv.0.weight_g", (1, 1, 128)), ("pos_conv.0.weight_v", (1024, 64, 128)), ("pos_conv.0.bias", (1024,)),
               ("layer_norm.weight", (1024,)), ("layer_norm.bias", (1024,))]
    for i in range(L):
        p = f"layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            fs_spec += [(f"{p}.self_attn.{n}.weight", (1024, 1024)), (f"{p}.self_attn.{n}.bias", (1024,))]
        fs_spec += [(f"{p}.self_attn_layer_norm.weight", (1024,)), (f"{p}.self_attn_layer_norm.bias", (1024,)),
                    (f"{p}.fc1.weight", (4096, 1024)), (f"{p}.fc1.bias", (4096,)), (f"{p}.fc2.weight", (1024, 4096)),
                    (f"{p}.fc2.bias", (1024,)), (f"{p}.final_layer_norm.weight", (1024,)),
                    (f"{p}.final_layer_norm.bias", (1024,))]
    fsd = weights.synth_state_dict([("enc." + k, s) for k, s in fs_spec], seed=14)
    fsd = {k[4:]: v for k, v in fsd.items()}
    hf = {}
    for k, v in fsd.items():
        k2 = k.replace("pos_conv.0.weight_g", "pos_conv_embed.conv.parametrizations.weight.original0")
        k2 = k2.replace("pos_conv.0.weight_v", "pos_conv_embed.conv.parametrizations.weight.original1")
        k2 = k2.replace("pos_conv.0.bias", "pos_conv_embed.conv.bias")
        k2 = k2.replace("self_attn_layer_norm", "layer_norm") if ".self_attn_layer_norm" in k2 else k2
        k2 = k2.replace(".self_attn.", ".attention.")
        k2 = k2.replace(".fc1.", ".feed_forward.intermediate_dense.").replace(".fc2.", ".feed_forward.output_dense.")
        hf[k2] = v
    missing = enc.load_state_dict(hf, strict=True)
    g = torch.Generator().manual_seed(404)
    x = torch.randn(2, 30, 1024, generator=g)
    pad = torch.zeros(2, 30, dtype=torch.bool)
    pad[1, 21:] = True
    with torch.no_grad():
        y = enc(x.clone(), attention_mask=~pad).last_hidden_state  # bool mask: HF indexes with ~mask
    np.savez_compressed(os.path.join(OUT, "hubert_standin.npz"), seed=14, layers=L, x=x.numpy(), lens=np.array([30, 21]),
                        out=y.numpy())
    return missing


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["frontend", "frontend_swish", "raven", "conformer", "vocoder", "vocoder_lrs3", "hubert_standin"]
    for w in which:
        print("making", w, flush=True)
        globals()["make_" + w]()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
