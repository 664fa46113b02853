#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own modules (CPU, fp32) on build-owned synthetic weights.

Runs only in the build container (needs /root/reference); the fixtures it writes are data (seed, shapes, inputs,
outputs) and travel with the repo - the reference itself never does.  Weights are not stored: both sides regenerate
them from parameter names/shapes with lip2speech_unit_amd.weights.synth_state_dict(seed).

  frontend.npz   avhubert/resnet.py ResEncoder('prelu')                 (loaded as a single file: the package imports fairseq)
  conformer.npz  espnet/nets/pytorch_backend/transformer/encoder.py Encoder.forward_after_frontend (12 x 512, rel_mha, macaron, cnn k=31)
  vocoder.npz    multi_input_vocoder/models_multi_input.py MelCodeGenerator (configs/lrs3/multi_input.json, weight norm removed)
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


def make_hubert_standin():
    from transformers import HubertConfig
    from transformers.models.hubert.modeling_hubert import HubertEncoderStableLayerNorm
    L = 3
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
    which = sys.argv[1:] or ["frontend", "conformer", "vocoder", "hubert_standin"]
    for w in which:
        print("making", w, flush=True)
        globals()["make_" + w]()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
