"""Oracle for the AV-HuBERT encoder half of stage 1: HubertEncoderWrapper.forward -> AVHubertModel.extract_finetune
(avhubert/hubert_asr.py:380-394, avhubert/hubert.py:694-745) with the fairseq TransformerEncoder it calls (:739).

fairseq is NOT vendored in the reference (fairseq @ afc77bd, README.md:31-34) and is not installed: transformer_encoder()
restates its published algorithm [recalled semantics, SURVEY.md section 8 row a7] - "parity unpinned" against the reference
itself; tests cross-check it against HuggingFace's independent port (HubertEncoderStableLayerNorm).
"""
import math

import torch
import torch.nn.functional as F

from . import frontend


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(sd, p, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def pos_conv_weight(sd, p):
    """nn.utils.weight_norm(conv, name='weight', dim=2): w = g * v / ||v|| with the norm over dims (0,1)."""
    if p + ".weight" in sd:
        return sd[p + ".weight"]
    g, v = sd[p + ".weight_g"], sd[p + ".weight_v"]
    return v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())


def mha(sd, p, x, key_padding_mask, heads):
    """fairseq MultiheadAttention, self-attention, separate q/k/v/out projections with bias; q scaled by d_h^-0.5;
    padded keys -> -inf before a float32 softmax.  x: [B,T,C]."""
    B, T, C = x.shape
    d = C // heads
    q = _lin(sd, p + ".q_proj", x) * (d ** -0.5)
    k = _lin(sd, p + ".k_proj", x)
    v = _lin(sd, p + ".v_proj", x)
    q = q.view(B, T, heads, d).transpose(1, 2)
    k = k.view(B, T, heads, d).transpose(1, 2)
    v = v.view(B, T, heads, d).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    a = torch.softmax(s.float(), dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, T, C)
    return _lin(sd, p + ".out_proj", o)


def transformer_encoder(sd, p, x, padding_mask, layers=24, heads=16, conv_pos=128, conv_groups=16, taps=None):
    """fairseq TransformerEncoder.forward with layer_norm_first=True (conf/pretrain/large_vox_iter5.yaml:96-101):
    zero padded rows; x += GELU(pos_conv(x)[..., :-1]); 24 x pre-LN {MHSA, FFN(GELU)} ; final LayerNorm."""
    x = x.clone()
    if padding_mask is not None:
        x[padding_mask] = 0
    w = pos_conv_weight(sd, p + ".pos_conv.0")
    xc = F.conv1d(x.transpose(1, 2), w, sd[p + ".pos_conv.0.bias"], padding=conv_pos // 2, groups=conv_groups)
    if conv_pos % 2 == 0:
        xc = xc[:, :, :-1]  # SamePad
    x = x + F.gelu(xc).transpose(1, 2)
    if taps is not None:
        taps["pos_conv"] = x
    for i in range(layers):
        lp = f"{p}.layers.{i}"
        h = _ln(sd, lp + ".self_attn_layer_norm", x)
        x = x + mha(sd, lp + ".self_attn", h, padding_mask, heads)
        h = _ln(sd, lp + ".final_layer_norm", x)
        x = x + _lin(sd, lp + ".fc2", F.gelu(_lin(sd, lp + ".fc1", h)))
        if taps is not None:
            taps[f"layer{i}"] = x
    return _ln(sd, p + ".layer_norm", x)


def extract_finetune(sd, video, padding_mask, layers=24, heads=16, taps=None, p="w2v_model"):
    """AVHubertModel.extract_finetune, video-only (hubert.py:694-745) on top of SubModel.forward (:324-332).
    video: [B,1,T,88,88]; padding_mask: [B,T] bool (True = pad) -> (x [B,T,C], padding_mask)."""
    fsd = {k[len(p) + len(".feature_extractor_video.resnet."):]: v for k, v in sd.items()
           if k.startswith(p + ".feature_extractor_video.resnet.")}
    feat = frontend.res_encoder(fsd, video)                      # [B,512,T]
    fv = _lin(sd, p + ".feature_extractor_video.proj", feat.transpose(1, 2)).transpose(1, 2)  # [B,C,T]  :327
    fa = fv.new_zeros(fv.shape)                                  # :708  audio stream absent
    feats = torch.cat([fa, fv], dim=1).transpose(1, 2)           # :714 audio first, :719
    feats = _ln(sd, p + ".layer_norm", feats)                    # :720 LayerNorm(2C)
    if taps is not None:
        taps["fused_ln"] = feats
    x = _lin(sd, p + ".post_extract_proj", feats)                # :727
    if taps is not None:
        taps["post_extract_proj"] = x
    x = transformer_encoder(sd, p + ".encoder", x, padding_mask, layers, heads, taps=taps)  # :739
    return x, padding_mask


def encoder_wrapper(sd, video, padding_mask, **kw):
    """HubertEncoderWrapper.forward hubert_asr.py:380-394: returns encoder_out [T,B,C]."""
    x, pm = extract_finetune(sd, video, padding_mask, **kw)
    return {"encoder_out": x.transpose(0, 1), "encoder_padding_mask": pm, "padding_mask": pm}
