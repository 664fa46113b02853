"""Oracle for the conformer half of stage 1: Conformer.forward (multi_target_lip2speech/model_avhubert.py:249-297) over
the vendored ESPnet encoder (espnet/nets/pytorch_backend/transformer/{encoder,encoder_layer,attention,embedding,
convolution,positionwise_feed_forward,layer_norm}.py)."""
import math

import torch
import torch.nn.functional as F


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-12)  # layer_norm.py:21


def rel_pos_table(T, d_model):
    """RelPositionalEncoding.extend_pe + forward slice (embedding.py:172-217): returns pos_emb [1, 2T-1, d];
    row k encodes relative position (T-1-k)."""
    rel = torch.arange(T - 1, -T, -1, dtype=torch.float32).unsqueeze(1)  # T-1 ... -(T-1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(2 * T - 1, d_model)
    pe[:, 0::2] = torch.sin(rel * div)
    pe[:, 1::2] = torch.cos(rel * div)
    return pe.unsqueeze(0)


def rel_shift(x):
    """attention.py:218-238."""
    b, h, t1, n = x.shape
    xp = torch.cat([x.new_zeros(b, h, t1, 1), x], dim=-1).view(b, h, n + 1, t1)
    return xp[:, :, 1:].view_as(x)[:, :, :, : n // 2 + 1]


def rel_mha(sd, p, x, pos_emb, mask, heads):
    """RelPositionMultiHeadedAttention.forward attention.py:240-280 + forward_attention :59-90.  mask: [B,1,T] bool."""
    B, T, C = x.shape
    d = C // heads
    q = _lin(sd, p + ".linear_q", x).view(B, T, heads, d)
    k = _lin(sd, p + ".linear_k", x).view(B, T, heads, d).transpose(1, 2)
    v = _lin(sd, p + ".linear_v", x).view(B, T, heads, d).transpose(1, 2)
    pp = F.linear(pos_emb, sd[p + ".linear_pos.weight"]).view(pos_emb.size(0), -1, heads, d).transpose(1, 2)
    qu = (q + sd[p + ".pos_bias_u"]).transpose(1, 2)
    qv = (q + sd[p + ".pos_bias_v"]).transpose(1, 2)
    ac = qu @ k.transpose(-2, -1)
    bd = rel_shift(qv @ pp.transpose(-2, -1))
    scores = (ac + bd) / math.sqrt(d)
    m = mask.unsqueeze(1).eq(0)
    scores = scores.masked_fill(m, torch.finfo(scores.dtype).min)
    attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
    o = (attn @ v).transpose(1, 2).contiguous().view(B, T, C)
    return _lin(sd, p + ".linear_out", o)


def conv_module(sd, p, x):
    """ConvolutionModule.forward convolution.py:53-65 (BatchNorm1d in eval mode)."""
    x = x.transpose(1, 2)
    x = F.conv1d(x, sd[p + ".pointwise_cov1.weight"], sd[p + ".pointwise_cov1.bias"])
    x = F.glu(x, dim=1)
    C = x.shape[1]
    k = sd[p + ".depthwise_conv.weight"].shape[-1]
    x = F.conv1d(x, sd[p + ".depthwise_conv.weight"], sd[p + ".depthwise_conv.bias"], padding=(k - 1) // 2, groups=C)
    x = F.batch_norm(x, sd[p + ".norm.running_mean"], sd[p + ".norm.running_var"], sd[p + ".norm.weight"],
                     sd[p + ".norm.bias"], False, 0.0, 1e-5)
    x = x * torch.sigmoid(x)
    x = F.conv1d(x, sd[p + ".pointwise_cov2.weight"], sd[p + ".pointwise_cov2.bias"])
    return x.transpose(1, 2)


def ffn(sd, p, x):
    """positionwise_feed_forward.py:28-30 (ReLU)."""
    return _lin(sd, p + ".w_2", torch.relu(_lin(sd, p + ".w_1", x)))


def encoder_layer(sd, p, x, pos_emb, mask, heads):
    """EncoderLayer.forward encoder_layer.py:75-149 (normalize_before, macaron, cnn module, no concat_after)."""
    x = x + 0.5 * ffn(sd, p + ".feed_forward_macaron", _ln(sd, p + ".norm_ff_macaron", x))
    x = x + rel_mha(sd, p + ".self_attn", _ln(sd, p + ".norm_mha", x), pos_emb, mask, heads)
    x = x + conv_module(sd, p + ".conv_module", _ln(sd, p + ".norm_conv", x))
    x = x + 0.5 * ffn(sd, p + ".feed_forward", _ln(sd, p + ".norm_ff", x))
    return _ln(sd, p + ".norm_final", x)


def espnet_encoder_after_frontend(sd, p, x, masks, layers=12, heads=8, taps=None):
    """Encoder.forward_after_frontend encoder.py:285-306 with embed = Linear(512,d)+RelPositionalEncoding :152-156."""
    x = _lin(sd, p + ".embed.0", x)
    d = x.shape[-1]
    x = x * math.sqrt(d)                       # embedding.py:211
    pos_emb = rel_pos_table(x.shape[1], d)
    for i in range(layers):
        x = encoder_layer(sd, f"{p}.encoders.{i}", x, pos_emb, masks, heads)
        if taps is not None:
            taps[f"block{i}"] = x
    return _ln(sd, p + ".after_norm", x), masks


def _heads(sd, p, x, masks, spk_emb, taps=None):
    """mel + unit heads of Conformer.forward (model_avhubert.py:266-292 == model.py:256-285)."""
    if taps is not None:
        taps["head_in"] = x                                                      # [B,2T,d]: what proj_out / mel_conv read
    padding_mask = ~masks.squeeze(-2)
    assert spk_emb.size(-1) == 256                                               # :268
    spk_x = torch.cat([spk_emb.unsqueeze(1).repeat(1, x.size(1), 1), x], dim=-1)  # :269
    h = spk_x.transpose(1, 2)
    for i in (0, 3, 6):                                                          # mel_conv :231-241 (dropout = id)
        h = F.gelu(F.conv1d(h, sd[f"{p}.mel_conv.{i}.weight"], sd[f"{p}.mel_conv.{i}.bias"], padding=1))
    mel = _lin(sd, p + ".mel_proj", h.transpose(1, 2))                           # :273
    B, T, D = mel.shape
    mel = mel.reshape(B, T, D // 2, 2).transpose(-1, -2).reshape(B, T * 2, D // 2)  # :276
    unit = _lin(sd, p + ".proj_out", x.transpose(0, 1))                          # :280-285  [2T,B,V]
    return {"encoder_out": unit, "encoder_padding_mask": padding_mask, "padding_mask": padding_mask,
            "encoder_out_mel": mel}


def conformer_forward(sd, source, padding_mask, spk_emb, layers=12, heads=8, p="conformer", taps=None):
    """Conformer.forward model_avhubert.py:249-297.  source [2T,B,1024], padding_mask [B,2T] bool, spk_emb [B,256]."""
    x = source.transpose(0, 1)
    x = _lin(sd, p + ".proj_in", x)                                              # :257-258
    x, masks = espnet_encoder_after_frontend(sd, p + ".encoder", x, ~padding_mask.unsqueeze(-2), layers, heads, taps)
    return _heads(sd, p, x, masks, spk_emb, taps)


def multi_target_forward(sd, video, padding_mask, spk_emb, layers=12, heads=8, p="encoder", taps=None):
    """`multi_target` Conformer.forward multi_target_lip2speech/model.py:238-285: Conv3dResNet (Swish) frontend on
    source['video'] [B,1,T,88,88], x2 time repeat (:244-245), no proj_in (d == 512, :216-219), ESPnet encoder, heads."""
    from . import frontend
    fsd = {k[len(p) + len(".encoder.frontend."):]: v for k, v in sd.items() if k.startswith(p + ".encoder.frontend.")}
    x = frontend.conv3d_resnet(fsd, video.squeeze(1))                            # :242
    x = x.repeat_interleave(2, dim=1)
    pm2 = padding_mask.repeat_interleave(2, dim=1)
    x, masks = espnet_encoder_after_frontend(sd, p + ".encoder", x, ~pm2.unsqueeze(-2), layers, heads, taps)
    return _heads(sd, p, x, masks, spk_emb, taps)


def auto_avsr_forward(sd, video, padding_mask, spk_emb, enc_layers=12, enc_heads=12, layers=12, heads=8, taps=None):
    """`multi_target_auto_avsr` (multi_target_lip2speech/model_auto_avsr.py:72-86,133-152): ESPnet Encoder.forward WITH its
    Conv3dResNet frontend (encoder.py:230-259) at d = 768, then the conformer head on the x2-repeated output (proj_in :181)."""
    from . import frontend
    p = "encoder.encoder"
    fsd = {k[len(p) + len(".frontend."):]: v for k, v in sd.items() if k.startswith(p + ".frontend.")}
    x = frontend.conv3d_resnet(fsd, video.squeeze(1))
    x, _ = espnet_encoder_after_frontend(sd, p, x, ~padding_mask.unsqueeze(-2), enc_layers, enc_heads, taps)
    return _head_stack(sd, x, padding_mask, spk_emb, layers, heads, taps)


def _head_stack(sd, x, padding_mask, spk_emb, layers, heads, taps):
    """The conformer head on the x2-repeated encoder output; of its intermediates only "head_in" reaches `taps` (the block
    names would collide with the encoder's)."""
    ht = {} if taps is not None else None
    out = conformer_forward(sd, x.transpose(0, 1).repeat_interleave(2, dim=0), padding_mask.repeat_interleave(2, dim=1), spk_emb,
                            layers, heads, taps=ht)
    if taps is not None:
        taps["head_in"] = ht["head_in"]
    return out


def raven_encoder_after_frontend(sd, p, x, masks, layers=24, heads=16, taps=None):
    """raven/_espnet/.../transformer/encoder.py:260-327 after the frontend, as model_raven.py:107-132 configures it:
    embed = Linear + RelPositionalEncoding ('vanilla_linear'), blocks of encoder_layer.py:175-243 (normalize_before, no macaron,
    no conv module, layerscale, ff_bn_pre: x += gamma_mha * RelMHA(LN(x)); x += gamma_ff * FFN(BatchNorm1d(x))), after_norm.
    The attention / embedding / feed-forward modules of that vendored copy are identical to espnet/'s."""
    x = _lin(sd, p + ".embed.0", x)
    d = x.shape[-1]
    x = x * math.sqrt(d)
    pos_emb = rel_pos_table(x.shape[1], d)
    for i in range(layers):
        lp = f"{p}.encoders.{i}"
        x = x + sd[lp + ".gamma_mha"] * rel_mha(sd, lp + ".self_attn", _ln(sd, lp + ".norm_mha", x), pos_emb, masks, heads)
        h = F.batch_norm(x.transpose(1, 2), sd[lp + ".norm_ff.running_mean"], sd[lp + ".norm_ff.running_var"],
                         sd[lp + ".norm_ff.weight"], sd[lp + ".norm_ff.bias"], False, 0.0, 1e-5).transpose(1, 2)
        x = x + sd[lp + ".gamma_ff"] * ffn(sd, lp + ".feed_forward", h)
        if taps is not None:
            taps[f"block{i}"] = x
    return _ln(sd, p + ".after_norm", x), masks


def raven_forward(sd, video, padding_mask, spk_emb, enc_layers=24, enc_heads=16, layers=12, heads=8, taps=None):
    """`multi_target_raven` (multi_target_lip2speech/model_raven.py:77-91,134-152)."""
    from . import frontend
    p = "encoder.encoder"
    fsd = {k[len(p) + len(".frontend."):]: v for k, v in sd.items() if k.startswith(p + ".frontend.")}
    x = frontend.conv3d_resnet(fsd, video.squeeze(1))
    x, _ = raven_encoder_after_frontend(sd, p, x, ~padding_mask.unsqueeze(-2), enc_layers, enc_heads, taps)
    return _head_stack(sd, x, padding_mask, spk_emb, layers, heads, taps)
