"""Oracle for stage 2: MelCodeGenerator.forward (multi_input_vocoder/models_multi_input.py:60-97) over Generator.forward
and ResBlock1.forward (speech-resynthesis/models.py:98-114, :34-41), plus the int16 conversion of
multi_input_vocoder/inference.py:79-81."""
import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # models.py:13


def _w(sd, p):
    """remove_weight_norm()'d weight, or g*v/||v|| (weight_norm dim=0) when the state dict still carries weight_g/v."""
    if p + ".weight" in sd:
        return sd[p + ".weight"]
    g, v = sd[p + ".weight_g"], sd[p + ".weight_v"]
    return v * (g / v.pow(2).sum(dim=tuple(range(1, v.dim())), keepdim=True).sqrt())


def get_padding(k, d=1):
    return int((k * d - d) / 2)  # speech-resynthesis/utils.py:44-45


def resblock1(sd, p, x, k, dilations=(1, 3, 5)):
    for i, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, _w(sd, f"{p}.convs1.{i}"), sd[f"{p}.convs1.{i}.bias"], 1, get_padding(k, d), d)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, _w(sd, f"{p}.convs2.{i}"), sd[f"{p}.convs2.{i}.bias"], 1, get_padding(k, 1), 1)
        x = xt + x
    return x


def generator(sd, x, h, taps=None):
    """Generator.forward models.py:98-114."""
    nk = len(h["resblock_kernel_sizes"])
    x = F.conv1d(x, _w(sd, "conv_pre"), sd["conv_pre.bias"], 1, 3)
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, _w(sd, f"ups.{i}"), sd[f"ups.{i}.bias"], u, (k - u) // 2)
        xs = None
        for j, (rk, rd) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            r = resblock1(sd, f"resblocks.{i * nk + j}", x, rk, rd)
            xs = r if xs is None else xs + r
        x = xs / nk
        if taps is not None:
            taps[f"stage{i}"] = x
    x = F.leaky_relu(x)  # default slope 0.01 (models.py:110)
    x = F.conv1d(x, _w(sd, "conv_post"), sd["conv_post.bias"], 1, 3)
    return torch.tanh(x)


def mel_code_generator(sd, h, code, mel, spkr, taps=None):
    """MelCodeGenerator.forward models_multi_input.py:60-97 (multispkr + embedder_dim, no text labels).
    code [B,L] long, mel [B,80,2L], spkr [B,256] -> [B,1,320L]."""
    c = F.embedding(code, sd["dict.weight"])                                         # :67
    c = F.gelu(F.conv_transpose1d(c.permute(0, 2, 1), sd["layer.0.weight"], sd["layer.0.bias"], 2, 1)).permute(0, 2, 1)
    c = F.linear(c, sd["fc.weight"], sd["fc.bias"]).permute(0, 2, 1)                 # :70-71
    x = torch.cat([mel, c], dim=1)                                                   # :73
    s = F.linear(spkr, sd["spkr.weight"], sd["spkr.bias"])                           # :80
    s = s.unsqueeze(2).repeat(1, 1, x.shape[-1])                                     # _upsample models.py:158-177
    x = torch.cat([x, s], dim=1)                                                     # :82
    if taps is not None:
        taps["model_in"] = x
    return generator(sd, x, h, taps)


def to_int16(y):
    """multi_input_vocoder/inference.py:79-81: audio * MAX_WAV_VALUE then numpy astype('int16') (truncation)."""
    a = y.squeeze() * 32768.0
    return a.cpu().numpy().astype("int16")
