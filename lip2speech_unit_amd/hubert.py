"""AV-HuBERT encoder (inference part) on gfx950 — host-side mirror of avhubert/hubert.py + hubert_asr.py.

Keeps the reference's names and state_dict layout (`AVHubertModel` hubert.py:334-440, `SubModel` :317-332,
`extract_finetune` :694-745, `HubertEncoderWrapper` hubert_asr.py:375-409; the fairseq `TransformerEncoder` /
`TransformerSentenceEncoderLayer` / `MultiheadAttention` parameter names the checkpoints carry).  nn layers only hold
parameters; the arithmetic is liblip2speech_hip.so: tap-GEMMs with fused bias/GELU/residual epilogues, wavefront
LayerNorm, fused masked attention.  Residual stream fp32 in HBM, GEMM operands 16-bit, fp32 accumulation.
"""
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_GELU, F_DUAL, F_MASK, F_RES_POST, MODE_CONV1D
from .resnet import ResEncoder


@dataclass
class AVHubertConfig:
    """Subset of avhubert/hubert.py:64-315 that shapes inference; defaults = conf/pretrain/large_vox_iter5.yaml:88-101."""
    encoder_layers: int = 24
    encoder_embed_dim: int = 1024
    encoder_ffn_embed_dim: int = 4096
    encoder_attention_heads: int = 16
    conv_pos: int = 128
    conv_pos_groups: int = 16
    layer_norm_first: bool = True
    audio_feat_dim: int = 104
    modality_fuse: str = "concat"
    resnet_relu_type: str = "prelu"
    sub_encoder_layers: int = 0

    @classmethod
    def from_w2v_args(cls, w2v_args):
        """From the pre-training config a fine-tuned checkpoint embeds as `cfg.model.w2v_args` (model_avhubert.py:71-84,:98:
        `task_pretrain.build_model(w2v_args.model)`); fields absent there keep the large_vox_iter5 defaults."""
        from .plugin import cfg_get
        m = cfg_get(w2v_args, "model", None)
        if m is None:
            # an old-style checkpoint's `args` is ONE flat Namespace (the reference converts it with
            # convert_namespace_to_omegaconf, model_avhubert.py:79-81): the model fields sit on it directly.  Anything else
            # (no `model` group and none of the encoder's fields) cannot size the encoder - refuse instead of guessing "large".
            if any(cfg_get(w2v_args, k, None) is not None for k in ("encoder_layers", "encoder_embed_dim")):
                m = w2v_args
            else:
                raise ValueError("w2v_args carries neither a `model` group nor flat encoder fields (encoder_layers, encoder_embed_dim)")
        c = cls()
        for k, v in list(vars(c).items()):
            got = cfg_get(m, k, v)
            if isinstance(v, bool):       # bool("False") is True: parse the strings a yaml / argparse round trip leaves
                if isinstance(got, str):
                    if got.strip().lower() not in ("true", "false", "1", "0"):
                        raise ValueError(f"w2v_args.model.{k}: cannot read {got!r} as a bool")
                    got = got.strip().lower() in ("true", "1")
                setattr(c, k, bool(got))
            else:
                setattr(c, k, type(v)(got))
        return c


class MultiheadAttention(nn.Module):
    """Parameter holder with fairseq's names: {q,k,v,out}_proj."""

    def __init__(self, dim, heads):
        super().__init__()
        self.embed_dim, self.num_heads = dim, heads
        self.k_proj = nn.Linear(dim, dim)
        self.v_proj = nn.Linear(dim, dim)
        self.q_proj = nn.Linear(dim, dim)
        self.out_proj = nn.Linear(dim, dim)


class TransformerSentenceEncoderLayer(nn.Module):
    def __init__(self, dim, ffn, heads):
        super().__init__()
        self.self_attn = MultiheadAttention(dim, heads)
        self.self_attn_layer_norm = nn.LayerNorm(dim, eps=1e-5)
        self.fc1 = nn.Linear(dim, ffn)
        self.fc2 = nn.Linear(ffn, dim)
        self.final_layer_norm = nn.LayerNorm(dim, eps=1e-5)


class _WeightNormConv1d(nn.Module):
    """Conv1d under nn.utils.weight_norm(dim=2): parameters weight_g [1,1,k], weight_v [Cout,Cin/g,k], bias."""

    def __init__(self, cin, cout, k, groups):
        super().__init__()
        self.weight_g = nn.Parameter(torch.ones(1, 1, k))
        self.weight_v = nn.Parameter(torch.zeros(cout, cin // groups, k))
        self.bias = nn.Parameter(torch.zeros(cout))


class TransformerEncoder(nn.Module):
    """fairseq TransformerEncoder (called at hubert.py:739): pos_conv + N pre-LN layers + final layer_norm."""

    def __init__(self, cfg: AVHubertConfig, dtype=ops.F16):
        super().__init__()
        d = cfg.encoder_embed_dim
        self.cfg = cfg
        self.pos_conv = nn.Sequential(_WeightNormConv1d(d, d, cfg.conv_pos, cfg.conv_pos_groups))
        self.layers = nn.ModuleList(
            [TransformerSentenceEncoderLayer(d, cfg.encoder_ffn_embed_dim, cfg.encoder_attention_heads)
             for _ in range(cfg.encoder_layers)])
        self.layer_norm = nn.LayerNorm(d, eps=1e-5)
        self.dtype = dtype
        self._packed = None
        if not cfg.layer_norm_first:
            raise NotImplementedError("only layer_norm_first=True (AV-HuBERT large) is built")

    def pack(self, dev):
        t16 = ops.torch_dtype(self.dtype)
        cfg = self.cfg
        d, G, k = cfg.encoder_embed_dim, cfg.conv_pos_groups, cfg.conv_pos
        pc = self.pos_conv[0]
        g, v = pc.weight_g.detach().float(), pc.weight_v.detach().float()
        w = v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())  # weight_norm dim=2
        cg = d // G
        P = {"pc_w": w.view(G, cg, cg, k).permute(0, 1, 3, 2).reshape(G, cg, k * cg).to(dev, t16).contiguous(),
             "pc_b": pc.bias.detach().float().to(dev).contiguous(), "layers": []}
        scale = (d // cfg.encoder_attention_heads) ** -0.5
        for L in self.layers:
            a = L.self_attn
            wq = a.q_proj.weight.detach().float() * scale
            bq = a.q_proj.bias.detach().float() * scale
            e = {
                "wqkv": torch.cat([wq, a.k_proj.weight.detach().float(), a.v_proj.weight.detach().float()], 0)
                .to(dev, t16).contiguous(),
                "bqkv": torch.cat([bq, a.k_proj.bias.detach().float(), a.v_proj.bias.detach().float()], 0)
                .to(dev).contiguous(),
                "wo": a.out_proj.weight.detach().to(dev, t16).contiguous(),
                "bo": a.out_proj.bias.detach().float().to(dev).contiguous(),
                "w1": L.fc1.weight.detach().to(dev, t16).contiguous(),
                "b1": L.fc1.bias.detach().float().to(dev).contiguous(),
                "w2": L.fc2.weight.detach().to(dev, t16).contiguous(),
                "b2": L.fc2.bias.detach().float().to(dev).contiguous(),
                "ln1": (L.self_attn_layer_norm.weight.detach().float().to(dev).contiguous(),
                        L.self_attn_layer_norm.bias.detach().float().to(dev).contiguous()),
                "ln2": (L.final_layer_norm.weight.detach().float().to(dev).contiguous(),
                        L.final_layer_norm.bias.detach().float().to(dev).contiguous()),
            }
            P["layers"].append(e)
        P["lnf"] = (self.layer_norm.weight.detach().float().to(dev).contiguous(),
                    self.layer_norm.bias.detach().float().to(dev).contiguous())
        self._packed = P

    def forward_rows(self, x32, x16, lens, B, T):
        """x32/x16: [B*T, d] fp32 / 16-bit copies of the input with padded rows already zeroed.  Returns fp32 [B*T, d]."""
        dev = x32.device
        if self._packed is None or self._packed["pc_b"].device != dev:
            self.pack(dev)
        P, dt, cfg = self._packed, self.dtype, self.cfg
        t16 = ops.torch_dtype(dt)
        d, H, F = cfg.encoder_embed_dim, cfg.encoder_attention_heads, cfg.encoder_ffn_embed_dim
        G, k = cfg.conv_pos_groups, cfg.conv_pos
        cg = d // G
        M = B * T
        x = torch.empty(M, d, device=dev, dtype=torch.float32)
        # x = x + GELU(pos_conv(x)); SamePad drops the trailing step of the even kernel, i.e. taps reach t-64 .. t+63
        ops.tapgemm(x16, P["pc_w"], x, M=M, N=cg, Cin=cg, ntaps=k, lda=d, ldc=d, mode=MODE_CONV1D, T_out=T, T_in=T,
                    stride=1, dil=1, off=-(k // 2), bias=P["pc_b"], act=ACT_GELU, R=x32, ldr=d, flags=F_RES_POST,
                    dtype=dt, groups=G, a_gstride=cg, c_gstride=cg, w_gstride=cg * k * cg)
        h = torch.empty(M, d, device=dev, dtype=t16)
        qkv = torch.empty(M, 3 * d, device=dev, dtype=t16)
        att = torch.empty(M, d, device=dev, dtype=t16)
        f = torch.empty(M, F, device=dev, dtype=t16)
        # pre-LN layers (hubert.py:739-743): every residual update is followed by a LayerNorm - the layer's own ln2, then the next
        # layer's ln1 (the encoder's final layer_norm behind the last layer) - which ops.residual_linear applies behind the update
        # (at one-clip M inside the split-K reduction's launch)
        out = torch.empty(M, d, device=dev, dtype=torch.float32)
        layers = P["layers"]
        if not layers:
            ops.layernorm(x, P["lnf"][0], P["lnf"][1], 1e-5, out, M=M, C=d, dtype=dt)
            return out
        ops.layernorm(x, layers[0]["ln1"][0], layers[0]["ln1"][1], 1e-5, h, M=M, C=d, dtype=dt)
        for li, e in enumerate(layers):
            ops.tapgemm(h, e["wqkv"], qkv, M=M, N=3 * d, Cin=d, bias=e["bqkv"], dtype=dt)
            ops.attention(qkv, att, B=B, T=T, H=H, lens=lens, len_mul=1, dtype=dt)
            ops.residual_linear(att, e["wo"], e["bo"], x, M=M, N=d, K=d, dtype=dt, cache=e, key="wo",
                                ln=(e["ln2"][0], e["ln2"][1], 1e-5, h))
            ops.tapgemm(h, e["w1"], f, M=M, N=F, Cin=d, bias=e["b1"], act=ACT_GELU, dtype=dt)
            last = li + 1 == len(layers)
            nxt = P["lnf"] if last else layers[li + 1]["ln1"]
            ops.residual_linear(f, e["w2"], e["b2"], x, M=M, N=d, K=F, dtype=dt, cache=e, key="w2",
                                ln=(nxt[0], nxt[1], 1e-5, out if last else h))
        return out


class SubModel(nn.Module):
    """avhubert/hubert.py:317-332 with sub_encoder_layers == 0 (encoder=None)."""

    def __init__(self, resnet=None, input_dim=None, cfg: AVHubertConfig = None):
        super().__init__()
        self.resnet = resnet
        self.proj = nn.Linear(input_dim, cfg.encoder_embed_dim)
        if cfg.sub_encoder_layers > 0:
            raise NotImplementedError("sub_encoder_layers > 0 is not used by AV-HuBERT large")
        self.encoder = None


class AVHubertModel(nn.Module):
    """Inference subset of avhubert/hubert.py:334-440: video-only extract_finetune."""

    def __init__(self, cfg: AVHubertConfig = None, dtype=ops.F16):
        super().__init__()
        cfg = cfg or AVHubertConfig()
        self.cfg = cfg
        d = cfg.encoder_embed_dim
        resnet = ResEncoder(relu_type=cfg.resnet_relu_type, weights=None, dtype=dtype)
        self.feature_extractor_audio = SubModel(resnet=None, input_dim=cfg.audio_feat_dim, cfg=cfg)
        self.feature_extractor_video = SubModel(resnet=resnet, input_dim=resnet.backend_out, cfg=cfg)
        if cfg.modality_fuse != "concat":
            raise NotImplementedError("modality_fuse='add' is not used by the lip2speech checkpoints")
        self.embed = 2 * d
        self.encoder_embed_dim = d
        self.post_extract_proj = nn.Linear(self.embed, d)
        self.mask_emb = nn.Parameter(torch.zeros(cfg.audio_feat_dim))  # kept for checkpoint key parity (unused)
        self.encoder = TransformerEncoder(cfg, dtype=dtype)
        self.layer_norm = nn.LayerNorm(self.embed, eps=1e-5)
        self.dtype = dtype
        self._packed = None

    def pack(self, dev):
        t16 = ops.torch_dtype(self.dtype)
        fv = self.feature_extractor_video
        self._packed = {
            "wp": fv.proj.weight.detach().to(dev, t16).contiguous(),
            "bp": fv.proj.bias.detach().float().to(dev).contiguous(),
            "ln": (self.layer_norm.weight.detach().float().to(dev).contiguous(),
                   self.layer_norm.bias.detach().float().to(dev).contiguous()),
            "wpe": self.post_extract_proj.weight.detach().to(dev, t16).contiguous(),
            "bpe": self.post_extract_proj.bias.detach().float().to(dev).contiguous(),
        }
        fv.resnet.pack(dev)
        self.encoder.pack(dev)

    def repack(self):
        self._packed = None
        self.encoder._packed = None
        self.feature_extractor_video.resnet._packed = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.repack()
        return r

    @staticmethod
    def lens_from_padding_mask(padding_mask, B, T, dev):
        if padding_mask is not None:
            padding_mask = padding_mask.to(torch.bool).contiguous()      # no-ops for the collater's bool mask
        return ops.lens_from_mask(padding_mask, B, T, dev)                # one launch, no ATen kernels inside a captured step

    def forward_padding_mask(self, features, padding_mask):
        """hubert.py:564-574 — identity when the mask is already at the feature rate (always true on this path)."""
        return padding_mask

    def extract_rows(self, video, padding_mask):
        """Hot path: video [B,1,T,88,88], padding_mask [B,T] bool or None -> (fp32 [B*T, d] rows (b,t), lens int32 [B])."""
        dev = video.device
        if self._packed is None or self._packed["bp"].device != dev:
            self.pack(dev)
        P, dt = self._packed, self.dtype
        t16 = ops.torch_dtype(dt)
        d = self.encoder_embed_dim
        feat, B, T = self.feature_extractor_video.resnet.forward_rows(video)         # hubert.py:707 -> :324-326
        M = B * T
        lens = self.lens_from_padding_mask(padding_mask, B, T, dev)
        fv = torch.empty(M, d, device=dev, dtype=torch.float32)
        ops.tapgemm(feat, P["wp"], fv, M=M, N=d, Cin=feat.shape[1], bias=P["bp"], dtype=dt)  # :327
        fused = torch.empty(M, 2 * d, device=dev, dtype=t16)
        ops.layernorm(fv, P["ln"][0], P["ln"][1], 1e-5, fused, M=M, C=d, zero_prefix=d, dtype=dt)  # :708-720
        x32 = torch.empty(M, d, device=dev, dtype=torch.float32)
        x16 = torch.empty(M, d, device=dev, dtype=t16)
        # post_extract_proj :727; rows of padded frames are zeroed here (TransformerEncoder: x[padding_mask] = 0)
        ops.tapgemm(fused, P["wpe"], x32, M=M, N=d, Cin=2 * d, bias=P["bpe"], C2=x16, ldc2=d, lens=lens, mask_T=T,
                    mask_mul=1, flags=F_MASK | F_DUAL, slope2=1.0, dtype=dt)
        out = self.encoder.forward_rows(x32, x16, lens, B, T)                         # :739
        return out, lens, B, T

    def extract_finetune(self, source, padding_mask=None, mask=False, ret_conv=False, output_layer=None):
        """hubert.py:694-745 (eval, video only): returns (x [B,T,C] fp32, padding_mask)."""
        if source.get("audio") is not None:
            raise NotImplementedError("audio modality is outside the lip2speech inference path (modalities=['video'])")
        if mask or ret_conv or output_layer is not None:
            raise NotImplementedError("mask / ret_conv / output_layer are training-time options")
        out, lens, B, T = self.extract_rows(source["video"], padding_mask)
        return out.view(B, T, -1), padding_mask

    def remove_pretraining_modules(self):
        pass


class HubertEncoderWrapper(nn.Module):
    """avhubert/hubert_asr.py:375-409."""

    def __init__(self, w2v_model):
        super().__init__()
        self.w2v_model = w2v_model

    def forward(self, source, padding_mask, **kwargs):
        x, padding_mask = self.w2v_model.extract_finetune(source=source, padding_mask=padding_mask)
        return {"encoder_out": x.transpose(0, 1), "encoder_padding_mask": padding_mask, "padding_mask": padding_mask}

    def forward_torchscript(self, net_input):
        return self.forward(**{k: v for k, v in net_input.items() if k in ("source", "padding_mask")})

    def reorder_encoder_out(self, encoder_out, new_order):
        if encoder_out["encoder_out"] is not None:
            encoder_out["encoder_out"] = encoder_out["encoder_out"].index_select(1, new_order)
        for k in ("encoder_padding_mask", "padding_mask"):
            if encoder_out[k] is not None:
                encoder_out[k] = encoder_out[k].index_select(0, new_order)
        return encoder_out

    def max_positions(self):
        return None
