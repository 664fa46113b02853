"""Conformer + unit/mel heads on gfx950 — host-side mirror of multi_target_lip2speech/model_avhubert.py::Conformer
(:182-319) over the vendored ESPnet encoder (espnet/nets/pytorch_backend/transformer/encoder.py:55-306,
encoder_layer.py:16-149, attention.py:194-280, convolution.py:14-73, positionwise_feed_forward.py:12-30).

Same class / parameter names (state_dict layout `conformer.{proj_in, encoder.{embed.0, encoders.N...., after_norm},
proj_out, mel_conv.{0,3,6}, mel_proj}`); arithmetic in liblip2speech_hip.so.  Batched clips reproduce what the reference
computes for each clip alone (it runs batch_size=1, inference.py:161): padded keys are masked in attention and rows
beyond a clip's length are zeroed before every temporal conv (depthwise k=31, mel_conv k=3).
"""
import math
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_GELU, ACT_RELU, F_MASK, F_RES_POST, MODE_CONV1D


@dataclass
class ConformerConfig:
    """Fields of MultiTargetEncoderModelConfig used by Conformer.__init__ (model.py:32-63, lrs3 yaml values)."""
    conformer_embed_dim: int = 512
    conformer_attention_heads: int = 8
    conformer_ffn_embed_dim: int = 2048
    conformer_layers: int = 12
    cnn_module_kernel: int = 31
    encoder_embed_dim: int = 1024   # w2v_args.model.encoder_embed_dim
    decoder_embed_dim: int = 204    # len(tgt_dict), model_avhubert.py:112
    spk_dim: int = 256
    mel_dim: int = 160

    @classmethod
    def from_model_cfg(cls, cfg):
        """From a fairseq model config (MultiTargetEncoderModelConfig as saved in a checkpoint: DictConfig / dataclass /
        Namespace): the conformer_* fields Conformer.__init__ reads (model_avhubert.py:187-203); anything absent keeps its
        default."""
        from .plugin import cfg_get
        c = cls()
        for k in ("conformer_embed_dim", "conformer_attention_heads", "conformer_ffn_embed_dim", "conformer_layers"):
            setattr(c, k, int(cfg_get(cfg, k, getattr(c, k))))
        return c


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d, hidden):
        super().__init__()
        self.w_1 = nn.Linear(d, hidden)
        self.w_2 = nn.Linear(hidden, d)


class RelPositionMultiHeadedAttention(nn.Module):
    def __init__(self, heads, d):
        super().__init__()
        self.h, self.d_k = heads, d // heads
        self.linear_q = nn.Linear(d, d)
        self.linear_k = nn.Linear(d, d)
        self.linear_v = nn.Linear(d, d)
        self.linear_out = nn.Linear(d, d)
        self.linear_pos = nn.Linear(d, d, bias=False)
        self.pos_bias_u = nn.Parameter(torch.zeros(heads, d // heads))
        self.pos_bias_v = nn.Parameter(torch.zeros(heads, d // heads))


class ConvolutionModule(nn.Module):
    def __init__(self, ch, k):
        super().__init__()
        self.pointwise_cov1 = nn.Conv1d(ch, 2 * ch, 1)
        self.depthwise_conv = nn.Conv1d(ch, ch, k, padding=(k - 1) // 2, groups=ch)
        self.norm = nn.BatchNorm1d(ch)
        self.pointwise_cov2 = nn.Conv1d(ch, ch, 1)


class EncoderLayer(nn.Module):
    def __init__(self, d, heads, hidden, k):
        super().__init__()
        self.self_attn = RelPositionMultiHeadedAttention(heads, d)
        self.feed_forward = PositionwiseFeedForward(d, hidden)
        self.conv_module = ConvolutionModule(d, k)
        self.norm_ff = nn.LayerNorm(d, eps=1e-12)
        self.norm_mha = nn.LayerNorm(d, eps=1e-12)
        self.feed_forward_macaron = PositionwiseFeedForward(d, hidden)
        self.norm_ff_macaron = nn.LayerNorm(d, eps=1e-12)
        self.norm_conv = nn.LayerNorm(d, eps=1e-12)
        self.norm_final = nn.LayerNorm(d, eps=1e-12)


class RavenEncoderLayer(nn.Module):
    """raven/_espnet/nets/pytorch_backend/transformer/encoder_layer.py:66-256 as model_raven.py:107-132 builds it: no macaron
    branch, no conv module, rel-pos attention, layer-scale (gamma_mha, gamma_ff) and BatchNorm1d in front of the feed-forward
    (ff_bn_pre)."""

    def __init__(self, d, heads, hidden):
        super().__init__()
        self.self_attn = RelPositionMultiHeadedAttention(heads, d)
        self.feed_forward = PositionwiseFeedForward(d, hidden)
        self.norm_ff = nn.BatchNorm1d(d)
        self.norm_mha = nn.LayerNorm(d, eps=1e-12)
        self.gamma_ff = nn.Parameter(0.1 * torch.ones(d))
        self.gamma_mha = nn.Parameter(0.1 * torch.ones(d))


class Encoder(nn.Module):
    """ESPnet Encoder (encoder.py:55-306) with input_layer 'conv3d' (embed.0 Linear(512,d) + rel-pos), macaron conformer
    blocks (rel_mha, cnn module k=31), after_norm.  `frontend` is None for the conformer head (model_avhubert.py:206) and a
    Conv3dResNet for the `multi_target` / Auto-AVSR encoders."""

    def __init__(self, d, heads, hidden, blocks, k, idim=512, raven=False):
        super().__init__()
        self.frontend = None
        self.raven = raven   # RAVEn's transformer variant of the layer (see RavenEncoderLayer)
        self.embed = nn.Sequential(nn.Linear(idim, d))  # index 1 (RelPositionalEncoding) has no parameters
        self.encoders = nn.ModuleList([RavenEncoderLayer(d, heads, hidden) if raven else EncoderLayer(d, heads, hidden, k)
                                       for _ in range(blocks)])
        self.after_norm = nn.LayerNorm(d, eps=1e-12)
        self.d, self.heads, self.hidden, self.k = d, heads, hidden, k
        self._packed = None
        self._pos_cache = {}
        if d // heads != 64:
            raise NotImplementedError("the attention kernels are built for 64-dim heads (512/8, 768/12, 1024/16)")

    def pack(self, dev, dtype):
        t16 = ops.torch_dtype(dtype)
        d = self.d
        inv = 1.0 / math.sqrt(d // self.heads)  # attention.py:276, folded into q, u, v

        def w16(t):
            return t.detach().float().to(dev, t16).contiguous()

        P = {"layers": [], "dtype": dtype}
        P["w_emb"], P["b_emb"] = w16(self.embed[0].weight), _f32(self.embed[0].bias, dev)
        pos_w = []
        for L in self.encoders if self.raven else ():
            # layer-scale folded into the output projections, the eval-mode BatchNorm1d into the feed-forward's first Linear:
            # w_1(a*x + c) = (w_1 * a) x + (w_1 c + b_1)
            a, bn = L.self_attn, L.norm_ff
            g_m, g_f = L.gamma_mha.detach().float(), L.gamma_ff.detach().float()
            sc = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
            sh = bn.bias.detach().float() - bn.running_mean.detach().float() * sc
            w1 = L.feed_forward.w_1.weight.detach().float()
            e = {
                "ff": (w16(w1 * sc[None, :]), (w1 @ sh + L.feed_forward.w_1.bias.detach().float()).to(dev).contiguous(),
                       w16(L.feed_forward.w_2.weight.detach().float() * g_f[:, None]),
                       (L.feed_forward.w_2.bias.detach().float() * g_f).to(dev).contiguous()),
                "wqkv": w16(torch.cat([a.linear_q.weight.detach().float() * inv, a.linear_k.weight.detach().float(),
                                       a.linear_v.weight.detach().float()], 0)),
                "bqkv": torch.cat([a.linear_q.bias.detach().float() * inv, a.linear_k.bias.detach().float(),
                                   a.linear_v.bias.detach().float()], 0).to(dev).contiguous(),
                "u": (a.pos_bias_u.detach().float() * inv).to(dev).contiguous(),
                "v": (a.pos_bias_v.detach().float() * inv).to(dev).contiguous(),
                "wo": w16(a.linear_out.weight.detach().float() * g_m[:, None]),
                "bo": (a.linear_out.bias.detach().float() * g_m).to(dev).contiguous(),
                "n_mha": _ln(L.norm_mha, dev),
            }
            pos_w.append(a.linear_pos.weight.detach().float())
            P["layers"].append(e)
        for L in () if self.raven else self.encoders:
            a, c = L.self_attn, L.conv_module
            bn = c.norm
            sc = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
            sh = bn.bias.detach().float() - bn.running_mean.detach().float() * sc
            dw = c.depthwise_conv.weight.detach().float()[:, 0, :] * sc[:, None]       # [C,k]
            db = c.depthwise_conv.bias.detach().float() * sc + sh
            e = {
                "ffm": (w16(L.feed_forward_macaron.w_1.weight), _f32(L.feed_forward_macaron.w_1.bias, dev),
                        w16(L.feed_forward_macaron.w_2.weight), _f32(L.feed_forward_macaron.w_2.bias, dev)),
                "ff": (w16(L.feed_forward.w_1.weight), _f32(L.feed_forward.w_1.bias, dev),
                       w16(L.feed_forward.w_2.weight), _f32(L.feed_forward.w_2.bias, dev)),
                "wqkv": w16(torch.cat([a.linear_q.weight.detach().float() * inv, a.linear_k.weight.detach().float(),
                                       a.linear_v.weight.detach().float()], 0)),
                "bqkv": torch.cat([a.linear_q.bias.detach().float() * inv, a.linear_k.bias.detach().float(),
                                   a.linear_v.bias.detach().float()], 0).to(dev).contiguous(),
                "u": (a.pos_bias_u.detach().float() * inv).to(dev).contiguous(),
                "v": (a.pos_bias_v.detach().float() * inv).to(dev).contiguous(),
                "wo": w16(a.linear_out.weight), "bo": _f32(a.linear_out.bias, dev),
                "pw1": w16(c.pointwise_cov1.weight[:, :, 0]), "pb1": _f32(c.pointwise_cov1.bias, dev),
                "dw": dw.t().contiguous().to(dev), "db": db.to(dev).contiguous(),
                "pw2": w16(c.pointwise_cov2.weight[:, :, 0]), "pb2": _f32(c.pointwise_cov2.bias, dev),
                "n_ffm": _ln(L.norm_ff_macaron, dev), "n_mha": _ln(L.norm_mha, dev), "n_conv": _ln(L.norm_conv, dev),
                "n_ff": _ln(L.norm_ff, dev), "n_fin": _ln(L.norm_final, dev),
            }
            pos_w.append(a.linear_pos.weight.detach().float())
            P["layers"].append(e)
        P["w_pos"] = w16(torch.cat(pos_w, 0))                                           # [layers*d, d]
        P["n_after"] = _ln(self.after_norm, dev)
        self._packed = P
        self._pos_cache = {}
        return P

    def _pos_proj(self, T, dev):
        """linear_pos(pos_emb) for all layers at once (attention.py:257): [2T-1, layers*d] 16-bit, cached per T.  Building it
        copies a host table, which a hipGraph capture of a new bucket length could not do - and a captured graph keeps the
        raw device pointer of its entry, so entries are NEVER evicted (each is (2T-1) * layers * d * 2 bytes: 4.9 MB at
        T = 200, 29 MB at the service's 24-s limit); the cache is dropped only with the weights (pack / load_state_dict),
        which invalidates captured graphs anyway."""
        key = (T, str(dev))
        if key not in self._pos_cache:
            P, d = self._packed, self.d
            dt = P["dtype"]
            t16 = ops.torch_dtype(dt)
            pe = rel_pos_table(T, d, dev).to(t16).contiguous()
            nl = len(P["layers"])
            out = torch.empty(2 * T - 1, nl * d, device=dev, dtype=t16)
            ops.tapgemm(pe, P["w_pos"], out, M=2 * T - 1, N=nl * d, Cin=d, dtype=dt)
            self._pos_cache[key] = out
        return self._pos_cache[key]

    def forward_rows(self, xin, lens, B, T, len_mul, dtype):
        """forward_after_frontend (encoder.py:285-306) up to, not including, after_norm: xin [B*T, idim] 16-bit rows (b,t) ->
        the fp32 residual stream [B*T, d] after the last block.  lens: int32 [B], valid rows = lens*len_mul."""
        dev = xin.device
        if self._packed is None or self._packed["b_emb"].device != dev or self._packed["dtype"] != dtype:
            self.pack(dev, dtype)
        P, dt = self._packed, dtype
        t16 = ops.torch_dtype(dt)
        d, H, F, k = self.d, self.heads, self.hidden, self.k
        M = B * T
        nl = len(P["layers"])
        x = torch.empty(M, d, device=dev, dtype=torch.float32)
        # embed.0 then x * sqrt(d) (embedding.py:211)
        ops.tapgemm(xin, P["w_emb"], x, M=M, N=d, Cin=xin.shape[1], bias=P["b_emb"], alpha=math.sqrt(d), dtype=dt)
        pos = self._pos_proj(T, dev)
        h = torch.empty(M, d, device=dev, dtype=t16)
        f = torch.empty(M, F, device=dev, dtype=t16)
        qkv = torch.empty(M, 3 * d, device=dev, dtype=t16)
        att = torch.empty(M, d, device=dev, dtype=t16)
        glu_in = torch.empty(M, 2 * d, device=dev, dtype=t16)
        cv = torch.empty(M, d, device=dev, dtype=t16)

        def half_ffn(e, name, norm, after, y):
            """x += 0.5 * FFN(LayerNorm(x)) and y = `after`(x): the LayerNorm that follows the update (it rides in the split-K
            reduction's launch at one-clip M, ops.residual_linear)"""
            w = e[name]
            ops.layernorm(x, norm[0], norm[1], 1e-12, h, M=M, C=d, dtype=dt)
            ops.tapgemm(h, w[0], f, M=M, N=F, Cin=d, bias=w[1], act=ACT_RELU, dtype=dt)
            ops.residual_linear(f, w[2], w[3], x, M=M, N=d, K=F, alpha=0.5, dtype=dt, cache=e, key=name,
                                ln=(after[0], after[1], 1e-12, y))

        for li, e in enumerate(P["layers"] if self.raven else ()):
            # raven encoder_layer.py:175-243: x += gamma_mha * MHA(LN(x)) ; x += gamma_ff * FFN(BN(x))  (gamma / BN folded)
            ops.layernorm(x, e["n_mha"][0], e["n_mha"][1], 1e-12, h, M=M, C=d, dtype=dt)
            ops.tapgemm(h, e["wqkv"], qkv, M=M, N=3 * d, Cin=d, bias=e["bqkv"], dtype=dt)
            ops.attention(qkv, att, B=B, T=T, H=H, pos=pos[:, li * d:], ldp=nl * d, bias_u=e["u"], bias_v=e["v"],
                          lens=lens, len_mul=len_mul, dtype=dt)
            ops.tapgemm(att, e["wo"], x, M=M, N=d, Cin=d, bias=e["bo"], R=x, ldr=d, flags=F_RES_POST, dtype=dt)
            ops.cast_f32_to_16(x, h, M, d, dt)
            w = e["ff"]
            ops.tapgemm(h, w[0], f, M=M, N=F, Cin=d, bias=w[1], act=ACT_RELU, dtype=dt)
            ops.tapgemm(f, w[2], x, M=M, N=d, Cin=F, bias=w[3], R=x, ldr=d, flags=F_RES_POST, dtype=dt)
        for li, e in enumerate(() if self.raven else P["layers"]):
            half_ffn(e, "ffm", e["n_ffm"], e["n_mha"], h)                               # encoder_layer.py:89-95, norm_mha of :98
            ops.tapgemm(h, e["wqkv"], qkv, M=M, N=3 * d, Cin=d, bias=e["bqkv"], dtype=dt)
            ops.attention(qkv, att, B=B, T=T, H=H, pos=pos[:, li * d:], ldp=nl * d, bias_u=e["u"], bias_v=e["v"],
                          lens=lens, len_mul=len_mul, dtype=dt)
            ops.tapgemm(att, e["wo"], x, M=M, N=d, Cin=d, bias=e["bo"], R=x, ldr=d, flags=F_RES_POST, dtype=dt)
            ops.layernorm(x, e["n_conv"][0], e["n_conv"][1], 1e-12, h, M=M, C=d, dtype=dt)  # :124-130
            ops.tapgemm(h, e["pw1"], glu_in, M=M, N=2 * d, Cin=d, bias=e["pb1"], dtype=dt)
            ops.glu_dwconv_swish(glu_in, e["dw"], e["db"], cv, B=B, T=T, C=d, k=k, lens=lens, len_mul=len_mul, dtype=dt)
            ops.tapgemm(cv, e["pw2"], x, M=M, N=d, Cin=d, bias=e["pb2"], R=x, ldr=d, flags=F_RES_POST, dtype=dt)
            half_ffn(e, "ff", e["n_ff"], e["n_fin"], x)                                 # :133-138, norm_final of :140-141 (in place)
        return x


def _f32(t, dev):
    return t.detach().float().to(dev).contiguous()


def _ln(m, dev):
    return (_f32(m.weight, dev), _f32(m.bias, dev))


def rel_pos_table(T, d, dev):
    """embedding.py:172-217: pos_emb[1, 2T-1, d], row k <-> relative position T-1-k (sin on even, cos on odd dims)."""
    rel = torch.arange(T - 1, -T, -1, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(2 * T - 1, d)
    pe[:, 0::2] = torch.sin(rel * div)
    pe[:, 1::2] = torch.cos(rel * div)
    return pe.to(dev)


class Conformer(nn.Module):
    """model_avhubert.py:182-319."""

    def __init__(self, cfg: ConformerConfig = None, dtype=ops.F16):
        super().__init__()
        cfg = cfg or ConformerConfig()
        self.cfg = cfg
        d = cfg.conformer_embed_dim
        self.encoder = Encoder(d, cfg.conformer_attention_heads, cfg.conformer_ffn_embed_dim, cfg.conformer_layers,
                               cfg.cnn_module_kernel)
        self.proj_in = nn.Linear(cfg.encoder_embed_dim, d) if cfg.encoder_embed_dim != d else None
        self.proj_out = nn.Linear(d, cfg.decoder_embed_dim) if cfg.decoder_embed_dim != d else None
        self.mel_conv = nn.Sequential(
            nn.Conv1d(d + cfg.spk_dim, d, 3, 1, 1), nn.Dropout(0.0), nn.GELU(),
            nn.Conv1d(d, d, 3, 1, 1), nn.Dropout(0.0), nn.GELU(),
            nn.Conv1d(d, d, 3, 1, 1), nn.Dropout(0.0), nn.GELU())
        self.mel_proj = nn.Linear(d, cfg.mel_dim)
        self.dtype = dtype
        self._packed = None

    @property
    def _pos_cache(self):
        return self.encoder._pos_cache

    @_pos_cache.setter
    def _pos_cache(self, v):
        self.encoder._pos_cache = v

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._packed = None
        self.encoder._packed, self.encoder._pos_cache = None, {}
        return r

    def pack(self, dev):
        t16 = ops.torch_dtype(self.dtype)
        cfg = self.cfg

        def w16(t):
            return t.detach().float().to(dev, t16).contiguous()

        P = {}
        if self.proj_in is not None:
            P["w_in"], P["b_in"] = w16(self.proj_in.weight), _f32(self.proj_in.bias, dev)
        self.encoder.pack(dev, self.dtype)
        P["n_after"] = self.encoder._packed["n_after"]
        P["mel"] = []
        for i in (0, 3, 6):
            cv = self.mel_conv[i]
            P["mel"].append((w16(cv.weight.detach().permute(0, 2, 1).reshape(cv.weight.shape[0], -1)),
                             _f32(cv.bias, dev), cv.weight.shape[1]))
        # mel_proj rows permuted so column j*80+k of the output is feature 2k+j: the reshape/transpose of
        # model_avhubert.py:276 becomes a free view [B, 2T, 160] -> [B, 4T, 80]
        D = cfg.mel_dim
        perm = torch.tensor([2 * k + j for j in range(2) for k in range(D // 2)])
        P["w_mel"], P["b_mel"] = w16(self.mel_proj.weight.detach()[perm]), _f32(self.mel_proj.bias.detach()[perm], dev)
        if self.proj_out is not None:
            P["w_out"], P["b_out"] = w16(self.proj_out.weight), _f32(self.proj_out.bias, dev)
        P["dev_probe"] = P["b_mel"]
        self._packed = P

    def forward_rows(self, src16, lens, B, T, spk_emb, len_mul=2):
        """src16: [B*T, 1024] 16-bit rows (b,t) at the 50 Hz rate; lens: int32 [B] in video frames (len_mul=2).
        Returns (logits fp32 [B*T, V], mel fp32 [B*T, 160] == [B, 2T, 80], y16 [B*T, d])."""
        dev = src16.device
        if self._packed is None or self._packed["dev_probe"].device != dev:
            self.pack(dev)
        P, dt, cfg = self._packed, self.dtype, self.cfg
        t16 = ops.torch_dtype(dt)
        d = cfg.conformer_embed_dim
        M = B * T
        if self.proj_in is not None:
            xin = torch.empty(M, d, device=dev, dtype=t16)
            ops.tapgemm(src16, P["w_in"], xin, M=M, N=d, Cin=src16.shape[1], bias=P["b_in"], dtype=dt)  # :257-258
        else:
            xin = src16
        x = self.encoder.forward_rows(xin, lens, B, T, len_mul, dt)                      # :259-265 forward_after_frontend
        # after_norm (encoder.py:303-304): y16 feeds proj_out; a masked copy lands in the mel-head concat buffer
        y16 = torch.empty(M, d, device=dev, dtype=t16)
        cat = torch.empty(M, d + cfg.spk_dim, device=dev, dtype=t16)
        ops.layernorm(x, P["n_after"][0], P["n_after"][1], 1e-12, y16, M=M, C=d, dtype=dt)
        ops.layernorm(x, P["n_after"][0], P["n_after"][1], 1e-12, cat[:, cfg.spk_dim:], M=M, C=d, ldy=d + cfg.spk_dim,
                      lens=lens, len_mul=len_mul, mask_T=T, dtype=dt)
        assert spk_emb.size(-1) == cfg.spk_dim                                           # model_avhubert.py:268
        spk = spk_emb.contiguous()
        if spk.dtype not in (torch.float32, t16):
            spk = spk.float()
        ops.broadcast_rows(spk, cat, B=B, T=T, C=cfg.spk_dim, ldy=d + cfg.spk_dim, col0=0, lens=lens, len_mul=len_mul,
                           dtype=dt)                                                     # :269
        cur = cat
        for (w, b, cin) in P["mel"]:                                                     # :231-241, :273
            nxt = torch.empty(M, d, device=dev, dtype=t16)
            ops.tapgemm(cur, w, nxt, M=M, N=d, Cin=cin, ntaps=3, mode=MODE_CONV1D, T_out=T, T_in=T, stride=1, dil=1,
                        off=-1, bias=b, act=ACT_GELU, lens=lens, mask_T=T, mask_mul=len_mul, flags=F_MASK, dtype=dt)
            cur = nxt
        mel = torch.empty(M, cfg.mel_dim, device=dev, dtype=torch.float32)
        ops.tapgemm(cur, P["w_mel"], mel, M=M, N=cfg.mel_dim, Cin=d, bias=P["b_mel"], dtype=dt)
        if self.proj_out is not None:
            V = cfg.decoder_embed_dim
            logits = torch.empty(M, V, device=dev, dtype=torch.float32)
            ops.tapgemm(y16, P["w_out"], logits, M=M, N=V, Cin=d, bias=P["b_out"], dtype=dt)  # :285
        else:
            logits = None
        return logits, mel, y16

    def forward(self, source, padding_mask, spk_emb=None, tbc=True, **kwargs):
        """model_avhubert.py:249-297: source [2T,B,1024] (tbc) or [B,2T,1024]; padding_mask [B,2T] bool."""
        x = source.transpose(0, 1) if tbc else source
        B, T, C = x.shape
        dev = x.device
        t16 = ops.torch_dtype(self.dtype)
        src16 = x.to(t16).contiguous().view(B * T, C)
        lens = ops.lens_from_mask(None if padding_mask is None else padding_mask.to(torch.bool).contiguous(), B, T, dev)
        if spk_emb is None:
            raise NotImplementedError("the mel head of the released checkpoints is built with the 256-d speaker embedding")
        logits, mel, _ = self.forward_rows(src16, lens, B, T, spk_emb, len_mul=1)
        V = logits.shape[1]
        unit = logits.view(B, T, V)
        unit = unit.transpose(0, 1) if tbc else unit
        return {"encoder_out": unit, "encoder_padding_mask": padding_mask, "padding_mask": padding_mask,
                "encoder_out_mel": mel.view(B, 2 * T, self.cfg.mel_dim // 2)}

    def reorder_encoder_out(self, encoder_out, new_order):
        if encoder_out["encoder_out"] is not None:
            encoder_out["encoder_out"] = encoder_out["encoder_out"].index_select(1, new_order)
        for k in ("encoder_padding_mask", "padding_mask"):
            if encoder_out[k] is not None:
                encoder_out[k] = encoder_out[k].index_select(0, new_order)
        return encoder_out

    def max_positions(self):
        return None
