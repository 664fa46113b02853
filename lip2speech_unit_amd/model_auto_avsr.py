"""`multi_target_auto_avsr` model — host-side mirror of multi_target_lip2speech/model_auto_avsr.py:28-134: the Auto-AVSR
visual encoder (ESPnet Conv3dResNet frontend + 12 macaron conformer blocks, d = 768, 12 heads, ffn 3072; defaults
model.py:47-52) in place of AV-HuBERT, the same 12 x 512 conformer head behind a Linear(768, 512) (`proj_in`, :181).

Nothing new on the device: the encoder is the ESPnet `Encoder` the conformer head already runs on (`conformer.py::Encoder`:
rel-pos attention with 64-dim heads, GLU / depthwise-31 conv module, ReLU feed-forwards) at another width, fed by the
Swish frontend of `conv3d_extractor.py`.  State_dict layout as the reference: `encoder.encoder.{frontend,embed.0,encoders.N,
after_norm}` (so `self.encoder.load_state_dict(auto_avsr_state)` of :39-48 works on an Auto-AVSR checkpoint) and `conformer.*`.
"""
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import ops
from .conformer import Conformer, ConformerConfig, Encoder
from .conv3d_extractor import Conv3dResNet
from .model import MultiTargetAutoAVSREncoderModelConfig, env_dtype
from .plugin import ModelBase, cfg_get, register_model


@dataclass
class AutoAVSRConfig:
    """model.py:47-52 (Auto-AVSR model config values)."""
    encoder_attention_dim: int = 768
    encoder_attention_heads: int = 12
    encoder_linear_units: int = 3072
    encoder_num_blocks: int = 12

    @classmethod
    def from_model_cfg(cls, cfg):
        c = cls()
        for k in vars(c):
            setattr(c, k, int(cfg_get(cfg, k, getattr(c, k))))
        return c


class AutoAVSREncoder(nn.Module):
    """model_auto_avsr.py:97-152."""

    def __init__(self, cfg: AutoAVSRConfig = None, dtype=ops.F16):
        super().__init__()
        cfg = cfg or AutoAVSRConfig()
        self.cfg = cfg
        self.encoder = Encoder(cfg.encoder_attention_dim, cfg.encoder_attention_heads, cfg.encoder_linear_units,
                               cfg.encoder_num_blocks, 31)
        self.encoder.frontend = Conv3dResNet(relu_type="swish", dtype=dtype)
        self.dtype = dtype

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.encoder._packed, self.encoder._pos_cache = None, {}
        self.encoder.frontend._packed = None
        return r

    def extract_rows(self, video, padding_mask):
        """Encoder.forward (espnet encoder.py:230-259): video [B,1,T,88,88] -> (fp32 [B*T, d] rows (b,t), lens int32 [B], B, T)."""
        enc, dt = self.encoder, self.dtype
        feat, B, T = enc.frontend.forward_rows(video)                       # [B*T, 512] 16-bit
        dev = feat.device
        lens = ops.lens_from_mask(None if padding_mask is None else padding_mask.to(torch.bool).contiguous(), B, T, dev)
        x = enc.forward_rows(feat, lens, B, T, 1, dt)
        out = torch.empty(B * T, enc.d, device=dev, dtype=torch.float32)
        na = enc._packed["n_after"]
        ops.layernorm(x, na[0], na[1], 1e-12, out, M=B * T, C=enc.d, dtype=dt)   # after_norm :255-257
        return out, lens, B, T

    def forward(self, source, padding_mask, spk_emb=None, **kwargs):
        out, lens, B, T = self.extract_rows(source["video"], padding_mask)
        return {"encoder_out": out.view(B, T, -1).transpose(0, 1), "encoder_padding_mask": padding_mask,
                "padding_mask": padding_mask}


@register_model("multi_target_auto_avsr", dataclass=MultiTargetAutoAVSREncoderModelConfig)   # model_auto_avsr.py:28
class MultiTargetAutoAVSREncoderModel(ModelBase):
    """model_auto_avsr.py:28-95."""

    def __init__(self, encoder, tgt_dict=None, cfg=None, conformer=None):
        super().__init__()
        self.encoder = encoder
        self.conformer = conformer
        self.cfg = cfg
        self.tgt_dict = tgt_dict

    @classmethod
    def build_model(cls, cfg=None, task=None, dtype=None, encoder_cfg: AutoAVSRConfig = None,
                    conformer_cfg: ConformerConfig = None):
        dtype = env_dtype() if dtype is None else dtype
        encoder_cfg = encoder_cfg or AutoAVSRConfig.from_model_cfg(cfg)
        conformer_cfg = conformer_cfg or ConformerConfig.from_model_cfg(cfg)
        tgt_dict = getattr(task, "target_dictionary", None) if task is not None else None
        if tgt_dict is not None:
            conformer_cfg.decoder_embed_dim = len(tgt_dict)                # :64
        conformer_cfg.encoder_embed_dim = encoder_cfg.encoder_attention_dim  # proj_in = Linear(encoder_attention_dim, d) :181
        conformer = Conformer(conformer_cfg, dtype=dtype)
        if conformer.proj_in is None:                                       # the reference builds it unconditionally
            conformer.proj_in = nn.Linear(encoder_cfg.encoder_attention_dim, conformer_cfg.conformer_embed_dim)
        return cls(AutoAVSREncoder(encoder_cfg, dtype=dtype), tgt_dict, cfg, conformer)

    def load_state_dict(self, state_dict, strict=True, model_cfg=None, args=None):
        r = nn.Module.load_state_dict(self, state_dict, strict=strict)
        self.encoder.encoder._packed, self.encoder.encoder._pos_cache = None, {}
        self.encoder.encoder.frontend._packed = None
        self.conformer._packed = None
        return r

    def forward(self, **kwargs):
        out = self.encoder(source=kwargs["source"], padding_mask=kwargs["padding_mask"])
        out = self.conformer(source=out["encoder_out"].repeat_interleave(2, dim=0),
                             padding_mask=out["encoder_padding_mask"].repeat_interleave(2, dim=1),
                             spk_emb=kwargs["spk_emb"])
        out["encoder_out"] = out["encoder_out"].transpose(0, 1).contiguous()
        return out

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        logits = net_output["encoder_out"].float()
        return torch.log_softmax(logits, dim=-1) if log_probs else torch.softmax(logits, dim=-1)

    def max_positions(self):
        return None

    def prepare_for_inference_(self, cfg=None):
        self.eval()

    def half(self):
        return self
