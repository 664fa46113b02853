"""`multi_target_raven` model — host-side mirror of multi_target_lip2speech/model_raven.py:28-160: the RAVEn visual encoder
(raven/_espnet Conv3dResNet-Swish frontend + 24 transformer blocks, d = 1024, 16 heads, ffn 4096: rel-pos attention,
layer-scale, BatchNorm1d-pre feed-forward, no macaron / conv module; defaults model.py:57-63) in place of AV-HuBERT, the same
12 x 512 conformer head behind `proj_in` = Linear(1024, 512).

On the device this is the rel-pos attention + Linear kernels of the conformer head: `conformer.py::Encoder(raven=True)` folds
the layer-scale vectors into the output projections and the eval-mode BatchNorm into the feed-forward's first Linear.
State_dict layout as the reference: `encoder.encoder.{frontend,embed.0,encoders.N.{self_attn,feed_forward,norm_ff,norm_mha,
gamma_ff,gamma_mha},after_norm}` and `conformer.*`.
"""
from dataclasses import dataclass

import torch.nn as nn

from . import ops
from .conformer import Conformer, ConformerConfig, Encoder
from .conv3d_extractor import Conv3dResNet
from .model import MultiTargetRAVENEncoderModelConfig, env_dtype
from .model_auto_avsr import AutoAVSREncoder, MultiTargetAutoAVSREncoderModel
from .plugin import cfg_get, register_model


@dataclass
class RAVENConfig:
    """model.py:57-63 (RAVEn large config values)."""
    encoder_idim: int = 512
    encoder_attention_dim: int = 1024
    encoder_attention_heads: int = 16
    encoder_linear_units: int = 4096
    encoder_num_blocks: int = 24

    @classmethod
    def from_model_cfg(cls, cfg):
        c = cls()
        for k in vars(c):
            setattr(c, k, int(cfg_get(cfg, k, getattr(c, k))))
        return c


class RAVENEncoder(AutoAVSREncoder):
    """model_raven.py:103-160 (`extract_rows` / `forward` are AutoAVSREncoder's: frontend -> embed -> blocks -> after_norm)."""

    def __init__(self, cfg: RAVENConfig = None, dtype=ops.F16):
        nn.Module.__init__(self)
        cfg = cfg or RAVENConfig()
        self.cfg = cfg
        self.encoder = Encoder(cfg.encoder_attention_dim, cfg.encoder_attention_heads, cfg.encoder_linear_units,
                               cfg.encoder_num_blocks, 31, idim=cfg.encoder_idim, raven=True)
        self.encoder.frontend = Conv3dResNet(relu_type="swish", dtype=dtype)
        self.dtype = dtype


@register_model("multi_target_raven", dataclass=MultiTargetRAVENEncoderModelConfig)          # model_raven.py:34
class MultiTargetRAVENEncoderModel(MultiTargetAutoAVSREncoderModel):
    """model_raven.py:28-100."""

    @classmethod
    def build_model(cls, cfg=None, task=None, dtype=None, encoder_cfg: RAVENConfig = None,
                    conformer_cfg: ConformerConfig = None):
        dtype = env_dtype() if dtype is None else dtype
        encoder_cfg = encoder_cfg or RAVENConfig.from_model_cfg(cfg)
        conformer_cfg = conformer_cfg or ConformerConfig.from_model_cfg(cfg)
        tgt_dict = getattr(task, "target_dictionary", None) if task is not None else None
        if tgt_dict is not None:
            conformer_cfg.decoder_embed_dim = len(tgt_dict)
        conformer_cfg.encoder_embed_dim = encoder_cfg.encoder_attention_dim   # proj_in = Linear(encoder_attention_dim, d)
        conformer = Conformer(conformer_cfg, dtype=dtype)
        if conformer.proj_in is None:
            conformer.proj_in = nn.Linear(encoder_cfg.encoder_attention_dim, conformer_cfg.conformer_embed_dim)
        return cls(RAVENEncoder(encoder_cfg, dtype=dtype), tgt_dict, cfg, conformer)
