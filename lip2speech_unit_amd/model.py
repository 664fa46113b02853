"""`multi_target` model — host-side mirror of multi_target_lip2speech/model.py:66-252: the conformer head of
`multi_target_avhubert` fed by ESPnet's Conv3dResNet (Swish) frontend instead of AV-HuBERT.

Structure and state_dict layout follow the reference: `MultiTargetEncoderModel.encoder` IS the `Conformer`
(FairseqEncoderModel, :66-71), whose ESPnet `Encoder` keeps its frontend (`encoder.encoder.frontend.*`, :187-206: here the
line that drops it is commented out); conformer_embed_dim == 512, so there is no proj_in (:216-219).
`Conformer.forward(source, padding_mask, spk_emb)` :238-285: frontend on source['video'] -> x2 time repeat -> encoder ->
mel / unit heads.
"""
import os
from dataclasses import dataclass
from typing import Any, Optional

import torch
import torch.nn as nn

from . import ops
from .conformer import Conformer as _ConformerHead, ConformerConfig
from .conv3d_extractor import Conv3dResNet
from .plugin import DataclassBase, ModelBase, cfg_get, interpolation, register_model


@dataclass
class MultiTargetEncoderModelConfig(DataclassBase):
    """model.py:32-44 over AVHubertSeq2SeqConfig / AVHubertAsrConfig (avhubert/hubert_asr.py:36-146,:193-249), field for
    field: fairseq merges the model config saved in a checkpoint into the registered dataclass, so every field a released
    checkpoint carries has to exist here.  The inference path reads `w2v_args`, `w2v_path` and the `conformer_*` sizes;
    the training-time fields (dropouts, masking, the unused seq2seq decoder sizes) are carried, not interpreted.
    `mask_selection` / `mask_channel_selection` are plain strings here (a ChoiceEnum in fairseq)."""
    # AVHubertAsrConfig
    w2v_path: str = ""
    no_pretrained_weights: bool = False
    dropout_input: float = 0.0
    final_dropout: float = 0.0
    dropout: float = 0.0
    attention_dropout: float = 0.0
    activation_dropout: float = 0.0
    apply_mask: bool = False
    mask_length: int = 10
    mask_prob: float = 0.5
    mask_selection: str = "static"
    mask_other: float = 0
    no_mask_overlap: bool = False
    mask_channel_length: int = 10
    mask_channel_prob: float = 0.0
    mask_channel_selection: str = "static"
    mask_channel_other: float = 0
    no_mask_channel_overlap: bool = False
    freeze_finetune_updates: int = 0
    feature_grad_mult: float = 0.0
    layerdrop: float = 0.0
    normalize: bool = interpolation("task.normalize", False)
    data: str = interpolation("task.data", "")
    w2v_args: Any = None
    # AVHubertSeq2SeqConfig
    decoder_embed_dim: int = 768
    decoder_ffn_embed_dim: int = 3072
    decoder_layers: int = 6
    decoder_layerdrop: float = 0.0
    decoder_attention_heads: int = 4
    decoder_learned_pos: bool = False
    decoder_normalize_before: bool = False
    no_token_positional_embeddings: bool = False
    decoder_dropout: float = 0.0
    decoder_attention_dropout: float = 0.0
    decoder_activation_dropout: float = 0.0
    max_target_positions: int = 2048
    share_decoder_input_output_embed: bool = False
    no_scale_embedding: bool = True
    # model.py:32-44
    checkpoint_path: Optional[str] = None
    use_conformer: bool = False
    conformer_layers: int = 12
    conformer_embed_dim: int = 512
    conformer_ffn_embed_dim: int = 2048
    conformer_attention_heads: int = 8
    conformer_dropout: float = 0.1
    conformer_attention_dropout: float = 0.1
    conformer_layer_norm_first: bool = True
    text_supervision: bool = bool(int(os.environ.get("TEXT_SUPERVISION", 0)))


@dataclass
class MultiTargetAutoAVSREncoderModelConfig(MultiTargetEncoderModelConfig):
    """model.py:46-53 (Auto-AVSR encoder sizes, environment-overridable as in the reference)."""
    avsr_checkpoint_path: Optional[str] = os.environ.get("AVSR_CHECKPOINT_PATH")
    encoder_attention_dim: int = int(os.environ.get("ENCODER_ATTN_DIM", 768))
    encoder_attention_heads: int = int(os.environ.get("ENCODER_ATTN_HEADS", 12))
    encoder_linear_units: int = int(os.environ.get("ENCODER_LIN_UNITS", 3072))
    encoder_num_blocks: int = int(os.environ.get("ENCODER_NUM_BLOCKS", 12))


@dataclass
class MultiTargetRAVENEncoderModelConfig(MultiTargetEncoderModelConfig):
    """model.py:56-63 (RAVEn large sizes)."""
    raven_checkpoint_path: Optional[str] = os.environ.get("RAVEN_CHECKPOINT_PATH")
    encoder_idim: int = int(os.environ.get("ENCODER_IDIM", 512))
    encoder_attention_dim: int = int(os.environ.get("ENCODER_ATTN_DIM", 1024))
    encoder_attention_heads: int = int(os.environ.get("ENCODER_ATTN_HEADS", 16))
    encoder_linear_units: int = int(os.environ.get("ENCODER_LIN_UNITS", 4096))
    encoder_num_blocks: int = int(os.environ.get("ENCODER_NUM_BLOCKS", 24))


def env_dtype():
    """fairseq's `build_model(cfg, task)` has no precision argument: L2S_DTYPE=bf16 selects bf16 operands (default fp16)."""
    return ops.BF16 if os.environ.get("L2S_DTYPE", "f16").lower() in ("bf16", "bfloat16") else ops.F16

AVSR_FRONTEND_WEIGHT_SUM = -27874.6481   # model.py:137-144: known-answer check of the pretrained Auto-AVSR frontend


class Conformer(_ConformerHead):
    """model.py:184-252 (the conformer WITH its visual frontend)."""

    def __init__(self, cfg: ConformerConfig = None, dtype=ops.F16):
        cfg = cfg or ConformerConfig()
        cfg.encoder_embed_dim = cfg.conformer_embed_dim      # 512 != d never holds: proj_in is None (:216-219)
        super().__init__(cfg, dtype=dtype)
        self.encoder.frontend = Conv3dResNet(relu_type="swish", dtype=dtype)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.encoder.frontend._packed = None
        return r

    def frontend_weight_checksum(self):
        s = 0.0
        for name, p in self.encoder.frontend.named_parameters():
            if "weight" in name:
                s += p.detach().cpu().numpy().sum()
        return round(float(s), 4)

    def forward_video_rows(self, video, padding_mask, spk_emb):
        """Hot path: video [B,1,T,88,88] / [B,T,88,88] -> (logits fp32 [B*2T, V], mel fp32 [B*2T, 160], lens int32 [B], B, T)."""
        fe = self.encoder.frontend
        feat, B, T = fe.forward_rows(video)                                   # :242  [B*T, 512] 16-bit
        dev = feat.device
        lens = ops.lens_from_mask(None if padding_mask is None else padding_mask.to(torch.bool).contiguous(), B, T, dev)
        f32 = torch.empty(B * T, 512, device=dev, dtype=torch.float32)
        ops.cast_16_to_f32(feat, f32, B * T, 512, self.dtype)
        src16 = torch.empty(B * 2 * T, 512, device=dev, dtype=ops.torch_dtype(self.dtype))
        ops.repeat2_cast(f32, src16, B, T, 512, self.dtype)                   # :244-245 repeat_interleave(2)
        logits, mel, _ = self.forward_rows(src16, lens, B, 2 * T, spk_emb, len_mul=2)
        return logits, mel, lens, B, T

    def forward(self, source, padding_mask, spk_emb=None, tbc=True, **kwargs):
        logits, mel, lens, B, T = self.forward_video_rows(source["video"], padding_mask, spk_emb)
        V = logits.shape[1]
        unit = logits.view(B, 2 * T, V)
        pm2 = padding_mask.repeat_interleave(2, dim=1) if padding_mask is not None else None
        return {"encoder_out": unit.transpose(0, 1) if tbc else unit, "encoder_padding_mask": pm2, "padding_mask": pm2,
                "encoder_out_mel": mel.view(B, 4 * T, self.cfg.mel_dim // 2)}

    def forward_torchscript(self, net_input):
        return self.forward(**{k: v for k, v in net_input.items() if k in ("source", "padding_mask", "spk_emb")})


@register_model("multi_target", dataclass=MultiTargetAutoAVSREncoderModelConfig)     # model.py:66 (sic: the AVSR config)
class MultiTargetEncoderModel(ModelBase):
    """model.py:66-107."""

    def __init__(self, conformer, tgt_dict=None, cfg=None):
        super().__init__()
        self.encoder = conformer
        self.cfg = cfg
        self.tgt_dict = tgt_dict

    @classmethod
    def build_model(cls, cfg=None, task=None, dtype=None, conformer_cfg: ConformerConfig = None):
        dtype = env_dtype() if dtype is None else dtype
        conformer_cfg = conformer_cfg or ConformerConfig.from_model_cfg(cfg)
        tgt_dict = getattr(task, "target_dictionary", None) if task is not None else None
        if tgt_dict is not None:
            conformer_cfg.decoder_embed_dim = len(tgt_dict)                   # :77
        return cls(Conformer(conformer_cfg, dtype=dtype), tgt_dict, cfg)

    def forward(self, **kwargs):
        out = self.encoder(**kwargs)
        out["encoder_out"] = out["encoder_out"].transpose(0, 1).contiguous()  # :89
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
