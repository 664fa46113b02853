"""`multi_target` model — host-side mirror of multi_target_lip2speech/model.py:66-252: the conformer head of
`multi_target_avhubert` fed by ESPnet's Conv3dResNet (Swish) frontend instead of AV-HuBERT.

Structure and state_dict layout follow the reference: `MultiTargetEncoderModel.encoder` IS the `Conformer`
(FairseqEncoderModel, :66-71), whose ESPnet `Encoder` keeps its frontend (`encoder.encoder.frontend.*`, :187-206: here the
line that drops it is commented out); conformer_embed_dim == 512, so there is no proj_in (:216-219).
`Conformer.forward(source, padding_mask, spk_emb)` :238-285: frontend on source['video'] -> x2 time repeat -> encoder ->
mel / unit heads.
"""
import torch
import torch.nn as nn

from . import ops
from .conformer import Conformer as _ConformerHead, ConformerConfig
from .conv3d_extractor import Conv3dResNet

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
        if padding_mask is None:
            lens = torch.full((B,), T, device=dev, dtype=torch.int32)
        else:
            lens = (T - padding_mask.to(torch.int32).sum(-1)).to(torch.int32).contiguous()
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


class MultiTargetEncoderModel(nn.Module):
    """model.py:66-107."""

    def __init__(self, conformer, tgt_dict=None, cfg=None):
        super().__init__()
        self.encoder = conformer
        self.cfg = cfg
        self.tgt_dict = tgt_dict

    @classmethod
    def build_model(cls, cfg=None, task=None, dtype=ops.F16, conformer_cfg: ConformerConfig = None):
        conformer_cfg = conformer_cfg or ConformerConfig()
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


try:  # optional fairseq plugin registration (fairseq is not installed in the build image)
    from fairseq.models import register_model  # type: ignore

    register_model("multi_target")(MultiTargetEncoderModel)
except Exception:  # pragma: no cover
    pass
