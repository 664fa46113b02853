"""`multi_target_avhubert` model — host-side mirror of multi_target_lip2speech/model_avhubert.py:27-180.

`MultiTargetAVHubertEncoderModel` keeps the attributes the generator and inference.py use (`.encoder`, `.conformer`,
`get_normalized_probs`, `max_positions`, `prepare_for_inference_`, `half/cuda/eval`, state_dict layout
`encoder.w2v_model.*` / `conformer.*`).  When fairseq is importable the class is also registered under the
reference's model name so `--user-dir` loading resolves to this implementation.
"""
import torch
import torch.nn as nn

from . import ops
from .conformer import Conformer, ConformerConfig
from .hubert import AVHubertConfig, AVHubertModel, HubertEncoderWrapper
from .model import MultiTargetEncoderModelConfig, env_dtype
from .plugin import ModelBase, cfg_get, register_model


# keys a real checkpoint may lack / carry beyond this module's parameters (everything else is an error):
#  * mask_emb is deleted before the nested load (model_avhubert.py:105) and unused in eval;
#  * the audio sub-model is never reached with modalities=['video'] (conf/decode.yaml:23);
#  * pre-training heads are dropped by remove_pretraining_modules() (:108); BatchNorm's num_batches_tracked is bookkeeping;
#  * `multi_target` checkpoints keep the (unused) ESPnet frontend under conformer.encoder.frontend;
#  * text-supervision heads (TEXT_SUPERVISION=1, model_avhubert.py:208-229) are outside the inference path.
ALLOWED_MISSING = ("encoder.w2v_model.mask_emb", "encoder.w2v_model.feature_extractor_audio.")
ALLOWED_UNEXPECTED = ("encoder.w2v_model.final_proj", "encoder.w2v_model.label_embs_concat", "encoder.w2v_model.mask_emb",
                      "num_batches_tracked", "conformer.encoder.frontend.", "conformer.ctc", "conformer.text_",
                      "encoder.w2v_model.feature_extractor_audio.", "encoder.w2v_model.feature_extractor_video.encoder.")
RESNET_WEIGHT_SUM = -13260.4916   # model_avhubert.py:119-123 (large_vox_iter5.pt frontend, frozen in fine-tuning)


class CheckpointMismatch(RuntimeError):
    pass


@register_model("multi_target_avhubert", dataclass=MultiTargetEncoderModelConfig)       # model_avhubert.py:27
class MultiTargetAVHubertEncoderModel(ModelBase):
    def __init__(self, encoder, tgt_dict=None, cfg=None, conformer=None):
        super().__init__()
        self.encoder = encoder
        self.conformer = conformer
        self.cfg = cfg
        self.tgt_dict = tgt_dict

    @classmethod
    def build_model(cls, cfg=None, task=None, dtype=None, w2v_cfg: AVHubertConfig = None,
                    conformer_cfg: ConformerConfig = None):
        """model_avhubert.py:47-126 without the checkpoint side effects: builds the AV-HuBERT encoder and the conformer
        with len(tgt_dict) output units; weights come from load_state_dict().  `cfg` is what fairseq hands over (the
        checkpoint's saved model config): the encoder is sized from its embedded pre-training config `cfg.w2v_args.model`
        — or, when that is absent, from the config stored in the `cfg.w2v_path` checkpoint (:71-84) — and the conformer
        from its `conformer_*` fields.  Explicit `w2v_cfg` / `conformer_cfg` / `dtype` arguments win."""
        dtype = env_dtype() if dtype is None else dtype
        if w2v_cfg is None:
            w2v_args = cfg_get(cfg, "w2v_args", None)
            w2v_path = cfg_get(cfg, "w2v_path", "")
            if w2v_args is None and w2v_path:
                # the user's own pre-training checkpoint: fairseq checkpoints embed omegaconf / argparse objects, which torch >= 2.6
                # refuses to unpickle under the default weights_only=True
                state = torch.load(w2v_path, map_location="cpu", weights_only=False)
                w2v_args = state.get("cfg", None)
                if w2v_args is None:
                    w2v_args = state.get("args", None)      # old-style checkpoints: ONE flat argparse Namespace, no `model` group
                if w2v_args is None:
                    raise CheckpointMismatch(f"{w2v_path}: neither `cfg` nor `args` inside - cannot size the AV-HuBERT encoder")
            w2v_cfg = AVHubertConfig.from_w2v_args(w2v_args) if w2v_args is not None else AVHubertConfig()
        conformer_cfg = conformer_cfg or ConformerConfig.from_model_cfg(cfg)
        tgt_dict = getattr(task, "target_dictionary", None) if task is not None else None
        if tgt_dict is not None:
            conformer_cfg.decoder_embed_dim = len(tgt_dict)                      # :112
        conformer_cfg.encoder_embed_dim = w2v_cfg.encoder_embed_dim
        encoder = HubertEncoderWrapper(AVHubertModel(w2v_cfg, dtype=dtype))
        conformer = Conformer(conformer_cfg, dtype=dtype)
        return cls(encoder, tgt_dict, cfg, conformer)

    def load_state_dict(self, state_dict, strict=True, model_cfg=None, args=None):
        # (BaseFairseqModel.load_state_dict's signature; its upgrade / prune hooks have nothing to do for this model)
        r = nn.Module.load_state_dict(self, state_dict, strict=strict)
        self.encoder.w2v_model.repack()
        self.conformer._packed, self.conformer._pos_cache = None, {}
        return r

    def load_checkpoint_state(self, state_dict, check_resnet_sum=True, expected_resnet_sum=RESNET_WEIGHT_SUM):
        """Load a released checkpoint's `state["model"]` (multi_target_lip2speech/inference.py:116 ->
        model_avhubert.py:71-123).  Unlike a bare strict=False load, a key that is missing or unknown outside the two
        allow-lists raises: a parameter left at its init value (weight_v = 0 -> NaN weight norm, random layers) would still
        produce pred_unit/pred_mel files.  The reference's known-answer guard on the frozen frontend is asserted too."""
        r = self.load_state_dict(state_dict, strict=False)
        missing = [k for k in r.missing_keys if not any(a in k for a in ALLOWED_MISSING)]
        unexpected = [k for k in r.unexpected_keys if not any(a in k for a in ALLOWED_UNEXPECTED)]
        if missing or unexpected:
            raise CheckpointMismatch(
                f"checkpoint does not match multi_target_avhubert: {len(missing)} missing (e.g. {missing[:4]}), "
                f"{len(unexpected)} unexpected (e.g. {unexpected[:4]})")
        if check_resnet_sum:
            got = self.resnet_weight_checksum()
            if got != round(expected_resnet_sum, 4):
                raise CheckpointMismatch(f"resnet weight checksum {got} != {expected_resnet_sum} (model_avhubert.py:119-123)")
        return r

    def resnet_weight_checksum(self):
        """The reference's only known-answer check (model_avhubert.py:119-123): sum of all resnet parameters, expected
        -13260.4916 for large_vox_iter5.pt."""
        s = 0.0
        for name, p in self.encoder.named_parameters():
            if "resnet" in name:
                s += p.detach().cpu().numpy().sum()
        return round(float(s), 4)

    def forward(self, **kwargs):
        """model_avhubert.py:128-155 (training-time entry; the generator bypasses it)."""
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
        # common.fp16 (inference.py:155-156): the HIP path already computes with 16-bit operands; parameters stay fp32
        return self

    def reorder_encoder_out(self, encoder_out, new_order):
        return self.conformer.reorder_encoder_out(encoder_out, new_order)
