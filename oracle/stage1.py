"""Oracle for stage 1 end to end, as MultiTargetSequenceGenerator._generate strings it together
(multi_target_lip2speech/sequence_generator.py:40-507): encoder -> x2 time repeat -> conformer -> mel slice -> decode."""
import torch

from . import avhubert, conformer, decode


def split_state_dict(sd):
    enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    con = {k: v for k, v in sd.items() if k.startswith("conformer.")}
    return enc, con


def generate(sd, video, padding_mask, spk_emb, enc_layers=24, enc_heads=16, conf_layers=12, conf_heads=8, beam=1,
             temperature=1.0, use_beam_search=False, taps=None):
    """video [B,1,T,88,88]; padding_mask [B,T] bool; spk_emb [B,256].
    Returns dict(tokens=list of LongTensor[L+1], mels=list of [4*src_len,80], logits [2T,B,V], encoder_out); a `taps` dict
    receives the conformer's intermediate tensors ("head_in" = the [B,2T,512] rows the unit / mel heads read)."""
    enc_sd, con_sd = split_state_dict(sd)
    B, T = padding_mask.shape
    src_lengths = T - padding_mask.long().sum(-1)                                   # :64-65
    target_lengths = src_lengths * 2                                                # :109
    eo = avhubert.encoder_wrapper(enc_sd, video, padding_mask, layers=enc_layers, heads=enc_heads)   # :126
    co = conformer.conformer_forward(con_sd, eo["encoder_out"].repeat_interleave(2, dim=0),          # :128-134
                                     eo["encoder_padding_mask"].repeat_interleave(2, dim=1), spk_emb,
                                     layers=conf_layers, heads=conf_heads, taps=taps)
    mels = [m[: int(n) * 2] for m, n in zip(co["encoder_out_mel"], target_lengths)]                  # :136-139
    if use_beam_search:
        fin = decode.beam_search_decode(co["encoder_out"], target_lengths.tolist(), beam_size=beam,
                                        temperature=temperature)
        hyps = [f[0] for f in fin]
    else:
        hyps = decode.greedy_decode(co["encoder_out"], target_lengths.tolist(), temperature=temperature)
    return {"tokens": [h["tokens"] for h in hyps], "scores": [h["score"] for h in hyps], "mels": mels,
            "logits": co["encoder_out"], "encoder_out": eo["encoder_out"], "target_lengths": target_lengths}
