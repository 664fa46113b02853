"""Unit decoder on gfx950 — host-side mirror of multi_target_lip2speech/sequence_generator.py
(`MultiTargetSequenceGenerator._generate` :40-507 over avhubert/sequence_generator.py `SequenceGenerator`).

Same constructor keywords (hubert_pretraining.py:385-400 + fp16), `generate(models, sample, ...)` entry, and return value
`(finalized, sample)`: finalized[b][0] = {"tokens" [L+1] ending in EOS, "score", "attention", "alignment",
"positional_scores"}; `sample` gains `target_lengths` (:109) and `mels` (:136-139) and has `target` patched (:110-117).

The reference's 2T+1-step python loop is non-autoregressive (lprobs come from encoder_out[step], :253-256): step scores
are independent of the history, so hypothesis 0 of the beam search equals the per-step argmax over unit ids.  One HIP
kernel (l2s_greedy_decode) produces it for the whole batch.  The lower-ranked hypotheses, which inference.py:254 never
reads, are materialised on request (`nbest` > 1, up to `beam_size`) by l2s_beam_decode: the reference's beam search itself
(BeamSearch.step + finalize_hypos), one wavefront per clip, hypotheses in the reference's order (descending score).
"""
from typing import Dict, List, Optional

import torch

from . import ops


class MultiTargetSequenceGenerator:
    def __init__(self, models, tgt_dict, beam_size=1, max_len_a=0, max_len_b=200, max_len=0, min_len=1,
                 normalize_scores=True, len_penalty=1.0, unk_penalty=0.0, temperature=1.0, match_source_len=False,
                 no_repeat_ngram_size=0, search_strategy=None, eos=None, symbols_to_strip_from_output=None,
                 lm_model=None, lm_weight=1.0, nbest=1, use_hipgraph=False, frame_bucket=8, **kwargs):
        self.models = list(models) if isinstance(models, (list, tuple)) else [models]
        if len(self.models) != 1:
            raise NotImplementedError("ensembles are not used on the lip2speech inference path")
        self.model = self.models[0]
        self.tgt_dict = tgt_dict
        self.pad, self.unk, self.bos = tgt_dict.pad(), tgt_dict.unk(), tgt_dict.bos()
        self.eos = tgt_dict.eos() if eos is None else eos
        self.symbols_to_strip_from_output = (symbols_to_strip_from_output.union({self.eos})
                                             if symbols_to_strip_from_output is not None else {self.eos})
        self.vocab_size = len(tgt_dict)
        self.beam_size = min(beam_size, self.vocab_size - 1)
        self.max_len_a, self.max_len_b, self.min_len = max_len_a, max_len_b, min_len
        self.normalize_scores, self.len_penalty, self.unk_penalty = normalize_scores, len_penalty, unk_penalty
        self.temperature = temperature
        self.nbest = max(1, min(int(nbest), self.beam_size))   # hypotheses materialised per clip (the reference: beam_size)
        assert temperature > 0, "--temperature must be greater than 0"
        if lm_model is not None or no_repeat_ngram_size > 0:
            raise NotImplementedError("LM fusion / n-gram blocking are not part of the lip2speech decode config")
        self.kwargs = kwargs
        self.results_path = None
        # use_hipgraph: replay the device part of a batch from a hipGraph captured per (batch, frames) shape; frames are
        # zero-padded (and masked) up to a multiple of frame_bucket so ragged datasets hit few shapes - padding does not
        # change a clip's result
        self.use_hipgraph, self.frame_bucket = bool(use_hipgraph), max(1, int(frame_bucket))
        self._graphs = None
        if (self.pad, self.bos, self.eos, self.unk) != (1, 0, 2, 3):
            raise NotImplementedError("the decode kernel assumes fairseq's special ids bos=0,pad=1,eos=2,unk=3")

    def cuda(self):
        self.model.cuda()
        return self

    @torch.no_grad()
    def generate(self, models, sample: Dict, **kwargs):
        return self._generate(sample, **kwargs)

    def _device_part(self, video, padding_mask, spk_emb):
        """Everything of a batch that runs on the device, with no host synchronisation (hipGraph-capturable): encoder ->
        x2 repeat -> conformer -> heads -> greedy decode.  Returns (logits [B*2T,V], mel [B*2T,160], lens [B], tokens
        [B,2T+1], lprobs [B,2T+1], score [B])."""
        model = self.model
        if getattr(model, "conformer", None) is None:
            # `multi_target` (model.py:66-252): the encoder IS the conformer with its Conv3dResNet frontend (:126 only)
            logits, mel, lens, B, T = model.encoder.forward_video_rows(video, padding_mask, spk_emb)
        else:
            # AV-HuBERT (multi_target_avhubert) or the Auto-AVSR / RAVEn encoders: all hand over fp32 rows
            enc_mod = model.encoder.w2v_model if hasattr(model.encoder, "w2v_model") else model.encoder
            enc, lens, B, T = enc_mod.extract_rows(video, padding_mask)                 # :126 forward_encoder
            dt = model.conformer.dtype
            src16 = torch.empty(B * 2 * T, enc.shape[1], device=enc.device, dtype=ops.torch_dtype(dt))
            ops.repeat2_cast(enc, src16, B, T, enc.shape[1], dt)                        # :130-131 repeat_interleave(2)
            logits, mel, _ = model.conformer.forward_rows(src16, lens, B, 2 * T, spk_emb, len_mul=2)  # :128-134
        T2, V = 2 * T, logits.shape[1]
        tokens = torch.empty(B, T2 + 1, device=logits.device, dtype=torch.int32)
        lprobs = torch.empty(B, T2 + 1, device=logits.device, dtype=torch.float32)
        score = torch.empty(B, device=logits.device, dtype=torch.float32)
        ops.greedy_decode(logits, tokens, lprobs, score, B=B, T2=T2, V=V, lens=lens, len_mul=2,
                          temperature=self.temperature, lenpen=self.len_penalty if self.normalize_scores else 0.0)
        return logits, mel, lens, tokens, lprobs, score

    def _generate(self, sample, prefix_tokens: Optional[torch.Tensor] = None, constraints=None,
                  bos_token: Optional[int] = None):
        if constraints is not None:
            raise NotImplementedError("Target-side constraints were provided, but search method doesn't support them")
        if prefix_tokens is not None:
            raise NotImplementedError("prefix tokens are not used on the lip2speech inference path (prefix_size=0)")
        net_input = sample["net_input"]
        src = net_input["source"]
        if src.get("audio") is not None:
            raise NotImplementedError("modalities=['video'] only (conf/decode.yaml:23)")
        video, padding_mask = src["video"], net_input["padding_mask"]
        spk_emb = net_input["spk_emb"]
        if self.use_hipgraph:
            from .pipeline import GraphCache
            if self._graphs is None:
                self._graphs = GraphCache(self._device_part)
            Bv, Tv = video.shape[0], video.shape[2] if video.dim() == 5 else video.shape[1]
            Tp = -(-Tv // self.frame_bucket) * self.frame_bucket
            pm = padding_mask if padding_mask is not None else torch.zeros(Bv, Tv, dtype=torch.bool, device=video.device)
            if Tp != Tv:
                video = torch.nn.functional.pad(video, (0, 0, 0, 0, 0, Tp - Tv))
                pm = torch.nn.functional.pad(pm, (0, Tp - Tv), value=True)
            outs = self._graphs(video.contiguous(), pm.contiguous(), spk_emb.contiguous())
            logits, mel, lens, tokens, lprobs, score = (t.clone() for t in outs)   # the capture's outputs are reused next call
        else:
            logits, mel, lens, tokens, lprobs, score = self._device_part(video, padding_mask, spk_emb)
        B, T2 = tokens.shape[0], tokens.shape[1] - 1
        T, V = T2 // 2, logits.shape[1]

        src_lengths = lens.to(torch.long)
        sample["target_lengths"] = src_lengths * 2                                  # :109
        tl = sample["target_lengths"].tolist()                                       # one host sync per batch
        max_len = max(tl)
        if sample.get("target") is not None:                                        # :110-117 (ground truth, eval only)
            tgt = sample["target"][:, :max_len]
            for i, n in enumerate(tl):
                tgt[i][n:] = self.pad
                row = tgt[i]
                for j in (row == self.eos).nonzero().flatten().tolist():
                    row[j] = row[j - 1]
            sample["target"] = tgt
        self.last_mel = mel.view(B, 2 * T2, -1)                                     # device copy for the fused stage-2 hand-off
        mels = self.last_mel.cpu().numpy()                                          # :136-139
        sample["mels"] = [m[: 2 * n] for m, n in zip(mels, tl)]
        tokens64 = tokens.to(torch.long)
        finalized: List[List[Dict[str, torch.Tensor]]] = []
        if self.nbest > 1:
            # avhubert/sequence_generator.py:605-721: up to beam_size hypotheses per sentence, sorted by score (:497-505)
            if self.beam_size > 64:
                raise NotImplementedError("n-best output is built for beam <= 64 (conf/decode.yaml: beam 50)")
            btok, bpos, bscore, nhyp = ops.beam_decode(
                logits, B=B, T2=T2, V=V, beam=self.beam_size, lens=lens, len_mul=2, temperature=self.temperature,
                lenpen=self.len_penalty if self.normalize_scores else 0.0)
            btok64, nh = btok.to(torch.long), nhyp.tolist()
            for b, n in enumerate(tl):
                finalized.append([{
                    "tokens": btok64[b, h, : n + 1], "score": bscore[b, h], "attention": torch.empty(0),
                    "alignment": torch.empty(0), "positional_scores": bpos[b, h, : n + 1],
                } for h in range(min(self.nbest, nh[b]))])
            self.last_logits = logits.view(B, T2, V)
            return finalized, sample
        for b, n in enumerate(tl):
            finalized.append([{
                "tokens": tokens64[b, : n + 1],
                "score": score[b],
                "attention": torch.empty(0),
                "alignment": torch.empty(0),
                "positional_scores": lprobs[b, : n + 1],
            }])
        self.last_logits = logits.view(B, T2, V)
        return finalized, sample
