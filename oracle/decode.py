"""Oracle for the unit decode loop: multi_target_lip2speech/sequence_generator.py:173-507 with finalize_hypos /
is_finished of avhubert/sequence_generator.py:605-739 and fairseq's BeamSearch.step / get_normalized_probs.

fairseq's pieces are un-vendored third-party code (fairseq @ afc77bd): BeamSearch.step is restated from its published
algorithm [recalled]: step 0 uses beam 0 only, later steps add the cumulative beam score, then top-(2*beam) over
beam*V candidates.  "parity unpinned" against the reference; pinned by property tests (hypothesis 0 == masked argmax).
Pure torch, written for small cases (one python iteration per step, like the reference).
"""
import math

import torch

PAD, BOS, EOS, UNK = 1, 0, 2, 3


def beam_search_decode(enc_out, target_lengths, beam_size=5, temperature=1.0, len_penalty=1.0, min_len=1,
                       normalize_scores=True):
    """enc_out: [T2, B, V] logits (conformer 'encoder_out'); target_lengths: [B] ints (= 2*src_len).
    Returns finalized: list over sentences of list of dicts sorted by score (tokens end with EOS)."""
    T2, B, V = enc_out.shape
    beam_size = min(beam_size, V - 1)
    max_len = int(max(target_lengths))
    finalized = [[] for _ in range(B)]
    for b in range(B):  # sentences are independent in the reference (batch bookkeeping only) - decode one at a time
        trg_len = int(target_lengths[b])
        scores = torch.zeros(beam_size, max_len + 1)
        tokens = torch.full((beam_size, max_len + 2), PAD, dtype=torch.long)
        tokens[:, 0] = EOS
        cands_to_ignore = torch.zeros(beam_size, dtype=torch.bool)
        done = False
        for step in range(max_len + 1):
            if step >= max_len:
                lprobs = torch.zeros(beam_size, V)
            else:
                row = enc_out[step, b].float() / temperature                      # :256 div_(temperature)
                lprobs = torch.log_softmax(row, dim=-1).unsqueeze(0).repeat(beam_size, 1)
            lprobs[lprobs != lprobs] = -math.inf                                  # :274
            lprobs[:, PAD] = -math.inf                                            # :276
            lprobs[:, BOS] = -math.inf                                            # :280-282
            lprobs[:, EOS] = -math.inf
            lprobs[:, UNK] = -math.inf
            if step >= max_len or trg_len <= step:                               # :286-298
                lprobs[:, :EOS] = -math.inf
                lprobs[:, EOS + 1:] = -math.inf
                lprobs[:, EOS] = 0
            elif step < min_len:                                                  # :309-311
                lprobs[:, EOS] = -math.inf
            # fairseq BeamSearch.step
            if step == 0:
                cand = lprobs[0:1, :]
            else:
                cand = lprobs + scores[:, step - 1].unsqueeze(-1)
            k = min(beam_size * 2, cand.numel() - 1)
            cand_scores, idx = torch.topk(cand.reshape(-1), k)
            cand_beams, cand_indices = idx // V, idx.fmod(V)
            eos_mask = cand_indices.eq(EOS) & cand_scores.ne(-math.inf)           # :348-350
            eos_mask[:beam_size][cands_to_ignore] = False
            for i in range(min(beam_size, k)):                                    # finalize_hypos
                if eos_mask[i] and len(finalized[b]) < beam_size:
                    beam = int(cand_beams[i])
                    toks = tokens[beam, 1: step + 2].clone()
                    toks[step] = EOS
                    pos = scores[beam, : step + 1].clone()
                    pos[step] = cand_scores[i]
                    pos[1:] = pos[1:] - pos[:-1].clone()
                    sc = cand_scores[i] / ((step + 1) ** len_penalty) if normalize_scores else cand_scores[i]
                    finalized[b].append({"tokens": toks, "score": sc, "positional_scores": pos})
            if len(finalized[b]) >= beam_size or step >= max_len:
                done = True
                break
            # active hypotheses: first beam_size non-EOS candidates (:436-482)
            em = eos_mask.clone()
            em[:beam_size] = em[:beam_size] | cands_to_ignore
            active_mask = em.long() * (2 * beam_size) + torch.arange(k)
            new_ignore, active = torch.topk(active_mask, k=beam_size, largest=False)
            cands_to_ignore = new_ignore.ge(2 * beam_size)
            ab = cand_beams[active]
            tokens[:, : step + 1] = tokens[ab, : step + 1]
            tokens[:, step + 1] = cand_indices[active]
            if step > 0:
                scores[:, :step] = scores[ab, :step]
            scores[:, step] = cand_scores[active]
        assert done
        finalized[b].sort(key=lambda hyp: -float(hyp["score"]))                   # :497-505
    return finalized


def greedy_decode(enc_out, target_lengths, temperature=1.0, len_penalty=1.0):
    """What hypothesis 0 of the loop above reduces to: per-step argmax over unit ids [4,V) of the log-softmax."""
    T2, B, V = enc_out.shape
    out = []
    for b in range(B):
        L = int(target_lengths[b])
        lp = torch.log_softmax(enc_out[:L, b].float() / temperature, dim=-1)
        lp[lp != lp] = -math.inf
        m, arg = lp[:, 4:].max(dim=-1)
        toks = torch.cat([arg + 4, torch.tensor([EOS])])
        pos = torch.cat([m, torch.zeros(1)])
        out.append({"tokens": toks, "score": pos.sum() / ((L + 1) ** len_penalty), "positional_scores": pos})
    return out
