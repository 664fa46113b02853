"""Test helper: the "decisive" synthetic regime of lip2speech_unit_amd/weights.py (structured frames, weak residual branches,
nearest-centroid unit head) wired to the CPU oracle.  The head is fitted on the ORACLE's fp32 head input for the clips the
test checks, then the oracle is run again with the fitted head: both sides (oracle, HIP path) load the same state dict."""
import torch

from lip2speech_unit_amd import weights

BRANCH_SCALE = 0.25


def frames_from_u8(u8, crop=88):
    """hubert_dataset.py:242-245 on uint8 [B,T,96,96] -> fp32 [B,1,T,88,88]."""
    d = (u8.shape[-1] - crop) // 2
    return ((u8[:, :, d:d + crop, d:d + crop].float() / 255.0 - 0.421) / 0.165).unsqueeze(1).contiguous()


def fit_decisive_head(sd, run_oracle, clip_ids, head="conformer.proj_out"):
    """run_oracle(sd, clip, taps) -> the oracle's result dict for that clip run ALONE (taps receives "head_in").
    Returns (state dict with the fitted unit head, {clip: oracle result under that state dict})."""
    rows = []
    with torch.no_grad():
        for b in clip_ids:
            taps = {}
            run_oracle(sd, b, taps)
            rows.append(taps["head_in"].reshape(-1, taps["head_in"].shape[-1]))
        w, bias = weights.nearest_centroid_head(torch.cat(rows), n_units=sd[head + ".weight"].shape[0] - 4)
        sd = dict(sd)
        sd[head + ".weight"], sd[head + ".bias"] = w, bias
        refs = {b: run_oracle(sd, b, None) for b in clip_ids}
    return sd, refs


def margins(logits_2d):
    """Top-2 gap over the unit ids (columns 4..) of oracle logits [L, V]."""
    top2 = logits_2d[:, 4:].topk(2, -1).values
    return top2[:, 0] - top2[:, 1]
