"""Clip-parallel sharding across the GPUs of one node (one process per GPU, torch.distributed: backend "nccl" is RCCL
over xGMI on ROCm, "gloo" on CPU for tests).

The reference shards the dataset by rank with no collective (multi_target_lip2speech/inference.py:174-175: num_shards /
shard_id) and lets every rank overwrite the same summary files (:297-311).  Here clips are dealt to ranks by sorted
length (balances sum T and sum T^2), each rank runs its own batches, and results are collated with ONE padded
all_gather per batch (variable-length unit ids; KB-sized, latency-bound - never a ring all-reduce).
"""
import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """Returns (rank, world_size, local_rank); initialises the default process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_by_length(lengths: Sequence[int], world_size: int, rank: int) -> List[int]:
    """Indices of the clips this rank owns: sort by length (desc, stable) and deal round-robin in a serpentine order so
    every rank gets the same number of clips (+-1) and near-equal total frames."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    mine = []
    for pos, idx in enumerate(order):
        rnd, slot = divmod(pos, world_size)
        owner = slot if rnd % 2 == 0 else world_size - 1 - slot
        if owner == rank:
            mine.append(idx)
    return mine


def gather_padded(tokens: torch.Tensor, lengths: torch.Tensor, pad_value: int = 1, static_shape: bool = False):
    """all_gather of a rank-local [b, Lmax] int tensor + [b] lengths.
    static_shape=True (every rank holds the same [b, Lmax], e.g. the fixed-shape bench): ONE all_gather_into_tensor, no
    host synchronisation.  Otherwise ranks may hold different b / Lmax: the shapes are first agreed on with a 16-byte
    all_reduce(MAX) read back on the host (a second, tiny collective + one sync), then the same single all_gather moves the
    payload.  Returns (tokens [world*bmax, Lmax], lengths [world*bmax]); rows past a rank's own b have length 0."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return tokens, lengths
    world = dist.get_world_size()
    if static_shape:
        bmax, lmax = tokens.shape
    else:
        shape = torch.tensor([tokens.shape[0], tokens.shape[1]], device=tokens.device, dtype=torch.int64)
        dist.all_reduce(shape, op=dist.ReduceOp.MAX)
        bmax, lmax = int(shape[0]), int(shape[1])
    buf = torch.full((bmax, lmax + 1), pad_value, device=tokens.device, dtype=torch.int32)
    buf[: tokens.shape[0], : tokens.shape[1]] = tokens.to(torch.int32)
    buf[:, lmax] = 0
    buf[: tokens.shape[0], lmax] = lengths.to(torch.int32)   # lengths ride in the last column: one collective
    out = torch.empty(world * bmax, lmax + 1, device=tokens.device, dtype=torch.int32)
    dist.all_gather_into_tensor(out, buf)
    return out[:, :lmax], out[:, lmax]


def gather_results(records: list) -> list:
    """Run-level collation of the CLI's per-clip records (dataset index, utt_id, ref, hypo strings - a few hundred bytes a
    clip): every rank contributes its list, every rank gets the concatenation sorted by dataset index, so ONE hypo-<fid>.json /
    wer.<fid> pair in dataset order is written by rank 0 (the reference lets ranks overwrite each other,
    multi_target_lip2speech/inference.py:297-311).  Off the data path: one all_gather_object per run."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return sorted(records, key=lambda r: r[0])
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, records)
    return sorted((r for p in parts for r in p), key=lambda r: r[0])


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def all_ranks(value: float, device) -> List[float]:
    """Every rank's value, in rank order (the bench prints each rank's own step time beside the max: skew shows at once)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [value]
    t = torch.tensor([value], device=device, dtype=torch.float64)
    out = torch.empty(dist.get_world_size(), device=device, dtype=torch.float64)
    dist.all_gather_into_tensor(out, t)
    return [float(x) for x in out]
