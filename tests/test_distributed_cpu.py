"""N>1 path on CPU: world_size-2 gloo run of the clip sharding + padded all_gather collation."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lip2speech_unit_amd import distributed as l2s_dist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = l2s_dist.init_from_env("gloo")
    lengths = [100, 37, 250, 25, 180, 60, 99]
    mine = l2s_dist.shard_by_length(lengths, w, r)
    L = max(2 * lengths[i] for i in mine)
    toks = torch.full((len(mine), L + 1), 1, dtype=torch.int32)
    for j, i in enumerate(mine):
        n = 2 * lengths[i]
        toks[j, :n] = 4 + (torch.arange(n) + i) % 200
        toks[j, n] = 2
    lens = torch.tensor([2 * lengths[i] for i in mine], dtype=torch.int32)
    all_t, all_l = l2s_dist.gather_padded(toks, lens)
    t = l2s_dist.max_over_ranks(float(rank + 1), "cpu")
    # fixed-shape collation (the bench): one collective, no shape agreement
    st, sl = l2s_dist.gather_padded(torch.full((2, 9), 4 + rank, dtype=torch.int32), torch.full((2,), 8 + rank, dtype=torch.int32),
                                    static_shape=True)
    assert st.shape == (2 * w, 9) and st[:, 0].tolist() == [4, 4, 5, 5] and sl.tolist() == [8, 8, 9, 9]
    # run-level collation of the CLI's records: every rank gets all of them in dataset order
    recs = l2s_dist.gather_results([(i, f"utt{i}", "r", f"h{rank}") for i in mine])
    owner = {i: o for o in range(w) for i in l2s_dist.shard_by_length(lengths, w, o)}
    assert [x[0] for x in recs] == list(range(len(lengths))) and all(x[3] == f"h{owner[x[0]]}" for x in recs)
    l2s_dist.barrier()
    # plain python lists through the queue: a torch tensor would travel as a file descriptor that the parent has to fetch
    # from this process while it is still alive (the race that made this test flaky)
    q.put((rank, mine, all_t.tolist(), all_l.tolist(), t))
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    lengths = [100, 37, 250, 25, 180, 60, 99]
    owned = sorted(res[0][1] + res[1][1])
    assert owned == list(range(7))                                   # a partition
    assert abs(len(res[0][1]) - len(res[1][1])) <= 1
    tot = [sum(lengths[i] for i in r[1]) for r in res]
    assert abs(tot[0] - tot[1]) <= max(lengths)                     # balanced frames
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]         # every rank sees the same collation
    all_t, all_l = torch.tensor(res[0][2], dtype=torch.int32), torch.tensor(res[0][3], dtype=torch.int32)
    bmax = all_t.shape[0] // world
    for r in range(world):
        for j, i in enumerate(res[r][1]):
            n = 2 * lengths[i]
            row = all_t[r * bmax + j]
            assert int(all_l[r * bmax + j]) == n
            assert torch.equal(row[:n], (4 + (torch.arange(n) + i) % 200).int()) and int(row[n]) == 2
    assert res[0][4] == 2.0 and res[1][4] == 2.0


def test_single_process_passthrough():
    t, l = torch.ones(2, 5, dtype=torch.int32), torch.tensor([5, 3], dtype=torch.int32)
    a, b = l2s_dist.gather_padded(t, l)
    assert a is t and b is l
    assert l2s_dist.shard_by_length([5, 9, 1], 1, 0) == [1, 0, 2]
