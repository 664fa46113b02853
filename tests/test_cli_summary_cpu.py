"""The stage-1 CLI's N > 1 behaviour on CPU (world-size-2 gloo): clips dealt to the ranks by sorted length, every rank writes the
per-clip files of ITS clips, the per-clip records are gathered once and rank 0 alone writes ONE hypo-<fid>.json / wer.<fid> in
dataset order (the reference lets the ranks overwrite each other, multi_target_lip2speech/inference.py:297-311).  The device part
is a test double (a generator that answers from the frames' content): the host loop, the sharding and the collation are the real
`inference.decode_dataset`."""
import glob
import json
import logging
import os
import socket
from types import SimpleNamespace

import numpy as np
import torch
import torch.multiprocessing as mp

from tests import _synth_dataset

FRAMES = (12, 9, 5, 14, 7, 10, 3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StubGenerator:
    """Hypothesis of a clip = its GROUND-TRUTH units for even dataset ids, all-unit-7 for odd ones: the summary's accuracy is
    then a known number; mels are zeros of the right length."""

    def generate(self, models, sample, **kw):
        pm = sample["net_input"]["padding_mask"]
        lens = (pm.shape[1] - pm.long().sum(-1)) * 2
        sample["target_lengths"] = lens
        sample["mels"] = [np.zeros((2 * int(n), 80), dtype=np.float32) for n in lens]
        hypos = []
        for i, n in enumerate(lens.tolist()):
            if int(sample["id"][i]) % 2 == 0 and sample["target"] is not None:
                toks = sample["target"][i][:n].clone()
            else:
                toks = torch.full((n,), 4 + 7, dtype=torch.long)
            hypos.append([{"tokens": torch.cat([toks.long(), torch.tensor([2])])}])
        return hypos, sample


def _worker(rank, world, port, root, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from lip2speech_unit_amd import distributed as l2s_dist
    from lip2speech_unit_amd import inference as s1
    from lip2speech_unit_amd.task import Lip2SpeechTask, decode_config
    r, w, _ = l2s_dist.init_from_env("gloo")
    lab = os.path.join(root, "label")
    task = Lip2SpeechTask(decode_config(data=lab, label_dir=lab, fp16=False))
    ds = task.load_dataset("test")
    cfg = dict(s1.DEFAULTS)
    cfg["dataset.batch_size"] = 2
    gen = (_StubGenerator(), SimpleNamespace(beam=50, nbest=1))
    res = s1.decode_dataset(cfg, task, SimpleNamespace(), ds, out, logging.getLogger("t"), r, w, generator=gen)
    l2s_dist.barrier()
    assert len(res["utt_id"]) == len(FRAMES)            # every rank holds the whole, ordered record list
    torch.distributed.destroy_process_group()


def test_cli_world2_writes_one_summary_in_dataset_order(tmp_path):
    root, out = str(tmp_path / "data"), str(tmp_path / "results")
    os.makedirs(out)
    _synth_dataset.make(root, frames=FRAMES)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=_worker, args=(r, world, port, root, out)) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(timeout=180)
        assert p.exitcode == 0
    hypo = glob.glob(os.path.join(out, "hypo-*.json"))
    wer = glob.glob(os.path.join(out, "wer.*"))
    assert len(hypo) == 1 and len(wer) == 1, (hypo, wer)         # ONE pair, written by rank 0
    res = json.load(open(hypo[0]))
    want = [f"test/spk{i % 2}/{i:05d}" for i in range(len(FRAMES))]
    assert res["utt_id"] == want                                 # dataset order, every clip exactly once
    units = [ln.split() for ln in open(os.path.join(root, "label", "test.unt")).read().strip().split("\n")]
    n_equal = n_total = 0
    for i, T in enumerate(FRAMES):
        ref = units[i][: 2 * T]
        hyp = res["hypo"][i].split()
        assert res["ref"][i].split() == ref
        assert hyp == (ref if i % 2 == 0 else ["7"] * (2 * T)), i
        n_equal += sum(a == b for a, b in zip(hyp, ref))
        n_total += len(ref)
        assert os.path.exists(os.path.join(out, "pred_unit", want[i] + ".txt"))      # per-clip files: written by the owning rank
        assert np.load(os.path.join(out, "pred_mel", want[i] + ".npy")).shape == (4 * T, 80)
    txt = open(wer[0]).read()
    assert f"Accuracy: {100 * n_equal / n_total}" in txt and "beam: 50" in txt
