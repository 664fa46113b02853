"""fairseq plugin protocol for this path: which base classes the task / model / dataset classes are built on and how
they get registered (multi_target_lip2speech/task.py:48, model_avhubert.py:27, model.py:66, model_auto_avsr.py:28,
model_raven.py:34 register through `fairseq.tasks.register_task` / `fairseq.models.register_model` with `dataclass=`).

fairseq's decorators refuse a class that does not extend `FairseqTask` / `BaseFairseqModel` and a `dataclass=` that does
not extend `FairseqDataclass` (ValueError), so the bases are chosen HERE, once, at import time:

* fairseq importable  -> TaskBase = FairseqTask, ModelBase = BaseFairseqModel, DataclassBase = FairseqDataclass,
  DatasetBase = FairseqDataset, and `register_task` / `register_model` below forward to fairseq's decorators.  Any error
  they raise propagates: a plugin that cannot register must not look loaded.
* fairseq absent (this build image: it is an un-vendored dependency of the reference) -> small stand-alone bases with the
  same constructor signatures, and the decorators only fill the local registries.

Only `ImportError` selects the second mode.  The local registries are filled in both modes; the CLIs resolve
`lip2speech` / `multi_target_*` through them, tests read them.  Nothing here has run against a real fairseq (none exists
in the image); the fairseq mode is exercised by tests/test_plugin_cpu.py against a stand-in that restates the decorators'
type checks.
"""
import logging
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

logger = logging.getLogger(__name__)

TASK_REGISTRY = {}    # name -> (class, config dataclass)
MODEL_REGISTRY = {}   # name -> (class, config dataclass)

try:
    from fairseq.data import FairseqDataset as DatasetBase
    from fairseq.dataclass import FairseqDataclass as DataclassBase
    from fairseq.models import BaseFairseqModel as ModelBase
    from fairseq.models import register_model as _fairseq_register_model
    from fairseq.tasks import FairseqTask as TaskBase
    from fairseq.tasks import register_task as _fairseq_register_task
    HAVE_FAIRSEQ = True
except ImportError:
    HAVE_FAIRSEQ = False
    _fairseq_register_model = _fairseq_register_task = None

    @dataclass
    class DataclassBase:
        """fairseq.dataclass.FairseqDataclass: one optional `_name` field in front of the subclass's own."""
        _name: Optional[str] = None

    class TaskBase:
        """The part of FairseqTask.__init__/setup_task the path relies on."""

        def __init__(self, cfg, **kwargs):
            self.cfg = cfg
            self.datasets = {}

        @classmethod
        def setup_task(cls, cfg, **kwargs):
            return cls(cfg, **kwargs)

        def dataset(self, split):
            return self.datasets[split]

    class ModelBase(nn.Module):
        """BaseFairseqModel: an nn.Module whose constructor takes no arguments."""

    class DatasetBase(torch.utils.data.Dataset):
        pass

logger.info("lip2speech_unit_amd plugin mode: %s", "fairseq (classes extend FairseqTask / BaseFairseqModel)"
            if HAVE_FAIRSEQ else "stand-alone (fairseq not importable; local registries only)")


def interpolation(key, default):
    """`II("task.normalize")`-style defaults (avhubert/hubert_asr.py:136-137): an omegaconf interpolation under fairseq (the
    config store resolves it), the plain default otherwise."""
    if HAVE_FAIRSEQ:
        from omegaconf import II
        return II(key)
    return default


def register_task(name, dataclass=None):
    def deco(cls):
        if HAVE_FAIRSEQ:
            cls = _fairseq_register_task(name, dataclass=dataclass)(cls)
        TASK_REGISTRY[name] = (cls, dataclass)
        return cls
    return deco


def register_model(name, dataclass=None):
    def deco(cls):
        if HAVE_FAIRSEQ:
            cls = _fairseq_register_model(name, dataclass=dataclass)(cls)
        MODEL_REGISTRY[name] = (cls, dataclass)
        return cls
    return deco


def cfg_get(cfg, key, default=None):
    """Read `key` from a dataclass instance, an omegaconf DictConfig, an argparse Namespace or a dict."""
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    v = getattr(cfg, key, default)   # a mandatory value left at `???` raises omegaconf's error: loud on purpose
    return default if v is None else v
