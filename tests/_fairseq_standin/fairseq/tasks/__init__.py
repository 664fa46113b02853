import dataclasses

from fairseq.dataclass import FairseqDataclass

TASK_REGISTRY = {}
TASK_DATACLASS_REGISTRY = {}
TASK_CLASS_NAMES = set()


class FairseqTask:
    def __init__(self, cfg, **kwargs):
        self.cfg = cfg
        self.datasets = {}
        self.dataset_to_epoch_iter = {}

    @classmethod
    def setup_task(cls, cfg, **kwargs):
        return cls(cfg, **kwargs)

    def dataset(self, split):
        return self.datasets[split]

    @property
    def target_dictionary(self):
        raise NotImplementedError

    def build_model(self, cfg):
        from fairseq import models
        return models.build_model(cfg, self)


def register_task(name, dataclass=None):
    def register_task_cls(cls):
        if name in TASK_REGISTRY:
            raise ValueError(f"Cannot register duplicate task ({name})")
        if not issubclass(cls, FairseqTask):
            raise ValueError(f"Task ({name}: {cls.__name__}) must extend FairseqTask")
        if cls.__name__ in TASK_CLASS_NAMES:
            raise ValueError(f"Cannot register task with duplicate class name ({cls.__name__})")
        if dataclass is not None and not issubclass(dataclass, FairseqDataclass):
            raise ValueError(f"Dataclass {dataclass} must extend FairseqDataclass")
        TASK_REGISTRY[name] = cls
        TASK_CLASS_NAMES.add(cls.__name__)
        cls.__dataclass = dataclass
        if dataclass is not None:
            TASK_DATACLASS_REGISTRY[name] = dataclass
            node = dataclass()          # the config store instantiates the defaults
            node._name = name
        return cls
    return register_task_cls


def merge_with_parent(dc, cfg):
    """omegaconf struct merge: every key of `cfg` must be a field of the dataclass."""
    known = {f.name for f in dataclasses.fields(dc)}
    for k, v in cfg.items():
        if k not in known:
            raise KeyError(f"Key '{k}' not in '{type(dc).__name__}'")
        setattr(dc, k, v)
    return dc


def setup_task(cfg, **kwargs):
    name = cfg["_name"]
    dc = TASK_DATACLASS_REGISTRY[name]
    return TASK_REGISTRY[name].setup_task(merge_with_parent(dc(), cfg), **kwargs)
