import torch.nn as nn

from fairseq.dataclass import FairseqDataclass

MODEL_REGISTRY = {}
MODEL_DATACLASS_REGISTRY = {}
ARCH_MODEL_REGISTRY = {}


class BaseFairseqModel(nn.Module):
    def __init__(self):
        super().__init__()
        self._is_generation_fast = False

    @classmethod
    def build_model(cls, args, task):
        raise NotImplementedError("Model must implement the build_model method")

    def load_state_dict(self, state_dict, strict=True, model_cfg=None, args=None):
        return super().load_state_dict(state_dict, strict)


class FairseqEncoder(nn.Module):
    def __init__(self, dictionary):
        super().__init__()
        self.dictionary = dictionary


class FairseqEncoderModel(BaseFairseqModel):
    def __init__(self, encoder):
        super().__init__()
        self.encoder = encoder
        assert isinstance(self.encoder, FairseqEncoder)


def register_model(name, dataclass=None):
    def register_model_cls(cls):
        if name in MODEL_REGISTRY:
            raise ValueError(f"Cannot register duplicate model ({name})")
        if not issubclass(cls, BaseFairseqModel):
            raise ValueError(f"Model ({name}: {cls.__name__}) must extend BaseFairseqModel")
        if dataclass is not None and not issubclass(dataclass, FairseqDataclass):
            raise ValueError(f"Dataclass {dataclass} must extend FairseqDataclass")
        MODEL_REGISTRY[name] = cls
        cls.__dataclass = dataclass
        if dataclass is not None:
            MODEL_DATACLASS_REGISTRY[name] = dataclass
            node = dataclass()
            node._name = name
            ARCH_MODEL_REGISTRY[name] = cls
        return cls
    return register_model_cls


def build_model(cfg, task):
    from fairseq.tasks import merge_with_parent
    name = cfg["_name"]
    dc = MODEL_DATACLASS_REGISTRY[name]
    return ARCH_MODEL_REGISTRY[name].build_model(merge_with_parent(dc(), cfg), task)
