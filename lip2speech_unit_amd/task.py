"""`lip2speech` task surface — host-side mirror of multi_target_lip2speech/task.py:28-116 and the pieces of
avhubert/hubert_pretraining.py it inherits (dictionary loading :196-202, build_generator :282-400).

fairseq is an un-vendored dependency of the reference and is absent here, so `UnitDictionary` restates the part of
fairseq.data.Dictionary the path uses (specials <s>,<pad>,</s>,<unk> = 0,1,2,3 then the symbols of dict.unt.txt).
"""
import os
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from .plugin import DataclassBase, TaskBase, cfg_get, interpolation, register_task


class UnitDictionary:
    def __init__(self, symbols: Optional[List[str]] = None):
        self.symbols = ["<s>", "<pad>", "</s>", "<unk>"]
        self.indices = {s: i for i, s in enumerate(self.symbols)}
        self.nspecial = 4
        for s in symbols or []:
            self.add_symbol(s)

    @classmethod
    def load(cls, path):
        """dict.unt.txt: one '<symbol> <count>' per line (datasets/lrs3/label/dict.unt.txt)."""
        d = cls()
        with open(path, "r", encoding="utf-8") as f:
            for line in f:
                line = line.rstrip()
                if not line:
                    continue
                sym = line.rsplit(" ", 1)[0]
                d.add_symbol(sym)
        return d

    def add_symbol(self, s):
        if s not in self.indices:
            self.indices[s] = len(self.symbols)
            self.symbols.append(s)
        return self.indices[s]

    def __len__(self):
        return len(self.symbols)

    def bos(self):
        return 0

    def pad(self):
        return 1

    def eos(self):
        return 2

    def unk(self):
        return 3

    def index(self, s):
        return self.indices.get(s, self.unk())

    def encode_line(self, line, append_eos=True, add_if_not_exist=False):
        ids = [self.index(w) for w in line.strip().split()]
        if append_eos:
            ids.append(self.eos())
        return torch.tensor(ids, dtype=torch.int32)

    def string(self, tensor, extra_symbols_to_ignore=None):
        ignore = set(extra_symbols_to_ignore or [])
        ignore.add(self.eos())
        return " ".join(self.symbols[int(i)] for i in tensor if int(i) not in ignore)


class LabelEncoderUnit:
    """task.py:28-35."""

    def __init__(self, dictionary):
        self.dictionary = dictionary

    def __call__(self, label: str):
        return self.dictionary.encode_line(label, append_eos=True, add_if_not_exist=False).long()

    def decode(self, tok, symbols_ignore=None):
        return self.dictionary.string(tok, extra_symbols_to_ignore=symbols_ignore)


@dataclass
class Lip2SpeechConfig(DataclassBase):
    """task.py:38-45 on top of EVERY field of AVHubertPretrainingConfig (hubert_pretraining.py:62-158): fairseq merges a
    checkpoint's saved task config into this dataclass in struct mode, so a field the checkpoint carries and this class
    lacks would fail `tasks.setup_task`.  Defaults are the reference's except `data` (MISSING there; "" here so the
    stand-alone CLI can construct it) — the fields the inference path does not read are carried, not interpreted."""
    data: str = ""
    labels: List[str] = field(default_factory=lambda: ["ltr"])
    label_dir: Optional[str] = None
    label_rate: int = -1
    sample_rate: int = 16_000
    normalize: bool = False
    enable_padding: bool = False
    max_sample_size: Optional[int] = None
    min_sample_size: Optional[int] = None
    max_trim_sample_size: Optional[int] = interpolation("task.max_sample_size", None)
    single_target: Optional[bool] = False
    random_crop: Optional[bool] = True
    pad_audio: Optional[bool] = False
    pdb: Optional[bool] = False
    stack_order_audio: int = 1
    skip_verify: Optional[bool] = False
    image_aug: bool = False
    image_crop_size: int = 88
    image_mean: float = 0.421
    image_std: float = 0.165
    modalities: Optional[List[str]] = field(default_factory=lambda: ["audio", "video"])
    is_s2s: bool = False
    tokenizer_bpe_name: Optional[str] = None
    tokenizer_bpe_model: Optional[str] = None
    noise_wav: Optional[str] = None
    noise_prob: float = 0
    noise_snr: Optional[str] = "0"
    noise_num: int = 1
    fine_tuning: bool = False
    # task.py:38-45
    time_mask: bool = False
    random_erase: bool = False
    fp16: bool = False
    text_supervision: bool = bool(int(os.environ.get("TEXT_SUPERVISION", 0)))
    grayscale_transform: bool = bool(int(os.environ.get("GRAYSCALE_TRANSFORM", 0)))
    skip_aug: bool = bool(int(os.environ.get("SKIP_AUG", 0)))


def decode_config(data=None, label_dir=None, fp16=False, labels=("unt",), modalities=("video",)):
    """The task config a released stage-1 checkpoint carries for this path (conf/decode.yaml:21-25 + the fine-tuning yaml:
    unit labels at 50 Hz over 25 fps video), for callers that have no saved config to start from."""
    return Lip2SpeechConfig(data=data or "", label_dir=label_dir, labels=list(labels), sample_rate=25, label_rate=50,
                            modalities=list(modalities), fine_tuning=True, fp16=fp16)


@register_task("lip2speech", dataclass=Lip2SpeechConfig)
class Lip2SpeechTask(TaskBase):
    """task.py:48-116 over hubert_pretraining.py:160-400 (the dictionary is loaded eagerly: the reference defers it through
    fairseq's task state, :175-176)."""

    def __init__(self, cfg: Lip2SpeechConfig, dictionary: Optional[UnitDictionary] = None):
        super().__init__(cfg)
        self.fine_tuning = True          # the path only exists fine-tuned (target_dictionary, not dictionaries)
        if cfg_get(cfg, "text_supervision", False):
            raise NotImplementedError("TEXT_SUPERVISION=1 is outside the lip2speech inference path")
        if dictionary is None:
            dictionary = UnitDictionary.load(os.path.join(self.get_label_dir(), f"dict.{cfg.labels[0]}.txt"))
        self._dict = dictionary

    @classmethod
    def setup_task(cls, cfg, **kw):                      # hubert_pretraining.py:212-219
        return cls(cfg)

    def get_label_dir(self):
        return self.cfg.label_dir if self.cfg.label_dir is not None else self.cfg.data

    @property
    def target_dictionary(self):
        return self._dict

    @property
    def source_dictionary(self):
        return None

    @property
    def dictionaries(self):
        return [self._dict]

    def max_positions(self):
        return (2 ** 31 - 1, 2 ** 31 - 1)

    def load_dataset(self, split: str, **kwargs):
        from .data import MultiTargetDataset
        self.datasets[split] = MultiTargetDataset(
            f"{self.cfg.data}/{split}.tsv", label_path=f"{self.get_label_dir()}/{split}.{self.cfg.labels[0]}",
            label_processor=LabelEncoderUnit(self._dict), pad=self._dict.pad(), image_mean=self.cfg.image_mean,
            image_std=self.cfg.image_std, image_crop_size=self.cfg.image_crop_size)
        return self.datasets[split]

    def dataset(self, split):
        return self.datasets[split]

    def build_generator(self, models, args, seq_gen_cls=None, extra_gen_cls_kwargs=None,
                        prefix_allowed_tokens_fn=None):
        """task.py:107-116 -> hubert_pretraining.py:385-400."""
        if seq_gen_cls is None:
            from .sequence_generator import MultiTargetSequenceGenerator
            seq_gen_cls = MultiTargetSequenceGenerator
        extra = dict(extra_gen_cls_kwargs or {})
        extra["fp16"] = self.cfg.fp16
        return seq_gen_cls(
            models, self.target_dictionary,
            beam_size=getattr(args, "beam", 5), max_len_a=getattr(args, "max_len_a", 0),
            max_len_b=getattr(args, "max_len_b", 200), min_len=getattr(args, "min_len", 1),
            normalize_scores=(not getattr(args, "unnormalized", False)), len_penalty=getattr(args, "lenpen", 1),
            unk_penalty=getattr(args, "unkpen", 0), temperature=getattr(args, "temperature", 1.0),
            match_source_len=getattr(args, "match_source_len", False),
            no_repeat_ngram_size=getattr(args, "no_repeat_ngram_size", 0), search_strategy=None,
            nbest=getattr(args, "nbest", 1), use_hipgraph=getattr(args, "hipgraph", False), **extra)

    def inference_step(self, generator, models, sample, prefix_tokens=None, constraints=None):
        with torch.no_grad():
            return generator.generate(models, sample, prefix_tokens=prefix_tokens, constraints=constraints)
