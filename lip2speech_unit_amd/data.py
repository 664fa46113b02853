"""Input boundary of both stages (host side; file I/O only, no arithmetic beyond the reference's normalisation).

Stage 1 — `MultiTargetDataset` follows multi_target_lip2speech/dataset.py:37-257 on top of avhubert/hubert_dataset.py
(:242-245 transform, :317-321 load, :395-479 collate): tsv manifest + `.unt` labels, frames -> /255 -> CenterCrop(88) ->
(x-0.421)/0.165 -> [B,1,T,88,88] zero-padded to the longest clip with `padding_mask` (True = pad), `spk_emb`/`mel` sidecars.
Stage 2 — `parse_manifest`, `load_code_dict`, `code_to_sequence`, `MelCodeDataset` follow
multi_input_vocoder/dataset_multi_input.py:41-141,198-291 with segment_size = -1 (inference).

Video decode itself is outside the path: mp4 is read with OpenCV when it is importable, otherwise a
sibling `<clip>.npy` uint8 [T,H,W] array is used; nothing else is attempted.
"""
import os
import wave
from typing import List

import numpy as np
import torch

from .plugin import DatasetBase


def load_video(path: str) -> np.ndarray:
    """avhubert/utils.py:13-30: grayscale uint8 frames [T,H,W]."""
    npy = os.path.splitext(path)[0] + ".npy"
    if os.path.exists(npy):
        return np.load(npy)
    try:
        import cv2  # type: ignore
    except ImportError as e:
        raise RuntimeError(f"cannot decode {path}: OpenCV is not installed and no {npy} sidecar exists") from e
    for attempt in range(3):
        cap = cv2.VideoCapture(path)
        frames = []
        while True:
            ret, frame = cap.read()
            if not ret:
                break
            frames.append(cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY))
        if frames:
            return np.stack(frames)
    raise ValueError(f"Unable to load {path}")


def center_crop(frames: np.ndarray, size: int) -> np.ndarray:
    """avhubert/utils.py:75-95 (delta = int(round(w - tw) / 2.))."""
    t, h, w = frames.shape
    dw, dh = int(round(w - size) / 2.), int(round(h - size) / 2.)
    return frames[:, dh:dh + size, dw:dw + size]


def normalize_frames(frames_u8: np.ndarray, crop=88, mean=0.421, std=0.165) -> np.ndarray:
    """hubert_dataset.py:242-245: Normalize(0,255) -> CenterCrop -> Normalize(mean,std); returns fp32 [T,crop,crop]."""
    x = frames_u8.astype(np.float32) / 255.0
    x = center_crop(x, crop)
    return (x - mean) / std


class MultiTargetDataset(DatasetBase):
    """A FairseqDataset when fairseq is importable (plugin.py), so `task.get_batch_iterator` of the reference's decode loop
    (inference.py:164-179) can batch it: `num_tokens` / `size` / `ordered_indices` follow hubert_dataset.py:533-552."""

    def __init__(self, manifest_path, label_path=None, label_processor=None, pad=1, image_mean=0.421, image_std=0.165,
                 image_crop_size=88):
        with open(manifest_path) as f:
            self.root = f.readline().strip()
            rows = [ln.rstrip("\n").split("\t") for ln in f if ln.strip()]
        # hubert_dataset.py load_audio_visual: (id, video path, audio path, n_frames, n_samples)
        self.names = [(r[1], r[2] + ":" + r[0]) for r in rows]
        self.ids = [r[0] for r in rows]
        self.sizes = [int(r[-2]) for r in rows]
        self.labels = None
        if label_path is not None and os.path.exists(label_path):
            with open(label_path) as f:
                self.labels = [ln.rstrip("\n") for ln in f]
            assert len(self.labels) == len(rows), "label file and manifest disagree"
        self.label_processor = label_processor
        self.label_processors = [label_processor]
        self.pad = pad
        self.mean, self.std, self.crop = image_mean, image_std, image_crop_size

    def __len__(self):
        return len(self.ids)

    def num_tokens(self, index):
        return self.sizes[index]

    def size(self, index):
        return self.sizes[index]

    def ordered_indices(self):
        # shuffle=False at inference (inference.py:179): longest first, ties in manifest order
        return np.lexsort((np.arange(len(self)), self.sizes))[::-1]

    def _sidecar(self, video_fn, kind):
        return os.path.join(self.root, video_fn).replace("/video/", f"/{kind}/")[:-4] + ".npy"  # dataset.py:197-212

    def __getitem__(self, index):
        video_fn = self.names[index][0]
        frames = load_video(os.path.join(self.root, video_fn))
        feats = normalize_frames(frames, self.crop, self.mean, self.std)
        sample = {"id": index, "fid": self.ids[index], "names": self.names[index],
                  "video_source": torch.from_numpy(np.ascontiguousarray(feats)), "audio_source": None}
        if self.labels is not None and self.label_processor is not None:
            sample["label_list"] = [self.label_processor(self.labels[index])]
        for kind in ("mel", "spk_emb"):
            p = self._sidecar(video_fn, kind)
            if not os.path.exists(p):
                raise FileNotFoundError(f"{p} does not exist")
            sample[kind] = torch.from_numpy(np.load(p).astype(np.float32))
        return sample

    def collater(self, samples: List[dict]):
        """hubert_dataset.py:395-479 + dataset.py:242-257 (pad_audio=True: pad to the longest clip)."""
        B = len(samples)
        T = max(s["video_source"].shape[0] for s in samples)
        H, W = samples[0]["video_source"].shape[1:]
        video = torch.zeros(B, 1, T, H, W)
        padding_mask = torch.zeros(B, T, dtype=torch.bool)
        for i, s in enumerate(samples):
            n = s["video_source"].shape[0]
            video[i, 0, :n] = s["video_source"]
            padding_mask[i, n:] = True
        batch = {"id": torch.tensor([s["id"] for s in samples]), "utt_id": [s["fid"] for s in samples],
                 "names": [s["names"] for s in samples],
                 "net_input": {"source": {"audio": None, "video": video}, "padding_mask": padding_mask,
                               "spk_emb": torch.stack([s["spk_emb"] for s in samples])},
                 "input_lengths": torch.tensor([s["video_source"].shape[0] for s in samples], dtype=torch.int32)}
        if "label_list" in samples[0]:
            labs = [s["label_list"][0] for s in samples]
            L = max(len(x) for x in labs)
            tgt = torch.full((B, L), self.pad, dtype=torch.long)
            for i, x in enumerate(labs):
                tgt[i, : len(x)] = x
            batch["target"] = tgt
            batch["target_lengths"] = torch.tensor([len(x) for x in labs])
            batch["ntokens"] = int(sum(len(x) for x in labs))
        else:
            batch["target"] = None
        mlen = max(len(s["mel"]) for s in samples)
        batch["mel"] = torch.stack([torch.nn.functional.pad(s["mel"], [0, 0, 0, mlen - len(s["mel"])]) for s in samples])
        return batch


# ---- stage 2 -----------------------------------------------------------------------------------------------------
def parse_manifest(manifest_path, max_keep=None, min_keep=None):
    """dataset_multi_input.py:41-110: returns (audio_files, mel_files, codes)."""
    audio_files, mels, codes = [], [], []
    code_path = os.path.splitext(manifest_path)[0] + ".unt"
    with open(manifest_path) as f, open(code_path) as f_c:
        root = f.readline().strip()
        for line, line_code in zip(f, f_c):
            items = line.strip().split("\t")
            code = line_code.strip().split("|")[-1]
            sz = int(items[-2])
            diff = len(code.split()) - sz * 2
            assert -2 <= diff <= 2, "code length != video length * 2"
            if (min_keep is not None and sz < min_keep) or (max_keep is not None and sz > max_keep):
                continue
            audio_path = os.path.join(root, items[2])
            audio_files.append(audio_path)
            mels.append(audio_path.replace("/audio/", "/mel/")[:-4] + ".npy")
            codes.append(code)
    return audio_files, mels, codes


def load_code_dict(path):
    """dataset_multi_input.py:118-125: symbol -> line index (raw unit id, NOT the +4 fairseq token id)."""
    with open(path) as f:
        syms = [line.rstrip().rsplit(" ", 1)[0] for line in f]
    d = {c: i for i, c in enumerate(syms)}
    assert set(d.values()) == set(range(len(d)))
    return d


def code_to_sequence(code, code_dict, collapse_code=False):
    """dataset_multi_input.py:128-141."""
    if collapse_code:
        seq, prev = [], None
        for c in code:
            if c in code_dict and c != prev:
                seq.append(code_dict[c])
                prev = c
        return seq
    return [code_dict[c] for c in code if c in code_dict]


def audio_num_samples(path, pad=None):
    """The reference reads the wav only for its length (dataset_multi_input.py:201-213,222-239)."""
    with wave.open(path, "rb") as w:
        n = w.getnframes()
    if pad:
        n += pad - (n % pad)
    return n


class MelCodeDataset:
    def __init__(self, file_list, code_hop_size=320, mel_hop_size=160, code_dict_path=None, pad=None):
        self.audio_files, self.mel_files, self.codes = file_list
        self.code_hop_size, self.mel_hop_size, self.pad = code_hop_size, mel_hop_size, pad
        self.code_dict = load_code_dict(code_dict_path)
        self.speaker_emb_files = [f.replace("/audio/", "/spk_emb/")[:-4] + ".npy" for f in self.audio_files]

    def __len__(self):
        return len(self.audio_files)

    def __getitem__(self, index):
        """dataset_multi_input.py:198-291 with segment_size=-1: (feats{code,mel,spkr}, None, filename, None)."""
        filename = self.audio_files[index]
        n_audio = audio_num_samples(filename, self.pad)
        code = np.array(code_to_sequence(self.codes[index].split(), self.code_dict))
        code_length = min(n_audio // self.code_hop_size, code.shape[0])
        code = code[:code_length]
        mel = np.load(self.mel_files[index])
        mel_length = min(n_audio // self.mel_hop_size, mel.shape[0])
        mel = mel[:mel_length]
        cut = min(mel_length * self.mel_hop_size, code_length * self.code_hop_size)
        mel = mel[: cut // self.mel_hop_size]
        code = code[: cut // self.code_hop_size]
        assert cut // self.code_hop_size == code.shape[0], "Code audio mismatch"
        assert cut // self.mel_hop_size == mel.shape[0], "Mel audio mismatch"
        feats = {"code": code.astype(np.int64), "mel": np.ascontiguousarray(mel.transpose(1, 0)).astype(np.float32),
                 "spkr": np.load(self.speaker_emb_files[index]).astype(np.float32)}
        return feats, None, str(filename), None
