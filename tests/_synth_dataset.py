"""Builds a tiny on-disk dataset in the reference's layout (datasets/lrs3: label/{split}.tsv,.unt, dict.unt.txt, video/,
audio/, mel/, spk_emb/) with synthetic content; frames are stored as the .npy sidecar the loader accepts."""
import os
import wave

import numpy as np


def make(root, frames=(12, 9, 5), seed=0):
    rng = np.random.default_rng(seed)
    lab = os.path.join(root, "label")
    os.makedirs(lab, exist_ok=True)
    rows, units = [], []
    for i, T in enumerate(frames):
        utt = f"test/spk{i % 2}/{i:05d}"
        for kind in ("video", "audio", "mel", "spk_emb"):
            os.makedirs(os.path.join(root, kind, os.path.dirname(utt)), exist_ok=True)
        np.save(os.path.join(root, "video", utt + ".npy"), rng.integers(0, 256, (T, 96, 96), dtype=np.uint8))
        n_samp = T * 640 + int(rng.integers(0, 300))
        with wave.open(os.path.join(root, "audio", utt + ".wav"), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes((rng.standard_normal(n_samp) * 1000).astype(np.int16).tobytes())
        np.save(os.path.join(root, "mel", utt + ".npy"), (-11.5 + 11.4 * rng.random((4 * T + 2, 80))).astype(np.float32))
        e = np.maximum(rng.standard_normal(256), 0).astype(np.float32)
        np.save(os.path.join(root, "spk_emb", utt + ".npy"), e / (np.linalg.norm(e) + 1e-6))
        rows.append(f"{utt}\tvideo/{utt}.mp4\taudio/{utt}.wav\t{T}\t{n_samp}")
        units.append(" ".join(str(int(u)) for u in rng.integers(0, 200, 2 * T + (i % 2))))
    with open(os.path.join(lab, "test.tsv"), "w") as f:
        f.write(root + "\n" + "\n".join(rows) + "\n")
    with open(os.path.join(lab, "test.unt"), "w") as f:
        f.write("\n".join(units) + "\n")
    with open(os.path.join(lab, "dict.unt.txt"), "w") as f:
        f.write("".join(f"{i} 1\n" for i in range(200)))
    return lab
