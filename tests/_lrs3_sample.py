"""Lays the committed LRS3-sample fixtures (tests/golden/lrs3_sample label files + the two clips of vocoder_lrs3.npz) out
as a dataset directory in the reference's layout (datasets/lrs3: label/, audio/, mel/, spk_emb/).  The wav files carry
only the right length: the vocoder loader reads them for nothing else (dataset_multi_input.py:201-213,222-239)."""
import os
import shutil
import wave

import numpy as np


def materialise(root, golden_dir):
    src = os.path.join(golden_dir, "lrs3_sample")
    lab = os.path.join(root, "label")
    os.makedirs(lab, exist_ok=True)
    rows = open(os.path.join(src, "test.tsv")).read().splitlines()[1:]
    with open(os.path.join(lab, "test.tsv"), "w") as f:      # first line = dataset root (the author's path in the sample)
        f.write(root + "\n" + "\n".join(rows) + "\n")
    for fn in ("test.unt", "dict.unt.txt"):
        shutil.copyfile(os.path.join(src, fn), os.path.join(lab, fn))
    g = np.load(os.path.join(golden_dir, "vocoder_lrs3.npz"))
    for ci, clip in enumerate(g["clips"]):
        clip = str(clip)
        for kind in ("audio", "mel", "spk_emb"):
            os.makedirs(os.path.join(root, kind, os.path.dirname(clip)), exist_ok=True)
        with wave.open(os.path.join(root, "audio", clip + ".wav"), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes(np.zeros(int(g[f"c{ci}_n_audio"]), np.int16).tobytes())
        np.save(os.path.join(root, "mel", clip + ".npy"), g[f"c{ci}_mel_raw"])
        np.save(os.path.join(root, "spk_emb", clip + ".npy"), g[f"c{ci}_spk"])
    return lab, [r.split("\t")[0] for r in rows], g
