"""Fixture of the Stage-1 loader contract (data/dataset.py:16-42).  torchvision is not installed in the authoring
container, so the reference loader cannot run: the vectors come from oracle/loader_oracle.py (a restatement of the four
torchvision calls, formulas cited there) -- "parity unpinned" by a reference run, pinned against regressions.
    python tests/golden/gen_loader_golden.py      -> tests/golden/stage1_loader.npz"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import loader_oracle as LO

CASES = [((32, 20), 4), ((20, 32), 4), ((33, 17), 3), ((17, 33), 2), ((16, 16), 2), ((45, 28), 8), ((7, 9), 5)]   # (w, h), scale


def image(w, h, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h // 3 + 1, w // 3 + 1, 3), dtype=np.uint8)
    return Image.fromarray(a).resize((w, h), Image.BILINEAR)


if __name__ == "__main__":
    out = {}
    for i, ((w, h), s) in enumerate(CASES):
        img = image(w, h, 40 + i)
        out[f"in{i}"] = np.asarray(img)
        x = LO.load(img, s)
        u8 = np.round((x + 1) * 127.5).astype(np.uint8)
        assert np.array_equal(((u8.astype(np.float32) / np.float32(255)) - np.float32(0.5)) / np.float32(0.5), x)
        out[f"out{i}_u8"] = u8          # x == ((u8 / 255) - 0.5) / 0.5 in float32, exactly
    np.savez_compressed(os.path.join(HERE, "stage1_loader.npz"), **out)
    print({k: v.shape for k, v in out.items()})
