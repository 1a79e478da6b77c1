"""CPU restatement of the Stage-1 input loader (reference: data/dataset.py:16-42) -- TEST INFRASTRUCTURE.

The reference's loader is four torchvision calls (torchvision is pinned to 0.20.x by the reference's torch 2.5.1 and is
NOT installed in the authoring container, so the loader itself cannot be run here: PARITY UNPINNED by a reference run; the
fixture tests/golden/stage1_loader.npz is produced by this restatement and pins the product against regressions):

  trans_fn.resize(img, int)      torchvision/transforms/functional.py `_compute_resized_output_size`: the SHORTER side becomes
                                 `size`, the longer int(size * long / short); PIL resize with the given filter (:16-19)
  trans_fn.center_crop(img, int) top = int(round((H - size) / 2.0)), left = int(round((W - size) / 2.0))      (:20)
  ToTensor                       uint8 HWC -> float32 CHW / 255                                                (:31)
  Normalize(0.5, 0.5)            (x - 0.5) / 0.5                                                               (:32)
"""
import numpy as np
from PIL import Image


def load(img, scale, resample=Image.BICUBIC):
    """PIL RGB image -> float32 [1, 3, S, S] in [-1, 1] as a numpy array."""
    w, h = img.size
    size = int(max(w, h) * scale)
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
    img = img.resize((new_w, new_h), resample)
    top, left = int(round((new_h - size) / 2.0)), int(round((new_w - size) / 2.0))
    img = img.crop((left, top, left + size, top + size))
    x = np.asarray(img, dtype=np.uint8).astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    return ((x - np.float32(0.5)) / np.float32(0.5))[None]
