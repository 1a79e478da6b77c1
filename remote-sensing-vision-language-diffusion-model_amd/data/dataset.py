"""Stage-1 input contract (reference: data/dataset.py:7-42): LR PIL image -> bicubic resize of the SHORTER
side to max(w,h)*scale (torchvision `resize(img, int)` semantics), centre-crop to a square of that size,
ToTensor, (x-0.5)/0.5.  Implemented with PIL + numpy (torchvision is not a dependency)."""
import numpy as np
import torch
from PIL import Image


def resize_and_convert(img, scale, resample=Image.BICUBIC):
    w, h = img.size
    target = int(max(w, h) * scale)
    # torchvision.transforms.functional.resize(img, int): the shorter side becomes `target`, aspect kept
    if w <= h:
        nw, nh = target, int(target * h / w)
    else:
        nh, nw = target, int(target * w / h)
    img = img.resize((nw, nh), resample)
    left, top = int(round((nw - target) / 2.0)), int(round((nh - target) / 2.0))
    return img.crop((left, top, left + target, top + target))


def load_sr_input(image_path, scale=1, resample=Image.BICUBIC):
    """-> {'SR': fp32 [1,3,S,S] in [-1,1], 'Index': 0}, what ``next(iter(dataloader(...)))`` yields (:30-42)."""
    img = resize_and_convert(Image.open(image_path).convert("RGB"), scale, resample)
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
    return {"SR": ((x - 0.5) / 0.5).unsqueeze(0), "Index": torch.tensor([0])}


def dataloader(image_path, scale=1, resample=Image.BICUBIC):
    return [load_sr_input(image_path, scale, resample)]
