"""Goldens of the host-side pre/post functions from the REFERENCE: models/util.py PIL2Tensor /
Tensor2PIL and utils/tensor2img.py tensor2img, on small synthetic images.  Authoring container only:
    python tests/golden/gen_host_golden.py        -> tests/golden/host_prepost.npz
(data/dataset.py needs torchvision, which this container lacks: the Stage-1 loader is not pinned here.)"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_shims

ref_shims.install()
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
from PIL import Image

from models.util import PIL2Tensor, Tensor2PIL
from utils.tensor2img import tensor2img

CASES = [  # (w, h), upscale, min_size, fix_resize
    ((37, 53), 1, 128, None), ((50, 40), 2, 64, None), ((96, 64), 1, 64, None), ((45, 31), 3, 64, 100), ((130, 70), 1, 64, None),
]


def image(w, h, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h // 4 + 1, w // 4 + 1, 3), dtype=np.uint8)
    return Image.fromarray(a).resize((w, h), Image.BILINEAR)   # smooth content: bicubic overshoot stays moderate


def main():
    out = {}
    for i, ((w, h), up, ms, fr) in enumerate(CASES):
        img = image(w, h, 100 + i)
        out[f"in{i}"] = np.asarray(img)
        x, h0, w0 = PIL2Tensor(img, upscale=up, min_size=ms, fix_resize=fr)
        u8 = ((x + 1) * 127.5).round()
        u8n = u8.numpy().astype(np.uint8)
        assert torch.equal(torch.tensor(u8n / 255 * 2 - 1, dtype=torch.float32), x)
        out[f"p2t{i}_u8"] = u8n        # x == float32(u8 / 255 * 2 - 1 in float64) exactly: that is how the reference builds it
        out[f"p2t{i}_hw"] = np.array([h0, w0])
        g = torch.Generator().manual_seed(i)
        noise = (0.3 * torch.randn(x.shape, generator=g)).half()                # stored in 16 bits, applied as such
        out[f"noise{i}"] = noise.numpy()
        y = (x + noise.float()).clamp(-1.2, 1.2)                                # a "sample": slightly out of range
        out[f"t2p{i}"] = np.asarray(Tensor2PIL(y, h0, w0))
        out[f"t2i{i}"] = tensor2img(y.unsqueeze(0).clone())                    # [1,3,H,W] -> squeeze -> HWC uint8
    out["t2i_2d"] = tensor2img(torch.linspace(-1.5, 1.5, 48).reshape(6, 8))
    out["t2i_f32"] = tensor2img(torch.linspace(-1, 1, 3 * 4 * 5).reshape(3, 4, 5), out_type=np.float32)
    np.savez_compressed(os.path.join(HERE, "host_prepost.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
