"""Stage-1 -> Stage-2 hand-off (reference: utils/tensor2img.py:4-21): clamp to [-1,1], map to [0,1],
x255, round -> uint8 HWC.  The 8-bit quantisation is part of the reference's semantics."""
import numpy as np


def tensor2img(tensor, out_type=np.uint8, min_max=(-1, 1)):
    tensor = tensor.squeeze().float().cpu().clamp_(*min_max)
    tensor = (tensor - min_max[0]) / (min_max[1] - min_max[0])
    if tensor.dim() == 3:
        img_np = np.transpose(tensor.numpy(), (1, 2, 0))
    elif tensor.dim() == 2:
        img_np = tensor.numpy()
    else:
        raise TypeError(f"Only 3D and 2D tensors are supported here (the reference's 4-D branch calls an "
                        f"unimported make_grid). Got {tensor.dim()}D")
    if out_type == np.uint8:
        img_np = (img_np * 255.0).round()
    return img_np.astype(out_type)
