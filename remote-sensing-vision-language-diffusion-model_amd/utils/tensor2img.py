"""Stage-1 -> Stage-2 hand-off (reference: utils/tensor2img.py:4-21): clamp to [-1,1], map to [0,1],
x255, round -> uint8 HWC.  The 8-bit quantisation is part of the reference's semantics."""
import numpy as np


def tensor2img(tensor, out_type=np.uint8, min_max=(-1, 1)):
    """``[3,H,W]`` / ``[1,3,H,W]`` -> HWC, ``[H,W]`` -> HW; uint8 output is rounded, any other dtype keeps [0, 1] values."""
    lo, hi = min_max
    unit = (tensor.squeeze().float().cpu().clamp_(lo, hi) - lo) / (hi - lo)
    rank = unit.dim()
    if rank not in (2, 3):
        raise TypeError(f"Only 3D and 2D tensors are supported here (the reference's 4-D branch calls an "
                        f"unimported make_grid). Got {rank}D")
    arr = unit.numpy() if rank == 2 else unit.numpy().transpose(1, 2, 0)
    if out_type == np.uint8:
        arr = np.round(arr * 255.0)
    return arr.astype(out_type)
