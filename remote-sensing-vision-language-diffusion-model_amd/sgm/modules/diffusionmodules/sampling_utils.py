"""reference: sgm/modules/diffusionmodules/sampling_utils.py (inference subset)."""
from .... import ops
from ...util import append_dims


class NoDynamicThresholding:
    def __call__(self, uncond, cond, scale):
        """uncond + scale*(cond - uncond); ``scale`` is uniform over the batch on this path."""
        s = float(scale.reshape(-1)[0]) if hasattr(scale, "reshape") else float(scale)
        return ops.lerp_f32(uncond.contiguous(), cond.contiguous(), s)


def to_d(x, sigma, denoised):
    return (x - denoised) / append_dims(sigma, x.ndim)
