"""Network wrapper (reference: sgm/modules/diffusionmodules/wrappers.py:68-110).

``ControlWrapper.forward(x, t, c, control_scale, fbcache_mode, partial_info)`` runs the ControlNet
and then the UNet, like the reference — with one structural difference: on an ``*_stage2`` call
the reference re-runs the ControlNet and throws the result away (the UNet takes ``control`` from
``partial_info``, SR_modules.py:694); here the ControlNet is not launched at all in that case.
``dtype`` is assigned by SR_backbone (SR_model.py:41) and selects the compute type (fp16 / bf16 storage, or fp32 operands).
Inputs may be fp32 NCHW (reference layout; converted once) or already-packed 16-bit NHWC.
Full outputs are the UNet's fp32 NHWC epsilon ``[N,H,W,8]`` (4 channels + padding); the denoiser
turns them back into fp32 NCHW."""
import torch
import torch.nn as nn

from .... import ops
from ...._lib import RsvldError

OPENAIUNETWRAPPER = "rsvld_amd.sgm.modules.diffusionmodules.wrappers.ControlWrapper"


class ControlWrapper(nn.Module):
    def __init__(self, diffusion_model, compile_model: bool = False, dtype=torch.float32):
        super().__init__()
        self.diffusion_model = diffusion_model
        self.control_model = None
        self.dtype = dtype
        self.split = None      # with dtype fp32: the ops.SplitPolicy of the split-operand precision (SR_backbone.set_precision(..., "split")), else None

    def load_control_model(self, control_model):
        self.control_model = control_model

    def _compute_dtype(self):
        """fp16 / bf16: the 16-bit kernel family; fp32 (``diffusion_dtype: fp32``): the fp32-operand family (csrc/f32.hip)."""
        return self.dtype if self.dtype in (torch.bfloat16, torch.float32) else torch.float16

    def _nhwc(self, t, dt):
        if t.dtype in (torch.float16, torch.bfloat16) and t.dim() == 4 and t.shape[-1] % 8 == 0:
            if dt == torch.float32:
                raise RsvldError("a 16-bit NHWC tensor was handed to the fp32 network")
            return t if t.dtype == dt else t.to(dt)
        if t.dtype == torch.float32 and getattr(t, "_nhwc", False):
            if dt != torch.float32:
                raise RsvldError("an fp32 NHWC tensor was handed to the 16-bit network")
            return t
        return ops.nchw_to_nhwc(t, dt)

    def forward(self, x, t, c, control_scale=1, fbcache_mode="none", partial_info=None, **kwargs):
        with ops.f32_split(self.split if self._compute_dtype() == torch.float32 else None):
            return self._forward(x, t, c, control_scale, fbcache_mode, partial_info, **kwargs)

    def _forward(self, x, t, c, control_scale=1, fbcache_mode="none", partial_info=None, **kwargs):
        if not x.is_cuda:
            raise RsvldError("ControlWrapper runs on the GPU only")
        dt = self._compute_dtype()
        for net in (self.diffusion_model, self.control_model):
            net.set_compute_dtype(dt)            # packed weights are per dtype (kept across switches)
        context = c.get("crossattn", None)
        if context is not None and context.dtype != dt:
            cached = getattr(self, "_ctx_cast", None)
            if cached is None or cached[0] is not context or cached[1] != context._version:
                cached = self._ctx_cast = (context, context._version, context.to(dt).contiguous())
            context = cached[2]
        y = c.get("vector", None)
        xh = self._nhwc(x, dt)
        if "stage2" in fbcache_mode:
            control = None
        else:
            control = self.control_model(x=self._nhwc(c.get("control", None), dt), timesteps=t, xt=xh,
                                         context=context, y=y)
        return self.diffusion_model(xh, timesteps=t, context=context, y=y, control=control,
                                    control_scale=control_scale, fbcache_mode=fbcache_mode,
                                    partial_info=partial_info, **kwargs)
