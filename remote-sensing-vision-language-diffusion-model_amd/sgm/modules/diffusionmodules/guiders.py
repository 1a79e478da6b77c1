"""Classifier-free guidance (reference: sgm/modules/diffusionmodules/guiders.py:10-101).

``prepare_inputs`` builds the [uc; c] batch.  The conditioning entries do not change during a
sampling loop, so their concatenation is memoised on the identity of the (c, uc) tensors: the SAME
concatenated tensor objects are handed to the network every step, which is what lets the
cross-attention K/V projections of the text context be computed once per image
(sgm/modules/attention.py, HipNet.cache_context_kv)."""
from functools import partial

import torch

from ...util import default, instantiate_from_config

_CAT_KEYS = ["vector", "crossattn", "concat", "control", "control_vector", "mask_x"]
_NO_DYN = {"target": "rsvld_amd.sgm.modules.diffusionmodules.sampling_utils.NoDynamicThresholding"}


class _CFGBase:
    def __init__(self, dyn_thresh_config=None):
        self.dyn_thresh = instantiate_from_config(default(dyn_thresh_config, _NO_DYN))
        self._memo = None

    def __call__(self, x, sigma):
        x_u, x_c = x.chunk(2)
        return self.dyn_thresh(x_u, x_c, self.scale_schedule(sigma))

    def prepare_inputs(self, x, s, c, uc):
        sig = tuple((k, id(c[k]), id(uc[k]), getattr(c[k], "_version", 0), getattr(uc[k], "_version", 0)) for k in c)
        if self._memo is None or self._memo[0] != sig:
            c_out = {}
            for k in c:
                if k in _CAT_KEYS:
                    c_out[k] = torch.cat((uc[k], c[k]), 0)
                else:
                    assert c[k] == uc[k]
                    c_out[k] = c[k]
            self._memo = (sig, c_out, [c[k] for k in c], [uc[k] for k in c])  # keep the sources alive: ids stay unique
        return torch.cat([x] * 2), torch.cat([s] * 2), self._memo[1]


class VanillaCFG(_CFGBase):
    def __init__(self, scale, dyn_thresh_config=None):
        super().__init__(dyn_thresh_config)
        self.scale_schedule = partial(lambda scale, sigma: scale, scale)


class LinearCFG(_CFGBase):
    def __init__(self, scale, scale_min=None, dyn_thresh_config=None):
        super().__init__(dyn_thresh_config)
        if scale_min is None:
            scale_min = scale
        self.scale_schedule = partial(lambda scale, scale_min, sigma: (scale - scale_min) * sigma / 14.6146 + scale_min,
                                      scale, scale_min)


class IdentityGuider:
    def __call__(self, x, sigma):
        return x

    def prepare_inputs(self, x, s, c, uc):
        return x, s, dict(c)
