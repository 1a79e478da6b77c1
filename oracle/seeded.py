"""Deterministic synthetic weights (test infrastructure).

No checkpoints are available offline, so golden vectors are produced with weights drawn from a
seed by this recipe, applied to the REFERENCE modules when the fixtures are generated and to the
product / oracle when they are checked.  Only parameter names + shapes + the seed matter, so the
weights themselves never need to be committed.
"""
import math

import torch


def seeded_state_dict(named_shapes, seed, zero_ok=False):
    """``named_shapes``: iterable of (name, shape) in state-dict order -> {name: fp32 tensor}.

    * 1-D ``*.weight`` (norm gains): 1 + 0.1 N(0,1);  ``*.bias``: 0.05 N(0,1)
    * >= 2-D weights: N(0, 1/fan_in) (variance preserving; zero-initialised reference tensors such as
      sgm ``zero_module`` outputs would make every golden trivially 0, so they are overwritten too)
    One generator is consumed in iteration order, so the ORDER of names is part of the recipe.
    """
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in named_shapes:
        shape = tuple(shape)
        if len(shape) == 0:
            out[name] = torch.zeros(shape)
        elif len(shape) == 1:
            r = torch.randn(shape, generator=g)
            out[name] = 1.0 + 0.1 * r if name.endswith("weight") else 0.05 * r
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            out[name] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
    return out


def seed_module(module, seed, skip=()):
    """Overwrite every PARAMETER of ``module`` with the recipe (buffers untouched)."""
    names = [(k, tuple(v.shape)) for k, v in module.named_parameters() if not any(k.startswith(s) for s in skip)]
    sd = seeded_state_dict(names, seed)
    with torch.no_grad():
        for k, p in module.named_parameters():
            if k in sd:
                p.copy_(sd[k])
    return sd


def synthetic_image(shape, seed, smooth=4):
    """Image-like fp32 tensor in [-1, 1]: uniform noise, box low-pass, min-max normalised."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(shape, generator=g)
    if smooth > 1:
        x = torch.nn.functional.avg_pool2d(x, smooth, stride=1, padding=smooth // 2)[..., :shape[-2], :shape[-1]]
    lo, hi = x.amin(dim=(-3, -2, -1), keepdim=True), x.amax(dim=(-3, -2, -1), keepdim=True)
    return ((x - lo) / (hi - lo) * 2 - 1).contiguous()
