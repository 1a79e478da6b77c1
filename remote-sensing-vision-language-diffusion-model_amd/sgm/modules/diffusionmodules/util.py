"""Small helpers of sgm/modules/diffusionmodules/util.py used by the inference path."""
import torch
from torch import nn

from .... import ops


class GroupNorm32(nn.GroupNorm):
    """util.py:273-276 — a plain GroupNorm(32, C, eps=1e-5, affine)."""


def normalization(channels):
    return GroupNorm32(32, channels)


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def conv_nd(dims, *args, **kwargs):
    if dims != 2:
        raise ValueError(f"unsupported dimensions: {dims} (the SR networks are 2-D)")
    return nn.Conv2d(*args, **kwargs)


def linear(*args, **kwargs):
    return nn.Linear(*args, **kwargs)


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """util.py:206-230 -> fp32 ``[N, dim]`` = [cos | sin], computed by rsvld_sinusoidal_embedding."""
    if repeat_only or dim % 2 or max_period != 10000:
        raise NotImplementedError
    return ops.sinusoidal(timesteps.float(), dim, 1)


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2):
    """util.py: 'linear' = linspace(sqrt(start), sqrt(end))**2 in float64 (the LDM convention)."""
    if schedule != "linear":
        raise NotImplementedError(schedule)
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    return betas.numpy()
