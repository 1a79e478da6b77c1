"""Stage-2 sampler (reference: sgm/modules/diffusionmodules/sampling.py:23-75, 530-694).

``RestoreEDMSampler`` keeps the reference's public contract —
    init_loop(x, cond, uc, num_steps) -> (x, s_in, sigmas, num_sigmas, cond, uc)
    step(x, i, s_in, sigmas, denoiser, cond, uc, x_center, control_scale, ..., threshold) -> (x, thr)
— its RNG draw order (one ``randn_like`` per step with gamma > 0), the first-block-cache control
flow of ``denoise`` (:548-596: a miss REPLACES the threshold by the measured diff, a hit reuses the
previous guided prediction) and the fp32 sampler arithmetic.

Differences that are structure, not semantics: sigmas live on the HOST (the per-step scalars are
python floats, no device sync); the churn injection, CFG combine, restore pull and Euler update are
kernels of librsvld_hip.so (rsvld_axpy_f32 / rsvld_lerp_f32 / rsvld_euler_step); the cache
similarity is one deterministic reduction kernel + one device->host copy per step.
"""
from typing import Dict, Union

import torch

from .... import ops
from ....models.modules.DFBCache import get_can_use_cache_multi, get_current_cache_context
from ...util import default, instantiate_from_config

DEFAULT_GUIDER = {"target": "rsvld_amd.sgm.modules.diffusionmodules.guiders.IdentityGuider"}


class BaseDiffusionSampler:
    def __init__(self, discretization_config, num_steps=None, guider_config=None, verbose=False, device="cuda"):
        self.num_steps = num_steps
        self.discretization = instantiate_from_config(discretization_config)
        self.guider = instantiate_from_config(default(guider_config, DEFAULT_GUIDER))
        self.verbose = verbose
        self.device = device

    def prepare_sampling_loop(self, x, cond, uc=None, num_steps=None):
        # the schedule is a host table: 51 floats do not belong on the GPU (the reference puts them on
        # `self.device` and then syncs through .item() / python comparisons every step)
        sigmas = self.discretization(self.num_steps if num_steps is None else num_steps, device="cpu")
        uc = default(uc, cond)
        x = ops.axpy_f32(None, x.contiguous(), float(torch.sqrt(1.0 + sigmas[0] ** 2.0)))   # x *= sqrt(1 + sigma_0^2)
        num_sigmas = len(sigmas)
        s_in = torch.ones([x.shape[0]])
        return x, s_in, sigmas, num_sigmas, cond, uc


class RestoreEDMSampler(BaseDiffusionSampler):
    def __init__(self, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, restore_cfg=4.0,
                 restore_cfg_s_tmin=0.05, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.s_churn, self.s_tmin, self.s_tmax, self.s_noise = s_churn, s_tmin, s_tmax, s_noise
        self.restore_cfg, self.restore_cfg_s_tmin = restore_cfg, restore_cfg_s_tmin
        self.sigma_max = 14.6146
        self.fb_mode = "input_stage"
        self.noise_fn = torch.randn_like   # SR_backbone may swap in a CPU-generator draw (noise_source="cpu")

    # ---- sampling.py:548-596
    def denoise(self, x, denoiser, sigma, cond, uc, control_scale=1.0, threshold=0.1):
        if threshold <= 0:
            denoised = denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc), control_scale=control_scale,
                                fbcache_mode="none", partial_info=None)
            return self.guider(denoised, sigma), threshold

        context = get_current_cache_context()
        partial_info = denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc), control_scale=control_scale,
                                fbcache_mode=self.fb_mode + "1", partial_info=None)
        can_use_cache, cache_th = get_can_use_cache_multi(partial_info["h"], threshold=threshold, parallelized=False)
        if can_use_cache:
            if context.final_decode is not None:
                return context.final_decode, threshold
            raise RuntimeError("feature cache hit without a cached prediction")  # unreachable: step 0 always misses
        context.prev = partial_info["h"]   # stage 2 never writes into it, so no clone is needed
        denoised = denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc), control_scale=control_scale,
                            fbcache_mode=self.fb_mode + "2", partial_info=partial_info)
        denoised = self.guider(denoised, sigma)
        context.final_decode = denoised
        return denoised, cache_th

    # ---- sampling.py:598-621
    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc=None, gamma=0.0, x_center=None, eps_noise=None,
                     control_scale=1.0, use_linear_control_scale=False, control_scale_start=0.0, threshold=0.1):
        sigma_hat = sigma * (gamma + 1.0)
        if gamma > 0:
            eps = eps_noise if eps_noise is not None else self.noise_fn(x)
            x = ops.axpy_f32(x, eps, float(self.s_noise * (sigma_hat[0] ** 2 - sigma[0] ** 2) ** 0.5))
        if use_linear_control_scale:
            control_scale = (float(sigma[0]) / self.sigma_max) * (control_scale_start - control_scale) + control_scale

        denoised, threshold = self.denoise(x, denoiser, sigma_hat, cond, uc, control_scale=control_scale,
                                           threshold=threshold)
        restore_w, center = 0.0, None
        if float(next_sigma[0]) > self.restore_cfg_s_tmin and self.restore_cfg > 0:
            restore_w, center = float((sigma[0] / self.sigma_max) ** self.restore_cfg), x_center
        x = ops.euler_step(x, denoised, center, restore_w, float(sigma_hat[0]), float(next_sigma[0] - sigma_hat[0]))
        return x, threshold

    def init_loop(self, x, cond, uc=None, num_steps=None):
        return self.prepare_sampling_loop(x, cond, uc, num_steps)

    def _gamma(self, sigmas, i):
        return min(self.s_churn / (len(sigmas) - 1), 2 ** 0.5 - 1) if self.s_tmin <= float(sigmas[i]) <= self.s_tmax else 0.0

    def step(self, x, i, s_in, sigmas, denoiser, cond, uc, x_center=None, control_scale=1.0,
             use_linear_control_scale=False, control_scale_start=0.0, threshold=0.1):
        s_in = s_in.cpu()
        return self.sampler_step(sigma=s_in * sigmas[i], next_sigma=s_in * sigmas[i + 1], denoiser=denoiser, x=x,
                                 cond=cond, uc=uc, gamma=self._gamma(sigmas, i), x_center=x_center,
                                 control_scale=control_scale, use_linear_control_scale=use_linear_control_scale,
                                 control_scale_start=control_scale_start, threshold=threshold)

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, x_center=None, control_scale=1.0,
                 use_linear_control_scale=False, control_scale_start=0.0, threshold=0.1):
        x, s_in, sigmas, num_sigmas, cond, uc = self.prepare_sampling_loop(x, cond, uc, num_steps)
        th = threshold
        for i in range(num_sigmas - 1):
            x, th = self.step(x, i, s_in, sigmas, denoiser, cond, uc, x_center, control_scale,
                              use_linear_control_scale, control_scale_start, th)
        return x, th
