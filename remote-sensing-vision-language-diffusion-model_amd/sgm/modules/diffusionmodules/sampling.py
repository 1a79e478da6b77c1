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

import numpy as np
import torch

from .... import ops
from ....models.modules.DFBCache import (get_can_use_cache_multi, get_current_cache_context, relative_l1,
                                         select_partial_info)
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
        if isinstance(threshold, (list, tuple)):
            return self._denoise_per_image(x, denoiser, sigma, cond, uc, control_scale, list(threshold))
        if threshold <= 0:
            denoised = denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc), control_scale=control_scale,
                                fbcache_mode="none", partial_info=None)
            return self.guider(denoised, sigma), threshold

        context = get_current_cache_context()
        partial_info = denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc), control_scale=control_scale,
                                fbcache_mode=self.fb_mode + "1", partial_info=None)
        first = context.prev is None
        can_use_cache, cache_th = get_can_use_cache_multi(partial_info["h"], threshold=threshold, parallelized=False)
        if getattr(context, "trace", None) is not None:   # [(threshold, measured diff, hit)] per image, one entry per step
            context.trace.append([(float(threshold), None if first else float(cache_th), bool(can_use_cache))])
        if can_use_cache:
            if context.final_decode is not None:
                return context.final_decode, threshold
            raise RuntimeError("feature cache hit without a cached prediction")  # unreachable: step 0 always misses
        context.prev = partial_info["h"].clone()   # (:581) a replayed graph overwrites its static output on the next step
        denoised = denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc), control_scale=control_scale,
                            fbcache_mode=self.fb_mode + "2", partial_info=partial_info)
        denoised = self.guider(denoised, sigma)
        context.final_decode = denoised
        return denoised, cache_th

    def _denoise_per_image(self, x, denoiser, sigma, cond, uc, control_scale, thresholds):
        """A batch of B independent images with the feature cache ON: the reference only ever samples
        one image (infer.py:172,199), so its whole-tensor test (:548-596, DFBCache.py:98-134) IS a per-image test.
        Here the first UNet half runs on the full CFG batch, every image takes its own hit / miss decision against its
        own threshold (rows b and B+b of the cache key), the second half runs on the sub-batch that missed, and the
        guided predictions are scattered into the cached tensor.  Image b sees exactly the control flow, thresholds and
        (batch-invariant kernels) values of a batch-of-1 run."""
        B = x.shape[0]
        context = get_current_cache_context()
        x2, s2, c2 = self.guider.prepare_inputs(x, sigma, cond, uc)
        partial_info = denoiser(x2, s2, c2, control_scale=control_scale, fbcache_mode=self.fb_mode + "1", partial_info=None)
        h = partial_info["h"]
        if context.prev is None:
            diffs, hit = [None] * B, [False] * B                      # step 0: nothing to compare with (:123-124)
        else:
            diffs = relative_l1(context.prev, h, images=B)
            hit = [d < t for d, t in zip(diffs, thresholds)]
        trace = getattr(context, "trace", None)
        if trace is not None:
            trace.append([(float(t), None if d is None else float(d), bool(u)) for t, d, u in zip(thresholds, diffs, hit)])
        miss = [b for b in range(B) if not hit[b]]
        new_thr = [t if (hit[b] or diffs[b] is None) else diffs[b] for b, t in enumerate(thresholds)]
        if not miss:
            return context.final_decode, new_thr
        if len(miss) == B:
            context.prev = h.clone()
            denoised = self.guider(denoiser(x2, s2, c2, control_scale=control_scale, fbcache_mode=self.fb_mode + "2",
                                            partial_info=partial_info), sigma)
            context.final_decode = denoised
            return denoised, new_thr
        idx = torch.tensor(miss, device=x.device)
        rows = torch.cat([idx, idx + B])                                # [uc rows of the missed images; c rows]
        context.prev = context.prev.index_copy(0, rows, h.index_select(0, rows))
        sub = select_partial_info(partial_info, rows)
        xs = x.index_select(0, idx)
        ss = sigma[torch.tensor(miss)]
        with ops.plan_units(len(miss)):                                 # the sub-batch holds len(miss) images
            den = denoiser(torch.cat([xs] * 2), torch.cat([ss] * 2), c2, control_scale=control_scale,
                           fbcache_mode=self.fb_mode + "2", partial_info=sub)
        den = self.guider(den, ss)
        context.final_decode = context.final_decode.index_copy(0, idx, den)
        return context.final_decode, new_thr

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


# ---------------------------------------------------------------------------------------------------------------
# Latent-tile sampling (sampling.py:697-757, 830-863).  NOTE on the reference: as shipped, TiledRestoreEDMSampler.__call__
# cannot run -- RestoreEDMSampler.sampler_step was changed to return ``(x, threshold)`` and to default ``threshold=0.1``
# (the feature cache, which needs a cache context), while the tile loop still multiplies its return value by the tile
# mask (:733) and opens no cache context; ``gaussian_weights`` hard-codes ``device='cuda'``.  Nothing in the reference
# selects the class.  It is provided here with the evident intent: every tile takes one un-cached step (threshold 0) and
# the tensor part of the result is blended.  ``_sliding_windows`` is pinned to the reference's function
# (tests/golden/tiled_sampler_windows.json); the mask and the loop follow the source line by line (oracle/s2_oracle.py).
def gaussian_weights(tile_width, tile_height, nbatches, device=None):
    """sampling.py:830-847 -> float64 ``[nbatches, 4, tile_height, tile_width]``.  var = 0.01; note the reference's
    asymmetry: the x midpoint is (w-1)/2, the y midpoint h/2."""
    var, norm = 0.01, np.sqrt(2 * np.pi * 0.01)
    mx, my = (tile_width - 1) / 2, tile_height / 2

    def bell(n, mid):      # element by element through numpy's SCALAR exp: the array form differs from it in the last bit of some entries,
        return [np.exp(-(i - mid) * (i - mid) / (n * n) / (2 * var)) / norm for i in range(n)]   # and the reference's plane is pinned bit for bit

    w = torch.tensor(np.outer(bell(tile_height, my), bell(tile_width, mx)), device=device)
    return torch.tile(w, (nbatches, 4, 1, 1))


def _sliding_windows(h, w, tile_size, tile_stride):
    """sampling.py:850-863: (hi, hi_end, wi, wi_end) row-major; a last window flush with the border when the stride does
    not land on it."""
    hi_list = list(range(0, h - tile_size + 1, tile_stride))
    if (h - tile_size) % tile_stride != 0:
        hi_list.append(h - tile_size)
    wi_list = list(range(0, w - tile_size + 1, tile_stride))
    if (w - tile_size) % tile_stride != 0:
        wi_list.append(w - tile_size)
    return [(hi, hi + tile_size, wi, wi + tile_size) for hi in hi_list for wi in wi_list]


class TiledRestoreEDMSampler(RestoreEDMSampler):
    def __init__(self, tile_size=128, tile_stride=64, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.tile_size, self.tile_stride = tile_size, tile_stride
        self.tile_weights = gaussian_weights(tile_size, tile_size, 1)   # float64, host; one fp32 [th, tw] plane goes to the device

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, x_center=None, control_scale=1.0,
                 use_linear_control_scale=False, control_scale_start=0.0):
        use_local_prompt = isinstance(cond, list)
        b, _, h, w = x.shape
        if h < self.tile_size or w < self.tile_size:
            raise ValueError(f"latent {h}x{w} is smaller than the tile ({self.tile_size})")
        tiles = _sliding_windows(h, w, self.tile_size, self.tile_stride)
        if use_local_prompt and len(cond) != len(tiles):
            raise ValueError("Number of local prompts should be equal to number of tiles")
        lq = (cond[0] if use_local_prompt else cond)["control"]
        weights = self.tile_weights[0, 0].to(device=x.device, dtype=torch.float32).contiguous()
        x, s_in, sigmas, num_sigmas, cond, uc = self.prepare_sampling_loop(x, cond, uc, num_steps)
        s_in = s_in.cpu()
        for i in range(num_sigmas - 1):
            gamma = self._gamma(sigmas, i)
            x_next, count = torch.zeros_like(x), torch.zeros_like(x)
            eps_noise = self.noise_fn(x)                      # one draw per step over the whole latent, sliced per tile (:723)
            for j, (hi, hi_end, wi, wi_end) in enumerate(tiles):
                sl = (slice(None), slice(None), slice(hi, hi_end), slice(wi, wi_end))
                ctl = lq[sl].contiguous()
                c_j = dict(cond[j] if use_local_prompt else cond, control=ctl)
                uc_j = dict(uc, control=ctl)
                x_tile, _ = self.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x[sl].contiguous(), c_j, uc_j,
                                              gamma, None if x_center is None else x_center[sl].contiguous(),
                                              eps_noise=eps_noise[sl].contiguous(), control_scale=control_scale,
                                              use_linear_control_scale=use_linear_control_scale,
                                              control_scale_start=control_scale_start, threshold=0.0)
                ops.tile_blend_accumulate(x_next, count, x_tile.contiguous(), weights, hi, wi)
            x = ops.tile_blend_finish(x_next, count)
        return x

