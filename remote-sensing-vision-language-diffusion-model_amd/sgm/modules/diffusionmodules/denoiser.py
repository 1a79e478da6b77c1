"""Denoiser wrappers (reference: sgm/modules/diffusionmodules/denoiser.py:7-78).

``DiscreteDenoiserWithControl`` keeps the reference call contract
    denoiser(network, input, sigma, cond, control_scale, fbcache_mode, partial_info)
with ``input`` fp32 NCHW and ``sigma`` a per-sample tensor.  The sigma -> nearest-of-1000 table
quantisation runs on the host (no device sync when ``sigma`` lives on the CPU, which is how
RestoreEDMSampler passes it); the two scalings are kernels: ``input * c_in`` is folded into the
NCHW->NHWC pack, ``net * c_out + input * c_skip`` is rsvld_denoiser_out."""
import torch
import torch.nn as nn

from .... import ops
from ...util import instantiate_from_config


class Denoiser(nn.Module):
    def __init__(self, weighting_config, scaling_config):
        super().__init__()
        self.weighting = instantiate_from_config(weighting_config)
        self.scaling = instantiate_from_config(scaling_config)

    def possibly_quantize_sigma(self, sigma):
        return sigma

    def possibly_quantize_c_noise(self, c_noise):
        return c_noise

    def w(self, sigma):
        return self.weighting(sigma)


class DiscreteDenoiser(Denoiser):
    def __init__(self, weighting_config, scaling_config, num_idx, discretization_config, do_append_zero=False,
                 quantize_c_noise=True, flip=True):
        super().__init__(weighting_config, scaling_config)
        sigmas = instantiate_from_config(discretization_config)(num_idx, do_append_zero=do_append_zero, flip=flip)
        self.register_buffer("sigmas", sigmas)
        self._sigmas_host = sigmas.clone()          # host copy: quantisation never syncs the device
        self.quantize_c_noise = quantize_c_noise

    def sigma_to_idx(self, sigma):
        s = sigma.detach().to("cpu", torch.float32)
        dists = s - self._sigmas_host[:, None]
        return dists.abs().argmin(dim=0).view(sigma.shape)

    def idx_to_sigma(self, idx):
        return self._sigmas_host[idx]

    def possibly_quantize_sigma(self, sigma):
        return self.idx_to_sigma(self.sigma_to_idx(sigma))

    def possibly_quantize_c_noise(self, c_noise):
        return self.sigma_to_idx(c_noise) if self.quantize_c_noise else c_noise


class DiscreteDenoiserWithControl(DiscreteDenoiser):
    def __call__(self, network, input, sigma, cond, control_scale, fbcache_mode=None, partial_info=None):
        sigma = self.possibly_quantize_sigma(sigma)                  # host fp32 [N]
        c_skip, c_out, c_in, c_noise = self.scaling(sigma)
        c_noise = self.possibly_quantize_c_noise(c_noise)            # int64 table index [N]
        if float(c_in.min()) != float(c_in.max()):
            raise NotImplementedError("per-sample sigmas inside one batch are not produced by RestoreEDMSampler")
        net_dtype = getattr(network, "dtype", torch.float16)      # fp32 = the fp32-operand kernel family (diffusion_dtype: fp32)
        x_in = ops.nchw_to_nhwc(input, net_dtype, scale=float(c_in[0]))
        out = network(x_in, c_noise.to(input.device, torch.float32), cond, control_scale, fbcache_mode, partial_info)
        if "stage1" in fbcache_mode:
            return out
        return ops.denoiser_out(out, input, float(c_out[0]), float(c_skip[0]))
