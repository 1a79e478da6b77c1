"""Pre-conditioning scalings (reference: sgm/modules/diffusionmodules/denoiser_scaling.py)."""
import torch


class EpsScaling:
    def __call__(self, sigma):
        c_skip = torch.ones_like(sigma, device=sigma.device)
        c_out = -sigma
        c_in = 1 / (sigma ** 2 + 1.0) ** 0.5
        c_noise = sigma.clone()
        return c_skip, c_out, c_in, c_noise


class EDMScaling:
    def __init__(self, sigma_data=0.5):
        self.sigma_data = sigma_data

    def __call__(self, sigma):
        sd = self.sigma_data
        return sd ** 2 / (sigma ** 2 + sd ** 2), sigma * sd / (sigma ** 2 + sd ** 2) ** 0.5, \
            1 / (sigma ** 2 + sd ** 2) ** 0.5, 0.25 * sigma.log()


class VScaling:
    def __call__(self, sigma):
        return 1.0 / (sigma ** 2 + 1.0), -sigma / (sigma ** 2 + 1.0) ** 0.5, 1.0 / (sigma ** 2 + 1.0) ** 0.5, sigma.clone()
