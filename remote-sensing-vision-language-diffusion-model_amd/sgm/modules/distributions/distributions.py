"""VAE posterior (reference: sgm/modules/distributions/distributions.py:24-41,71-72).

Works on the NHWC moments the HIP encoder produces.  ``sample()`` keeps the reference's RNG
contract: the noise is drawn with the DEFAULT CPU generator (``torch.randn(shape)`` then moved to
the device), between the device-generator draws of the sampler."""
import torch

from .... import ops


class DiagonalGaussianDistribution:
    def __init__(self, parameters_nhwc, channels=None, deterministic=False):
        self.parameters = parameters_nhwc
        self.channels = parameters_nhwc.shape[-1] // 2 if channels is None else channels
        self.deterministic = deterministic

    def sample(self, scale=1.0, noise=None):
        B, H, W, _ = self.parameters.shape
        if self.deterministic:
            return self.mode(scale)
        if noise is None:
            noise = torch.randn((B, self.channels, H, W))
        return ops.gaussian_sample(self.parameters, self.channels, noise.to(self.parameters.device), scale)

    def mode(self, scale=1.0):
        return ops.gaussian_sample(self.parameters, self.channels, None, scale)
