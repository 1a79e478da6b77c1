"""AutoencoderKL (reference: sgm/models/autoencoder.py:282-321): encoder / decoder + the 1x1
``quant_conv`` / ``post_quant_conv``; ``AutoencoderKLInferenceWrapper.encode`` returns a SAMPLE of
the posterior.  Image tensors cross this API as fp32 NCHW like in the reference; latents too."""
import torch
from torch import nn

from ... import ops
from ...hipnn import HipNet
from ..modules.diffusionmodules.model import Decoder, Encoder
from ..modules.distributions.distributions import DiagonalGaussianDistribution


class AutoencoderKL(HipNet):
    compute_dtype = torch.bfloat16
    split = False     # with compute_dtype fp32: the split-operand precision mode (``ae_dtype: split``)

    def __init__(self, embed_dim: int, **kwargs):
        super().__init__()
        ddconfig = kwargs.pop("ddconfig")
        ckpt_path = kwargs.pop("ckpt_path", None)
        kwargs.pop("lossconfig", None)
        kwargs.pop("monitor", None)
        assert ddconfig["double_z"]
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        if ckpt_path is not None:
            sd = torch.load(ckpt_path, map_location="cpu")
            self.load_state_dict(sd.get("state_dict", sd), strict=False)

    def set_compute_dtype(self, dt):
        """bf16 (the reference's default), fp16 (tests only: real SDXL-VAE activations overflow it) or fp32 (``ae_dtype:
        fp32``: the fp32-operand kernel family, csrc/f32.hip)."""
        for m in (self.encoder, self.decoder, getattr(self, "denoise_encoder", None)):
            if m is not None:
                HipNet.set_compute_dtype(m, dt)
        HipNet.set_compute_dtype(self, dt)

    def moments(self, x, encoder=None):
        """posterior parameters, fp32 NHWC ``[B, h, w, 8]`` (mean | logvar)."""
        with ops.f32_split(self.split if self.compute_dtype == torch.float32 else None):
            h = (encoder or self.encoder)(x)
            return ops.conv2d(h, self.pk(self.quant_conv), pad=0, out_f32=True)

    def encode(self, x):
        assert not self.training, f"{self.__class__.__name__} only supports inference currently"
        return DiagonalGaussianDistribution(self.moments(x), channels=self.embed_dim)

    def decode(self, z, **decoder_kwargs):
        """z fp32 NCHW ``[B,4,h,w]`` -> fp32 NCHW image ``[B,3,8h,8w]``."""
        with ops.f32_split(self.split if self.compute_dtype == torch.float32 else None):
            zin = ops.nchw_to_nhwc(z.float().contiguous(), self.compute_dtype)
            h = ops.conv2d(zin, self.pk(self.post_quant_conv), pad=0)
            dec = self.decoder(h, **decoder_kwargs)
            return ops.nhwc_to_nchw(dec, channels=self.decoder.out_ch)


class AutoencoderKLInferenceWrapper(AutoencoderKL):
    def encode(self, x):
        return super().encode(x).sample()
