"""Colour correction from the pre-cleaned image (reference: utils/colorfix.py:44-119), fp32 NCHW.

``wavelet_decomposition`` is 5 a-trous levels of the 3x3 [1,2,1]x[1,2,1]/16 blur with dilation 2^i and
replicate padding; each level is ONE kernel that writes the low band and accumulates the high band
(rsvld_wavelet_blur), so a reconstruction costs 10 passes over the image plus one add."""
from .. import ops


def wavelet_blur(image, radius):
    return ops.wavelet_blur(image.contiguous(), radius)


def wavelet_decomposition(image, levels=5):
    import torch
    image = image.contiguous()
    high = torch.zeros_like(image)
    for i in range(levels):
        image = ops.wavelet_blur(image, 2 ** i, high_accum=high)
    return high, image


def wavelet_reconstruction(content_feat, style_feat):
    content_high, _ = wavelet_decomposition(content_feat.float())
    _, style_low = wavelet_decomposition(style_feat.float())
    return ops.add_f32(content_high, style_low)


def adaptive_instance_normalization(content_feat, style_feat):
    return ops.adain(content_feat.float().contiguous(), style_feat.float().contiguous())
