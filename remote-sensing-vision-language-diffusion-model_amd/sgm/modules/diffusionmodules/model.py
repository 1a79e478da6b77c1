"""SD-VAE encoder / decoder on the MI355X kernel library.

Parameter containers named like sgm/modules/diffusionmodules/model.py (ResnetBlock :91-148,
Up/Downsample :55-88, MemoryEfficientAttnBlock :201-262 == AttnBlock parameter-wise,
Encoder :482-596, Decoder :599-743), so ``first_stage_model.*`` checkpoint keys load unchanged and
``utils.tilevae.VAEHook`` finds the attributes it walks (:405-499).  bf16 storage (the reference
refuses fp16 here, SR_model.py:28-29), fp32 accumulate / statistics; with ``compute_dtype = torch.float32``
(``ae_dtype: fp32``) the same code runs on the fp32-operand kernel family (csrc/f32.hip, unfused GroupNorm):

  ResnetBlock   GN(eps 1e-6)+swish -> Conv3x3 -> GN+swish -> Conv3x3 (+ x or nin_shortcut(x) in the epilogue)
  Downsample    the asymmetric (0,1,0,1) zero pad is a property of the gather, not a padded copy
  Upsample      nearest x2 folded into the conv's gather
  AttnBlock     fused q|k|v 1x1 GEMM -> single-head d=512 flash attention -> proj_out (+x)
"""
import math

import torch
from torch import nn

from .... import ops
from ...._lib import RsvldError
from ....hipnn import HipNet


def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if not with_conv:
            raise NotImplementedError("resamp_with_conv=True in the shipped ddconfig")
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def run(self, rt, x):
        return ops.conv2d(x, rt.pk(self.conv), pad=1, upsample=True)


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if not with_conv:
            raise NotImplementedError("resamp_with_conv=True in the shipped ddconfig")
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def run(self, rt, x):
        return ops.conv2d(x, rt.pk(self.conv), stride=2, pad=(0, 0, 1, 1))   # F.pad(x, (0,1,0,1)) + stride 2


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        if temb_channels > 0 or conv_shortcut:
            raise NotImplementedError("the VAE uses temb_ch = 0 and 1x1 shortcuts")
        self.in_channels = in_channels
        self.out_channels = out_channels = in_channels if out_channels is None else out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def run(self, rt, x, norm1_stats=None, norm2_stats=None):
        """``norm*_stats``: optional externally supplied (mean, var) ``[B,32,2]`` — the tiled VAE's
        cross-tile GroupNorm statistics (utils/tilevae.py:599-674)."""
        skip = ops.conv2d(x, rt.pk(self.nin_shortcut), pad=0) if self.in_channels != self.out_channels else x
        if norm1_stats is None and norm2_stats is None:   # GN+swish fused into each conv's input staging
            n1, n2 = self.norm1, self.norm2
            h = ops.conv2d(x, rt.pk(self.conv1), pad=1, stats=True, norm=(n1.weight, n1.bias, n1.num_groups, n1.eps, True))
            return ops.conv2d(h, rt.pk(self.conv2), pad=1, residual=skip, stats=True,
                              norm=(n2.weight, n2.bias, n2.num_groups, n2.eps, True))
        h = _gn(x, self.norm1, True, norm1_stats)
        h = ops.conv2d(h, rt.pk(self.conv1), pad=1)
        h = _gn(h, self.norm2, True, norm2_stats)
        return ops.conv2d(h, rt.pk(self.conv2), pad=1, residual=skip)


def _gn(x, norm, silu, stats=None):
    """(planes=True: honoured in the split precision only -- every GroupNorm of the VAE feeds a convolution)"""
    if stats is None:
        return ops.group_norm(x, norm.weight, norm.bias, norm.num_groups, norm.eps, silu=silu, planes=True)
    return ops.group_norm_apply(x, stats, norm.weight, norm.bias, norm.num_groups, norm.eps, silu=silu, planes=True)


class AttnBlock(nn.Module):
    """Single-head self-attention over all positions, d = in_channels (= 512)."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)

    def run(self, rt, x, norm_stats=None):
        B, H, W, Cc = x.shape
        h = _gn(x, self.norm, False, norm_stats)
        qkv = ops.conv2d(h, rt.pk_cat([self.q, self.k, self.v], "qkv"), pad=0, out_planes=True).reshape(B, H * W, 3 * Cc)
        o = ops.attention(qkv[:, :, :Cc], qkv[:, :, Cc:2 * Cc], qkv[:, :, 2 * Cc:], heads=1, scale=Cc ** -0.5)
        return ops.conv2d(o.reshape(B, H, W, Cc), rt.pk(self.proj_out), pad=0, residual=x)


MemoryEfficientAttnBlock = AttnBlock


def make_attn(in_channels, attn_type="vanilla", attn_kwargs=None):
    if attn_type in ("vanilla", "vanilla-xformers"):
        return AttnBlock(in_channels)
    if attn_type == "none":
        return nn.Identity()
    raise NotImplementedError(f"attn_type {attn_type}")


class _VAEHalf(HipNet):
    compute_dtype = torch.bfloat16

    def _in(self, x):
        """fp32 NCHW (reference layout) or packed NHWC -> NHWC compute dtype."""
        if not x.is_cuda:
            raise RsvldError("the VAE runs on the GPU only")
        if x.dim() == 4 and x.dtype in (torch.float16, torch.bfloat16) and x.shape[-1] % 8 == 0:
            if self.compute_dtype == torch.float32:
                raise RsvldError("a 16-bit NHWC tensor was handed to the fp32 VAE")
            return x if x.dtype == self.compute_dtype else x.to(self.compute_dtype)
        if x.dtype == torch.float32 and getattr(x, "_nhwc", False):   # produced by the fp32 kernel family: already NHWC
            if self.compute_dtype != torch.float32:
                raise RsvldError("an fp32 NHWC tensor was handed to the 16-bit VAE")
            return x
        return ops.nchw_to_nhwc(x, self.compute_dtype)


class Encoder(_VAEHalf):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = nn.Conv2d(in_channels, ch, kernel_size=3, stride=1, padding=1)
        curr_res, in_ch_mult = resolution, (1,) + tuple(ch_mult)
        self.in_ch_mult = in_ch_mult
        self.down = nn.ModuleList()
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res //= 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        """-> NHWC ``[B, H/8, W/8, 2*z]`` in the compute dtype."""
        h = ops.conv2d(self._in(x), self.pk(self.conv_in), pad=1)
        for i_level in range(self.num_resolutions):
            for i_block in range(self.num_res_blocks):
                h = self.down[i_level].block[i_block].run(self, h)
                if len(self.down[i_level].attn) > 0:
                    h = self.down[i_level].attn[i_block].run(self, h)
            if i_level != self.num_resolutions - 1:
                h = self.down[i_level].downsample.run(self, h)
        h = self.mid.block_1.run(self, h)
        h = self.mid.attn_1.run(self, h)
        h = self.mid.block_2.run(self, h)
        n = self.norm_out
        return ops.conv2d(h, self.pk(self.conv_out), pad=1, norm=(n.weight, n.bias, n.num_groups, n.eps, True))


class Decoder(_VAEHalf):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.give_pre_end, self.tanh_out, self.out_ch = give_pre_end, tanh_out, out_ch
        if tanh_out:
            raise NotImplementedError("tanh_out is not used by the shipped ddconfig")
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res *= 2
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)

    def forward(self, z, **kwargs):
        """z NHWC (or fp32 NCHW) -> fp32 NHWC ``[B, 8H, 8W, 8]`` (3 image channels + padding)."""
        self.last_z_shape = z.shape
        h = ops.conv2d(self._in(z), self.pk(self.conv_in), pad=1)
        h = self.mid.block_1.run(self, h)
        h = self.mid.attn_1.run(self, h)
        h = self.mid.block_2.run(self, h)
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                h = self.up[i_level].block[i_block].run(self, h)
                if len(self.up[i_level].attn) > 0:
                    h = self.up[i_level].attn[i_block].run(self, h)
            if i_level != 0:
                h = self.up[i_level].upsample.run(self, h)
        if self.give_pre_end:
            return h
        n = self.norm_out
        return ops.conv2d(h, self.pk(self.conv_out), pad=1, out_f32=True, norm=(n.weight, n.bias, n.num_groups, n.eps, True))
