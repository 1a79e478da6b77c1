"""Stage-2 networks on the MI355X kernel library: GLVControl (ControlNet encoder), LightGLVUNet
(SDXL UNet + 12 control adapters), ZeroSFT, ZeroCrossAttn.

Drop-in for models/modules/SR_modules.py: same constructor kwargs (yaml :24-41, :46-64), same
parameter names, same ``partial_info`` protocol for the first-block cache (:660-730).  Internally
every tensor is 16-bit NHWC; ``forward`` takes / returns NHWC tensors (ControlWrapper converts at
the sgm boundary).

ZeroSFT (:59-110) as kernels:
   h' = h + zero_conv(c)                    1x1 conv, residual in the epilogue
   actv = SiLU(conv3x3(c))                  SiLU in the epilogue
   gamma|beta = conv3x3(actv)               ONE conv with zero_mul and zero_add stacked along Cout
   out = GN32([h_ori | h']) * (1+gamma) + beta     two-source GroupNorm writing the concatenated
                                                   tensor, modulation in the same pass
ZeroCrossAttn (:113-149): GN both inputs, q GEMM / fused k|v GEMM, d=64 flash attention over the
N x N decoder-token x control-token pairs, to_out GEMM with ``x_in + control_scale * (.)`` epilogue.
"""
import torch
from torch import nn

from ... import ops
from ..._lib import ACT_SILU, RsvldError
from ...hipnn import HipNet
from ...sgm.modules.attention import CrossAttention, SpatialTransformer
from ...sgm.modules.diffusionmodules.openaimodel import (Downsample, ResBlock, TimestepBlock,
                                                        TimestepEmbedSequential, UNetModel, Upsample)
from ...sgm.modules.diffusionmodules.util import conv_nd, normalization, zero_module


class ZeroSFT(nn.Module):
    def __init__(self, label_nc, norm_nc, concat_channels=0, norm=True, mask=False):
        super().__init__()
        if not norm:
            raise NotImplementedError("ZeroSFT is always built with norm=True (SR_modules.py:571-573)")
        self.norm = norm
        self.param_free_norm = normalization(norm_nc + concat_channels)
        nhidden = 128
        self.mlp_shared = nn.Sequential(nn.Conv2d(label_nc, nhidden, kernel_size=3, padding=1), nn.SiLU())
        self.zero_mul = zero_module(nn.Conv2d(nhidden, norm_nc + concat_channels, kernel_size=3, padding=1))
        self.zero_add = zero_module(nn.Conv2d(nhidden, norm_nc + concat_channels, kernel_size=3, padding=1))
        self.zero_conv = zero_module(conv_nd(2, label_nc, norm_nc, 1, 1, 0))
        self.pre_concat = bool(concat_channels != 0)
        self.mask = mask

    def run(self, rt, c, h, h_ori=None, control_scale=1):
        assert self.mask is False
        if h_ori is not None and not self.pre_concat:
            raise NotImplementedError("post-concat ZeroSFT is not instantiated by LightGLVUNet")
        cat = h_ori is not None and self.pre_concat
        c = ops.maybe_planes(c)                                                      # (split precision: two convs read it)
        hz = ops.conv2d(c, rt.pk(self.zero_conv), pad=0, residual=h)                 # h + zero_conv(c)
        actv = ops.conv2d(c, rt.pk(self.mlp_shared[0]), pad=1, act=ACT_SILU, out_planes=True)
        gb = ops.conv2d(actv, rt.pk_cat([self.zero_mul, self.zero_add], "gamma_beta"), pad=1)
        Cn = self.param_free_norm.num_channels
        gn = self.param_free_norm
        x1, x2 = (h_ori, hz) if cat else (hz, None)
        out = ops.group_norm(x1, gn.weight, gn.bias, gn.num_groups, gn.eps, x2=x2,
                             mod_scale1p=gb[..., :Cn], mod_shift=gb[..., Cn:])
        if control_scale != 1:
            h_raw = ops.concat_c(h_ori, h) if cat else h
            out = ops.axpby(out, h_raw, float(control_scale), 1.0 - float(control_scale))
        return out


class ZeroCrossAttn(nn.Module):
    def __init__(self, context_dim, query_dim, zero_out=True, mask=False):
        super().__init__()
        self.attn = CrossAttention(query_dim=query_dim, context_dim=context_dim, heads=query_dim // 64, dim_head=64)
        self.norm1 = normalization(query_dim)
        self.norm2 = normalization(context_dim)
        self.mask = mask

    def run(self, rt, context, x, control_scale=1):
        assert self.mask is False
        B, H, W, Cq = x.shape
        n1, n2 = self.norm1, self.norm2
        xq = ops.group_norm(x, n1.weight, n1.bias, n1.num_groups, n1.eps, planes=True, group="qkv").reshape(B, H * W, Cq)   # feed linears only
        ctx = ops.group_norm(context, n2.weight, n2.bias, n2.num_groups, n2.eps, planes=True, group="qkv")
        ctx = ctx.reshape(B, ctx.shape[1] * ctx.shape[2], ctx.shape[3])
        cache, rt.cache_context_kv = rt.cache_context_kv, False   # the control features change every step
        try:
            out = self.attn.run(rt, xq, ctx, residual=x.reshape(B, H * W, Cq), alpha=float(control_scale))
        finally:
            rt.cache_context_kv = cache
        return out.reshape(B, H, W, Cq)


class GLVControl(UNetModel):
    """ControlNet encoder (SR_modules.py:152-537): the SDXL encoder half + ``input_hint_block``; returns
    the 9 input-block features + the middle feature."""

    def __init__(self, in_channels, model_channels, out_channels, *args, input_upscale=1, **kwargs):
        super().__init__(in_channels, model_channels, out_channels, *args, **kwargs)
        if input_upscale != 1:
            raise NotImplementedError("input_upscale != 1 is not used by juggernautXL.yaml")
        self.input_upscale = input_upscale
        self.input_hint_block = TimestepEmbedSequential(zero_module(conv_nd(2, in_channels, model_channels, 3, padding=1)))

    def _build_decoder(self, *a, **k):   # the ControlNet has no decoder and no `out`
        pass

    def forward(self, x, timesteps, xt, context=None, y=None, **kwargs):
        """x: LQ latent (control), xt: noisy latent — NHWC 16-bit.  -> list of 10 NHWC feature maps."""
        emb_rows = self.emb_rows(self.embed(timesteps, y))
        hint = self.input_hint_block.run(self, x, emb_rows, context)
        hs, h = [], xt
        for i, module in enumerate(self.input_blocks):
            if i == 0:   # h = conv_in(xt) + guided_hint, the add fused in the conv epilogue (:521-524)
                conv = module[0]
                h = ops.conv2d(h, self.pk(conv), pad=conv.padding[0], residual=hint)
            else:
                h = module.run(self, h, emb_rows, context)
            hs.append(h)
        h = self.middle_block.run(self, h, emb_rows, context)
        hs.append(h)
        return hs


class LightGLVUNet(UNetModel):
    def __init__(self, mode="", project_type="ZeroSFT", project_channel_scale=1, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if mode == "XL-base":
            cond_output_channels = [320] * 4 + [640] * 3 + [1280] * 3
            project_channels = [160] * 4 + [320] * 3 + [640] * 3
            concat_channels = [320] * 2 + [640] * 3 + [1280] * 4 + [0]
            cross_attn_insert_idx = [6, 3]
            self.progressive_mask_nums = [0, 3, 7, 11]
        elif mode == "XL-refine":
            cond_output_channels = [384] * 4 + [768] * 3 + [1536] * 6
            project_channels = [192] * 4 + [384] * 3 + [768] * 6
            concat_channels = [384] * 2 + [768] * 3 + [1536] * 7 + [0]
            cross_attn_insert_idx = [9, 6, 3]
            self.progressive_mask_nums = [0, 3, 6, 10, 14]
        else:
            raise NotImplementedError
        if project_type != "ZeroSFT":
            raise NotImplementedError("project_type: ZeroSFT (juggernautXL.yaml:47)")
        project_channels = [int(c * project_channel_scale) for c in project_channels]
        self.cache_threshold = 0.1
        self.project_modules = nn.ModuleList()
        for i in range(len(cond_output_channels)):
            self.project_modules.append(ZeroSFT(project_channels[i], cond_output_channels[i],
                                                concat_channels=concat_channels[i]))
        for i in cross_attn_insert_idx:
            self.project_modules.insert(i, ZeroCrossAttn(cond_output_channels[i], concat_channels[i]))

    # ---- the two halves of a forward, split where the first-block cache looks (:660-730)
    def _encode(self, x, emb_rows, context):
        hs, h = [], x
        for module in self.input_blocks:
            h = module.run(self, h, emb_rows, context)
            hs.append(h)
        return h, hs

    def _decode(self, h, hs, emb_rows, context, control, control_scale, adapter_idx, control_idx, run_middle=True):
        pm = self.project_modules
        if run_middle:
            h = self.middle_block.run(self, h, emb_rows, context)
        h = pm[adapter_idx].run(self, control[control_idx], h, control_scale=control_scale)
        adapter_idx -= 1
        control_idx -= 1
        for module in self.output_blocks:
            _h = hs.pop()
            h = pm[adapter_idx].run(self, control[control_idx], _h, h, control_scale=control_scale)
            adapter_idx -= 1
            if len(module) == 3:
                assert isinstance(module[2], Upsample)
                for layer in module[:2]:
                    if isinstance(layer, TimestepBlock):
                        h = layer.run(self, h, emb_rows)
                    elif isinstance(layer, SpatialTransformer):
                        h = layer.run(self, h, context)
                    else:
                        h = layer.run(self, h)
                h = pm[adapter_idx].run(self, control[control_idx], h, control_scale=control_scale)
                adapter_idx -= 1
                h = module[2].run(self, h)
            else:
                h = module.run(self, h, emb_rows, context)
            control_idx -= 1
        return self.run_out(h)

    @torch.no_grad()
    def forward(self, x, timesteps=None, context=None, y=None, control=None, control_scale=1.0,
                fbcache_mode="none", partial_info=None, **kwargs):
        assert (y is not None) == (self.num_classes is not None)
        if fbcache_mode in ("none", "input_stage1", "middle_stage1"):
            emb_rows = self.emb_rows(self.embed(timesteps, y))
        adapter0, control0 = len(self.project_modules) - 1, len(control) - 1 if control is not None else None

        if fbcache_mode == "none":
            h, hs = self._encode(x, emb_rows, context)
            return self._decode(h, hs, emb_rows, context, control, control_scale, adapter0, control0)
        if fbcache_mode == "input_stage1":
            h, hs = self._encode(x, emb_rows, context)
            return {"mode": "input", "h": h, "hs": hs, "emb": emb_rows, "context": context, "control": control,
                    "adapter_idx": adapter0, "control_idx": control0}
        if fbcache_mode == "input_stage2":
            if partial_info is None or partial_info.get("mode", "") != "input":
                raise ValueError("input_stage2 requires partial_info from input_stage1")
            p = partial_info
            return self._decode(p["h"], p["hs"], p["emb"], p["context"], p["control"], control_scale,
                                p["adapter_idx"], p["control_idx"])
        if fbcache_mode == "middle_stage1":
            h, hs = self._encode(x, emb_rows, context)
            h = self.middle_block.run(self, h, emb_rows, context)
            return {"mode": "middle", "h": h, "hs": hs, "emb": emb_rows, "context": context, "control": control,
                    "adapter_idx": adapter0, "control_idx": control0}
        if fbcache_mode == "middle_stage2":
            if partial_info is None or partial_info.get("mode", "") != "middle":
                raise ValueError("middle_stage2 requires partial_info from middle_stage1")
            p = partial_info
            return self._decode(p["h"], p["hs"], p["emb"], p["context"], p["control"], control_scale,
                                p["adapter_idx"], p["control_idx"], run_middle=False)
        if fbcache_mode in ("output_stage1", "output_stage2"):
            raise NotImplementedError("fb_mode 'output_stage' is never selected (sampling.py:541-543 fixes 'input_stage')")
        raise ValueError(f"Unknown fbcache_mode={fbcache_mode}")
