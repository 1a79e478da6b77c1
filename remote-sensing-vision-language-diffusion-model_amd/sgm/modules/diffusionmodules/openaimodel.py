"""SDXL UNet building blocks on the MI355X kernel library.

Parameter containers named like sgm/modules/diffusionmodules/openaimodel.py (ResBlock :207-350,
Upsample :102-145, Downsample :164-204, TimestepEmbedSequential :75-99, UNetModel :500-1010);
activations are 16-bit NHWC.

  ResBlock     GN32+SiLU -> Conv3x3 (+ emb projection as the epilogue's per-image row vector)
               -> GN32+SiLU -> Conv3x3 with the skip (identity or 1x1 conv) added in its epilogue
  Upsample     nearest x2 folded into the conv's input gather (no fp32 round trip, no 4x tensor)
  Downsample   stride-2 conv
The emb_layers Linear of every ResBlock of a network is evaluated in one stacked launch
(HipNet.emb_rows).
"""
import torch
from torch import nn

from .... import ops
from ....hipnn import HipNet
from ...util import default, exists
from ..attention import SpatialTransformer
from .util import conv_nd, linear, normalization, timestep_embedding, zero_module


class TimestepBlock(nn.Module):
    """Marker: ``run(rt, x, emb_rows)`` takes the timestep embedding."""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    def run(self, rt, x, emb_rows, context=None):
        for layer in self:
            if isinstance(layer, TimestepBlock):
                x = layer.run(rt, x, emb_rows)
            elif isinstance(layer, SpatialTransformer):
                x = layer.run(rt, x, context)
            elif isinstance(layer, nn.Conv2d):
                x = ops.conv2d(x, rt.pk(layer), pad=layer.padding[0])
            else:
                x = layer.run(rt, x)
        return x


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1, third_up=False):
        super().__init__()
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, use_conv, dims
        if not use_conv:
            raise NotImplementedError("conv_resample=True in every shipped config")
        self.conv = conv_nd(dims, self.channels, self.out_channels, 3, padding=padding)

    def run(self, rt, x):
        return ops.conv2d(x, rt.pk(self.conv), pad=self.conv.padding[0], upsample=True)


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1, third_down=False):
        super().__init__()
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, use_conv, dims
        if not use_conv:
            raise NotImplementedError("conv_resample=True in every shipped config")
        self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=padding)

    def run(self, rt, x):
        return ops.conv2d(x, rt.pk(self.op), stride=2, pad=self.op.padding[0])


class ResBlock(TimestepBlock):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False,
                 dims=2, use_checkpoint=False, up=False, down=False, kernel_size=3, exchange_temb_dims=False,
                 skip_t_emb=False):
        super().__init__()
        if up or down or use_scale_shift_norm or skip_t_emb or exchange_temb_dims:
            raise NotImplementedError("resblock_updown / scale-shift norm are not used by juggernautXL.yaml")
        self.channels, self.emb_channels, self.out_channels = channels, emb_channels, out_channels or channels
        pad = kernel_size // 2
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(),
                                       conv_nd(dims, channels, self.out_channels, kernel_size, padding=pad))
        self.emb_layers = nn.Sequential(nn.SiLU(), linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(conv_nd(dims, self.out_channels, self.out_channels, kernel_size,
                                                            padding=pad)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, kernel_size, padding=pad)
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)

    @property
    def _emb_linear(self):  # picked up by HipNet.emb_rows
        return self.emb_layers[1]

    def run(self, rt, x, emb_rows):
        gn1, conv1 = self.in_layers[0], self.in_layers[2]
        gn2, conv2 = self.out_layers[0], self.out_layers[3]
        h = ops.conv2d(x, rt.pk(conv1), pad=conv1.padding[0], rowvec=emb_rows[id(self)], stats=True,
                       norm=(gn1.weight, gn1.bias, gn1.num_groups, gn1.eps, True), norm_group="conv1")
        if isinstance(self.skip_connection, nn.Conv2d):
            skip = ops.conv2d(x, rt.pk(self.skip_connection), pad=self.skip_connection.padding[0])
        else:
            skip = x
        return ops.conv2d(h, rt.pk(conv2), pad=conv2.padding[0], residual=skip,
                          norm=(gn2.weight, gn2.bias, gn2.num_groups, gn2.eps, True), norm_group="conv2")


class UNetModel(HipNet):
    """Constructor-compatible with openaimodel.py:500-1010 for the SDXL-style options the shipped
    yaml uses (spatial transformer, linear projections, num_head_channels, sequential label_emb)."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False,
                 use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None, legacy=True,
                 disable_self_attentions=None, num_attention_blocks=None, disable_middle_self_attn=False,
                 use_linear_in_transformer=False, spatial_transformer_attn_type="softmax", adm_in_channels=None,
                 use_fairscale_checkpoint=False, offload_to_cpu=False, transformer_depth_middle=None, **ignored):
        super().__init__()
        if not use_spatial_transformer or context_dim is None:
            raise NotImplementedError("the SR networks are SDXL UNets with spatial transformers")
        if resblock_updown or use_scale_shift_norm or n_embed is not None or dims != 2:
            raise NotImplementedError("option not used by juggernautXL.yaml")
        if num_heads == -1 and num_head_channels == -1:
            raise ValueError("Either num_heads or num_head_channels has to be set")
        channel_mult = list(channel_mult)
        if isinstance(context_dim, (list, tuple)):
            context_dim = list(context_dim)
        if isinstance(transformer_depth, int):
            transformer_depth = len(channel_mult) * [transformer_depth]
        transformer_depth = list(transformer_depth)
        transformer_depth_middle = default(transformer_depth_middle, transformer_depth[-1])
        if isinstance(num_res_blocks, int):
            num_res_blocks = len(channel_mult) * [num_res_blocks]
        self.num_res_blocks = list(num_res_blocks)
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.attention_resolutions, self.channel_mult, self.num_classes = attention_resolutions, channel_mult, num_classes

        ted = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, ted), nn.SiLU(), linear(ted, ted))
        if num_classes is not None:
            if num_classes != "sequential":
                raise NotImplementedError("only num_classes: sequential (SDXL vector conditioning) is on the path")
            assert adm_in_channels is not None
            self.label_emb = nn.Sequential(nn.Sequential(linear(adm_in_channels, ted), nn.SiLU(), linear(ted, ted)))

        def heads_for(ch):
            if num_head_channels == -1:
                nh, dh = num_heads, ch // num_heads
            else:
                nh, dh = ch // num_head_channels, num_head_channels
            if legacy:
                dh = ch // nh
            return nh, dh

        def transformer(ch, depth, disabled_sa=False):
            nh, dh = heads_for(ch)
            return SpatialTransformer(ch, nh, dh, depth=depth, context_dim=context_dim, disable_self_attn=disabled_sa,
                                      use_linear=use_linear_in_transformer, attn_type=spatial_transformer_attn_type,
                                      use_checkpoint=use_checkpoint)

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
        chans, ch, ds = [model_channels], model_channels, 1
        for level, mult in enumerate(channel_mult):
            for nr in range(self.num_res_blocks[level]):
                layers = [ResBlock(ch, ted, dropout, out_channels=mult * model_channels, dims=dims)]
                ch = mult * model_channels
                if ds in attention_resolutions and (not exists(num_attention_blocks) or nr < num_attention_blocks[level]):
                    layers.append(transformer(ch, transformer_depth[level],
                                              disable_self_attentions[level] if exists(disable_self_attentions) else False))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(
            ResBlock(ch, ted, dropout, dims=dims),
            transformer(ch, transformer_depth_middle, disable_middle_self_attn),
            ResBlock(ch, ted, dropout, dims=dims))
        self._build_decoder(chans, ch, ds, ted, dropout, dims, conv_resample, transformer, transformer_depth,
                            num_attention_blocks, disable_self_attentions)

    def _build_decoder(self, chans, ch, ds, ted, dropout, dims, conv_resample, transformer, transformer_depth,
                       num_attention_blocks, disable_self_attentions):
        mc = self.model_channels
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(self.channel_mult))[::-1]:
            for i in range(self.num_res_blocks[level] + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, ted, dropout, out_channels=mc * mult, dims=dims)]
                ch = mc * mult
                if ds in self.attention_resolutions and (not exists(num_attention_blocks) or i < num_attention_blocks[level]):
                    layers.append(transformer(ch, transformer_depth[level],
                                              disable_self_attentions[level] if exists(disable_self_attentions) else False))
                if level and i == self.num_res_blocks[level]:
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), zero_module(conv_nd(dims, mc, self.out_channels, 3, padding=1)))

    # ---- shared forward pieces
    def embed(self, timesteps, y):
        """fp32 ``emb [N, 4*model_channels]`` = time_embed(t_emb) + label_emb(y)  (openaimodel.py:985-990)."""
        t_emb = timestep_embedding(timesteps, self.model_channels)
        te = self.time_embed
        emb = ops.linear_small(ops.linear_small(t_emb, te[0].weight, te[0].bias, 0, 1), te[2].weight, te[2].bias)
        if self.num_classes is not None:
            le = self.label_emb[0]
            lab = ops.linear_small(ops.linear_small(y.float().contiguous(), le[0].weight, le[0].bias, 0, 1),
                                   le[2].weight, le[2].bias)
            emb = ops.add_f32(emb, lab)
        return emb

    def run_out(self, h):
        gn, conv = self.out[0], self.out[2]
        return ops.conv2d(h, self.pk(conv), pad=conv.padding[0], out_f32=True,
                          norm=(gn.weight, gn.bias, gn.num_groups, gn.eps, True))

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        """Plain SDXL UNet forward (openaimodel.py:968-1010) on NHWC 16-bit ``x``; skip connections are
        consumed as second inputs of the decoder ResBlocks' GN / convs via a materialised concat."""
        emb_rows = self.emb_rows(self.embed(timesteps, y))
        hs, h = [], x
        for module in self.input_blocks:
            h = module.run(self, h, emb_rows, context)
            hs.append(h)
        h = self.middle_block.run(self, h, emb_rows, context)
        for module in self.output_blocks:
            h = module.run(self, ops.concat_c(h, hs.pop()), emb_rows, context)
        return self.run_out(h)
