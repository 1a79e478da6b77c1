"""Transformer blocks of the SDXL UNet / ControlNet on the MI355X kernel library.

Parameter containers named like sgm/modules/attention.py (CrossAttention :196-285 ==
MemoryEfficientCrossAttention :288-373 parameter-wise, BasicTransformerBlock :376-486,
SpatialTransformer :533-635, GEGLU/FeedForward :84-110); execution is NHWC / token-major 16-bit:

  self-attention   LN -> ONE fused q|k|v GEMM -> flash attention d=64 -> to_out GEMM (+bias, +x)
  cross-attention  LN -> q GEMM; context -> fused k|v GEMM (cached per context tensor: the text
                   embedding does not change over the 50 sampler steps) -> attention -> to_out (+x)
  feed-forward     LN -> GEGLU GEMM with the x*gelu(gate) product in the GEMM epilogue -> GEMM (+x)
  SpatialTransformer  GN(eps 1e-6) -> proj_in GEMM -> blocks -> proj_out GEMM with the outer residual
The three residual adds of a block therefore cost no extra pass over the tokens.
"""
import math

import torch
from torch import nn

from ... import ops
from ..._lib import ACT_GEGLU
from ...models.modules.DFBCache import RowSubset


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def Normalize(in_channels):
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.0):
        super().__init__()
        if not glu:
            raise NotImplementedError("SDXL blocks use gated_ff=True (attention.py:387)")
        inner = int(dim * mult)
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out or dim))

    def run(self, rt, x_norm, residual):
        # (split precision: ``x_norm`` arrives as planes, or as fp16 when the policy's "ff" group is set -- then both layers run as
        #  fp16 activation x weight pair, the GEGLU product stays fp16 and only the residual stream is fp32)
        g = ops.linear(x_norm, rt.pk(self.net[0].proj, "geglu", geglu=True), act=ACT_GEGLU, out_planes=True, out_group="ff")   # feeds a linear only
        return ops.linear(g, rt.pk(self.net[2]), residual=residual, group="ff_out")


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0, backend=None, **kwargs):
        super().__init__()
        inner = dim_head * heads
        self.is_self = context_dim is None
        context_dim = query_dim if context_dim is None else context_dim
        self.scale, self.heads, self.dim_head = dim_head ** -0.5, heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))

    def run(self, rt, x, context=None, residual=None, alpha=1.0):
        """x ``[B,N,C]`` (already normalised), context ``[B,M,Cc]`` or None -> to_out(attn)*alpha + residual."""
        inner = self.heads * self.dim_head
        if context is None:
            qkv = ops.linear(x, rt.pk_cat([self.to_q, self.to_k, self.to_v], "qkv"), out_planes=True, out_group="attn")   # feeds the attention only
            q, k, v = qkv[..., :inner], qkv[..., inner:2 * inner], qkv[..., 2 * inner:]
        else:
            q = ops.linear(x, rt.pk(self.to_q), out_planes=True, out_group="attn")
            rows = None
            if isinstance(context, RowSubset):   # a sub-batch of a cached context: project the full one, pick rows
                context, rows = context.full, context.rows
            # valid only while the very same tensor object (held alive here) is passed, unmodified
            # (keyed on the precision too: fp32 and split share one pack dictionary, and a cached K/V is a tensor of ONE of them)
            ckey = ("ctx_kv", id(self), ops.precision_token())
            ent = rt._pk.get(ckey) if rt.cache_context_kv else None
            if ent is not None and ent[0] is context and ent[1] == context._version:
                kv = ent[2]
            else:
                kv = ops.linear(context, rt.pk_cat([self.to_k, self.to_v], "kv"), out_planes=True, out_group="attn")
                if rt.cache_context_kv:
                    rt._pk[ckey] = (context, context._version, kv)
            if rows is not None:
                kv = kv.index_select(0, rows)
            k, v = kv[..., :inner], kv[..., inner:]
        o = ops.attention(q, k, v, heads=self.heads, scale=self.scale)
        return ops.linear(o, rt.pk(self.to_out[0]), residual=residual, alpha=alpha, group="attn_out")


MemoryEfficientCrossAttention = CrossAttention  # same parameters; the kernel IS the memory-efficient path


class BasicTransformerBlock(nn.Module):
    ATTENTION_MODES = {"softmax": CrossAttention, "softmax-xformers": MemoryEfficientCrossAttention}

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False, attn_mode="softmax", sdp_backend=None):
        super().__init__()
        assert attn_mode in self.ATTENTION_MODES
        self.disable_self_attn = disable_self_attn
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                                    context_dim=context_dim if disable_self_attn else None)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head,
                                    dropout=dropout)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)

    def run(self, rt, x, context=None):
        n = ops.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, planes=True, group="qkv")
        x = self.attn1.run(rt, n, context if self.disable_self_attn else None, residual=x)
        n = ops.layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, planes=True, group="qkv")
        x = self.attn2.run(rt, n, context, residual=x)
        n = ops.layer_norm(x, self.norm3.weight, self.norm3.bias, self.norm3.eps, planes=True, group="ff")
        return self.ff.run(rt, n, residual=x)


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, disable_self_attn=False,
                 use_linear=False, attn_type="softmax", use_checkpoint=True, sdp_backend=None):
        super().__init__()
        if not use_linear:
            raise NotImplementedError("the SDXL config sets use_linear_in_transformer: True (juggernautXL.yaml:37,61)")
        if context_dim is not None and not isinstance(context_dim, (list, tuple)):
            context_dim = [context_dim]
        if isinstance(context_dim, (list, tuple)):
            if depth != len(context_dim):
                assert all(c == context_dim[0] for c in context_dim)
                context_dim = depth * [context_dim[0]]
        else:
            context_dim = [None] * depth
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.norm = Normalize(in_channels)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim[d],
                                  disable_self_attn=disable_self_attn, attn_mode=attn_type, checkpoint=use_checkpoint)
            for d in range(depth)])
        self.proj_out = zero_module(nn.Linear(inner, in_channels))
        self.use_linear = use_linear

    def run(self, rt, x, context=None):
        """x NHWC ``[B,H,W,C]`` -> same shape."""
        B, H, W, Cc = x.shape
        contexts = context if isinstance(context, list) else [context]
        h = ops.group_norm(x, self.norm.weight, self.norm.bias, self.norm.num_groups, self.norm.eps, planes=True, group="proj")
        h = ops.linear(h.reshape(B, H * W, Cc), rt.pk(self.proj_in))
        for i, blk in enumerate(self.transformer_blocks):
            h = blk.run(rt, h, contexts[i if len(contexts) > 1 else 0])
        h = ops.linear(h, rt.pk(self.proj_out), residual=x.reshape(B, H * W, Cc))
        return h.reshape(B, H, W, Cc)
