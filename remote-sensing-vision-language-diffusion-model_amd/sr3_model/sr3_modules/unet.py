"""Stage-1 (SR3) pixel-space UNet on the MI355X kernel library.

Drop-in for the reference ``models/sr3_model/sr3_modules/unet.py`` (UNet :162-261): same
constructor arguments, same parameter names (so ``I1000000_E800_gen.pth`` loads unchanged), same
``forward(x, time)`` contract.  What differs is everything underneath: activations are 16-bit
NHWC tensors that never leave HBM in NCHW form, and every layer is a kernel of librsvld_hip.so:

  Block            GN(32)+Swish (norm.hip) -> Conv3x3 implicit GEMM on MFMA (conv_igemm.hip)
  FeatureWiseAffine all 26 noise-level projections in ONE stacked launch, consumed as the
                   per-image row vector of block1's conv epilogue
  ResnetBlock      residual / 1x1 res_conv add fused in block2's conv epilogue; the skip
                   concatenation (unet.py:257) is never materialised (two-source GN + conv)
  SelfAttention    fused qkv 1x1 GEMM -> flash attention, d = 512 (attention.hip) -> out 1x1 GEMM
                   with the residual in its epilogue; no [B,1,h,w,h,w] score tensor (:133-141)
  Upsample         nearest x2 folded into the conv's gather;  Downsample: stride-2 gather

The nn.Conv2d / nn.GroupNorm / nn.Linear children only HOLD parameters under the reference's
names; their ``forward`` is never called.
"""
import math

import torch
from torch import nn

from ... import ops
from ..._lib import RsvldError


class PositionalEncoding(nn.Module):  # parameter-free; index 0 of noise_level_mlp (unet.py:19-32)
    def __init__(self, dim):
        super().__init__()
        self.dim = dim


class Swish(nn.Module):
    pass


class FeatureWiseAffine(nn.Module):
    """unet.py:35-51 — only the additive form is used by the shipped config."""

    def __init__(self, in_channels, out_channels, use_affine_level=False):
        super().__init__()
        if use_affine_level:
            raise NotImplementedError("use_affine_level=True is not used by configs/sr_sr3.json")
        self.use_affine_level = use_affine_level
        self.noise_func = nn.Sequential(nn.Linear(in_channels, out_channels))


class Upsample(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.up = nn.Upsample(scale_factor=2, mode="nearest")
        self.conv = nn.Conv2d(dim, dim, 3, padding=1)


class Downsample(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv2d(dim, dim, 3, 2, 1)


class Block(nn.Module):
    def __init__(self, dim, dim_out, groups=32, dropout=0):
        super().__init__()
        self.block = nn.Sequential(
            nn.GroupNorm(groups, dim), Swish(),
            nn.Dropout(dropout) if dropout != 0 else nn.Identity(),
            nn.Conv2d(dim, dim_out, 3, padding=1))


class ResnetBlock(nn.Module):
    def __init__(self, dim, dim_out, noise_level_emb_dim=None, dropout=0, use_affine_level=False, norm_groups=32):
        super().__init__()
        self.noise_func = FeatureWiseAffine(noise_level_emb_dim, dim_out, use_affine_level)
        self.block1 = Block(dim, dim_out, groups=norm_groups)
        self.block2 = Block(dim_out, dim_out, groups=norm_groups, dropout=dropout)
        self.res_conv = nn.Conv2d(dim, dim_out, 1) if dim != dim_out else nn.Identity()


class SelfAttention(nn.Module):
    def __init__(self, in_channel, n_head=1, norm_groups=32):
        super().__init__()
        self.n_head = n_head
        self.norm = nn.GroupNorm(norm_groups, in_channel)
        self.qkv = nn.Conv2d(in_channel, in_channel * 3, 1, bias=False)
        self.out = nn.Conv2d(in_channel, in_channel, 1)


class ResnetBlocWithAttn(nn.Module):
    def __init__(self, dim, dim_out, *, noise_level_emb_dim=None, norm_groups=32, dropout=0, with_attn=False):
        super().__init__()
        self.with_attn = with_attn
        self.res_block = ResnetBlock(dim, dim_out, noise_level_emb_dim, norm_groups=norm_groups, dropout=dropout)
        if with_attn:
            self.attn = SelfAttention(dim_out, norm_groups=norm_groups)


class UNet(nn.Module):
    def __init__(self, in_channel=6, out_channel=3, inner_channel=32, norm_groups=32,
                 channel_mults=(1, 2, 4, 8, 8), attn_res=(8), res_blocks=3, dropout=0,
                 with_noise_level_emb=True, image_size=128):
        super().__init__()
        if not with_noise_level_emb:
            raise NotImplementedError("the SR3 sampler always conditions on the noise level")
        if isinstance(attn_res, int):
            attn_res = (attn_res,)
        self.in_channel, self.out_channel = in_channel, out_channel if out_channel is not None else in_channel
        emb = inner_channel
        self.noise_level_mlp = nn.Sequential(
            PositionalEncoding(inner_channel), nn.Linear(inner_channel, inner_channel * 4), Swish(),
            nn.Linear(inner_channel * 4, inner_channel))

        # encoder: attention sites depend on the CONSTRUCTION-time resolution only (unet.py:198,221)
        widths = [inner_channel * m for m in channel_mults]
        res, cur = image_size, inner_channel
        skips = [cur]
        downs = [nn.Conv2d(in_channel, inner_channel, kernel_size=3, padding=1)]
        for level, width in enumerate(widths):
            attn_here = res in attn_res
            for _ in range(res_blocks):
                downs.append(ResnetBlocWithAttn(cur, width, noise_level_emb_dim=emb, norm_groups=norm_groups,
                                                dropout=dropout, with_attn=attn_here))
                cur = width
                skips.append(cur)
            if level != len(widths) - 1:
                downs.append(Downsample(cur))
                skips.append(cur)
                res //= 2
        self.downs = nn.ModuleList(downs)
        self.mid = nn.ModuleList([
            ResnetBlocWithAttn(cur, cur, noise_level_emb_dim=emb, norm_groups=norm_groups, dropout=dropout, with_attn=True),
            ResnetBlocWithAttn(cur, cur, noise_level_emb_dim=emb, norm_groups=norm_groups, dropout=dropout, with_attn=False)])
        ups = []
        for level in reversed(range(len(widths))):
            width, attn_here = widths[level], res in attn_res
            for _ in range(res_blocks + 1):
                ups.append(ResnetBlocWithAttn(cur + skips.pop(), width, noise_level_emb_dim=emb, norm_groups=norm_groups,
                                              dropout=dropout, with_attn=attn_here))
                cur = width
            if level != 0:
                ups.append(Upsample(cur))
                res *= 2
        self.ups = nn.ModuleList(ups)
        self.final_conv = Block(cur, self.out_channel, groups=norm_groups)

        self.compute_dtype = torch.float16   # storage / MFMA operand type of the activations
        self.pack_dtype = torch.float16      # type the weights are packed in: = compute_dtype, or fp32 masters under fp16 activations ("w2")
        self.split = None                    # with compute_dtype fp32: the ops.SplitPolicy of the split precision ("split"), else None
        self._pk = None                      # packed weights, built lazily on the parameters' device

    # ------------------------------------------------------------------ weight packing
    # ``pack_version`` counts invalidations: anything that holds raw pointers into the packed tensors (the sampler's
    # captured hipGraphs) keys itself on it and is rebuilt after a weight reload / move / dtype change.
    pack_version = 0

    def invalidate_packed(self):
        self._pk = None
        self.pack_version += 1

    def set_compute_dtype(self, dt, policy=None):
        """fp16 (default) / bf16: 16-bit storage, fp32 accumulation; fp32: the fp32-operand kernel family (what the reference's
        own Stage 1 computes in: it runs without autocast); "split": fp32 tensors, matrix products on split operands (three 16-bit
        MFMAs per product) under ``policy`` (an ``ops.SplitPolicy``, default ``ops.UNET_POLICY``); "w2" (round 5): fp16 tensors like
        "fp16", but every weight as the fp16 PAIR [W_lo | W_hi] (dtype RSVLD_F16W2, two MFMAs per product) -- the rounding of the
        WEIGHTS is the whole distance of "fp16" from the reference's CPU path after T = 50 steps."""
        split = None
        if dt == "split":               # captured hipGraphs bake the mode in: a new version drops them
            split = policy or ops.UNET_POLICY
            if not isinstance(split, ops.SplitPolicy):
                raise TypeError("set_compute_dtype: policy must be an rsvld_amd.ops.SplitPolicy")
        if split != self.split:
            self.split = split
            self.pack_version += 1
        pack = torch.float32 if dt == "w2" else None
        dt = {"fp16": torch.float16, "w2": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32, "split": torch.float32}.get(dt, dt)
        if dt not in (torch.float16, torch.bfloat16, torch.float32):
            raise ValueError(f"compute_dtype {dt!r}: fp16, w2, bf16, fp32 or split")
        pack = pack or dt
        if dt != self.compute_dtype or pack != self.pack_dtype:
            self.compute_dtype, self.pack_dtype = dt, pack
            self.invalidate_packed()

    def precision_key(self):
        return (str(self.compute_dtype), str(self.pack_dtype), None if self.split is None else self.split.key())

    def load_state_dict(self, *a, **k):
        self.invalidate_packed()
        return super().load_state_dict(*a, **k)

    def _load_from_state_dict(self, *a, **k):   # reached when a PARENT module's load_state_dict recurses into this one
        self.invalidate_packed()
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):  # .to(device) / .half() etc. move the fp32 masters
        self.invalidate_packed()
        return super()._apply(fn, *a, **k)

    def _res_blocks(self):
        for seq in (self.downs, self.mid, self.ups):
            for layer in seq:
                if isinstance(layer, ResnetBlocWithAttn):
                    yield layer

    def _pack(self):
        dev = self.noise_level_mlp[1].weight.device
        if dev.type != "cuda":
            raise RsvldError("SR3 UNet: parameters must be on the GPU (there is no CPU execution path)")
        dt = self.pack_dtype
        pk = {}

        def conv(m, **kw):
            pk[id(m)] = ops.pack_conv(m.weight, m.bias, dt, dev, **kw)

        conv(self.downs[0])
        # stacked noise-level projections: one [sum(C), emb] matrix, offsets per block
        ws, bs, off = [], [], 0
        for rb in self._res_blocks():
            lin = rb.res_block.noise_func.noise_func[0]
            pk[("nf", id(rb))] = (off, lin.out_features)
            ws.append(lin.weight.detach().float())
            bs.append(lin.bias.detach().float())
            off += lin.out_features
        pk["nf_w"] = torch.cat(ws, 0).contiguous().to(dev)
        pk["nf_b"] = torch.cat(bs, 0).contiguous().to(dev)
        for seq in (self.downs, self.mid, self.ups):
            for layer in seq:
                if isinstance(layer, (Downsample, Upsample)):
                    conv(layer.conv)
                elif isinstance(layer, ResnetBlocWithAttn):
                    rb = layer.res_block
                    c_in = rb.block1.block[0].num_channels
                    conv(rb.block1.block[3])
                    conv(rb.block2.block[3])
                    if isinstance(rb.res_conv, nn.Conv2d):
                        pk[("res", id(rb))] = c_in  # split decided at run time (skip concat or not)
                    if layer.with_attn:
                        if layer.attn.n_head != 1:
                            raise NotImplementedError("SR3 SelfAttention is built with n_head=1 (unet.py:151)")
                        self._pack_attention(pk, layer.attn, dt, dev)
        conv(self.final_conv.block[3])
        self._pk = pk
        return pk

    @staticmethod
    def _pack_attention(pk, at, dt, dev):
        """Single-head attention with keys and values taken from ONE tensor (csrc/attention.hip, "SH" note):
            softmax(q k^T / sqrt(C)) v,  q = n Wq^T, k = n Wk^T, v = n Wv^T   (unet.py:126-141, qkv without bias)
          = softmax((n Wq^T Wk) n^T / sqrt(C)) n  Wv^T
        so the query projection becomes Wk^T Wq (one C x C conv instead of the 3C-wide qkv conv), the keys AND values are
        the normalised input itself, and Wv moves into the output projection (W_out Wv).  Products in fp32 on the host."""
        w = at.qkv.weight.detach().to("cpu", torch.float32).flatten(1)          # [3C, C]
        C_ = w.shape[1]
        wq, wk, wv = w[:C_], w[C_:2 * C_], w[2 * C_:]
        if at.qkv.bias is not None:
            raise NotImplementedError("SR3 SelfAttention's qkv conv has no bias (unet.py:121)")
        pk[("attn_q", id(at))] = ops.pack_conv((wk.t() @ wq).contiguous(), None, dt, dev)
        wo = at.out.weight.detach().to("cpu", torch.float32).flatten(1)        # [C, C]
        pk[("attn_out", id(at))] = ops.pack_conv((wo @ wv).contiguous(), at.out.bias, dt, dev)

    def _res_conv_packed(self, rb, split):
        key = ("resw", id(rb), split)
        pk = self._pk
        if key not in pk:
            m = rb.res_conv
            pk[key] = ops.pack_conv(m.weight, m.bias, self.pack_dtype, m.weight.device, cin_split=split)
        return pk[key]

    # ------------------------------------------------------------------ forward pieces
    def _block(self, blk, x, x2=None, **conv_kw):
        """GN(32)+Swish+Conv3x3 as ONE fused conv call: statistics pass + conv with the normalisation applied
        while the input patch is staged (ops.conv2d norm=...); the skip concat [x | x2] is never built."""
        gn = blk.block[0]
        return ops.conv2d(x, self._pk[id(blk.block[3])], x2=x2, norm=(gn.weight, gn.bias, gn.num_groups, gn.eps, True),
                          **conv_kw)

    def _resblock(self, layer, x, x2, nf_all):
        rb = layer.res_block
        off, width = self._pk[("nf", id(layer))]
        rowvec = nf_all[:, off:off + width]
        h = self._block(rb.block1, x, x2, rowvec=rowvec, stats=True)     # its output feeds block2's GroupNorm
        if isinstance(rb.res_conv, nn.Conv2d):
            split = None if x2 is None else (x.shape[-1], x2.shape[-1])
            res = ops.conv2d(x, self._res_conv_packed(rb, split), x2=x2, pad=0)
        else:
            res = x
        h = self._block(rb.block2, h, residual=res, stats=not layer.with_attn)   # ... and the next block's, unless attention follows
        if layer.with_attn:
            h = self._attention(layer.attn, h)
        return h

    def _attention(self, at, x):
        B, H, W, Cc = x.shape
        # (planes / out_planes: honoured in the split precision only -- n feeds the query conv and the attention, q the attention)
        n = ops.group_norm(x, at.norm.weight, at.norm.bias, at.norm.num_groups, at.norm.eps, planes=True)
        q = ops.conv2d(n, self._pk[("attn_q", id(at))], pad=0, out_planes=True).reshape(B, H * W, Cc)
        kv = n.reshape(B, H * W, Cc)              # keys and values are the normalised input itself (_pack_attention)
        o = ops.attention(q, kv, kv, heads=1, scale=1.0 / math.sqrt(Cc))  # scale uses the full channel count (unet.py:135)
        return ops.conv2d(o.reshape(B, H, W, Cc), self._pk[("attn_out", id(at))], pad=0, residual=x)

    def forward_nhwc(self, x, noise_level):
        """x: 16-bit NHWC ``[B,H,W,pad8(in_channel)]``; noise_level fp32 ``[B,1]`` -> fp32 NHWC eps
        ``[B,H,W,pad8(out_channel)]`` (channels beyond out_channel are zero)."""
        with ops.f32_split(self.split if self.compute_dtype == torch.float32 else None):
            return self._forward_nhwc(x, noise_level)

    def _forward_nhwc(self, x, noise_level):
        if self._pk is None:
            self._pack()
        pk = self._pk
        mlp = self.noise_level_mlp
        pe = ops.sinusoidal(noise_level, mlp[0].dim, 0)
        t = ops.linear_small(pe, mlp[1].weight, mlp[1].bias, 0, 1)   # Linear + Swish
        t = ops.linear_small(t, mlp[3].weight, mlp[3].bias)
        nf_all = ops.linear_small(t, pk["nf_w"], pk["nf_b"])         # every FeatureWiseAffine at once

        feats = []
        for layer in self.downs:
            if isinstance(layer, ResnetBlocWithAttn):
                x = self._resblock(layer, x, None, nf_all)
            elif isinstance(layer, Downsample):
                x = ops.conv2d(x, pk[id(layer.conv)], stride=2, pad=1)
            else:
                x = ops.conv2d(x, pk[id(layer)], pad=1)
            feats.append(x)
        for layer in self.mid:
            x = self._resblock(layer, x, None, nf_all)
        for layer in self.ups:
            if isinstance(layer, ResnetBlocWithAttn):
                x = self._resblock(layer, x, feats.pop(), nf_all)
            else:
                x = ops.conv2d(x, pk[id(layer.conv)], pad=1, upsample=True, stats=True)
        return self._block(self.final_conv, x, out_f32=True)

    def forward(self, x, time):
        """Reference contract (unet.py:236-261): fp32 NCHW ``[B,in,H,W]``, ``time [B,1]`` -> fp32 NCHW."""
        if not x.is_cuda:
            raise RsvldError("SR3 UNet runs on the GPU only")
        xin = ops.nchw_to_nhwc(x, self.compute_dtype)
        eps = self.forward_nhwc(xin, time.float())
        return ops.nhwc_to_nchw(eps, channels=self.out_channel)
