"""Tiled VAE with cross-tile GroupNorm (reference: utils/tilevae.py:472-971, ``VAEHook``).

Same contract: ``VAEHook(net, tile_size, is_decoder, ...)`` replaces ``net.forward``; inputs whose larger
side is <= 2*pad + tile_size run untiled through ``net.original_forward``; otherwise the image / latent is
split by ``split_tiles`` (pad 32 px encoder, 11 latent px decoder, :686,717-774 — same integer bboxes),
every tile runs the network with tile-LOCAL attention and conv borders from the pad overlap, and every
GroupNorm uses statistics MERGED across tiles by the reference's formula (:629-648: pixel-weighted mean of
the per-tile means AND of the per-tile variances — not the pooled variance), then the valid regions are
cropped (:556-567) into the result.  So tiled != untiled, by design; the parity oracle is the tiled path.

What is different is the machine mapping.  The reference walks a per-tile task queue, parks tiles in host
memory and visits them in zig-zag order to keep ONE tile on the GPU (:846,893-957).  With 288 GB of HBM every
tile stays resident, and tiles of equal shape (a 4096^2 image has 64 encoder tiles in 4 shapes) are STACKED along
the batch: one launch per layer and shape class instead of one per layer and tile (round 3: a 512 -> 4096 image went
through ~60 k GroupNorm launches and 12 k device copies).  Each shape class is a Python generator that yields at
each GroupNorm; the driver advances all classes to the same norm layer, measures per-tile statistics with
rsvld_groupnorm_stats (fp32, deterministic; one row per stacked tile), merges them over ALL tiles in tile order,
and resumes every class with rsvld_groupnorm_apply.  Every launch is planned for ONE tile (ops.plan_units), so a
tile's values do not depend on what it is stacked with: ``stack_tiles = False`` (one generator per tile) gives
bit-identical results and is kept for that test.  No host round trips, no per-layer copies.
"""
import math

import torch

from .. import ops


def get_best_tile_size(lowerbound, upperbound):  # tilevae.py:702-715
    divider = 32
    while divider >= 2:
        remainer = lowerbound % divider
        if remainer == 0:
            return lowerbound
        candidate = lowerbound - remainer + divider
        if candidate <= upperbound:
            return candidate
        divider //= 2
    return lowerbound


def split_tiles(h, w, tile_size, pad, is_decoder):
    """tilevae.py:717-774 -> (input bboxes, output bboxes), each [x1, x2, y1, y2]."""
    in_bboxes, out_bboxes = [], []
    nh = max(math.ceil((h - 2 * pad) / tile_size), 1)
    nw = max(math.ceil((w - 2 * pad) / tile_size), 1)
    th = get_best_tile_size(math.ceil((h - 2 * pad) / nh), tile_size)
    tw = get_best_tile_size(math.ceil((w - 2 * pad) / nw), tile_size)
    for i in range(nh):
        for j in range(nw):
            ib = [pad + j * tw, min(pad + (j + 1) * tw, w), pad + i * th, min(pad + (i + 1) * th, h)]
            ob = [ib[0] if ib[0] > pad else 0, ib[1] if ib[1] < w - pad else w,
                  ib[2] if ib[2] > pad else 0, ib[3] if ib[3] < h - pad else h]
            out_bboxes.append([x * 8 if is_decoder else x // 8 for x in ob])
            in_bboxes.append([max(0, ib[0] - pad), min(w, ib[1] + pad), max(0, ib[2] - pad), min(h, ib[3] + pad)])
    return in_bboxes, out_bboxes


def crop_margins(input_bbox, target_bbox, is_decoder):
    """tilevae.py:556-567 -> (left, right, top, bottom) margins as used in x[:, :, m2:H+m3, m0:W+m1]."""
    padded = [i * 8 if is_decoder else i // 8 for i in input_bbox]
    return [target_bbox[i] - padded[i] for i in range(4)]


def merge_stats(stats, pixels):
    """GroupNormParam.summary (:629-648): stats = fp32 [B,32,2] (mean, var) per tile (a list, or one [T,B,32,2] tensor in
    tile order), pixels = h*w of each tile at this layer.  weight_t = (p_t / max p) / sum(p / max p)."""
    st = torch.stack(stats, 0) if isinstance(stats, (list, tuple)) else stats   # [T,B,32,2]
    p = torch.tensor(pixels, dtype=torch.float32, device=st.device) / max(pixels)
    p = (p / p.sum()).view(-1, 1, 1, 1)
    w = st * p
    # one reduction PER IMAGE over a contiguous [T,32,2] block: the order of a torch reduction depends on the shape it is given, and with
    # T = 64 tiles (4096^2) a [T,B,32,2] sum gave image 0 other last bits at B = 2 than at B = 1 (found by tests/test_gpu_batch16.py at
    # configs[3]'s size; the 4-tile images of the 1024^2 test happened to agree) -- batch members must equal their batch-of-1 runs
    return torch.stack([w[:, b].contiguous().sum(0) for b in range(w.shape[1])], 0)


# ---- per-tile programs (generators yielding (tensor, norm_layer) at every GroupNorm) -----------------
def _norm(x, norm, silu):
    stats = yield (x, norm)
    return ops.group_norm_apply(x, stats, norm.weight, norm.bias, norm.num_groups, norm.eps, silu=silu, planes=True)   # feeds a conv


def _resblock(rt, blk, x):
    skip = ops.conv2d(x, rt.pk(blk.nin_shortcut), pad=0) if blk.in_channels != blk.out_channels else x
    h = yield from _norm(x, blk.norm1, True)
    h = ops.conv2d(h, rt.pk(blk.conv1), pad=1)
    h = yield from _norm(h, blk.norm2, True)
    return ops.conv2d(h, rt.pk(blk.conv2), pad=1, residual=skip)


def _attn(rt, at, x):
    B, H, W, Cc = x.shape
    h = yield from _norm(x, at.norm, False)
    qkv = ops.conv2d(h, rt.pk_cat([at.q, at.k, at.v], "qkv"), pad=0, out_planes=True).reshape(B, H * W, 3 * Cc)
    o = ops.attention(qkv[:, :, :Cc], qkv[:, :, Cc:2 * Cc], qkv[:, :, 2 * Cc:], heads=1, scale=Cc ** -0.5)
    return ops.conv2d(o.reshape(B, H, W, Cc), rt.pk(at.proj_out), pad=0, residual=x)


def _mid(net, h):
    h = yield from _resblock(net, net.mid.block_1, h)
    h = yield from _attn(net, net.mid.attn_1, h)
    h = yield from _resblock(net, net.mid.block_2, h)
    return h


def _tile_program(net, x, is_decoder):
    """The order of tilevae.py:472-499 / 404-470 (conv_in, [mid], levels, [mid], norm_out, conv_out)."""
    h = ops.conv2d(x, net.pk(net.conv_in), pad=1)
    if is_decoder:
        h = yield from _mid(net, h)
        for lv in reversed(range(net.num_resolutions)):
            for b in range(net.num_res_blocks + 1):
                h = yield from _resblock(net, net.up[lv].block[b], h)
            if lv != 0:
                h = net.up[lv].upsample.run(net, h)
    else:
        for lv in range(net.num_resolutions):
            for b in range(net.num_res_blocks):
                h = yield from _resblock(net, net.down[lv].block[b], h)
            if lv != net.num_resolutions - 1:
                h = net.down[lv].downsample.run(net, h)
        h = yield from _mid(net, h)
    if is_decoder and net.give_pre_end:
        return h
    h = yield from _norm(h, net.norm_out, True)
    return ops.conv2d(h, net.pk(net.conv_out), pad=1, out_f32=is_decoder)


class VAEHook:
    stack_tiles = True      # False: one launch per tile and layer (the round-2 form; bit-identical, kept as the A/B reference)

    def __init__(self, net, tile_size, is_decoder, fast_decoder=False, fast_encoder=False, color_fix=False, to_gpu=False):
        if fast_decoder or fast_encoder or color_fix:
            raise NotImplementedError("SR_backbone.init_tile_vae installs the hooks with fast modes and color_fix off")
        self.net, self.tile_size, self.is_decoder = net, tile_size, is_decoder
        self.to_gpu = to_gpu
        self.pad = 11 if is_decoder else 32

    def __call__(self, x):
        net = self.net
        x = net._in(x)                                   # NHWC, compute dtype
        B, H, W, _ = x.shape
        if max(H, W) <= self.pad * 2 + self.tile_size:
            return net.original_forward(x)
        return self.vae_tile_forward(x)

    @torch.no_grad()
    def vae_tile_forward(self, z):
        net, dec = self.net, self.is_decoder
        B, H, W, _ = z.shape
        net.last_z_shape = z.shape
        in_bboxes, out_bboxes = split_tiles(H, W, self.tile_size, self.pad, dec)
        T = len(in_bboxes)
        # shape classes: tiles of equal (h, w) run as one stacked batch, tile-major ([t0 b0, t0 b1, .., t1 b0, ..])
        classes = {}
        for i, b in enumerate(in_bboxes):
            key = (b[3] - b[2], b[1] - b[0]) if self.stack_tiles else i
            classes.setdefault(key, []).append(i)
        groups = list(classes.values())
        unit = ops.context().plan_div                             # independent units the caller has already stacked along B
        gens, pending = [], []
        for idx in groups:
            x = torch.cat([z[:, in_bboxes[i][2]:in_bboxes[i][3], in_bboxes[i][0]:in_bboxes[i][1], :] for i in idx], 0)
            gens.append(_tile_program(net, x, dec))
        for g, idx in zip(gens, groups):
            with ops.plan_units(unit * len(idx)):        # every launch planned for ONE tile of ONE unit
                pending.append(next(g))                  # advanced to the first GroupNorm
        outs = [None] * len(groups)
        order = torch.tensor([i for idx in groups for i in idx], device=z.device)
        while any(p is not None for p in pending):
            live = [k for k, p in enumerate(pending) if p is not None]
            assert len(live) == len(groups)              # all classes run the same layer sequence
            norm = pending[live[0]][1]
            stats, pixels = [], [0] * T
            for k in live:
                x = pending[k][0]
                with ops.plan_units(unit * len(groups[k])):
                    st = ops.group_norm_stats(x, norm.num_groups)          # [len(idx) * B, 32, 2]
                stats.append(st.view(len(groups[k]), B, *st.shape[1:]))
                for i in groups[k]:
                    pixels[i] = x.shape[1] * x.shape[2]
            st_all = torch.empty((T, B) + tuple(stats[0].shape[2:]), device=z.device, dtype=stats[0].dtype)
            st_all[order] = torch.cat(stats, 0)                            # back to tile order: the merge sums in that order
            merged = merge_stats(st_all, pixels)                           # [B, 32, 2]
            for k in live:
                try:
                    with ops.plan_units(unit * len(groups[k])):
                        pending[k] = gens[k].send(merged.repeat(len(groups[k]), 1, 1))
                except StopIteration as done:
                    pending[k], outs[k] = None, done.value
        oh, ow = (H * 8, W * 8) if dec else (H // 8, W // 8)
        result = torch.zeros((B, oh, ow, outs[0].shape[-1]), device=z.device, dtype=outs[0].dtype)
        for out, idx in zip(outs, groups):
            for j, i in enumerate(idx):
                t, ib, ob = out[j * B:(j + 1) * B], in_bboxes[i], out_bboxes[i]
                m = crop_margins(ib, ob, dec)
                result[:, ob[2]:ob[3], ob[0]:ob[1], :] = t[:, m[2]:t.shape[1] + m[3], m[0]:t.shape[2] + m[1], :]
        return result
