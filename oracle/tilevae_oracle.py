"""CPU fp32 restatement of the tiled VAE (utils/tilevae.py: split_tiles :717-774, GroupNormParam :599-674,
custom_group_norm :524-553, crop_valid_region :556-567, vae_tile_forward :819-971).  TEST INFRASTRUCTURE.
Tiles run layer-synchronously; results equal the reference's task-queue walk (pinned by
tests/golden/tilevae_*.npz, generated from the reference's own VAEHook)."""
import math

import torch
import torch.nn.functional as F

from . import s2_oracle as O


def get_best_tile_size(lowerbound, upperbound):
    divider = 32
    while divider >= 2:
        r = lowerbound % divider
        if r == 0:
            return lowerbound
        cand = lowerbound - r + divider
        if cand <= upperbound:
            return cand
        divider //= 2
    return lowerbound


def split_tiles(h, w, tile_size, pad, is_decoder):
    ins, outs = [], []
    nh, nw = max(math.ceil((h - 2 * pad) / tile_size), 1), max(math.ceil((w - 2 * pad) / tile_size), 1)
    th = get_best_tile_size(math.ceil((h - 2 * pad) / nh), tile_size)
    tw = get_best_tile_size(math.ceil((w - 2 * pad) / nw), tile_size)
    for i in range(nh):
        for j in range(nw):
            ib = [pad + j * tw, min(pad + (j + 1) * tw, w), pad + i * th, min(pad + (i + 1) * th, h)]
            ob = [ib[0] if ib[0] > pad else 0, ib[1] if ib[1] < w - pad else w, ib[2] if ib[2] > pad else 0,
                  ib[3] if ib[3] < h - pad else h]
            outs.append([x * 8 if is_decoder else x // 8 for x in ob])
            ins.append([max(0, ib[0] - pad), min(w, ib[1] + pad), max(0, ib[2] - pad), min(h, ib[3] + pad)])
    return ins, outs


def var_mean(x, groups=32):  # get_var_mean :511-521 -> (var, mean) of shape [B*groups]
    b, c = x.shape[:2]
    v, m = torch.var_mean(x.contiguous().view(1, b * groups, c // groups, *x.shape[2:]), dim=[0, 2, 3, 4], unbiased=False)
    return v, m


def cross_tile_norm(sd, p, tiles, silu):
    """one GroupNorm layer over all tiles: summary() weighting, then custom_group_norm (eps 1e-6) + affine."""
    vs, ms, px = zip(*[(*var_mean(t), t.shape[2] * t.shape[3]) for t in tiles])
    w = torch.tensor(px, dtype=torch.float32) / max(px)
    w = (w / w.sum()).unsqueeze(1)
    var, mean = (torch.vstack(vs) * w).sum(0), (torch.vstack(ms) * w).sum(0)
    out = []
    for t in tiles:
        b, c = t.shape[:2]
        y = F.batch_norm(t.contiguous().view(1, b * 32, c // 32, *t.shape[2:]), mean, var, None, None, False, 0, 1e-6)
        y = y.view(t.shape) * sd[p + ".weight"].view(1, -1, 1, 1) + sd[p + ".bias"].view(1, -1, 1, 1)
        out.append(F.silu(y) if silu else y)
    return out


def _res(sd, p, tiles):
    skip = [O.conv(sd, p + ".nin_shortcut", t) if p + ".nin_shortcut.weight" in sd else t for t in tiles]
    h = [O.conv(sd, p + ".conv1", t, padding=1) for t in cross_tile_norm(sd, p + ".norm1", tiles, True)]
    h = [O.conv(sd, p + ".conv2", t, padding=1) for t in cross_tile_norm(sd, p + ".norm2", h, True)]
    return [a + b for a, b in zip(skip, h)]


def _attn(sd, p, tiles):
    out = []
    for t, n in zip(tiles, cross_tile_norm(sd, p + ".norm", tiles, False)):
        B, Cc, H, W = t.shape
        q, k, v = (O.conv(sd, f"{p}.{c}", n).reshape(B, Cc, H * W).transpose(1, 2) for c in "qkv")
        o = O.attention_core(q, k, v, 1).transpose(1, 2).reshape(B, Cc, H, W)
        out.append(t + O.conv(sd, p + ".proj_out", o))
    return out


def _mid(sd, P, tiles):
    return _res(sd, P + "mid.block_2", _attn(sd, P + "mid.attn_1", _res(sd, P + "mid.block_1", tiles)))


def tiled_forward(sd, x, tile_size, is_decoder, P):
    """x fp32 NCHW (image for the encoder, post_quant latent features for the decoder)."""
    pad = 11 if is_decoder else 32
    H, W = x.shape[2:]
    ins, outs = split_tiles(H, W, tile_size, pad, is_decoder)
    tiles = [O.conv(sd, P + "conv_in", x[:, :, b[2]:b[3], b[0]:b[1]], padding=1) for b in ins]
    if is_decoder:
        tiles = _mid(sd, P, tiles)
        n = O.count(sd, P + "up")
        for lv in reversed(range(n)):
            for b in range(O.count(sd, f"{P}up.{lv}.block")):
                tiles = _res(sd, f"{P}up.{lv}.block.{b}", tiles)
            if lv != 0:
                tiles = [O.conv(sd, f"{P}up.{lv}.upsample.conv", F.interpolate(t, scale_factor=2.0, mode="nearest"), padding=1) for t in tiles]
    else:
        n = O.count(sd, P + "down")
        for lv in range(n):
            for b in range(O.count(sd, f"{P}down.{lv}.block")):
                tiles = _res(sd, f"{P}down.{lv}.block.{b}", tiles)
            if lv != n - 1:
                tiles = [O.conv(sd, f"{P}down.{lv}.downsample.conv", F.pad(t, (0, 1, 0, 1)), stride=2) for t in tiles]
        tiles = _mid(sd, P, tiles)
    tiles = [O.conv(sd, P + "conv_out", t, padding=1) for t in cross_tile_norm(sd, P + "norm_out", tiles, True)]
    oh, ow = (H * 8, W * 8) if is_decoder else (H // 8, W // 8)
    res = torch.zeros(x.shape[0], tiles[0].shape[1], oh, ow)
    for t, ib, ob in zip(tiles, ins, outs):
        padded = [i * 8 if is_decoder else i // 8 for i in ib]
        m = [ob[i] - padded[i] for i in range(4)]
        res[:, :, ob[2]:ob[3], ob[0]:ob[1]] = t[:, :, m[2]:t.shape[2] + m[3], m[0]:t.shape[3] + m[1]]
    return res
