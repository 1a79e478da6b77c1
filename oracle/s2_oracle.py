"""CPU fp32 restatement of Stage 2: SDXL-style UNet + ControlNet + adapters, denoiser, CFG, the
RestoreEDM sampler with the first-block cache, the SD-VAE, the posterior and the colour fix.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional code over a plain state dict whose keys
are the reference's (``model.diffusion_model.*``, ``model.control_model.*``, ``first_stage_model.*``);
each function cites the reference lines it restates (paths relative to /root/reference).
Pinned by tests/golden/s2_*.npz (outputs of the reference's own modules on CPU, fp32 — on a CPU-only
box ``torch.autocast("cuda")`` disables itself, so this is the "CPU sgm path" of BASELINE.json).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------ primitives
def gn(sd, p, x, eps, groups=32):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps=eps)


def conv(sd, p, x, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def timestep_embedding(t, dim):  # sgm/modules/diffusionmodules/util.py:206-230
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


# ------------------------------------------------------------------------------ sgm blocks
def resblock(sd, p, x, emb):  # openaimodel.py:207-350 (no up/down, no scale-shift)
    h = conv(sd, p + ".in_layers.2", F.silu(gn(sd, p + ".in_layers.0", x, 1e-5)), padding=1)
    h = h + lin(sd, p + ".emb_layers.1", F.silu(emb))[:, :, None, None]
    h = conv(sd, p + ".out_layers.3", F.silu(gn(sd, p + ".out_layers.0", h, 1e-5)), padding=1)
    if p + ".skip_connection.weight" in sd:
        x = conv(sd, p + ".skip_connection", x)
    return x + h


def attention_core(q, k, v, heads):  # attention.py:196-285 (SDPA, scale = d_head^-0.5)
    B, N, Cq = q.shape
    d = Cq // heads
    q, k, v = (t.view(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    a = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1)
    return (a @ v).transpose(1, 2).reshape(B, N, Cq)


def cross_attention(sd, p, x, ctx, heads):
    ctx = x if ctx is None else ctx
    o = attention_core(lin(sd, p + ".to_q", x), lin(sd, p + ".to_k", ctx), lin(sd, p + ".to_v", ctx), heads)
    return lin(sd, p + ".to_out.0", o)


def layer_norm(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def basic_block(sd, p, x, ctx, heads):  # attention.py:376-486
    x = cross_attention(sd, p + ".attn1", layer_norm(sd, p + ".norm1", x), None, heads) + x
    x = cross_attention(sd, p + ".attn2", layer_norm(sd, p + ".norm2", x), ctx, heads) + x
    h = lin(sd, p + ".ff.net.0.proj", layer_norm(sd, p + ".norm3", x))   # GEGLU, attention.py:84-96
    a, gate = h.chunk(2, dim=-1)
    return lin(sd, p + ".ff.net.2", a * F.gelu(gate)) + x


def spatial_transformer(sd, p, x, ctx):  # attention.py:533-635, use_linear=True
    B, Cc, H, W = x.shape
    heads = Cc // 64
    h = gn(sd, p + ".norm", x, 1e-6)
    h = lin(sd, p + ".proj_in", h.permute(0, 2, 3, 1).reshape(B, H * W, Cc))
    i = 0
    while f"{p}.transformer_blocks.{i}.norm1.weight" in sd:
        h = basic_block(sd, f"{p}.transformer_blocks.{i}", h, ctx, heads)
        i += 1
    h = lin(sd, p + ".proj_out", h)
    return h.reshape(B, H, W, Cc).permute(0, 3, 1, 2) + x


def seq_block(sd, p, x, emb, ctx):
    """TimestepEmbedSequential (openaimodel.py:75-99): children told apart by their parameter names."""
    i = 0
    while True:
        q = f"{p}.{i}"
        if q + ".in_layers.0.weight" in sd:
            x = resblock(sd, q, x, emb)
        elif q + ".proj_in.weight" in sd:
            x = spatial_transformer(sd, q, x, ctx)
        elif q + ".op.weight" in sd:        # Downsample :164-204
            x = conv(sd, q + ".op", x, stride=2, padding=1)
        elif q + ".conv.weight" in sd:      # Upsample :102-145
            x = conv(sd, q + ".conv", F.interpolate(x, scale_factor=2, mode="nearest"), padding=1)
        elif q + ".weight" in sd and sd[q + ".weight"].dim() == 4:
            x = conv(sd, q, x, padding=1)
        else:
            return x
        i += 1


def count(sd, prefix):
    n = 0
    while any(k.startswith(f"{prefix}.{n}.") for k in sd):
        n += 1
    return n


def embed(sd, p, t, y, model_channels=320):  # openaimodel.py:985-990
    emb = lin(sd, p + "time_embed.2", F.silu(lin(sd, p + "time_embed.0", timestep_embedding(t, model_channels))))
    return emb + lin(sd, p + "label_emb.0.2", F.silu(lin(sd, p + "label_emb.0.0", y)))


# ------------------------------------------------------------------------------ SR modules
def zero_sft(sd, p, c, h, h_ori=None, control_scale=1.0):  # models/modules/SR_modules.py:59-110
    pre_concat = sd[p + ".param_free_norm.weight"].shape[0] != sd[p + ".zero_conv.weight"].shape[0]
    cat = h_ori is not None and pre_concat
    h_raw = torch.cat([h_ori, h], dim=1) if cat else h
    h = h + conv(sd, p + ".zero_conv", c)
    if cat:
        h = torch.cat([h_ori, h], dim=1)
    actv = F.silu(conv(sd, p + ".mlp_shared.0", c, padding=1))
    gamma, beta = conv(sd, p + ".zero_mul", actv, padding=1), conv(sd, p + ".zero_add", actv, padding=1)
    h = gn(sd, p + ".param_free_norm", h, 1e-5) * (gamma + 1) + beta
    return h * control_scale + h_raw * (1 - control_scale)


def zero_cross_attn(sd, p, context, x, control_scale=1.0):  # SR_modules.py:113-149
    B, Cc, H, W = x.shape
    xq = gn(sd, p + ".norm1", x, 1e-5).permute(0, 2, 3, 1).reshape(B, H * W, Cc)
    ctx = gn(sd, p + ".norm2", context, 1e-5)
    ctx = ctx.permute(0, 2, 3, 1).reshape(B, -1, ctx.shape[1])
    o = cross_attention(sd, p + ".attn", xq, ctx, Cc // 64)
    return x + o.reshape(B, H, W, Cc).permute(0, 3, 1, 2) * control_scale


def project(sd, p, *args, **kw):
    return (zero_cross_attn if p + ".attn.to_q.weight" in sd else zero_sft)(sd, p, *args, **kw)


def glv_control(sd, x, t, xt, ctx, y, P="model.control_model."):  # SR_modules.py:496-537
    emb = embed(sd, P, t, y)
    hint = conv(sd, P + "input_hint_block.0", x, padding=1)
    hs, h = [], xt
    for i in range(count(sd, P + "input_blocks")):
        h = seq_block(sd, f"{P}input_blocks.{i}", h, emb, ctx)
        if i == 0:
            h = h + hint
        hs.append(h)
    hs.append(seq_block(sd, P + "middle_block", h, emb, ctx))
    return hs


def light_unet_stage1(sd, x, t, ctx, y, P="model.diffusion_model."):  # SR_modules.py:660-683
    emb = embed(sd, P, t, y)
    hs, h = [], x
    for i in range(count(sd, P + "input_blocks")):
        h = seq_block(sd, f"{P}input_blocks.{i}", h, emb, ctx)
        hs.append(h)
    return {"h": h, "hs": hs, "emb": emb}


def light_unet_stage2(sd, part, ctx, control, control_scale=1.0, P="model.diffusion_model."):  # :686-730
    h, hs, emb = part["h"], list(part["hs"]), part["emb"]
    a, ci = count(sd, P + "project_modules") - 1, len(control) - 1
    h = seq_block(sd, P + "middle_block", h, emb, ctx)
    h = project(sd, f"{P}project_modules.{a}", control[ci], h, control_scale=control_scale)
    a, ci = a - 1, ci - 1
    for i in range(count(sd, P + "output_blocks")):
        q = f"{P}output_blocks.{i}"
        h = project(sd, f"{P}project_modules.{a}", control[ci], hs.pop(), h, control_scale=control_scale)
        a -= 1
        if count(sd, q) == 3:   # ResBlock, SpatialTransformer, Upsample: an extra adapter before the upsample
            h = resblock(sd, q + ".0", h, emb)
            h = spatial_transformer(sd, q + ".1", h, ctx)
            h = project(sd, f"{P}project_modules.{a}", control[ci], h, control_scale=control_scale)
            a -= 1
            h = conv(sd, q + ".2.conv", F.interpolate(h, scale_factor=2, mode="nearest"), padding=1)
        else:
            h = seq_block(sd, q, h, emb, ctx)
        ci -= 1
    return conv(sd, P + "out.2", F.silu(gn(sd, P + "out.0", h, 1e-5)), padding=1)


def light_unet(sd, x, t, ctx, y, control, control_scale=1.0):  # fbcache_mode="none", :597-657
    return light_unet_stage2(sd, light_unet_stage1(sd, x, t, ctx, y), ctx, control, control_scale)


# ------------------------------------------------------------------------------ schedule / denoiser / guider
def ddpm_sigmas_table(num=1000, linear_start=0.00085, linear_end=0.0120):  # discretizer.py:42-69 + util.py:19-32
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num, dtype=torch.float64) ** 2).numpy()
    return np.cumprod(1.0 - betas, axis=0)


def legacy_ddpm_sigmas(n, append_zero=True, flip=False):
    ac = ddpm_sigmas_table()
    if n < 1000:
        ac = ac[np.linspace(999, 0, n, endpoint=False).astype(int)[::-1]]
    s = torch.flip(torch.tensor((1 - ac) / ac, dtype=torch.float32) ** 0.5, (0,))
    if append_zero:
        s = torch.cat([s, s.new_zeros([1])])
    return torch.flip(s, (0,)) if flip else s


def quantize_sigma(sigma, table):  # denoiser.py:49-57
    idx = (sigma - table[:, None]).abs().argmin(dim=0).view(sigma.shape)
    return table[idx], idx


def network(sd, x, t, c, control_scale, mode="none", part=None):  # wrappers.py:84-110 (fp32 on CPU)
    ctx, y = c["crossattn"], c["vector"]
    if mode == "input_stage2":
        return light_unet_stage2(sd, part, part["context"], part["control"], control_scale)
    control = glv_control(sd, c["control"], t, x, ctx, y)
    if mode == "input_stage1":
        p = light_unet_stage1(sd, x, t, ctx, y)
        p.update(context=ctx, control=control)
        return p
    return light_unet(sd, x, t, ctx, y, control, control_scale)


def denoiser(sd, table, inp, sigma, c, control_scale, mode="none", part=None):  # denoiser.py:66-78, EpsScaling
    sigma_q, idx = quantize_sigma(sigma, table)
    s4 = sigma_q[:, None, None, None]
    c_in = 1 / (s4 ** 2 + 1.0) ** 0.5
    out = network(sd, inp * c_in, idx, c, control_scale, mode, part)
    return out if mode == "input_stage1" else out * (-s4) + inp


def cfg_inputs(x, s, c, uc):  # guiders.py:65-74
    c_out = {k: torch.cat((uc[k], c[k]), 0) for k in c}
    return torch.cat([x] * 2), torch.cat([s] * 2), c_out


def linear_cfg(x, sigma, scale, scale_min):  # guiders.py:44-63
    x_u, x_c = x.chunk(2)
    sv = (scale - scale_min) * sigma / 14.6146 + scale_min
    return x_u + sv.view(-1, 1, 1, 1) * (x_c - x_u)


def relative_l1(t1, t2):  # models/modules/DFBCache.py:98-112
    return float(((t1 - t2).abs().mean() / (t1.abs().mean() + 1e-6)))


class Cache:
    def __init__(self):
        self.prev, self.final_decode = None, None


def restore_edm_denoise(sd, table, cache, x, sigma, c, uc, scale, scale_min, control_scale, threshold, trace=None):
    """sampling.py:548-596 -> (denoised, new_threshold)."""
    if threshold <= 0:
        d = denoiser(sd, table, *cfg_inputs(x, sigma, c, uc), control_scale)
        return linear_cfg(d, sigma, scale, scale_min), threshold
    part = denoiser(sd, table, *cfg_inputs(x, sigma, c, uc), control_scale, "input_stage1")
    if cache.prev is not None:
        diff = relative_l1(cache.prev, part["h"])
        if diff < threshold:
            if trace is not None:
                trace.append((threshold, diff, True))
            return cache.final_decode, threshold
        new_thr = diff
    else:
        new_thr = threshold
    if trace is not None:
        trace.append((threshold, new_thr, False))
    cache.prev = part["h"].clone()
    d = denoiser(sd, table, *cfg_inputs(x, sigma, c, uc), control_scale, "input_stage2", part)
    d = linear_cfg(d, sigma, scale, scale_min)
    cache.final_decode = d.clone()
    return d, new_thr


def restore_edm_step(sd, table, cache, x, i, sigmas, c, uc, x_center, opt, threshold, randn, trace=None):
    """sampling.py:659-694 + :598-621.  opt: s_churn, s_noise, restore_cfg, scale, scale_min, control_scale."""
    B = x.shape[0]
    gamma = min(opt["s_churn"] / (len(sigmas) - 1), 2 ** 0.5 - 1)
    sigma, nxt = torch.ones(B) * sigmas[i], torch.ones(B) * sigmas[i + 1]
    sigma_hat = sigma * (gamma + 1.0)
    if gamma > 0:
        eps = randn(x.shape) * opt["s_noise"]
        x = x + eps * ((sigma_hat ** 2 - sigma ** 2)[:, None, None, None]) ** 0.5
    control_scale = opt["control_scale"]
    if opt.get("use_linear_control_scale"):   # :608-609
        control_scale = (float(sigma[0]) / 14.6146) * (opt.get("control_scale_start", 0.0) - control_scale) + control_scale
    d, threshold = restore_edm_denoise(sd, table, cache, x, sigma_hat, c, uc, opt["scale"], opt["scale_min"],
                                       control_scale, threshold, trace)
    if nxt[0] > 0.05 and opt["restore_cfg"] > 0:
        d = d - (d - x_center) * ((sigma.view(-1, 1, 1, 1) / 14.6146) ** opt["restore_cfg"])
    dd = (x - d) / sigma_hat[:, None, None, None]
    return x + dd * (nxt - sigma_hat)[:, None, None, None], threshold


# ---- latent-tile sampling (sampling.py:697-757, 830-863).  The reference's TiledRestoreEDMSampler.__call__ cannot run as
# shipped (its tile loop multiplies the (x, threshold) tuple that sampler_step now returns, :719-733, and sampler_step's
# default threshold 0.1 enters the feature cache without a cache context): this restatement follows the source line by
# line with the one evident repair -- every tile takes an UN-cached step and the tensor is blended.  PARITY UNPINNED for
# the loop (no reference output exists); sliding_windows is pinned (tests/golden/tiled_sampler_windows.json).
def sliding_windows(h, w, tile_size, tile_stride):  # :850-863
    hi_list = list(range(0, h - tile_size + 1, tile_stride))
    if (h - tile_size) % tile_stride != 0:
        hi_list.append(h - tile_size)
    wi_list = list(range(0, w - tile_size + 1, tile_stride))
    if (w - tile_size) % tile_stride != 0:
        wi_list.append(w - tile_size)
    return [(hi, hi + tile_size, wi, wi + tile_size) for hi in hi_list for wi in wi_list]


def gaussian_weights(tile_width, tile_height):  # :830-847 -> float64 [tile_height, tile_width] (the reference tiles it to [nb,4,h,w])
    # numpy's SCALAR exp / sqrt, one element at a time, as the reference evaluates them (math.exp differs from it in the last bit for some
    # arguments: found by the bit-exact pin against the reference's own plane, tests/golden/tiled_sampler_mask.npz)
    import numpy as np
    var = 0.01
    midpoint = (tile_width - 1) / 2
    x_probs = [np.exp(-(x - midpoint) * (x - midpoint) / (tile_width * tile_width) / (2 * var)) / np.sqrt(2 * np.pi * var)
               for x in range(tile_width)]
    midpoint = tile_height / 2          # sic: no "- 1" for the rows
    y_probs = [np.exp(-(y - midpoint) * (y - midpoint) / (tile_height * tile_height) / (2 * var)) / np.sqrt(2 * np.pi * var)
               for y in range(tile_height)]
    return torch.tensor(np.outer(y_probs, x_probs))


def tiled_restore_edm(sd, table, x, sigmas, c, uc, x_center, opt, randn, tile_size, tile_stride):
    """:704-757.  x: the initial noise (scaled here by sqrt(1 + sigma_0^2), :44-55); opt as restore_edm_step."""
    B, _, h, w = x.shape
    tiles = sliding_windows(h, w, tile_size, tile_stride)
    tw = gaussian_weights(tile_size, tile_size)[None, None].repeat(B, 4, 1, 1)
    lq = c["control"]
    x = x * torch.sqrt(1.0 + sigmas[0] ** 2.0)
    for i in range(len(sigmas) - 1):
        gamma = min(opt["s_churn"] / (len(sigmas) - 1), 2 ** 0.5 - 1)
        x_next, count = torch.zeros_like(x), torch.zeros_like(x)
        eps_noise = randn(x.shape)
        sigma, nxt = torch.ones(B) * sigmas[i], torch.ones(B) * sigmas[i + 1]
        sigma_hat = sigma * (gamma + 1.0)
        for hi, he, wi, we in tiles:
            xt = x[:, :, hi:he, wi:we]
            cj, ucj = dict(c, control=lq[:, :, hi:he, wi:we]), dict(uc, control=lq[:, :, hi:he, wi:we])
            if gamma > 0:
                xt = xt + eps_noise[:, :, hi:he, wi:we] * opt["s_noise"] * ((sigma_hat ** 2 - sigma ** 2)[:, None, None, None]) ** 0.5
            d, _ = restore_edm_denoise(sd, table, None, xt, sigma_hat, cj, ucj, opt["scale"], opt["scale_min"], opt["control_scale"], 0.0)
            if nxt[0] > 0.05 and opt["restore_cfg"] > 0:
                d = d - (d - x_center[:, :, hi:he, wi:we]) * ((sigma.view(-1, 1, 1, 1) / 14.6146) ** opt["restore_cfg"])
            xo = xt + (xt - d) / sigma_hat[:, None, None, None] * (nxt - sigma_hat)[:, None, None, None]
            x_next[:, :, hi:he, wi:we] += xo * tw          # fp32 += float64 product, as in the reference (:733)
            count[:, :, hi:he, wi:we] += tw
        x_next /= count
        x = x_next
    return x


# ------------------------------------------------------------------------------ VAE (model.py)
def vae_resblock(sd, p, x):  # model.py:91-148, temb = None
    h = conv(sd, p + ".conv1", F.silu(gn(sd, p + ".norm1", x, 1e-6)), padding=1)
    h = conv(sd, p + ".conv2", F.silu(gn(sd, p + ".norm2", h, 1e-6)), padding=1)
    if p + ".nin_shortcut.weight" in sd:
        x = conv(sd, p + ".nin_shortcut", x)
    return x + h


def vae_attn(sd, p, x):  # model.py:158-198
    B, Cc, H, W = x.shape
    h = gn(sd, p + ".norm", x, 1e-6)
    q, k, v = (conv(sd, f"{p}.{n}", h).reshape(B, Cc, H * W).transpose(1, 2) for n in "qkv")
    o = attention_core(q, k, v, 1)
    return x + conv(sd, p + ".proj_out", o.transpose(1, 2).reshape(B, Cc, H, W))


def vae_encoder(sd, x, P="first_stage_model.encoder."):  # model.py:571-596
    h = conv(sd, P + "conv_in", x, padding=1)
    nlev = count(sd, P + "down")
    for lv in range(nlev):
        for b in range(count(sd, f"{P}down.{lv}.block")):
            h = vae_resblock(sd, f"{P}down.{lv}.block.{b}", h)
        if lv != nlev - 1:   # asymmetric pad (0,1,0,1) then stride-2 conv, model.py:81-85
            h = conv(sd, f"{P}down.{lv}.downsample.conv", F.pad(h, (0, 1, 0, 1)), stride=2)
    h = vae_resblock(sd, P + "mid.block_1", h)
    h = vae_attn(sd, P + "mid.attn_1", h)
    h = vae_resblock(sd, P + "mid.block_2", h)
    return conv(sd, P + "conv_out", F.silu(gn(sd, P + "norm_out", h, 1e-6)), padding=1)


def vae_decoder(sd, z, P="first_stage_model.decoder."):  # model.py:710-743
    h = conv(sd, P + "conv_in", z, padding=1)
    h = vae_resblock(sd, P + "mid.block_1", h)
    h = vae_attn(sd, P + "mid.attn_1", h)
    h = vae_resblock(sd, P + "mid.block_2", h)
    nlev = count(sd, P + "up")
    for lv in reversed(range(nlev)):
        for b in range(count(sd, f"{P}up.{lv}.block")):
            h = vae_resblock(sd, f"{P}up.{lv}.block.{b}", h)
        if lv != 0:
            h = conv(sd, f"{P}up.{lv}.upsample.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"), padding=1)
    return conv(sd, P + "conv_out", F.silu(gn(sd, P + "norm_out", h, 1e-6)), padding=1)


def posterior(moments, noise=None):  # distributions.py:24-41,71-72
    mean, logvar = torch.chunk(moments, 2, dim=1)
    if noise is None:
        return mean
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise


def encode_with_denoise(sd, x, scale=0.13025):  # SR_model.py:64-78, use_sample=False
    m = conv(sd, "first_stage_model.quant_conv", vae_encoder(sd, x, "first_stage_model.denoise_encoder."))
    return scale * posterior(m)


def encode_sample(sd, x, noise, scale=0.13025):  # SR_model.py:57-62
    return scale * posterior(conv(sd, "first_stage_model.quant_conv", vae_encoder(sd, x)), noise)


def decode(sd, z, scale=0.13025):  # SR_model.py:80-85
    return vae_decoder(sd, conv(sd, "first_stage_model.post_quant_conv", z / scale))


# ------------------------------------------------------------------------------ colour fix (utils/colorfix.py)
def wavelet_blur(image, radius):  # :73-92
    k = torch.tensor([[0.0625, 0.125, 0.0625], [0.125, 0.25, 0.125], [0.0625, 0.125, 0.0625]])[None, None].repeat(3, 1, 1, 1)
    return F.conv2d(F.pad(image, (radius,) * 4, mode="replicate"), k, groups=3, dilation=radius)


def wavelet_decomposition(image, levels=5):  # :94-106
    high = torch.zeros_like(image)
    for i in range(levels):
        low = wavelet_blur(image, 2 ** i)
        high += image - low
        image = low
    return high, low


def wavelet_reconstruction(content, style):  # :108-119
    return wavelet_decomposition(content)[0] + wavelet_decomposition(style)[1]


def adain(content, style, eps=1e-5):  # :44-71
    def ms(f):
        b, c = f.shape[:2]
        return f.reshape(b, c, -1).mean(2).reshape(b, c, 1, 1), (f.reshape(b, c, -1).var(2) + eps).sqrt().reshape(b, c, 1, 1)
    sm, ss = ms(style)
    cm, cs = ms(content)
    return (content - cm) / cs * ss + sm


# ------------------------------------------------------------------------------ just_sampling (SR_model.py:200-298)
def just_sampling(sd, x, cond, uncond, opt, randn=None, trace=None):
    """x fp32 [N,3,H,W]; cond/uncond: {'crossattn','vector'} for ONE image; opt: num_steps, s_churn, s_noise,
    cfg_scale, cfg_scale_start, use_linear_CFG, restoration_scale, control_scale, img_threshold, dec_img,
    color_fix_type.  ``randn(shape)`` supplies draws in the reference's order."""
    randn = randn or (lambda shape: torch.randn(shape))
    if opt.get("num_samples", 1) > 1:   # :231-235
        assert x.shape[0] == 1
        x = x.repeat(opt["num_samples"], 1, 1, 1)
    N = x.shape[0]
    table = legacy_ddpm_sigmas(1000, append_zero=False, flip=True)   # denoiser buffer (denoiser.py:41-44)
    _z = encode_with_denoise(sd, x)
    x_stage1 = decode(sd, _z)
    z_stage1 = encode_sample(sd, x_stage1, randn(_z.shape))
    rep = lambda v: v.repeat(N, *[1] * (v.ndim - 1))
    c = {"crossattn": rep(cond["crossattn"]), "vector": rep(cond["vector"]), "control": _z}
    uc = {"crossattn": rep(uncond["crossattn"]), "vector": rep(uncond["vector"]), "control": _z}
    z = randn(_z.shape)
    sigmas = legacy_ddpm_sigmas(opt["num_steps"])
    z = z * torch.sqrt(1.0 + sigmas[0] ** 2.0)
    sopt = dict(s_churn=opt["s_churn"], s_noise=opt["s_noise"], restore_cfg=opt["restoration_scale"],
                scale=opt["cfg_scale_start"] if opt["use_linear_CFG"] else opt["cfg_scale"], scale_min=opt["cfg_scale"],
                control_scale=opt["control_scale"], use_linear_control_scale=opt.get("use_linear_control_scale", False),
                control_scale_start=opt.get("control_scale_start", 0.0))
    cache, thr, x_center = Cache(), opt["img_threshold"], z_stage1
    for i in range(len(sigmas) - 1):
        z, thr = restore_edm_step(sd, table, cache, z, i, sigmas, c, uc, x_center, sopt, thr, randn, trace)
        x_center = z
        thr = thr * opt["dec_img"]
    out = decode(sd, z)
    if opt["color_fix_type"] == "Wavelet":
        out = wavelet_reconstruction(out, x_stage1)
    elif opt["color_fix_type"] == "AdaIn":
        out = adain(out, x_stage1)
    return out
