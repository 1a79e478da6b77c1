"""Stage-1 (SR3) parity on the GPU: the HIP path (through the C ABI) against
  * the committed golden vectors produced by the reference itself (tests/golden/gen_sr3_golden.py),
  * the CPU oracle on fresh seeded inputs.
Tolerances are stated per test; the compute type is fp16 storage / fp32 accumulate, the oracle fp32.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

WEIGHT_SEED = 1234


@pytest.fixture(scope="module")
def sr3(cuda):
    from oracle import seeded, sr3_oracle as O
    from rsvld_amd.sr3_model.sr3_modules.diffusion import GaussianDiffusion
    from rsvld_amd.sr3_model.sr3_modules.unet import UNet
    c = O.SR3_CFG
    unet = UNet(in_channel=c["in_channel"], out_channel=c["out_channel"], inner_channel=c["inner_channel"],
                norm_groups=c["norm_groups"], channel_mults=c["channel_mults"], attn_res=list(c["attn_res"]),
                res_blocks=c["res_blocks"], dropout=0.2, image_size=c["image_size"])
    net = GaussianDiffusion(unet, image_size=c["image_size"], channels=3, conditional=True)
    seeded.seed_module(net, WEIGHT_SEED)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net.to(cuda).eval()
    return net, sd


def _err(got, want):
    return float((got.float().cpu() - torch.as_tensor(want).float()).abs().max())


def test_unet_layers_teacher_forced(sr3, cuda, golden_dir):
    """Each tapped layer is fed the REFERENCE's input for that layer and compared with the
    reference's output: isolates per-layer error (fp16 operands, fp32 accumulate).  Tolerance:
    5e-3 x the layer's output range."""
    from rsvld_amd import ops
    net, _ = sr3
    unet = net.denoise_fn
    unet._pack()
    z = np.load(os.path.join(golden_dir, "sr3_unet_taps.npz"))
    lvl = torch.tensor(z["level"]).to(cuda)
    mlp = unet.noise_level_mlp
    pe = ops.sinusoidal(lvl, mlp[0].dim, 0)
    t = ops.linear_small(ops.linear_small(pe, mlp[1].weight, mlp[1].bias, 0, 1), mlp[3].weight, mlp[3].bias)
    assert _err(t, z["t_emb"]) < 1e-5
    nf_all = ops.linear_small(t, unet._pk["nf_w"], unet._pk["nf_b"])
    mods = dict(unet.named_modules())
    dt = unet.compute_dtype
    from rsvld_amd.sr3_model.sr3_modules import unet as U
    report = {}
    for name in ["downs.0", "downs.1", "downs.2", "downs.3", "downs.7", "mid.0", "mid.1", "ups.0", "ups.2", "ups.4",
                 "ups.13", "final_conv"]:
        xin = torch.tensor(z[name + ".in"]).to(cuda)
        want = z[name + ".out"]
        m = mods[name]
        x = ops.nchw_to_nhwc(xin, dt)
        if isinstance(m, U.ResnetBlocWithAttn):
            c_in = m.res_block.block1.block[0].num_channels
            if name.startswith("ups"):
                # reference input is the concatenation [x | skip]; split it where the skip starts
                c_skip = c_in - (m.res_block.block2.block[0].num_channels if name != "ups.0" else 512)
                c_x = c_in - c_skip
                xa = ops.nchw_to_nhwc(xin[:, :c_x].contiguous(), dt)
                xb = ops.nchw_to_nhwc(xin[:, c_x:].contiguous(), dt)
                y = unet._resblock(m, xa, xb, nf_all)
            else:
                y = unet._resblock(m, x, None, nf_all)
        elif isinstance(m, U.Downsample):
            y = ops.conv2d(x, unet._pk[id(m.conv)], stride=2, pad=1)
        elif isinstance(m, U.Upsample):
            y = ops.conv2d(x, unet._pk[id(m.conv)], pad=1, upsample=True)
        elif isinstance(m, U.Block):
            y = unet._block(m, x, out_f32=True)
        else:
            y = ops.conv2d(x, unet._pk[id(m)], pad=1)
        got = ops.nhwc_to_nchw(y, channels=want.shape[1])
        scale = float(np.abs(want).max())
        e = _err(got, want)
        report[name] = (e, scale)
        assert e <= 1.5e-3 * scale + 1e-4, f"{name}: max|d|={e:.3e} range={scale:.3e}"   # measured worst 7.0e-4 x range
    print("per-layer max|d| / range:", {k: f"{e:.2e}/{s:.2f}" for k, (e, s) in report.items()})


@pytest.mark.parametrize("tag", ["a", "b"])
def test_unet_forward_vs_reference_golden(sr3, cuda, golden_dir, tag):
    """Whole UNet forward (26 res-blocks, 4 attention sites) vs the reference's fp32 output.
    Tolerance: 1e-2 absolute / 1.5e-3 mean on an output of range ~2.3 (fp16 storage through ~60 layers; measured
    4.9e-3 / 7.3e-4)."""
    from oracle import seeded
    net, _ = sr3
    z = np.load(os.path.join(golden_dir, "sr3_unet_forward.npz"))
    shape = tuple(int(v) for v in z[f"{tag}.shape"])
    x = seeded.synthetic_image(shape, seed=int(z[f"{tag}.seed"]), smooth=int(z[f"{tag}.smooth"]))
    lv = torch.tensor(z[f"{tag}.level"])
    y = net.denoise_fn(x.to(cuda), lv.to(cuda))
    e = _err(y, z[f"{tag}.y"])
    print(f"unet forward {tag}: max|d| = {e:.3e}, mean|d| = "
          f"{float((y.cpu() - torch.tensor(z[f'{tag}.y'])).abs().mean()):.3e}")
    assert e < 1e-2 and float((y.cpu() - torch.tensor(z[f'{tag}.y'])).abs().mean()) < 1.5e-3


@pytest.mark.parametrize("t", [9, 1, 0])
def test_p_sample_vs_reference_golden(sr3, cuda, golden_dir, t):
    """One ancestral step with the reference's noise draw (CPU generator). Tolerance 3e-4 (measured 1.4e-4)."""
    net, _ = sr3
    net.set_new_noise_schedule(dict(schedule="linear", n_timestep=10, linear_start=1e-6, linear_end=1e-2), cuda)
    net.noise_source = "cpu"
    z = np.load(os.path.join(golden_dir, "sr3_p_sample.npz"))
    torch.manual_seed(int(z[f"t{t}.seed"]))
    out = net.p_sample(torch.tensor(z["x"]).to(cuda), t, condition_x=torch.tensor(z["cond"]).to(cuda))
    e = _err(out, z[f"t{t}.out"])
    print(f"p_sample t={t}: max|d| = {e:.3e}")
    assert e < 3e-4


@pytest.mark.parametrize("prec", ["fp16", "fp32", "split", "w2"])
def test_pipeline_config1_vs_reference_golden(sr3, cuda, golden_dir, prec):
    """BASELINE config 1: 64 -> 256 (x4), 1 image, 10 DDPM steps, torch seed 0, CPU noise order.
    The reference hands Stage 1's result to Stage 2 as uint8 (utils/tensor2img.py); report both the
    fp32 per-pixel error and the 8-bit agreement.  fp16 = the default 16-bit kernels; fp32 = ``compute_dtype: fp32``, the
    fp32-operand kernel family (the reference's own Stage 1 runs in fp32); w2 = fp16 tensors x fp16 weight pairs, the Stage-1
    precision of the benchmarked (tolerance) composition: north_star's 1e-3 is its bound."""
    net, _ = sr3
    net.denoise_fn.set_compute_dtype(prec)
    try:
        _pipeline_config1(net, cuda, golden_dir, prec)
    finally:
        net.denoise_fn.set_compute_dtype("fp16")


def _pipeline_config1(net, cuda, golden_dir, prec):
    from oracle import sr3_oracle as O
    net.set_new_noise_schedule(dict(schedule="linear", n_timestep=10, linear_start=1e-6, linear_end=1e-2), cuda)
    net.noise_source = "cpu"
    z = np.load(os.path.join(golden_dir, "sr3_pipeline_c1.npz"))
    torch.manual_seed(int(z["torch_seed"]))
    sr = net.super_resolution(torch.tensor(z["cond"]).to(cuda), continous=True)
    assert sr.shape == (11, 3, 256, 256)
    final = sr[-1:].cpu()
    want = torch.tensor(z["final"])
    d = (final - want).abs()
    u8a, u8b = O.tensor2img_u8(final), O.tensor2img_u8(want)
    lsb = np.abs(u8a.astype(int) - u8b.astype(int))
    print(f"config-1 pipeline: max|d| = {float(d.max()):.3e}, mean|d| = {float(d.mean()):.3e}, "
          f"uint8 equal = {float((lsb == 0).mean()):.4f}, max LSB diff = {int(lsb.max())}")
    # north_star: |d| < 1e-3 per pixel.  Measured: max 9.9e-4, mean 1.0e-4, 98.7 % of the 8-bit hand-off identical, 1 LSB.
    if prec == "fp32":
        assert float(d.max()) < 5e-5 and float(d.mean()) < 5e-6 and float((lsb == 0).mean()) >= 0.9995 and int(lsb.max()) <= 1
        return
    if prec == "w2":     # the benchmarked Stage-1 dtype: the bound is north_star's own (|d| < 1e-3 per pixel)
        assert float(d.max()) < 1e-3 and float(d.mean()) < 1e-4 and float((lsb == 0).mean()) >= 0.99 and int(lsb.max()) <= 1
        return
    # fp16 operands sit AT north_star's 1e-3 after 10 steps (8.3e-4 .. 9.9e-4 across boxes), not inside it with margin: the mode
    # that holds the criterion with margin is compute_dtype fp32 above (9.5e-7).  Bound = 1.5 x measured.
    assert float(d.max()) < 1.5e-3 and float(d.mean()) < 2e-4
    assert float((lsb == 0).mean()) >= 0.98 and int(lsb.max()) <= 1


@pytest.mark.parametrize("prec", ["fp16", "fp32", "split", "w2"])
def test_pipeline_t50_vs_reference_golden(sr3, cuda, golden_dir, prec):
    """The step count the metric is quoted on: the config-1 image through T = 50 ancestral steps (torch seed 0, CPU noise
    order) against the reference's own 50-step run (tests/golden/gen_steps50_golden.py): the final frame, the 11 kept
    frames and x_t after t = 40 / 25 / 10, i.e. how the 16-bit error grows along ``x_(t-1) = mean(x_t, eps) + sigma z``
    (diffusion.py:170-175).  Bounds = 2 x measured."""
    from oracle import sr3_oracle as O
    net, _ = sr3
    net.denoise_fn.set_compute_dtype(prec)
    probes, orig = {}, net.p_sample

    def spy(x, t, *a, **k):
        out = orig(x, t, *a, **k)
        if t in (40, 25, 10):
            probes[t] = out.cpu()
        return out

    try:
        net.set_new_noise_schedule(dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=1e-2), cuda)
        net.noise_source = "cpu"
        z = np.load(os.path.join(golden_dir, "sr3_pipeline_t50.npz"))
        cond = torch.tensor(np.load(os.path.join(golden_dir, "sr3_pipeline_c1.npz"))["cond"])
        net.p_sample = spy
        torch.manual_seed(int(z["torch_seed"]))
        sr = net.super_resolution(cond.to(cuda), continous=True)
    finally:
        net.__dict__.pop("p_sample", None)
        net.denoise_fn.set_compute_dtype("fp16")
    assert sr.shape == (11, 3, 256, 256)
    fm = float((sr.mean(dim=(1, 2, 3)).cpu() - torch.tensor(z["frames_mean"])).abs().max())
    drift = {t: _err(probes[t], z[f"x_after_t{t}"]) for t in (40, 25, 10)}
    final, want = sr[-1:].cpu(), torch.tensor(z["final"])
    d = (final - want).abs()
    lsb = np.abs(O.tensor2img_u8(final).astype(int) - O.tensor2img_u8(want).astype(int))
    print(f"Stage 1, T = 50 [{prec}]: max|d| after t=40/25/10 = {drift[40]:.3e} / {drift[25]:.3e} / {drift[10]:.3e}; final max|d| = "
          f"{float(d.max()):.3e}, mean|d| = {float(d.mean()):.3e}; kept-frame means {fm:.3e}; uint8 equal = "
          f"{float((lsb == 0).mean()):.4f}, max LSB diff = {int(lsb.max())}")
    if prec == "fp32":
        assert float(d.max()) < 1e-4 and float(d.mean()) < 1e-5 and int(lsb.max()) <= 1
        return
    if prec == "split":
        assert float(d.max()) < 3e-4 and float(d.mean()) < 3e-5 and int(lsb.max()) <= 1
        return
    if prec == "w2":     # the benchmarked Stage-1 dtype at the metric's step count: north_star's 1e-3 (bench.py measured 5.9e-4 / 5.5e-5)
        assert float(d.max()) < 1e-3 and float(d.mean()) < 1.1e-4 and int(lsb.max()) <= 1
        return
    assert float(d.max()) < S1_T50_MAX and float(d.mean()) < S1_T50_MEAN and int(lsb.max()) <= S1_T50_LSB
    assert float((lsb == 0).mean()) >= 0.95


# fp16, 50 steps: measured max 2.0e-3 / mean 1.95e-4 (drift 5.6e-4 after 10 steps of the chain, 1.2e-3 after 25, 1.8e-3 after 40),
# 97.5 % of the 8-bit hand-off identical, never more than 1 LSB apart (10 steps: 9.9e-4 / 1.0e-4 / 98.7 %)
S1_T50_MAX, S1_T50_MEAN, S1_T50_LSB = 4e-3, 4e-4, 1


def test_unet_forward_vs_oracle_fresh_input(sr3, cuda):
    """Same check against the travelling CPU oracle on an input no fixture holds (batch 2, 48x80)."""
    from oracle import seeded, sr3_oracle as O
    net, sd = sr3
    usd = {k[len("denoise_fn."):]: v for k, v in sd.items() if k.startswith("denoise_fn.")}
    x = seeded.synthetic_image((2, 6, 48, 80), seed=77, smooth=3)
    lv = torch.tensor([[0.2], [0.95]])
    want = O.unet_forward(usd, O.SR3_CFG, x, lv)
    got = net.denoise_fn(x.to(cuda), lv.to(cuda))
    e = _err(got, want)
    print(f"unet vs oracle 48x80: max|d| = {e:.3e}")
    assert e < 8e-3          # measured 4.0e-3


def test_batch_invariant_sampling(sr3, cuda):
    """``batch_invariant = True`` plans every launch for one image (rsvld_conv_desc.plan_div): images of a batch of 3 are
    bit-identical to their batch-of-1 runs (what data-parallel sharding with batches per rank relies on)."""
    from oracle import seeded
    net, _ = sr3
    net.set_new_noise_schedule(dict(schedule="linear", n_timestep=4, linear_start=1e-6, linear_end=1e-2), cuda)
    cond = torch.cat([seeded.synthetic_image((1, 3, 64, 64), seed=20 + i, smooth=3) for i in range(3)]).to(cuda)
    noises = [torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(70 + i)) for i in range(5)]

    def run(sl):
        it = iter(noises)
        net._randn = lambda shape, device: next(it)[sl].to(device)
        try:
            return net.super_resolution(cond[sl], continous=True)[-len(range(3)[sl]):].cpu()
        finally:
            del net._randn

    net.batch_invariant = True
    try:
        both = run(slice(0, 3))
        for b in range(3):
            one = run(slice(b, b + 1))
            assert torch.equal(both[b:b + 1], one), f"image {b}: batched vs single max|d| = {float((both[b:b+1] - one).abs().max()):.3e}"
    finally:
        net.batch_invariant = False
