"""CPU: the oracle (oracle/sr3_oracle.py) against golden vectors produced by the reference itself,
and the product's host-side logic (schedules, parameter names, weight packing) -- no GPU needed."""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import seeded, sr3_oracle as O

WEIGHT_SEED = 1234


@pytest.fixture(scope="module")
def weights():
    from rsvld_amd.sr3_model.sr3_modules.diffusion import GaussianDiffusion
    from rsvld_amd.sr3_model.sr3_modules.unet import UNet
    c = O.SR3_CFG
    unet = UNet(in_channel=c["in_channel"], out_channel=c["out_channel"], inner_channel=c["inner_channel"],
                norm_groups=c["norm_groups"], channel_mults=c["channel_mults"], attn_res=list(c["attn_res"]),
                res_blocks=c["res_blocks"], dropout=0.2, image_size=c["image_size"])
    net = GaussianDiffusion(unet, image_size=c["image_size"], channels=3, conditional=True)
    seeded.seed_module(net, WEIGHT_SEED)
    return net, {k: v.detach().clone() for k, v in net.state_dict().items()}


@pytest.mark.parametrize("T", [10, 50, 500])
def test_schedules_oracle_and_product(golden_dir, weights, T):
    z = np.load(os.path.join(golden_dir, "sr3_schedules.npz"))
    opt = dict(schedule="linear", n_timestep=T, linear_start=1e-6, linear_end=1e-2)
    sch = O.schedule(opt)
    net, _ = weights
    net.set_new_noise_schedule(opt, torch.device("cpu"))
    for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                 "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                 "posterior_mean_coef1", "posterior_mean_coef2"):
        want = z[f"T{T}.{name}"]
        assert np.array_equal(sch[name].numpy(), want), name           # bit-exact fp32 tables
        assert np.array_equal(getattr(net, name).numpy(), want), name
    assert np.array_equal(sch["sqrt_alphas_cumprod_prev"], z[f"T{T}.sqrt_alphas_cumprod_prev"])
    assert np.array_equal(net.sqrt_alphas_cumprod_prev, z[f"T{T}.sqrt_alphas_cumprod_prev"])
    assert net.num_timesteps == T


def test_oracle_unet_taps(golden_dir, weights):
    _, sd = weights
    usd = {k[len("denoise_fn."):]: v for k, v in sd.items() if k.startswith("denoise_fn.")}
    z = np.load(os.path.join(golden_dir, "sr3_unet_taps.npz"))
    taps = {}
    y = O.unet_forward(usd, O.SR3_CFG, torch.tensor(z["x"]), torch.tensor(z["level"]), taps=taps)
    assert float((y - torch.tensor(z["y"])).abs().max()) < 2e-5
    for name in ["downs.0", "downs.1", "downs.2", "downs.3", "downs.7", "mid.0", "mid.1", "ups.0", "ups.2", "ups.4",
                 "ups.13", "final_conv"]:
        assert float((taps[name] - torch.tensor(z[name + ".out"])).abs().max()) < 2e-5, name


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_unet_forward(golden_dir, weights, tag):
    _, sd = weights
    usd = {k[len("denoise_fn."):]: v for k, v in sd.items() if k.startswith("denoise_fn.")}
    z = np.load(os.path.join(golden_dir, "sr3_unet_forward.npz"))
    shape = tuple(int(v) for v in z[f"{tag}.shape"])
    x = seeded.synthetic_image(shape, seed=int(z[f"{tag}.seed"]), smooth=int(z[f"{tag}.smooth"]))
    y = O.unet_forward(usd, O.SR3_CFG, x, torch.tensor(z[f"{tag}.level"]))
    assert float((y - torch.tensor(z[f"{tag}.y"])).abs().max()) < 2e-5


@pytest.mark.parametrize("t", [9, 1, 0])
def test_oracle_p_sample(golden_dir, weights, t):
    _, sd = weights
    z = np.load(os.path.join(golden_dir, "sr3_p_sample.npz"))
    sch = O.schedule(dict(schedule="linear", n_timestep=10, linear_start=1e-6, linear_end=1e-2))
    x, cond = torch.tensor(z["x"]), torch.tensor(z["cond"])
    torch.manual_seed(int(z[f"t{t}.seed"]))
    noise = torch.randn_like(x) if t > 0 else None
    out = O.p_sample(sd, O.SR3_CFG, sch, x, t, cond, noise)
    assert float((out - torch.tensor(z[f"t{t}.out"])).abs().max()) < 1e-5


def test_oracle_pipeline_config1(golden_dir, weights):
    """BASELINE config 1 (the reference's CPU-runnable case) end to end on the oracle."""
    _, sd = weights
    z = np.load(os.path.join(golden_dir, "sr3_pipeline_c1.npz"))
    sch = O.schedule(dict(schedule="linear", n_timestep=10, linear_start=1e-6, linear_end=1e-2))
    cond = torch.tensor(z["cond"])
    lr = seeded.synthetic_image((1, 3, 64, 64), seed=int(z["lr_seed"]), smooth=4)
    regen = F.interpolate(lr, scale_factor=4, mode="bicubic", align_corners=False).clamp(-1, 1)
    assert float((regen - cond).abs().max()) < 1e-6   # the input recipe is reproducible
    torch.manual_seed(int(z["torch_seed"]))
    sr = O.p_sample_loop(sd, O.SR3_CFG, sch, cond, continous=True)
    assert sr.shape == (11, 3, 256, 256)
    assert float((sr[-1:] - torch.tensor(z["final"])).abs().max()) < 2e-5
    assert np.allclose(sr.mean(dim=(1, 2, 3)).numpy(), z["frames_mean"], atol=1e-5)


def test_parameter_names_match_reference_contract(golden_dir, weights):
    """State-dict keys/shapes are part of the drop-in boundary (checkpoint I1000000_E800_gen.pth)."""
    net, _ = weights
    with open(os.path.join(golden_dir, "sr3_param_names.json")) as f:
        want = json.load(f)
    got = [[k, list(v.shape)] for k, v in net.denoise_fn.state_dict().items()]
    assert got == want


def test_pack_conv_layout_cpu():
    """The K-major packed weights reproduce F.conv2d through a plain im2col GEMM (host logic)."""
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(0)
    w = torch.randn(20, 6, 3, 3, generator=g)
    b = torch.randn(20, generator=g)
    pc = ops.pack_conv(w, b, torch.float32, "cpu")
    assert (pc.cin_p, pc.cout_p, tuple(pc.w.shape)) == (8, 24, (24, 72))
    x = torch.randn(2, 6, 5, 7, generator=g)
    xp = F.pad(x.permute(0, 2, 3, 1), (0, 2, 1, 1, 1, 1))  # NHWC, C->8, spatial pad 1
    cols = torch.stack([xp[:, ky:ky + 5, kx:kx + 7, :] for ky in range(3) for kx in range(3)], dim=3)  # [B,H,W,9,8]
    y = cols.reshape(2, 5, 7, 72) @ pc.w.t() + pc.bias
    want = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1)
    assert torch.allclose(y[..., :20], want, atol=1e-4)
    assert float(y[..., 20:].abs().max()) == 0.0
    # two-source concat packing pads each source separately
    pc2 = ops.pack_conv(torch.randn(16, 10, 1, 1, generator=g), None, torch.float32, "cpu", cin_split=(4, 6))
    assert pc2.cin_p == 16 and float(pc2.w[:, 4:8].abs().max()) == 0.0 and float(pc2.w[:, 14:].abs().max()) == 0.0
    # GEGLU interleave: value j / gate j adjacent
    wl = torch.arange(8.0).view(8, 1).repeat(1, 8)
    pg = ops.pack_conv(wl, None, torch.float32, "cpu", geglu=True)
    assert pg.w[:, 0].tolist()[:8] == [0, 4, 1, 5, 2, 6, 3, 7]


def test_product_refuses_cpu_execution(weights):
    from rsvld_amd._lib import RsvldError
    net, _ = weights
    net.set_new_noise_schedule(dict(schedule="linear", n_timestep=10, linear_start=1e-6, linear_end=1e-2),
                               torch.device("cpu"))
    with pytest.raises(RsvldError):
        net.super_resolution(torch.zeros(1, 3, 16, 16))
    with pytest.raises(RsvldError):
        net.denoise_fn(torch.zeros(1, 6, 16, 16), torch.zeros(1, 1))
