"""CPU: the Stage-2 oracle (oracle/s2_oracle.py) against golden vectors produced by the reference itself
(tests/golden/gen_s2_golden.py), plus host-side logic of the product (schedules, sigma quantisation,
CFG schedule, parameter-name contract, yaml plugin registry)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import s2_common as S
from oracle import s2_oracle as O
from oracle import seeded


@pytest.fixture(scope="module")
def model():
    from rsvld_amd.sgm.util import instantiate_from_config
    m = instantiate_from_config({"target": "rsvld_amd.models.SR_model.SR_backbone", "params": S.product_params()})
    seeded.seed_module(m, S.WEIGHT_SEED)
    return m.eval(), {k: v.detach().clone() for k, v in m.state_dict().items()}


def _d(a, b):
    return float((torch.as_tensor(a).float() - torch.as_tensor(b).float()).abs().max())


def test_parameter_contract_and_plugin_registry(model, golden_dir):
    m, sd = model
    want = json.load(open(os.path.join(golden_dir, "s2_param_names.json")))
    assert [[k, list(v.shape)] for k, v in sd.items()] == want   # names, shapes AND order of the reference
    from rsvld_amd.models.modules.SR_modules import GLVControl, LightGLVUNet
    from rsvld_amd.sgm.modules.diffusionmodules.denoiser import DiscreteDenoiserWithControl
    from rsvld_amd.sgm.modules.diffusionmodules.wrappers import ControlWrapper
    assert isinstance(m.model, ControlWrapper) and isinstance(m.model.diffusion_model, LightGLVUNet)
    assert isinstance(m.model.control_model, GLVControl) and isinstance(m.denoiser, DiscreteDenoiserWithControl)
    assert m.model.dtype == torch.float16 and m.ae_dtype == torch.bfloat16


def test_schedules_oracle_and_product(model, golden_dir):
    m, _ = model
    z = np.load(os.path.join(golden_dir, "s2_schedules.npz"))
    from rsvld_amd.sgm.modules.diffusionmodules.discretizer import LegacyDDPMDiscretization
    from rsvld_amd.sgm.modules.diffusionmodules.guiders import LinearCFG
    disc = LegacyDDPMDiscretization()
    for n in (6, 50):
        assert np.array_equal(O.legacy_ddpm_sigmas(n).numpy(), z[f"sigmas{n}"])
        assert np.array_equal(disc(n, device="cpu").numpy(), z[f"sigmas{n}"])
    assert np.array_equal(m.denoiser.sigmas.numpy(), z["table"])
    assert np.array_equal(m.denoiser.sigma_to_idx(torch.tensor(z["probe"])).numpy(), z["probe_idx"])
    assert np.array_equal(O.quantize_sigma(torch.tensor(z["probe"]), torch.tensor(z["table"]))[1].numpy(), z["probe_idx"])
    got = LinearCFG(scale=4.0, scale_min=7.5).scale_schedule(torch.tensor(z["sigmas50"])).numpy()
    assert np.array_equal(got, z["cfg_scale"])


def test_oracle_ops(model, golden_dir):
    _, sd = model
    z = np.load(os.path.join(golden_dir, "s2_networks.npz"))
    P = "model.diffusion_model."
    emb, ctx = S.rnd((2, 1280), 50, 0.5), S.rnd((2, 77, 64), 51)
    x320, x640, x1280 = S.rnd((2, 320, 8, 8), 52), S.rnd((2, 640, 4, 4), 53), S.rnd((2, 1280, 4, 4), 54)
    got = {
        "res_320": O.resblock(sd, P + "input_blocks.1.0", x320, emb),
        "res_320_640": O.resblock(sd, P + "input_blocks.4.0", S.rnd((2, 320, 4, 4), 55), emb),
        "st_640": O.spatial_transformer(sd, P + "input_blocks.4.1", x640, ctx),
        "st_1280": O.spatial_transformer(sd, P + "input_blocks.7.1", x1280, ctx),
        "sft_mid": O.zero_sft(sd, P + "project_modules.11", x1280, S.rnd((2, 1280, 4, 4), 56)),
        "sft_cat": O.zero_sft(sd, P + "project_modules.10", x1280, S.rnd((2, 1280, 4, 4), 57), S.rnd((2, 1280, 4, 4), 58)),
        "sft_cat_cs": O.zero_sft(sd, P + "project_modules.0", x320, S.rnd((2, 320, 8, 8), 59), S.rnd((2, 320, 8, 8), 60), 0.7),
        "zca_7": O.zero_cross_attn(sd, P + "project_modules.7", x640, x1280),
        "zca_3": O.zero_cross_attn(sd, P + "project_modules.3", x320, S.rnd((2, 640, 8, 8), 61), 0.9),
    }
    for k, v in got.items():
        assert _d(v, z["op." + k]) < 5e-5, k
    t, y = torch.tensor([999.0, 19.0]), S.rnd((2, 32), 62)
    assert _d(O.embed(sd, P, t, y), z["op.emb"]) < 1e-5


def test_oracle_networks(model, golden_dir):
    _, sd = model
    z = np.load(os.path.join(golden_dir, "s2_networks.npz"))
    t, y, ctx = torch.tensor([999.0, 19.0]), S.rnd((2, 32), 62), S.rnd((2, 77, 64), 51)
    xt, xc = S.rnd((2, 4, 16, 16), 70), S.rnd((2, 4, 16, 16), 71, 0.5)
    control = O.glv_control(sd, xc, t, xt, ctx, y)
    assert _d(control[9], z["control.9"]) < 2e-4
    part = O.light_unet_stage1(sd, xt, t, ctx, y)
    assert _d(part["h"], z["unet.h"]) < 2e-4
    assert _d(O.light_unet_stage2(sd, part, ctx, control, 1.0), z["unet.out"]) < 5e-5
    assert _d(O.light_unet(sd, xt, t, ctx, y, control, 0.8), z["unet.out_cs08"]) < 5e-5


def test_oracle_vae_colorfix(model, golden_dir):
    _, sd = model
    z = np.load(os.path.join(golden_dir, "s2_vae_colorfix.npz"))
    img = seeded.synthetic_image((1, 3, 64, 64), seed=80, smooth=3)
    assert _d(O.conv(sd, "first_stage_model.quant_conv", O.vae_encoder(sd, img)), z["moments"]) < 5e-5
    assert _d(O.decode(sd, S.rnd((1, 4, 8, 8), 81)), z["decoded"]) < 5e-5
    assert _d(O.encode_with_denoise(sd, img), z["z_denoise"]) < 1e-5
    torch.manual_seed(5)
    assert _d(O.encode_sample(sd, img, torch.randn(1, 4, 8, 8)), z["z_sample_seed5"]) < 1e-5
    a, b = S.rnd((2, 3, 48, 40), 82), S.rnd((2, 3, 48, 40), 83, 0.5) + 0.2
    assert _d(O.wavelet_reconstruction(a, b), z["wavelet"]) < 1e-6
    assert _d(O.adain(a, b), z["adain"]) < 1e-5


@pytest.mark.parametrize("tag", ["restore", "lincs", "adain", "ns2"])
def test_oracle_pipeline_branches(model, golden_dir, tag):
    """The branches the default call does not take (restore pull, linear control scale, AdaIN, num_samples = 2),
    each against the reference's own output (tests/golden/s2_branches.npz)."""
    _, sd = model
    z = np.load(os.path.join(golden_dir, "s2_branches.npz"))
    img = seeded.synthetic_image((1, 3, 64, 64), seed=80, smooth=3)
    c, uc = S.cond_dicts()
    trace = []
    torch.manual_seed(7)
    out = O.just_sampling(sd, img, c, uc, dict(S.PIPE_OPT, **S.BRANCHES[tag]), trace=trace)
    assert _d(out, z[f"pipe.{tag}.final"]) < 1e-4
    want = z[f"pipe.{tag}.trace"]
    assert len(trace) == len(want)
    for (a, b, h), w in zip(trace, want):
        assert bool(w[2]) == h and abs(a - w[0]) < 1e-4 and abs(b - w[1]) < 1e-3


@pytest.mark.parametrize("i", [0, 1, 49])
def test_oracle_sampler_step(model, golden_dir, i):
    """G3: RestoreEDMSampler.step at the first, second and last step of a 50-step schedule, churn noise by seed:
    a miss, then a forced hit on a different latent (the cached prediction is reused)."""
    _, sd = model
    z = np.load(os.path.join(golden_dir, "s2_branches.npz"))
    o = S.STEP_OPT
    sigmas = O.legacy_ddpm_sigmas(o["num_steps"])
    assert np.array_equal(sigmas.numpy(), z["step.sigmas"])
    table = O.legacy_ddpm_sigmas(1000, append_zero=False, flip=True)
    c0, uc0 = S.cond_dicts()
    _z, x_center = S.rnd((1, 4, 8, 8), 201, 0.8), S.rnd((1, 4, 8, 8), 202, 0.8)
    c, uc = dict(c0, control=_z), dict(uc0, control=_z)
    sopt = dict(s_churn=o["s_churn"], s_noise=o["s_noise"], restore_cfg=o["restore_cfg"], scale=o["cfg_scale_start"],
                scale_min=o["cfg_scale"], control_scale=1.0)
    cache = O.Cache()
    torch.manual_seed(1000 + i)
    x_miss, t_miss = O.restore_edm_step(sd, table, cache, torch.tensor(z[f"step.i{i}.x_in"]), i, sigmas, c, uc, x_center, sopt,
                                        1e-9, torch.randn)
    torch.manual_seed(2000 + i)
    x_hit, t_hit = O.restore_edm_step(sd, table, cache, torch.tensor(z[f"step.i{i}.x_in2"]), i, sigmas, c, uc, x_center, sopt,
                                      1e9, torch.randn)
    assert _d(x_miss, z[f"step.i{i}.miss"]) < 2e-4 and _d(x_hit, z[f"step.i{i}.hit"]) < 2e-4
    assert [t_miss, t_hit] == list(z[f"step.i{i}.thr"])


@pytest.mark.parametrize("tag", ["cache", "nocache"])
def test_oracle_pipeline(model, golden_dir, tag):
    """just_sampling end to end (VAE x4, conditioner, 6 sampler steps, cache decisions, wavelet fix)."""
    _, sd = model
    z = np.load(os.path.join(golden_dir, "s2_pipeline.npz"))
    img = seeded.synthetic_image((1, 3, 64, 64), seed=80, smooth=3)
    c, uc = S.cond_dicts()
    trace = []
    torch.manual_seed(7)
    thr = S.PIPE_OPT["img_threshold"] if tag == "cache" else 0.0
    out = O.just_sampling(sd, img, c, uc, dict(S.PIPE_OPT, img_threshold=thr), trace=trace)
    assert _d(out, z[f"{tag}.final"]) < 1e-4
    want = z[f"{tag}.trace"]
    assert len(trace) == len(want)
    for (a, b, h), w in zip(trace, want):
        assert bool(w[2]) == h and abs(a - w[0]) < 1e-4 and abs(b - w[1]) < 1e-3


def test_create_sr_model_loads_both_checkpoints_in_reference_order(model, tmp_path, monkeypatch):
    """models/util.py:73-108: yaml -> instantiate, then SR_CKPT (.safetensors) and SR_CKPT_Q (.ckpt holding
    {'state_dict': ...}) with strict=False, the second overriding the first where both carry a key; default_setting is
    returned on request.  The (30 s) network build is replaced by the fixture's instance; everything else is the function's own."""
    import safetensors.torch
    import yaml
    import rsvld_amd.models.util as U
    m, sd = model
    keys = [k for k in sd if sd[k].numel() < 4096 and sd[k].dtype == torch.float32]
    base_keys = [k for k in keys if k.startswith("model.diffusion_model.")][:6] + [k for k in keys if k.startswith("first_stage_model.")][:3]
    adapter_keys = [k for k in keys if k.startswith("model.control_model.")][:6]
    shared = base_keys[0]
    base = {k: (sd[k] + 0.5).contiguous() for k in base_keys}
    adapter = {k: (sd[k] - 0.25).contiguous() for k in adapter_keys + [shared]}
    safetensors.torch.save_file(base, str(tmp_path / "base.safetensors"))
    torch.save({"state_dict": adapter}, str(tmp_path / "adapter.ckpt"))
    assert set(U.load_state_dict(str(tmp_path / "adapter.ckpt"))) == set(adapter)
    assert set(U.load_state_dict(str(tmp_path / "base.safetensors"))) == set(base)
    cfg = yaml.safe_load(open(S.YAML))
    cfg["SR_CKPT"], cfg["SR_CKPT_Q"] = str(tmp_path / "base.safetensors"), str(tmp_path / "adapter.ckpt")
    with open(tmp_path / "m.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    seen = {}

    def fake_instantiate(c):
        seen["target"] = c["target"]
        return m
    monkeypatch.setattr(U, "instantiate_from_config", fake_instantiate)
    try:
        got, default_setting = U.create_SR_model(str(tmp_path / "m.yaml"), load_default_setting=True)
        assert got is m and seen["target"] == "rsvld_amd.models.SR_model.SR_backbone"
        now = m.state_dict()
        for k in sd:
            want = sd[k] - 0.25 if k in adapter else (sd[k] + 0.5 if k in base else sd[k])
            assert torch.equal(now[k], want), k
        assert default_setting["s_cfg_Quality"] == 7.5
        cfg["SR_CKPT"] = str(tmp_path / "missing.safetensors")     # a configured but absent checkpoint is an error, never skipped
        with open(tmp_path / "m2.yaml", "w") as f:
            yaml.safe_dump(cfg, f)
        with pytest.raises(FileNotFoundError):
            U.create_SR_model(str(tmp_path / "m2.yaml"))
        cfg["SR_CKPT"] = None                                       # models/util.py:101-103: "There are no pretrained weights."
        with open(tmp_path / "m3.yaml", "w") as f:
            yaml.safe_dump(cfg, f)
        assert U.create_SR_model(str(tmp_path / "m3.yaml")) is None
        assert U.create_SR_model(str(tmp_path / "m3.yaml"), allow_random_init=True) is m
    finally:
        m.load_state_dict(sd)


def test_tiled_sampler_windows_and_weights(golden_dir):
    """sampling.py:830-863: window lists against the reference's own `_sliding_windows` (golden); the Gaussian tile mask of the
    product AND of the oracle bit for bit against the plane the reference's own `gaussian_weights` returned
    (tests/golden/tiled_sampler_mask.npz, float64; gen_tiled_golden.py calls it with only the device keyword dropped)."""
    from rsvld_amd.sgm.modules.diffusionmodules.sampling import _sliding_windows, gaussian_weights
    for case in json.load(open(os.path.join(golden_dir, "tiled_sampler_windows.json"))):
        want = [tuple(t) for t in case["windows"]]
        assert O.sliding_windows(*case["args"]) == want
        assert _sliding_windows(*case["args"]) == want
        h, w, ts, _ = case["args"]
        cover = torch.zeros(h, w)
        for hi, he, wi, we in want:
            assert he - hi == ts and we - wi == ts and 0 <= hi and he <= h and 0 <= wi and we <= w
            cover[hi:he, wi:we] += 1
        assert float(cover.min()) >= 1          # every latent pixel is inside some tile
    masks = np.load(os.path.join(golden_dir, "tiled_sampler_mask.npz"))
    for tw, th in ((16, 16), (128, 128), (9, 16)):
        got = gaussian_weights(tw, th, 2)
        want = O.gaussian_weights(tw, th)
        ref = torch.tensor(masks[f"mask_{tw}x{th}"])
        assert ref.dtype == torch.float64 and ref.shape == (th, tw)
        assert got.dtype == torch.float64 and got.shape == (2, 4, th, tw)
        assert torch.equal(want, ref), f"oracle mask {tw}x{th}: max|d| = {float((want - ref).abs().max()):.3e}"
        assert torch.equal(got[1, 3].cpu(), ref), f"product mask {tw}x{th}: max|d| = {float((got[1, 3].cpu() - ref).abs().max()):.3e}"
        assert torch.equal(got[0, 0], got[1, 2])
        assert torch.allclose(want[:, 0], want[:, -1], rtol=1e-14)        # columns: symmetric about (w-1)/2
        assert torch.allclose(want[1], want[-1], rtol=1e-14)              # rows: about h/2 (the reference's asymmetry)
        assert float(want[0, tw // 2]) < float(want[-1, tw // 2])


def test_tiled_sampler_oracle_single_tile_equals_untiled(model):
    """A latent of exactly one tile: the blend is (x*w)/w, so the tiled loop reduces to the un-cached RestoreEDMSampler
    step with the same draws and a FIXED x_center (the tiled sampler never updates it, sampling.py:714,726)."""
    _, sd = model
    table = O.legacy_ddpm_sigmas(1000, append_zero=False, flip=True)
    cd, ucd = S.cond_dicts()
    zc, x0, xc = S.rnd((1, 4, 16, 16), 201, 0.5), S.rnd((1, 4, 16, 16), 202), S.rnd((1, 4, 16, 16), 203, 0.5)
    c, uc = {**cd, "control": zc}, {**ucd, "control": zc}
    sigmas = O.legacy_ddpm_sigmas(2)
    opt = dict(s_churn=5, s_noise=1.003, restore_cfg=2.0, scale=4.0, scale_min=7.5, control_scale=1.0)
    draws = [S.rnd((1, 4, 16, 16), 210 + i) for i in range(2)]
    it = iter(draws)
    tiled = O.tiled_restore_edm(sd, table, x0, sigmas, c, uc, xc, opt, lambda shape: next(it), 16, 8)
    it = iter(draws)
    x = x0 * torch.sqrt(1.0 + sigmas[0] ** 2.0)
    for i in range(len(sigmas) - 1):
        x, _ = O.restore_edm_step(sd, table, None, x, i, sigmas, c, uc, xc, opt, 0.0, lambda shape: next(it))
    assert torch.isfinite(tiled).all()
    assert _d(tiled, x) < 1e-5 * max(1.0, float(x.abs().max()))


def test_vae_split_yaml_is_the_shipped_yaml_with_one_change():
    """model_configs/juggernautXL_vae_split.yaml = the shipped plugin registry with ``ae_dtype: split`` and nothing else changed."""
    import yaml
    base = yaml.safe_load(open(S.YAML))
    alt = yaml.safe_load(open(os.path.join(os.path.dirname(S.YAML), "juggernautXL_vae_split.yaml")))
    assert alt["model"]["params"]["ae_dtype"] == "split" and base["model"]["params"]["ae_dtype"] == "bf16"
    alt["model"]["params"]["ae_dtype"] = "bf16"
    assert alt == base
