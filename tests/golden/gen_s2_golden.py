"""Generate the Stage-2 golden vectors by running the REFERENCE itself (CPU, fp32) through import-only
shims for missing third-party wheels (ref_shims.py).  Authoring container only:
    python tests/golden/gen_s2_golden.py
Writes tests/golden/s2_*.npz.  Weights: oracle.seeded recipe applied to the reference SR_backbone
(all zero-initialised tensors are overwritten, SURVEY.md §8(c) "zero-init trap")."""
import copy
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_shims

AttrDict = ref_shims.install(xformers=not getattr(ref_shims, "NO_XFORMERS", False))   # (check_s2_no_xformers.py imports this module without the stand-in)

import numpy as np
import torch
import yaml

import s2_common as S
from oracle import s2_oracle as O
from oracle import seeded

torch.set_num_threads(8)


def build_reference():
    cfg = yaml.safe_load(open("/root/reference/model_configs/juggernautXL.yaml"))["model"]["params"]
    for k in ("control_stage_config", "network_config"):
        cfg[k]["params"].update(copy.deepcopy(S.SMALL))
    c, uc = S.cond_dicts()
    torch.save(c, "/tmp/s2_c.pth")
    torch.save(uc, "/tmp/s2_uc.pth")
    cfg["conditioner_config"] = {"target": "sgm.modules.PreparedConditioner",
                                 "params": {"cond_pth": "/tmp/s2_c.pth", "un_cond_pth": "/tmp/s2_uc.pth"}}
    cfg["sampler_config"]["params"]["device"] = "cpu"
    cfg["first_stage_config"]["params"]["ddconfig"]["attn_type"] = "vanilla"
    from models.SR_model import SR_backbone
    m = SR_backbone(**AttrDict(cfg))
    seeded.seed_module(m, S.WEIGHT_SEED, skip=("_lpips",))
    return m.eval()


def sub(t, n=6):
    """strided subsample + moments: a compact fingerprint of a large tensor"""
    f = t.detach().float()
    s = f.flatten()[:: max(1, f.numel() // 4096)]
    return np.concatenate([s.numpy(), np.array([f.mean(), f.abs().mean(), f.std()], dtype=np.float32)])


@torch.no_grad()
def main():
    ref = build_reference()
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items() if not k.startswith("_lpips")}
    unet, ctrl = ref.model.diffusion_model, ref.model.control_model
    out = {}

    # ---- (a) op-level goldens: reference sub-modules on seeded random inputs --------------------------
    emb = S.rnd((2, 1280), 50, 0.5)
    ctx = S.rnd((2, 77, 64), 51)
    x320, x640, x1280 = S.rnd((2, 320, 8, 8), 52), S.rnd((2, 640, 4, 4), 53), S.rnd((2, 1280, 4, 4), 54)
    P = "model.diffusion_model."
    ops = {
        "res_320": (unet.input_blocks[1][0](x320, emb), O.resblock(sd, P + "input_blocks.1.0", x320, emb)),
        "res_320_640": (unet.input_blocks[4][0](S.rnd((2, 320, 4, 4), 55), emb),
                        O.resblock(sd, P + "input_blocks.4.0", S.rnd((2, 320, 4, 4), 55), emb)),
        "st_640": (unet.input_blocks[4][1](x640, ctx), O.spatial_transformer(sd, P + "input_blocks.4.1", x640, ctx)),
        "st_1280": (unet.input_blocks[7][1](x1280, ctx), O.spatial_transformer(sd, P + "input_blocks.7.1", x1280, ctx)),
        "down_320": (unet.input_blocks[3][0](x320), O.conv(sd, P + "input_blocks.3.0.op", x320, stride=2, padding=1)),
        "up_1280": (unet.output_blocks[2][2](x1280),
                    O.conv(sd, P + "output_blocks.2.2.conv", torch.nn.functional.interpolate(x1280, scale_factor=2), padding=1)),
        "sft_mid": (unet.project_modules[11](x1280, S.rnd((2, 1280, 4, 4), 56), control_scale=1.0),
                    O.zero_sft(sd, P + "project_modules.11", x1280, S.rnd((2, 1280, 4, 4), 56))),
        "sft_cat": (unet.project_modules[10](x1280, S.rnd((2, 1280, 4, 4), 57), S.rnd((2, 1280, 4, 4), 58), control_scale=1.0),
                    O.zero_sft(sd, P + "project_modules.10", x1280, S.rnd((2, 1280, 4, 4), 57), S.rnd((2, 1280, 4, 4), 58))),
        "sft_cat_cs": (unet.project_modules[0](x320, S.rnd((2, 320, 8, 8), 59), S.rnd((2, 320, 8, 8), 60), control_scale=0.7),
                       O.zero_sft(sd, P + "project_modules.0", x320, S.rnd((2, 320, 8, 8), 59), S.rnd((2, 320, 8, 8), 60), 0.7)),
        "zca_7": (unet.project_modules[7](x640, x1280, control_scale=1.0), O.zero_cross_attn(sd, P + "project_modules.7", x640, x1280)),
        "zca_3": (unet.project_modules[3](x320, S.rnd((2, 640, 8, 8), 61), control_scale=0.9),
                  O.zero_cross_attn(sd, P + "project_modules.3", x320, S.rnd((2, 640, 8, 8), 61), 0.9)),
    }
    for k, (r, o) in ops.items():
        print(f"op {k:12s} oracle vs reference max|d| = {float((r - o).abs().max()):.2e}  range {float(r.abs().max()):.2f}")
        out["op." + k] = r.numpy()
    t = torch.tensor([999.0, 19.0])
    y = S.rnd((2, 32), 62)
    from sgm.modules.diffusionmodules.util import timestep_embedding
    e_ref = unet.time_embed(timestep_embedding(t, 320)) + unet.label_emb(y)
    print("emb oracle vs reference", float((e_ref - O.embed(sd, P, t, y)).abs().max()))
    out["op.emb"] = e_ref.numpy()

    # ---- (b,c) whole ControlNet / UNet forwards at L = 16, CFG pair --------------------------------
    xt, xc = S.rnd((2, 4, 16, 16), 70), S.rnd((2, 4, 16, 16), 71, 0.5)
    control = ctrl(x=xc, timesteps=t, xt=xt, context=ctx, y=y)
    co = O.glv_control(sd, xc, t, xt, ctx, y)
    print("control maps:", [tuple(c.shape) for c in control])
    print("glv_control oracle vs reference", max(float((a - b).abs().max()) for a, b in zip(control, co)))
    for i, c in enumerate(control):
        out[f"control.{i}.fp"] = sub(c)
    out["control.9"] = control[9].numpy()
    full = unet(xt, timesteps=t, context=ctx, y=y, control=control, control_scale=1.0, fbcache_mode="none")
    part = unet(xt, timesteps=t, context=ctx, y=y, control=control, control_scale=1.0, fbcache_mode="input_stage1")
    out["unet.h"] = part["h"].numpy().copy()
    two = unet(xt, timesteps=t, context=ctx, y=y, control=control, control_scale=1.0, fbcache_mode="input_stage2", partial_info=part)
    print("reference: none vs stage1∘stage2", float((full - two).abs().max()))
    print("light_unet oracle vs reference", float((full - O.light_unet(sd, xt, t, ctx, y, co)).abs().max()), "range", float(full.abs().max()))
    out["unet.out"] = full.numpy()
    full08 = unet(xt, timesteps=t, context=ctx, y=y, control=control, control_scale=0.8, fbcache_mode="none")
    out["unet.out_cs08"] = full08.numpy()
    np.savez_compressed(os.path.join(HERE, "s2_networks.npz"), **out)

    # ---- (d) schedules -----------------------------------------------------------------------------
    g = {}
    from sgm.modules.diffusionmodules.discretizer import LegacyDDPMDiscretization
    from sgm.modules.diffusionmodules.guiders import LinearCFG
    disc = LegacyDDPMDiscretization()
    for n in (6, 50):
        g[f"sigmas{n}"] = disc(n, device="cpu").numpy()
        assert np.array_equal(g[f"sigmas{n}"], O.legacy_ddpm_sigmas(n).numpy())
    g["table"] = ref.denoiser.sigmas.numpy()
    assert np.array_equal(g["table"], O.legacy_ddpm_sigmas(1000, append_zero=False, flip=True).numpy())
    probe = torch.tensor([14.6146 * 1.1, 14.6146, 7.0, 1.0, 0.1345, 0.03, 0.0])
    g["probe"], g["probe_idx"] = probe.numpy(), ref.denoiser.sigma_to_idx(probe).numpy()
    g["cfg_scale"] = LinearCFG(scale=4.0, scale_min=7.5).scale_schedule(torch.tensor(g["sigmas50"])).numpy()
    np.savez_compressed(os.path.join(HERE, "s2_schedules.npz"), **g)

    # ---- (f,g) VAE + colour fix --------------------------------------------------------------------
    v = {}
    img = seeded.synthetic_image((1, 3, 64, 64), seed=80, smooth=3)
    fs = ref.first_stage_model
    mom = fs.quant_conv(fs.encoder(img))
    v["moments"] = mom.numpy()
    print("vae_encoder oracle vs reference", float((mom - O.conv(sd, "first_stage_model.quant_conv", O.vae_encoder(sd, img))).abs().max()), "range", float(mom.abs().max()))
    z = S.rnd((1, 4, 8, 8), 81)
    dec = ref.decode_first_stage(z)
    v["decoded"] = dec.numpy()
    print("decode oracle vs reference", float((dec - O.decode(sd, z)).abs().max()), "range", float(dec.abs().max()))
    zd = ref.encode_first_stage_with_denoise(img, use_sample=False)
    v["z_denoise"] = zd.numpy()
    print("encode_with_denoise oracle vs reference", float((zd - O.encode_with_denoise(sd, img)).abs().max()))
    torch.manual_seed(5)
    zs = ref.encode_first_stage(img)
    torch.manual_seed(5)
    print("encode sample oracle vs reference", float((zs - O.encode_sample(sd, img, torch.randn(zs.shape))).abs().max()))
    v["z_sample_seed5"] = zs.numpy()
    from utils.colorfix import adaptive_instance_normalization, wavelet_reconstruction
    a, b = S.rnd((2, 3, 48, 40), 82), S.rnd((2, 3, 48, 40), 83, 0.5) + 0.2
    v["wavelet"] = wavelet_reconstruction(a, b).numpy()
    v["adain"] = adaptive_instance_normalization(a, b).numpy()
    print("wavelet oracle", float((torch.tensor(v["wavelet"]) - O.wavelet_reconstruction(a, b)).abs().max()),
          "adain oracle", float((torch.tensor(v["adain"]) - O.adain(a, b)).abs().max()))
    np.savez_compressed(os.path.join(HERE, "s2_vae_colorfix.npz"), **v)

    # ---- (e) pipeline: just_sampling on a 64x64 image, 6 steps, cache on, Wavelet -------------------
    import sgm.modules.diffusionmodules.sampling as RS
    trace = []
    orig = RS.get_can_use_cache_multi

    def spy(first, threshold, parallelized=False):
        use, d = orig(first, threshold=threshold, parallelized=parallelized)
        trace.append((float(threshold), float(d), bool(use)))
        return use, d

    RS.get_can_use_cache_multi = spy
    opt = S.PIPE_OPT
    pipe = {}
    for tag, thr in (("cache", opt["img_threshold"]), ("nocache", 0.0)):
        trace.clear()
        torch.manual_seed(7)
        res = ref.just_sampling(img, [""], p_p="", n_p="", img_threshold=thr, dec_img=opt["dec_img"], num_steps=opt["num_steps"],
                                restoration_scale=opt["restoration_scale"], s_churn=opt["s_churn"], s_noise=opt["s_noise"],
                                cfg_scale=opt["cfg_scale"], seed=-1, num_samples=1, control_scale=opt["control_scale"],
                                color_fix_type=opt["color_fix_type"], use_linear_CFG=opt["use_linear_CFG"],
                                use_linear_control_scale=False, cfg_scale_start=opt["cfg_scale_start"], control_scale_start=0.0)
        otr = []
        torch.manual_seed(7)
        c, uc = S.cond_dicts()
        ores = O.just_sampling(sd, img, c, uc, dict(opt, img_threshold=thr), trace=otr)
        print(f"pipeline[{tag}] oracle vs reference max|d| = {float((res - ores).abs().max()):.2e}; range {float(res.abs().max()):.2f}")
        print("   reference cache trace:", [(round(a, 4), round(b, 4), h) for a, b, h in trace])
        print("   oracle    cache trace:", [(round(a, 4), round(b, 4), h) for a, b, h in otr])
        pipe[f"{tag}.final"] = res.numpy()
        pipe[f"{tag}.trace"] = np.array([[a, b, float(h)] for a, b, h in trace], dtype=np.float64).reshape(-1, 3)
    RS.get_can_use_cache_multi = orig
    np.savez_compressed(os.path.join(HERE, "s2_pipeline.npz"), **pipe)
    # parameter-name contract of the whole SR_backbone at the reduced config
    import json
    json.dump([[k, list(v.shape)] for k, v in sd.items()], open(os.path.join(HERE, "s2_param_names.json"), "w"))
    print("done")


if __name__ == "__main__":
    main()
