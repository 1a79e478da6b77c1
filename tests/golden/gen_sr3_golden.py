"""Generate the Stage-1 (SR3) golden vectors by running the REFERENCE itself.

Run in the authoring container only (needs /root/reference, which never travels to the GPU box):
    python tests/golden/gen_sr3_golden.py
Writes tests/golden/sr3_*.npz.  Weights come from oracle.seeded (seed below) applied to the
reference modules, so only inputs' seeds and expected outputs are stored.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")

import numpy as np
import torch

from models.sr3_model.sr3_modules.diffusion import GaussianDiffusion  # reference
from models.sr3_model.sr3_modules.unet import ResnetBlocWithAttn, UNet  # reference

from oracle import seeded, sr3_oracle as O

WEIGHT_SEED = 1234
torch.set_num_threads(8)


def build_reference():
    c = O.SR3_CFG
    unet = UNet(in_channel=c["in_channel"], out_channel=c["out_channel"], inner_channel=c["inner_channel"],
                norm_groups=c["norm_groups"], channel_mults=c["channel_mults"], attn_res=list(c["attn_res"]),
                res_blocks=c["res_blocks"], dropout=0.2, image_size=c["image_size"])
    net = GaussianDiffusion(unet, image_size=c["image_size"], channels=3, loss_type="l1", conditional=True)
    seeded.seed_module(net, WEIGHT_SEED)
    net.eval()
    return net


def main():
    net = build_reference()
    unet = net.denoise_fn
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    usd = {k[len("denoise_fn."):]: v for k, v in sd.items() if k.startswith("denoise_fn.")}

    # ---- G1 schedules -----------------------------------------------------------------------
    g1 = {}
    for T in (10, 50, 500):
        opt = dict(schedule="linear", n_timestep=T, linear_start=1e-6, linear_end=1e-2)
        net.set_new_noise_schedule(opt, torch.device("cpu"))
        for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                     "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                     "posterior_mean_coef1", "posterior_mean_coef2"):
            g1[f"T{T}.{name}"] = getattr(net, name).numpy().copy()
        g1[f"T{T}.sqrt_alphas_cumprod_prev"] = net.sqrt_alphas_cumprod_prev.copy()
    np.savez_compressed(os.path.join(HERE, "sr3_schedules.npz"), **g1)

    # ---- G2 op-level taps: inputs and outputs of selected layers on a 16x16 forward ---------
    taps = {}
    pick = ["downs.0", "downs.1", "downs.2", "downs.3", "downs.7", "mid.0", "mid.1", "ups.0", "ups.2", "ups.4",
            "ups.13", "final_conv"]
    mods = dict(unet.named_modules())
    hooks = []
    for name in pick:
        def hook(m, inp, out, name=name):
            taps[name + ".in"] = inp[0].detach().numpy().copy()
            taps[name + ".out"] = out.detach().numpy().copy()
        hooks.append(mods[name].register_forward_hook(hook))
    x = seeded.synthetic_image((1, 6, 16, 16), seed=11, smooth=2)
    lvl = torch.tensor([[0.7]])
    with torch.no_grad():
        y = unet(x, lvl)
    for h in hooks:
        h.remove()
    taps["x"], taps["level"], taps["y"] = x.numpy(), lvl.numpy(), y.numpy()
    with torch.no_grad():
        t_emb = unet.noise_level_mlp(lvl)
    taps["t_emb"] = t_emb.numpy()
    yo = O.unet_forward(usd, O.SR3_CFG, x, lvl)
    print("oracle vs reference, 16x16 forward: max|d| =", float((yo - y).abs().max()))
    np.savez_compressed(os.path.join(HERE, "sr3_unet_taps.npz"), **taps)

    # ---- G2b whole-UNet forwards (ragged token counts: 96/8 = 12 -> 144 tokens) --------------
    g2 = {}
    for tag, shape, seed, level in (("a", (1, 6, 96, 96), 21, [0.35]), ("b", (2, 6, 64, 64), 22, [0.9, 0.9])):
        x = seeded.synthetic_image(shape, seed=seed, smooth=3)
        lv = torch.tensor(level).view(-1, 1)
        with torch.no_grad():
            y = unet(x, lv)
        yo = O.unet_forward(usd, O.SR3_CFG, x, lv)
        print(f"oracle vs reference, forward {tag} {shape}: max|d| =", float((yo - y).abs().max()),
              "out absmax", float(y.abs().max()))
        g2[f"{tag}.shape"], g2[f"{tag}.seed"], g2[f"{tag}.smooth"] = np.array(shape), np.array(seed), np.array(3)
        g2[f"{tag}.level"], g2[f"{tag}.y"] = lv.numpy(), y.numpy()
    np.savez_compressed(os.path.join(HERE, "sr3_unet_forward.npz"), **g2)

    # ---- G3 single ancestral steps, T = 10 ---------------------------------------------------
    opt10 = dict(schedule="linear", n_timestep=10, linear_start=1e-6, linear_end=1e-2)
    net.set_new_noise_schedule(opt10, torch.device("cpu"))
    sch = O.schedule(opt10)
    g3 = {}
    cond = seeded.synthetic_image((1, 3, 32, 32), seed=31, smooth=3)
    xg = torch.Generator().manual_seed(32)
    xt = torch.randn((1, 3, 32, 32), generator=xg)
    g3["cond"], g3["x"] = cond.numpy(), xt.numpy()
    for t in (9, 1, 0):
        torch.manual_seed(100 + t)
        with torch.no_grad():
            out = net.p_sample(xt.clone(), t, condition_x=cond)
        torch.manual_seed(100 + t)
        noise = torch.randn_like(xt) if t > 0 else None
        oo = O.p_sample(sd, O.SR3_CFG, sch, xt, t, cond, noise)
        print(f"oracle vs reference, p_sample t={t}: max|d| =", float((oo - out).abs().max()))
        g3[f"t{t}.out"] = out.numpy()
        g3[f"t{t}.seed"] = np.array(100 + t)
    np.savez_compressed(os.path.join(HERE, "sr3_p_sample.npz"), **g3)

    # ---- G6 BASELINE config 1: 64 -> 256 (x4), 1 image, 10 DDPM steps, seed 0 ----------------
    lr = seeded.synthetic_image((1, 3, 64, 64), seed=41, smooth=4)
    cond = torch.nn.functional.interpolate(lr, scale_factor=4, mode="bicubic", align_corners=False).clamp(-1, 1)
    torch.manual_seed(0)
    with torch.no_grad():
        sr = net.super_resolution(cond, continous=True)
    torch.manual_seed(0)
    so = O.p_sample_loop(sd, O.SR3_CFG, sch, cond, continous=True)
    print("oracle vs reference, config-1 pipeline: max|d| =", float((so - sr).abs().max()), tuple(sr.shape))
    np.savez_compressed(os.path.join(HERE, "sr3_pipeline_c1.npz"), lr_seed=np.array(41), torch_seed=np.array(0),
                        cond=cond.numpy().astype(np.float32), final=sr[-1:].numpy(),
                        frames_mean=sr.mean(dim=(1, 2, 3)).numpy())
    print("done")


if __name__ == "__main__":
    main()
