"""Measurement infrastructure (not product code): the distance of a precision mode from the REFERENCE's CPU path after the
metric's 50 + 50 steps, measured on the device this process runs on against the committed reference-generated goldens
(tests/golden/sr3_pipeline_t50.npz, s2_pipeline_50.npz; generator scripts next to them).  It is what
tests/test_gpu_sr3.py::test_pipeline_t50_vs_reference_golden and tests/test_gpu_steps50.py::test_just_sampling_50_steps_vs_reference_golden
assert, packaged so that ``bench.py`` can print the figure of the mode it has just timed in the same run
(``config.tolerance_mode.max_abs_err_50_steps``).  The goldens were produced with the seeded weights of ``oracle/seeded.py``, so
that seeding utility is used here as well -- as the checker's fixture, never on the timed path."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def stage1_t50(dev, prec, policy=None):
    """Stage 1 (SR3): the config-1 image through T = 50 ancestral steps, CPU noise order -> max / mean |delta| of the final frame."""
    from oracle import seeded, sr3_oracle as O
    from rsvld_amd.sr3_model.sr3_modules.diffusion import GaussianDiffusion
    from rsvld_amd.sr3_model.sr3_modules.unet import UNet
    c = O.SR3_CFG
    unet = UNet(in_channel=c["in_channel"], out_channel=c["out_channel"], inner_channel=c["inner_channel"], norm_groups=c["norm_groups"],
                channel_mults=c["channel_mults"], attn_res=list(c["attn_res"]), res_blocks=c["res_blocks"], dropout=0.2,
                image_size=c["image_size"])
    net = GaussianDiffusion(unet, image_size=c["image_size"], channels=3, conditional=True)
    seeded.seed_module(net, 1234)
    net.to(dev).eval()
    net.denoise_fn.set_compute_dtype(prec, policy=policy)
    net.set_new_noise_schedule(dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=1e-2), dev)
    net.noise_source = "cpu"
    z = np.load(os.path.join(GOLDEN, "sr3_pipeline_t50.npz"))
    cond = torch.tensor(np.load(os.path.join(GOLDEN, "sr3_pipeline_c1.npz"))["cond"])
    torch.manual_seed(int(z["torch_seed"]))
    sr = net.super_resolution(cond.to(dev), continous=True)
    d = (sr[-1:].cpu() - torch.tensor(z["final"])).abs()
    return {"max": float(d.max()), "mean": float(d.mean())}


def stage2_50(dev, ae, diff, cache=False, policy=None):
    """Stage 2: ``just_sampling`` over 50 EDM steps at 64^2 (reduced-depth networks of the goldens), cache off or 0.3."""
    import s2_common as S
    from oracle import seeded
    from rsvld_amd.sgm.util import instantiate_from_config
    m = instantiate_from_config({"target": "rsvld_amd.models.SR_model.SR_backbone", "params": S.product_params()})
    seeded.seed_module(m, S.WEIGHT_SEED)
    m.to(dev).eval()
    img = seeded.synthetic_image((1, 3, 64, 64), seed=80, smooth=3).to(dev)
    opt = dict(S.PIPE_OPT, num_steps=50, **({} if cache else {"img_threshold": 0.0}))
    m.noise_source = "cpu"
    m.set_precision(ae, diff, policy=policy)
    torch.manual_seed(7)
    out = m.just_sampling(img, [""], p_p="", n_p="", **opt).cpu()
    z = np.load(os.path.join(GOLDEN, "s2_pipeline_50.npz"))
    tag = "cache50" if cache else "nocache50"
    want = torch.tensor(z[f"{tag}.final"])
    d = (out - want).abs()
    res = {"max": float(d.max()), "mean": float(d.mean()), "range": float(want.abs().max())}
    if cache:
        got = [step[0] for step in m.cache_trace]
        res["cache_decisions_equal_reference"] = all(bool(w[2]) == g[2] for g, w in zip(got, z[f"{tag}.trace"]))
    return res


def errors_after_50_steps(dev, sr3_prec, ae, diff, policy=None, sr3_policy=None):
    """-> {stage1: {max, mean}, stage2: {max, mean, range}} vs the reference's CPU runs (same seeds, same noise order).
    ``policy`` / ``sr3_policy``: the ops.SplitPolicy of the Stage-2 UNets / of SR3 when their precision is "split" (None = the default)."""
    with torch.random.fork_rng(devices=[dev]):
        return {"stage1_T50_256px": stage1_t50(dev, sr3_prec, sr3_policy), "stage2_50_steps_64px": stage2_50(dev, ae, diff, policy=policy),
                "reference": "tests/golden/sr3_pipeline_t50.npz, s2_pipeline_50.npz (the reference's own CPU runs)"}


if __name__ == "__main__":
    import json
    dev = torch.device("cuda:0")
    from rsvld_amd import ops
    P = ops.SplitPolicy
    # name -> (SR3 precision, VAE, UNets, UNet policy, SR3 policy); split = the mode as shipped
    modes = {"shipped": ("fp16", "bf16", "fp16", None, None), "vae_split": ("fp16", "split", "fp16", None, None),
             "split": ("split", "split", "split", None, None),
             "split_s1w2": ("w2", "split", "split", None, None),                            # Stage 1: fp16 tensors x weight pairs
             "split_r04": ("split", "split", "split", P(f16_inputs=("attn",)), P(f16_inputs=("attn",))),   # round 4's composition
             "split_noqkv": ("split", "split", "split", P(f16_inputs=("attn", "attn_out", "ff"), f16_weights=()), None),
             "split_noconv": ("w2", "split", "split", P(f16_inputs=("attn", "attn_out", "ff", "qkv"), f16_weights=()), None),
             "split_conv": ("w2", "split", "split", P(f16_inputs=("attn", "attn_out", "ff", "qkv", "conv1", "conv2"), f16_weights=()), None),
             "split_conv2": ("w2", "split", "split", P(f16_inputs=("attn", "attn_out", "ff", "qkv", "conv2"), f16_weights=()), None),
             "split_proj": ("split", "split", "split", P(f16_inputs=("attn", "attn_out", "ff", "qkv", "proj"), f16_weights=()), None),
             "split_pairs": ("w2", "split", "split", P(f16_weights=()), None),                 # every weight of the fp16-input GEMMs as a pair (two MFMAs)
             "split_w1_qkv": ("w2", "split", "split", P(f16_weights=("qkv",)), None),          # to_q / to_k / to_v weights rounded to fp16 (one MFMA)
             "split_w1_geglu": ("w2", "split", "split", P(f16_weights=("geglu",)), None),      # the GEGLU projection's
             "split_w1_all": ("w2", "split", "split", P(f16_weights=("qkv", "geglu", "attn_out", "ff_out")), None),   # + to_out and ff.net.2 (RSVLD_F16W1)
             "split_w1_ao": ("w2", "split", "split", P(f16_weights=("qkv", "geglu", "attn_out")), None),
             "split_w1_fo": ("w2", "split", "split", P(f16_weights=("qkv", "geglu", "ff_out")), None),
             "split_noq8": ("w2", "split", "split", P(q8_convs=()), None),   # round 5's composition: the ResBlock convolutions as three bf16 MFMAs
             "split_full": ("split", "split", "split", ops.ALL_SPLIT, ops.ALL_SPLIT)}
    if "--only" in sys.argv:
        modes = {k: v for k, v in modes.items() if k in sys.argv[sys.argv.index("--only") + 1].split(",")}
    for name, (s1, ae, df, pol, pol1) in modes.items():
        print(name, json.dumps(errors_after_50_steps(dev, s1, ae, df, policy=pol, sr3_policy=pol1)), flush=True)
