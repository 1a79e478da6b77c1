"""Golden vectors at the step count the metric is quoted on (50 + 50), produced by running the REFERENCE itself
(CPU, fp32; import-only shims of ref_shims.py for the Stage-2 tree).  Authoring container only:
    python tests/golden/gen_steps50_golden.py
Writes tests/golden/sr3_pipeline_t50.npz and tests/golden/s2_pipeline_50.npz.

  sr3_pipeline_t50   GaussianDiffusion.super_resolution (models/sr3_model/sr3_modules/diffusion.py:177-201) with the
                     T = 50 'val' schedule on the config-1 image (64 -> 256, seed 0): final frame, the 11 kept
                     frames' means, and x_t after t = 40, 25, 10 (where along the chain an error first shows)
  s2_pipeline_50     SR_backbone.just_sampling (models/SR_model.py:200-298), 50 EDM steps on the 64x64 image of
                     s2_pipeline.npz, feature cache 0.3 and off: final image + the (threshold, diff, hit) trace
  ns2_cache          num_samples = 2 WITH the cache on, 6 steps: the reference takes ONE decision over the stacked
                     [2 * num_samples, ...] tensor (models/modules/DFBCache.py:98-112)
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_shims

AttrDict = ref_shims.install()

import numpy as np
import torch

import s2_common as S
from oracle import s2_oracle as O
from oracle import seeded, sr3_oracle as O1

torch.set_num_threads(8)


@torch.no_grad()
def stage1():
    from gen_sr3_golden import build_reference
    net = build_reference()
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    opt = dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=1e-2)
    net.set_new_noise_schedule(opt, torch.device("cpu"))
    lr = seeded.synthetic_image((1, 3, 64, 64), seed=41, smooth=4)
    cond = torch.nn.functional.interpolate(lr, scale_factor=4, mode="bicubic", align_corners=False).clamp(-1, 1)
    probes, orig = {}, net.p_sample

    def spy(x, t, clip_denoised=True, condition_x=None):
        out = orig(x, t, clip_denoised=clip_denoised, condition_x=condition_x)
        if t in (40, 25, 10):
            probes[t] = out.detach().clone()
        return out

    net.p_sample = spy
    torch.manual_seed(0)
    sr = net.super_resolution(cond, continous=True)
    net.p_sample = orig
    torch.manual_seed(0)
    so = O1.p_sample_loop(sd, O1.SR3_CFG, O1.schedule(opt), cond, continous=True)
    print("stage 1, T = 50: oracle vs reference max|d| =", float((so - sr).abs().max()), tuple(sr.shape))
    np.savez_compressed(os.path.join(HERE, "sr3_pipeline_t50.npz"), lr_seed=np.array(41), torch_seed=np.array(0),
                        final=sr[-1:].numpy(), frames_mean=sr.mean(dim=(1, 2, 3)).numpy(),
                        **{f"x_after_t{t}": v.numpy() for t, v in probes.items()})


@torch.no_grad()
def stage2():
    from gen_s2_golden import build_reference
    import sgm.modules.diffusionmodules.sampling as RS
    ref = build_reference()
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items() if not k.startswith("_lpips")}
    img = seeded.synthetic_image((1, 3, 64, 64), seed=80, smooth=3)
    opt = S.PIPE_OPT
    out = {}
    runs = (("cache50", dict(num_steps=50)), ("nocache50", dict(num_steps=50, img_threshold=0.0)),
            ("ns2_cache", dict(num_samples=2)))
    for tag, over in runs:
        kw = dict(p_p="", n_p="", img_threshold=opt["img_threshold"], dec_img=opt["dec_img"], num_steps=opt["num_steps"],
                  restoration_scale=opt["restoration_scale"], s_churn=opt["s_churn"], s_noise=opt["s_noise"],
                  cfg_scale=opt["cfg_scale"], seed=-1, num_samples=1, control_scale=opt["control_scale"],
                  color_fix_type=opt["color_fix_type"], use_linear_CFG=opt["use_linear_CFG"], use_linear_control_scale=False,
                  cfg_scale_start=opt["cfg_scale_start"], control_scale_start=0.0)
        kw.update(over)
        trace, orig = [], RS.get_can_use_cache_multi

        def spy(first, threshold, parallelized=False):
            use, d = orig(first, threshold=threshold, parallelized=parallelized)
            trace.append((float(threshold), float(d), bool(use)))
            return use, d

        RS.get_can_use_cache_multi = spy
        try:
            torch.manual_seed(7)
            res = ref.just_sampling(img, [""], **kw)
        finally:
            RS.get_can_use_cache_multi = orig
        out[f"{tag}.final"] = res.numpy()
        out[f"{tag}.trace"] = np.array([[a, b, float(h)] for a, b, h in trace], dtype=np.float64).reshape(-1, 3)
        hits = sum(h for _, _, h in trace)
        print(f"{tag}: shape {tuple(res.shape)}, range {float(res.abs().max()):.2f}, {hits} hits / {len(trace)} decisions")
        if over.get("num_samples", 1) == 1:
            otr = []
            torch.manual_seed(7)
            c, uc = S.cond_dicts()
            ores = O.just_sampling(sd, img, c, uc, dict(opt, **{k: v for k, v in over.items()}), trace=otr)
            print(f"   oracle vs reference max|d| = {float((res - ores).abs().max()):.2e}; traces equal: "
                  f"{[h for _, _, h in trace] == [h for _, _, h in otr]}")
    np.savez_compressed(os.path.join(HERE, "s2_pipeline_50.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["s1", "s2"]
    if "s1" in which:
        stage1()
    if "s2" in which:
        stage2()
    print("done")
