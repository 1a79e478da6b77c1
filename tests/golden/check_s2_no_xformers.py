"""Is the Stage-2 pin independent of the ``xformers.ops.memory_efficient_attention`` stand-in of ref_shims.py?
Authoring container only (imports /root/reference):
    python tests/golden/check_s2_no_xformers.py
Runs the REFERENCE with NO xformers module at all, so that it takes its own torch-SDPA branch
(sgm/modules/attention.py:397-402 falls back from "softmax-xformers" to "softmax" = CrossAttention's
F.scaled_dot_product_attention, :273-277; models/modules/SR_modules.py:121 likewise), regenerates ``control.9`` / ``unet.out`` /
``unet.h`` / ``unet.out_cs08`` and the 6-step just_sampling pipeline (cache on and off) and compares them with the COMMITTED goldens
(s2_networks.npz, s2_pipeline.npz), which were produced through the stand-in.  Prints max|d| per tensor; exits non-zero beyond 1e-6
(the two branches are the same mathematics through two tensor layouts: [B*heads, N, d] per-head calls against [B, heads, N, d])."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_shims

ref_shims.NO_XFORMERS = True
AttrDict = ref_shims.install(xformers=False)

import numpy as np
import torch

import s2_common as S
from oracle import seeded

torch.set_num_threads(8)


@torch.no_grad()
def main():
    import gen_s2_golden as G          # (its module-level install() finds the shims already in place and adds none for xformers)
    assert "xformers" not in sys.modules, "the stand-in is installed: this check must run without it"
    ref = G.build_reference()
    import sgm.modules.attention as A
    import models.modules.SR_modules as M
    assert not A.XFORMERS_IS_AVAILABLE and not M.XFORMERS_IS_AVAILBLE
    kinds = {type(m).__name__ for m in ref.model.modules() if "CrossAttention" in type(m).__name__}
    print("attention classes in the reference's Stage-2 networks without xformers:", sorted(kinds))
    assert "MemoryEfficientCrossAttention" not in kinds
    unet, ctrl = ref.model.diffusion_model, ref.model.control_model
    want = np.load(os.path.join(HERE, "s2_networks.npz"))
    t = torch.tensor([999.0, 19.0])
    y, ctx = S.rnd((2, 32), 62), S.rnd((2, 77, 64), 51)
    xt, xc = S.rnd((2, 4, 16, 16), 70), S.rnd((2, 4, 16, 16), 71, 0.5)
    control = ctrl(x=xc, timesteps=t, xt=xt, context=ctx, y=y)
    got = {"control.9": control[9]}
    for i, c in enumerate(control):
        got[f"control.{i}.fp"] = torch.tensor(G.sub(c))
    got["unet.out"] = unet(xt, timesteps=t, context=ctx, y=y, control=control, control_scale=1.0, fbcache_mode="none")
    got["unet.h"] = unet(xt, timesteps=t, context=ctx, y=y, control=control, control_scale=1.0, fbcache_mode="input_stage1")["h"]
    got["unet.out_cs08"] = unet(xt, timesteps=t, context=ctx, y=y, control=control, control_scale=0.8, fbcache_mode="none")
    worst = 0.0
    for k, v in got.items():
        d = float((v - torch.tensor(want[k])).abs().max())
        worst = max(worst, d)
        print(f"  {k:16s} SDPA branch vs committed golden (stand-in): max|d| = {d:.3e}  (range {float(np.abs(want[k]).max()):.2f})")
    # the 6-step pipeline, cache on and off, with the cache decisions
    import sgm.modules.diffusionmodules.sampling as RS
    pipe = np.load(os.path.join(HERE, "s2_pipeline.npz"))
    img = seeded.synthetic_image((1, 3, 64, 64), seed=80, smooth=3)
    trace, orig = [], RS.get_can_use_cache_multi

    def spy(first, threshold, parallelized=False):
        use, d = orig(first, threshold=threshold, parallelized=parallelized)
        trace.append((float(threshold), float(d), bool(use)))
        return use, d

    RS.get_can_use_cache_multi = spy
    opt = S.PIPE_OPT
    for tag, thr in (("cache", opt["img_threshold"]), ("nocache", 0.0)):
        trace.clear()
        torch.manual_seed(7)
        res = ref.just_sampling(img, [""], p_p="", n_p="", img_threshold=thr, dec_img=opt["dec_img"], num_steps=opt["num_steps"],
                                restoration_scale=opt["restoration_scale"], s_churn=opt["s_churn"], s_noise=opt["s_noise"],
                                cfg_scale=opt["cfg_scale"], seed=-1, num_samples=1, control_scale=opt["control_scale"],
                                color_fix_type=opt["color_fix_type"], use_linear_CFG=opt["use_linear_CFG"],
                                use_linear_control_scale=False, cfg_scale_start=opt["cfg_scale_start"], control_scale_start=0.0)
        d = float((res - torch.tensor(pipe[f"{tag}.final"])).abs().max())
        worst = max(worst, d)
        dec = [h for _, _, h in trace] == [bool(r[2]) for r in pipe[f"{tag}.trace"]]
        print(f"  pipeline[{tag}] 6 steps: max|d| = {d:.3e}; cache decisions equal the committed trace: {dec}")
        assert dec
    RS.get_can_use_cache_multi = orig
    print(f"worst max|d| over all tensors = {worst:.3e}")
    assert worst < 1e-6, worst
    print("the Stage-2 goldens do not depend on the xformers stand-in")


if __name__ == "__main__":
    main()
