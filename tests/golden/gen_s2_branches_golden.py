"""Golden vectors for the Stage-2 branches the default call does not take, and for the step / cache-logic fixture
groups G3 / G4 of SURVEY.md 8(c) -- produced by running the REFERENCE itself (CPU, fp32) through the import-only shims
of ref_shims.py.  Authoring container only:
    python tests/golden/gen_s2_branches_golden.py
Writes tests/golden/s2_branches.npz and tests/golden/s2_st_full.npz.

  pipe.restore   just_sampling with restoration_scale = 4 (sampling.py:614-616, the restore pull), cache on
  pipe.lincs     use_linear_control_scale = True, control_scale 1.0 <- control_scale_start 0.3 (sampling.py:608-609)
  pipe.adain     color_fix_type = "AdaIn" (utils/colorfix.py:44-71)
  pipe.ns2       num_samples = 2 (SR_model.py:231-235), cache off
  step.i{0,1,49}.{miss,hit}   RestoreEDMSampler.step of a 50-step schedule with the churn noise injected by seed:
                 a cache MISS on x_in (threshold tiny), then a forced HIT on a different latent x_in2 (threshold huge:
                 the prediction cached by the miss is reused) -> x_next, new threshold
  cache.*        G4: the hit/miss/threshold sequence of RestoreEDMSampler.denoise over a 10-step run on synthetic
                 first-block features (stub denoiser), incl. the threshold replacement after every miss
  st_full        one SpatialTransformer at the FULL juggernautXL size (1280 channels, depth 10, context 2048), 4x4 map
"""
import copy
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_shims

AttrDict = ref_shims.install()

import numpy as np
import torch

import s2_common as S
from gen_s2_golden import build_reference
from oracle import s2_oracle as O
from oracle import seeded

torch.set_num_threads(8)

STEP_OPT = dict(num_steps=50, s_churn=5, s_noise=1.003, restore_cfg=4.0, cfg_scale=7.5, cfg_scale_start=4.0)
CACHE_DIFF_SCALES = [None, 0.9, 0.5, 1.3, 0.2, 0.7, 1.25, 0.1, 2.0, 0.05]   # step k: h_k = h_prevcomputed * (1 + s) -> diff ~ s


def cache_sequence_inputs():
    """Synthetic first-block features for G4: step 0 random; step k = last COMPUTED feature * (1 + s_k)."""
    base = S.rnd((2, 8, 4, 4), 301)
    return base


@torch.no_grad()
def main():
    ref = build_reference()
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items() if not k.startswith("_lpips")}
    img = seeded.synthetic_image((1, 3, 64, 64), seed=80, smooth=3)
    opt = S.PIPE_OPT
    out = {}

    # ---------------------------------------------------------------- (a) just_sampling variants
    import sgm.modules.diffusionmodules.sampling as RS
    variants = {
        "restore": dict(restoration_scale=4.0),
        "lincs": dict(use_linear_control_scale=True, control_scale_start=0.3),
        "adain": dict(color_fix_type="AdaIn"),
        "ns2": dict(num_samples=2, img_threshold=0.0),
    }
    for tag, over in variants.items():
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
        out[f"pipe.{tag}.final"] = res.numpy()
        out[f"pipe.{tag}.trace"] = np.array([[a, b, float(h)] for a, b, h in trace], dtype=np.float64).reshape(-1, 3)
        print(f"pipe.{tag}: shape {tuple(res.shape)}, range {float(res.abs().max()):.2f}, trace "
              f"{[(round(a, 3), round(b, 3), h) for a, b, h in trace]}")

    # ---------------------------------------------------------------- (b) G3: sampler.step at i in {0, 1, 49}
    from models.modules.DFBCache import MyCacheContext, cache_context
    from sgm.util import instantiate_from_config
    sp = ref.sampler_config.params
    sp.num_steps = STEP_OPT["num_steps"]
    sp.guider_config.params.scale_min = STEP_OPT["cfg_scale"]
    sp.guider_config.params.scale = STEP_OPT["cfg_scale_start"]
    sp.restore_cfg, sp.s_churn, sp.s_noise = STEP_OPT["restore_cfg"], STEP_OPT["s_churn"], STEP_OPT["s_noise"]
    sampler = instantiate_from_config(ref.sampler_config)
    _z = S.rnd((1, 4, 8, 8), 201, 0.8)                      # LQ latent (control), x_center and the noisy latent: seeded
    x_center = S.rnd((1, 4, 8, 8), 202, 0.8)
    c_img, uc_img = ref.prepare_condition(_z, [""], "", "", 1)

    def denoiser(inp, sigma, c, *a, **k):
        return ref.denoiser(ref.model, inp, sigma, c, *a, **k)

    z0, s_in, sigmas, num_sigmas, c_img, uc_img = sampler.init_loop(S.rnd((1, 4, 8, 8), 203), c_img, uc=uc_img,
                                                                     num_steps=STEP_OPT["num_steps"])
    out["step.sigmas"] = sigmas.numpy()
    for i in (0, 1, 49):
        x_in = S.rnd((1, 4, 8, 8), 210 + i) * float(sigmas[i])
        x_in2 = x_in + 0.1 * float(sigmas[i]) * S.rnd((1, 4, 8, 8), 230 + i)
        out[f"step.i{i}.x_in"], out[f"step.i{i}.x_in2"] = x_in.numpy(), x_in2.numpy()
        with cache_context(MyCacheContext()):
            torch.manual_seed(1000 + i)
            x_miss, thr_miss = sampler.step(x_in, i, s_in, sigmas, denoiser, c_img, uc_img, x_center=x_center,
                                            control_scale=1.0, threshold=1e-9)          # prev is None -> computes
            torch.manual_seed(2000 + i)     # a DIFFERENT latent: a hit must reuse the prediction cached by the miss above
            x_hit, thr_hit = sampler.step(x_in2, i, s_in, sigmas, denoiser, c_img, uc_img, x_center=x_center,
                                          control_scale=1.0, threshold=1e9)             # any diff < 1e9 -> hit
        out[f"step.i{i}.miss"], out[f"step.i{i}.hit"] = x_miss.numpy(), x_hit.numpy()
        out[f"step.i{i}.thr"] = np.array([thr_miss, thr_hit], dtype=np.float64)
        with cache_context(MyCacheContext()):
            torch.manual_seed(2000 + i)
            x_re, _ = sampler.step(x_in2, i, s_in, sigmas, denoiser, c_img, uc_img, x_center=x_center, control_scale=1.0,
                                   threshold=1e-9)
        print(f"step i={i}: sigma {float(sigmas[i]):.4f} -> {float(sigmas[i + 1]):.4f}; hit vs recompute of the same input differ "
              f"by {float((x_hit - x_re).abs().max()):.3e}; thresholds out {thr_miss}, {thr_hit}")

    # ---------------------------------------------------------------- (c) G4: cache logic over a synthetic 10-step run
    base = cache_sequence_inputs()
    feats, decisions = [], []

    class Stub:
        """denoiser stand-in: '*1' calls return the step's first-block feature, '*2' calls a marker prediction"""

        def __init__(self):
            self.k, self.last_computed = 0, None

        def __call__(self, x, sigma, c, control_scale=1.0, fbcache_mode="none", partial_info=None):
            if fbcache_mode.endswith("1"):
                s = CACHE_DIFF_SCALES[self.k]
                h = base.clone() if s is None else self.last_computed * (1.0 + s)
                feats.append(h.numpy().copy())
                return {"h": h}
            self.last_computed = partial_info["h"]
            return torch.cat([torch.full((1, 4, 8, 8), float(self.k)), torch.full((1, 4, 8, 8), float(self.k) + 0.5)])

    stub = Stub()
    thr = 0.3
    with cache_context(MyCacheContext()):
        for k in range(len(CACHE_DIFF_SCALES)):
            stub.k = k
            den, new_thr = sampler.denoise(torch.zeros(1, 4, 8, 8), stub, torch.ones(1) * sigmas[k], c_img, uc_img,
                                           control_scale=1.0, threshold=thr)
            decisions.append([thr, new_thr, float(den.mean())])
            thr = new_thr
    out["cache.feats"] = np.stack(feats)
    out["cache.decisions"] = np.array(decisions, dtype=np.float64)      # rows: threshold in, threshold out, mean of the returned x0
    print("cache logic (thr_in, thr_out, mean x0):", [(round(a, 4), round(b, 4), round(c, 3)) for a, b, c in decisions])
    np.savez_compressed(os.path.join(HERE, "s2_branches.npz"), **out)

    # ---------------------------------------------------------------- (d) full-size SpatialTransformer
    from sgm.modules.attention import SpatialTransformer
    torch.manual_seed(0)
    st = SpatialTransformer(1280, 20, 64, depth=10, context_dim=2048, use_linear=True, attn_type="softmax",
                            use_checkpoint=False).eval()
    seeded.seed_module(st, 777)
    x = S.rnd((2, 1280, 4, 4), 401)
    ctx = S.rnd((2, 77, 2048), 402)
    y = st(x, ctx)
    ssd = {"st." + k: v for k, v in st.state_dict().items()}
    print("st_full: oracle vs reference", float((y - O.spatial_transformer(ssd, "st", x, ctx)).abs().max()), "range", float(y.abs().max()))
    np.savez_compressed(os.path.join(HERE, "s2_st_full.npz"), y=y.numpy())
    print("done")


if __name__ == "__main__":
    main()
