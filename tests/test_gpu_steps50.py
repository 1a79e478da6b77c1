"""Stage 2 at the step count the metric is quoted on: ``just_sampling`` over 50 EDM steps (cache 0.3 and off) against the
reference's own 50-step CPU runs (tests/golden/gen_steps50_golden.py -> s2_pipeline_50.npz), in the shipped precision
(fp16 UNets / bf16 VAE), with the VAE in fp32, and on the fp32-operand family -- i.e. what the drift through
``x += d * dt`` (sampling.py:618-620) over 50 stochastic steps is for each; and ``num_samples = 2`` WITH the cache on,
where the reference takes ONE decision over the stacked tensor (DFBCache.py:98-112, SR_model.py:231-235).
The 6-step counterparts are tests/test_gpu_s2.py; Stage 1 at T = 50 is tests/test_gpu_sr3.py."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import s2_common as S

pytestmark = pytest.mark.gpu

PREC = {"shipped": ("bf16", "fp16"), "vae32": ("fp32", "fp16"), "allfp32": ("fp32", "fp32"), "split": ("split", "split"),
        "split_full": ("split", "split")}   # split = the mode as shipped (ops.UNET_POLICY: attention operands, to_out and FeedForward inputs in fp16); split_full = ops.ALL_SPLIT (every product three MFMAs, the split attention kernels)
# (max|d|, mean|d|) on an output of range 2.6 = 2 x measured.  Measured on MI355X, cache off / on (6 steps: test_gpu_s2.PIPE_BOUNDS):
#   shipped (bf16 VAE, fp16 UNets)  3.17e-2 / 4.6e-3   3.21e-2 / 4.7e-3   -- the same as after 6 steps: the bf16 VAE passes dominate
#   vae32   (fp32 VAE, fp16 UNets)  6.0e-3  / 6.1e-4   8.3e-3  / 7.6e-4   -- 2 x the 6-step figure: the drift of fp16 over 50 steps
#   allfp32                          1.1e-5  / 1.5e-6   1.9e-5  / 1.8e-6
#   split   (ops.UNET_POLICY: fp32 streams, three-MFMA convolutions, the transformer GEMMs on fp16 inputs AND fp16 weights)  4.5e-4 / 7.0e-5   5.7e-4 / 7.3e-5;
#           the bar is north_star's 1e-3 (NOT 2 x measured); round 4 (attention operands only in fp16): 3.0e-4 / 4.3e-5
#   split_full (attention in the split kernels too)      4.8e-5 / 7.5e-6
BOUNDS50 = {"shipped": (6.5e-2, 9.5e-3), "vae32": (1.7e-2, 1.6e-3), "allfp32": (5e-5, 5e-6), "split": (1e-3, 1e-4), "split_full": (1e-4, 1.5e-5)}


@pytest.fixture(scope="module")
def model(cuda):
    from oracle import seeded
    from rsvld_amd.sgm.util import instantiate_from_config
    m = instantiate_from_config({"target": "rsvld_amd.models.SR_model.SR_backbone", "params": S.product_params()})
    seeded.seed_module(m, S.WEIGHT_SEED)
    return m.to(cuda).eval()


def _run(m, cuda, prec, **over):
    from oracle import seeded
    img = seeded.synthetic_image((1, 3, 64, 64), seed=80, smooth=3).to(cuda)
    opt = dict(S.PIPE_OPT, **over)
    from rsvld_amd import ops
    m.noise_source = "cpu"
    m.set_precision(*PREC[prec], policy=ops.ALL_SPLIT if prec == "split_full" else None)
    try:
        torch.manual_seed(7)
        out = m.just_sampling(img, [""], p_p="", n_p="", **opt)
    finally:
        m.noise_source = "device"
        m.set_precision("bf16", "fp16")
    return out.cpu(), [step[0] for step in m.cache_trace]


def _check_trace(got, want, prec, tag):
    """Every decision the reference took must be taken here; a 16-bit run may only differ where the reference's own
    measurement sat within 2 % of its threshold (an undecidable step at that precision)."""
    assert len(got) == len(want), (len(got), len(want))
    flips = []
    for i, ((thr, diff, hit), w) in enumerate(zip(got, want)):
        if bool(w[2]) != hit:
            flips.append((i, float(w[0]), float(w[1])))
    hits = sum(int(w[2]) for w in want)
    print(f"{tag}: {hits} hits / {len(want)} decisions in the reference; decisions that differ here: {flips}")
    if prec in ("allfp32", "split", "split_full"):
        assert not flips
    return flips


@pytest.mark.parametrize("prec", ["shipped", "vae32", "allfp32", "split", "split_full"])
@pytest.mark.parametrize("tag", ["nocache50", "cache50"])
def test_just_sampling_50_steps_vs_reference_golden(model, cuda, golden_dir, tag, prec):
    z = np.load(os.path.join(golden_dir, "s2_pipeline_50.npz"))
    out, trace = _run(model, cuda, prec, num_steps=50, **({"img_threshold": 0.0} if tag == "nocache50" else {}))
    want = torch.tensor(z[f"{tag}.final"])
    d = (out - want).abs()
    print(f"just_sampling, 50 steps [{tag}, {prec}]: max|d| = {float(d.max()):.3e}, mean|d| = {float(d.mean()):.3e} "
          f"(range {float(want.abs().max()):.2f})")
    flips = []
    if tag == "cache50":
        flips = _check_trace(trace, z[f"{tag}.trace"], prec, f"cache trace [{prec}]")
        for i, thr, diff in flips:      # only undecidable steps may flip in 16 bit
            assert abs(diff - thr) < 2e-2 * max(thr, 1e-6), f"step {i}: decision flipped at diff {diff} vs threshold {thr}"
    if not flips:
        assert float(d.max()) < BOUNDS50[prec][0] and float(d.mean()) < BOUNDS50[prec][1]


@pytest.mark.parametrize("prec", ["shipped", "allfp32"])
def test_num_samples_2_with_cache_takes_one_decision(model, cuda, golden_dir, prec):
    """The reference repeats the ONE image num_samples times and its cache test averages over the whole stacked
    [2 * num_samples, 1280, L/4, L/4] tensor: one (threshold, diff, hit) per step, shared by the samples."""
    z = np.load(os.path.join(golden_dir, "s2_pipeline_50.npz"))
    out, trace = _run(model, cuda, prec, num_samples=2)
    want, wt = torch.tensor(z["ns2_cache.final"]), z["ns2_cache.trace"]
    assert out.shape == want.shape == (2, 3, 64, 64)
    assert all(len(step) == 1 for step in model.cache_trace)              # ONE decision per step, not one per sample
    flips = _check_trace(trace, wt, prec, f"ns2_cache [{prec}]")
    assert not flips
    for (thr, diff, hit), w in zip(trace, wt):
        assert diff is None or abs(diff - w[1]) < (1e-4 if prec == "allfp32" else 2e-2) * max(1.0, w[1])
    d = (out - want).abs()
    print(f"just_sampling[num_samples = 2, cache on, {prec}]: max|d| = {float(d.max()):.3e}, mean|d| = {float(d.mean()):.3e}")
    if prec == "allfp32":
        assert float(d.max()) < 1e-4 and float(d.mean()) < 1e-5
    else:
        assert float(d.max()) < 8e-2 and float(d.mean()) < 1.2e-2
