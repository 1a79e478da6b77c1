"""The 1e-3 of ``north_star`` at FULL network depth and at the metric's step count.

The reference-generated 50-step goldens (tests/golden/s2_pipeline_50.npz) use reduced-depth networks (tests/golden/s2_common.py:
transformer_depth [1, 1, 2], context 64) because the reference's CPU run of the full juggernautXL networks over 50 steps does
not fit a test; what they cannot show is whether the distance of a 16-bit / split precision grows with the depth of the
transformer stacks (model_configs/juggernautXL.yaml:24-64: depth [0, 2, 10], context 2048, adm 2816).  Here the FULL networks
run all 50 EDM steps at latent 64 (512 x 512 input) on the device, once per precision, against the fp32-operand kernel family on
the same device -- that family is pinned to the reference's CPU path at ~1e-5 by the goldens (tests/test_gpu_steps50.py
[allfp32], tests/test_gpu_s2.py [allfp32]) -- with the feature cache off and at the reference's default threshold; and Stage 1
(one network size, models/sr3_model/sr3_modules/unet.py:162-261) runs BASELINE configs[1] (128 -> 512, batch 4) over all T = 50
ancestral steps the same way.  Same seeds, CPU noise order: every precision sees the same draws.

Bars: the tolerance-compliant mode (``split``: what bench.py's headline value is timed in) must stay inside 1e-3 max |delta| per
pixel and take the fp32 family's cache decisions; the reference's GPU policy (fp16 UNets, bf16 VAE) is measured and bounded at
2 x what was measured, as everywhere else in this suite.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

S2_MODES = {"fp32": ("fp32", "fp32"), "split": ("split", "split"), "shipped": ("bf16", "fp16")}


def _s2_run(m, cuda, prec, thr):
    import bench
    img = bench.synthetic_image((1, 3, 512, 512), seed=4321, smooth=4).to(cuda)
    m.noise_source = "cpu"
    m.set_precision(*S2_MODES[prec])
    try:
        torch.manual_seed(7)
        out = m.just_sampling(img, [""], **dict(bench.S2_KW, img_threshold=thr, num_steps=50))
        return out.cpu(), [step[0] for step in m.cache_trace]
    finally:
        m.noise_source = "device"
        m.set_precision("bf16", "fp16")


@pytest.fixture(scope="module")
def s2_runs(cuda, full_model):
    """All six 50-step runs (3 precisions x cache off / 0.3) of the full-size Stage 2 at latent 64, computed once."""
    import hashlib
    res = {}
    wsum = sum(float(p_.detach().double().sum()) for p_ in full_model.parameters())
    print(f"   weights fingerprint: {wsum!r}")
    for thr in (0.0, 0.3):
        for prec in S2_MODES:
            res[prec, thr] = _s2_run(full_model, cuda, prec, thr)
            # (fingerprints: the same tree must print the same digests whatever ran earlier in the process)
            print(f"   digest [{prec}, cache {thr}]: {hashlib.sha1(res[prec, thr][0].numpy().tobytes()).hexdigest()[:12]}")
    return res


@pytest.mark.parametrize("thr", [0.0, 0.3])
def test_stage2_full_depth_50_steps_split_inside_1e3(s2_runs, thr):
    want, wtrace = s2_runs["fp32", thr]
    got, trace = s2_runs["split", thr]
    d = (got - want).abs()
    rng = float(want.abs().max())
    print(f"Stage 2, FULL depth, 50 steps at latent 64, cache {thr}: split vs fp32 family max|d| = {float(d.max()):.3e}, "
          f"mean|d| = {float(d.mean()):.3e} (range {rng:.2f})")
    assert bool(torch.isfinite(want).all()) and rng > 0.5
    assert float(d.max()) < 1e-3 and float(d.mean()) < 1.5e-4
    if thr > 0:
        hits = [bool(t[2]) for t in wtrace]
        assert [bool(t[2]) for t in trace] == hits, "the split mode took a different cache decision than the fp32 family"
        print(f"   cache decisions equal ({sum(hits)} hits / {len(hits)})")


@pytest.mark.parametrize("thr", [0.0, 0.3])
def test_stage2_full_depth_50_steps_reference_gpu_policy(s2_runs, thr):
    """fp16 UNets / bf16 VAE (the reference's own GPU policy, SR_model.py:28-33, wrappers.py:90): measured, bounded at 2 x."""
    want, wtrace = s2_runs["fp32", thr]
    got, trace = s2_runs["shipped", thr]
    d = (got - want).abs()
    flips = [i for i, (a, b) in enumerate(zip(trace, wtrace)) if bool(a[2]) != bool(b[2])]
    print(f"Stage 2, FULL depth, 50 steps at latent 64, cache {thr}: fp16 / bf16 vs fp32 family max|d| = {float(d.max()):.3e}, "
          f"mean|d| = {float(d.mean()):.3e}; cache decisions that differ: {flips}")
    assert bool(torch.isfinite(got).all())
    if not flips:
        assert float(d.max()) < S2_SHIPPED_BOUND[0] and float(d.mean()) < S2_SHIPPED_BOUND[1]


# measured on MI355X (round 5): see DESIGN.md section 4; bounds = 2 x measured
S2_SHIPPED_BOUND = (1.2e-1, 1.5e-2)


@pytest.fixture(scope="module")
def s1_runs(cuda):
    """BASELINE configs[1] (128 -> 512 x4, batch 4) over all T = 50 ancestral steps in the four Stage-1 precisions."""
    import bench
    net, _ = bench.build_stage1(50)
    cond = bench.stage1_input([0, 1, 2, 3], 128, 4).to(cuda)
    net.noise_source = "cpu"
    net.use_graph = False
    res = {}
    for prec in ("fp32", "split", "w2", "fp16"):
        net.denoise_fn.set_compute_dtype(prec)
        torch.manual_seed(0)
        res[prec] = net.super_resolution(cond, continous=True)[-4:].cpu()
    return res


def test_stage1_config1_T50_batch4_split_inside_1e3(s1_runs):
    want = s1_runs["fp32"]
    assert want.shape == (4, 3, 512, 512) and bool(torch.isfinite(want).all())
    d = (s1_runs["split"] - want).abs()
    print(f"Stage 1, configs[1] (512^2, batch 4), T = 50: split vs fp32 family max|d| = {float(d.max()):.3e}, mean|d| = {float(d.mean()):.3e}")
    assert float(d.max()) < 1e-3 and float(d.mean()) < 1e-4
    # "w2" = what bench.py's headline runs Stage 1 in: fp16 tensors, every weight as the fp16 pair [W_lo | W_hi] (two MFMAs per product)
    dw = (s1_runs["w2"] - want).abs()
    print(f"Stage 1, configs[1] (512^2, batch 4), T = 50: w2 (fp16 x weight pairs) vs fp32 family max|d| = {float(dw.max()):.3e}, mean|d| = {float(dw.mean()):.3e}")
    assert float(dw.max()) < 1e-3 and float(dw.mean()) < 1e-4
    d16 = (s1_runs["fp16"] - want).abs()
    print(f"Stage 1, configs[1] (512^2, batch 4), T = 50: fp16 vs fp32 family max|d| = {float(d16.max()):.3e}, mean|d| = {float(d16.mean()):.3e}")
    assert float(d16.max()) < 1e-2 and float(d16.mean()) < 1e-3


@pytest.mark.parametrize("thr", [0.0, 0.3])
def test_stage2_full_depth_50_steps_latent128_inside_1e3(cuda, full_model, thr):
    """The same measurement one size up: 1024^2 input = latent 128 (16 384 tokens at level 0, 4 x the keys per softmax row of the latent-64
    runs above), tiled VAE, all 50 EDM steps, tolerance composition vs the fp32-operand family.  Together with the truncated runs at latent
    256 / 512 (tools/tolerance_at_scale.py, profiles/r06_tolerance_at_scale.txt) this is the growth law in the token count: the distance
    does not grow with it (tools/tolerance_at_scale.py on the final composition, untiled VAE: 8.2e-4 / 3.9e-5 cache off, 5.8e-4 / 4.6e-5 at 0.3, all 50
    decisions equal; the maximum over 3 M pixels scatters between runs of different summation order, the mean is the stable figure)."""
    import bench
    img = bench.synthetic_image((1, 3, 1024, 1024), seed=4321, smooth=4).to(cuda)
    m = full_model
    res = {}
    for prec in ("fp32", "split"):
        m.noise_source = "cpu"
        m.set_precision(*S2_MODES[prec])
        try:
            torch.manual_seed(7)
            out = m.just_sampling(img, [""], **dict(bench.S2_KW, img_threshold=thr, num_steps=50))
            res[prec] = (out.cpu(), [bool(step[0][2]) for step in m.cache_trace])
        finally:
            m.noise_source = "device"
            m.set_precision("bf16", "fp16")
    (want, wtrace), (got, trace) = res["fp32"], res["split"]
    d = (got - want).abs()
    print(f"Stage 2, FULL depth, 50 steps at latent 128, cache {thr}: split vs fp32 family max|d| = {float(d.max()):.3e}, "
          f"mean|d| = {float(d.mean()):.3e} (range {float(want.abs().max()):.2f}); cache hits {sum(wtrace)} / {len(wtrace)}")
    assert bool(torch.isfinite(want).all()) and float(d.max()) < 1e-3 and float(d.mean()) < 1.5e-4
    assert trace == wtrace, "the split mode took a different cache decision than the fp32 family"
