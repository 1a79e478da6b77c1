#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Precision of the headline (``--precision tolerance``, the default of c4 / c4full): the composition whose distance from the
reference's CPU path after 50 + 50 steps is inside north_star's 1e-3 -- measured in the same run against the reference-generated
goldens and printed as ``config.tolerance``; the reference's own GPU policy (fp16 UNets, bf16 VAE: what rounds 1-4 quoted as
``value``) is timed beside it, outside ``value``, as ``config.reference_gpu_policy``.

Default workload = the configuration BASELINE.json's metric is quoted on (``--workload c4``):
512x512 -> 4096x4096 x8 SR, Stage 1 (SR3, 50 ancestral DDPM steps at 4096^2) -> 8-bit hand-off ->
Stage 2 (50 EDM steps at latent 512, ControlNet, tiled VAE 512/64, Wavelet colour fix), one image
per GPU per pass, synthetic inputs, seeded random-init weights of the shipped architectures.

A whole image is ~70 s of GPU work, so one bench "step" is ONE sampler iteration of EACH stage at
the real shapes (the work of every iteration is identical: the feature cache is OFF for the headline,
because its hit rate on random weights means nothing).  After W untimed iterations of each stage the
timed region runs, bracketed by barrier + synchronize: the per-image fixed part (2 tiled encodes +
2 tiled decodes, hand-off, conditioner, colour fix, uint8 all-gather) ONCE and exactly K iterations
of each stage.  With the phase times taken at device-synchronised stamps inside that region
(max over ranks each):

    value = n_gpus / (50 * t_S1_iter + 50 * t_S2_iter + t_fixed)          [images / s, whole job]

After the timed region (outside ``value``) the same process EXECUTES one whole image -- all 50 + 50 iterations, warm -- and prints it beside
the projection (``config.whole_image_executed``: 76.56 s executed vs 76.48 s projected on the round-6 tree).  ``--workload c4full`` times
complete images instead (one image per step).  Other workloads: ``c2`` (BASELINE configs[1],
Stage 1 only 128->512 batch 4), ``c3`` (configs[2], Stage 2 at 2048^2 batch 8 with the per-image
feature cache), ``s2`` (Stage 2 only, any size).

The JSON line also carries
  roofline     : the dominant kernel of the workload: algorithmic FLOPs of its launches / their summed
                 HIP-event durations (events on the launch stream, one extra instrumented pass), and the
                 HBM-side bytes per launch from the committed rocprofv3 PMC passes of this same workload;
  cpu_baseline : the CPU oracle (oracle/, a port of the reference's CPU path) timed on this host's
                 cores on a bounded sample and extrapolated as stated in ``sample``.
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

METRIC = "512px x8 SR images/sec @50 steps"              # BASELINE.json metric (the c4 / c4full workloads only)
PEAK_TFLOPS_F16 = 2500.0   # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
S1_TF_PER_IMAGE_STEP = {256: 0.262, 512: 1.126, 1024: 5.77, 2048: 43.3, 4096: 496.3}   # BASELINE.md section 2 / SURVEY 8(d)
S2_TF_PER_IMAGE_STEP = {64: 4.76, 128: 20.3, 256: 107.8, 512: 865.9}                  # by latent side, CFG pair + ControlNet
PKG = os.path.join(ROOT, "remote-sensing-vision-language-diffusion-model_amd")

S2_KW = dict(p_p="", n_p="", dec_img=1.0, restoration_scale=-1, s_churn=5, s_noise=1.003, cfg_scale=7.5, control_scale=1.0,
             color_fix_type="Wavelet", use_linear_CFG=True, cfg_scale_start=4.0)     # infer.py:44-62 defaults


# --------------------------------------------------------------------------------------------- synthetic inputs / models
def synthetic_image(shape, seed, smooth=4):
    """Image-like fp32 tensor in [-1, 1]: seeded uniform noise, box low-pass, min-max normalised."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(shape, generator=g)
    if smooth > 1:
        x = torch.nn.functional.avg_pool2d(x, smooth, stride=1, padding=smooth // 2)[..., :shape[-2], :shape[-1]]
    lo, hi = x.amin(dim=(-3, -2, -1), keepdim=True), x.amax(dim=(-3, -2, -1), keepdim=True)
    return ((x - lo) / (hi - lo) * 2 - 1).contiguous()


def stage1_input(image_ids, lr_side, scale):
    """LR images -> the Stage-1 input contract (data/dataset.py:16-21,30-42): bicubic x scale, [-1,1]."""
    lr = torch.cat([synthetic_image((1, 3, lr_side, lr_side), seed=1234 + i, smooth=4) for i in image_ids], 0)
    return torch.nn.functional.interpolate(lr, scale_factor=scale, mode="bicubic", align_corners=False).clamp(-1, 1)


# --precision -> (Stage-1 compute dtype, VAE dtype, UNet + ControlNet dtype).
#   tolerance      the composition that meets north_star's |delta| < 1e-3 against the reference's CPU path after 50 + 50 steps (measured in
#                  the same run, config.tolerance): Stage 1 = fp16 tensors x fp16 weight PAIRS (two MFMAs per product: the weights are
#                  the whole distance of plain fp16); Stage 2 = fp32 residual streams, convolution / proj operands as bf16 hi + lo
#                  (three MFMAs), attention operands and the to_out / FeedForward / q|k|v inputs in fp16 x weight pairs
#                  (rsvld_amd.ops.UNET_POLICY), the VAE all three-MFMA.  The headline workload's default.
#   reference-gpu  the reference's own GPU policy (fp16 UNets under autocast, bf16 VAE: SR_model.py:28-33, wrappers.py:90); 2e-3 / 3e-2
#                  from its CPU path.  "default" is the same (the secondary workloads and the tests' full_model fixture use it).
#   split, fp32, vae-split: the other modes of rounds 3-4, secondary measurements.
PRECISIONS = {"tolerance": ("w2", "split", "split"), "reference-gpu": ("fp16", "bf16", "fp16"), "default": ("fp16", "bf16", "fp16"),
              "split": ("split", "split", "split"), "fp32": ("fp32", "fp32", "fp32"), "vae-split": ("fp16", "split", "fp16")}
PRECISION = "default"     # --precision (module default = what importing tests get: the reference's GPU policy)
UNET_POLICY = None        # --unet-f16-groups / --unet-f16-weights (experiments): the ops.SplitPolicy handed to set_precision (None = its default)


def apply_precision(net, m, name):
    """Switch both stages of a built pipeline (packed weights of every type are kept: switching back and forth packs each once)."""
    s1, ae, df = PRECISIONS[name]
    if net is not None:
        net.denoise_fn.set_compute_dtype(s1)
    if m is not None:
        m.set_precision(ae, df, policy=UNET_POLICY if df == "split" else None)


def precision_report(net, m):
    """What the two stages actually run: dtype names + the SplitPolicy keys (the bench line names the composition it timed)."""
    from rsvld_amd import ops
    unet = net.denoise_fn
    return {"stage1": {"activations": str(unet.compute_dtype).replace("torch.", ""), "weights_packed": str(unet.pack_dtype).replace("torch.", ""),
                       "weight_form": "fp16 pairs [W_lo | W_hi], 2 MFMAs per product" if (unet.compute_dtype == torch.float16 and unet.pack_dtype == torch.float32)
                       else ("bf16 triples, 3 MFMAs per product" if unet.split is not None else "plain"),
                       "policy": None if unet.split is None else unet.split.describe()},
            "stage2": {"vae": m._precision_names[0], "unet_controlnet": m._precision_names[1],
                       "vae_policy": None if m.first_stage_model.split is None else m.first_stage_model.split.describe(),
                       "unet_policy": None if m.model.split is None else m.model.split.describe()}}


def build_stage1(T):
    """The shipped Stage-1 option file (configs/sr_sr3.json) through the product's own factory
    (sr3_model.create_model, infer.py:96-103), seeded default init, ``T``-step 'val' schedule."""
    from rsvld_amd.configs import sr3 as SR3
    from rsvld_amd.sr3_model import create_model
    from rsvld_amd.utils import logger as Logger
    opt = Logger.parse(SR3.SR3_Config(), allow_random_init=True)
    opt["path"]["resume_state"] = None          # no checkpoints offline
    torch.manual_seed(0)
    model = create_model(opt)
    sched = dict(opt["model"]["beta_schedule"]["val"], n_timestep=T)
    model.set_new_noise_schedule(sched, schedule_phase="val")
    apply_precision(model.netG, None, PRECISION)
    return model.netG, opt


def stage2_params(live_conditioner_on=None):
    """model.params of the shipped yaml.  Conditioner: the reference's cached-embedding class (PreparedConditioner, seeded
    embeddings), or -- ``live_conditioner_on=device`` -- the yaml's own GeneralConditionerWithControl over full-size seeded
    CLIP-L / OpenCLIP-bigG text towers (tools/synthetic_models.py: no checkpoints offline)."""
    import yaml
    cfg = yaml.safe_load(open(os.path.join(PKG, "model_configs", "juggernautXL.yaml")))["model"]["params"]
    if live_conditioner_on is not None:
        from tools import synthetic_models as SM
        cfg["conditioner_config"] = SM.live_conditioner_config(live_conditioner_on)
        return cfg
    g2, g3 = torch.Generator().manual_seed(2), torch.Generator().manual_seed(3)
    cfg["conditioner_config"] = {"target": "rsvld_amd.sgm.modules.PreparedConditioner", "params": {
        "cond_pth": {"crossattn": torch.randn(1, 77, 2048, generator=g2), "vector": torch.randn(1, 2816, generator=g2)},
        "un_cond_pth": {"crossattn": torch.randn(1, 77, 2048, generator=g3), "vector": torch.randn(1, 2816, generator=g3)}}}
    return cfg


@contextlib.contextmanager
def _host_init_skipped():
    """nn.Linear / nn.Conv2d constructors draw their default init on the HOST: 3.9 B values = 30 s of one rank's build, 71 s when two
    ranks share the cores (profiles/r03_bench_c4_2ranks_gloo.json), minutes at eight.  The bench seeds on the DEVICE instead
    (``_seed_stage2_on_device``), so the host draws are switched off while the modules are constructed; ``zero_module`` calls are
    recorded (their tensors are re-drawn: an all-zero projection would make the network's output 0)."""
    import importlib
    import pkgutil
    import torch.nn.init as init
    import rsvld_amd
    # Every module of the package that defines or imports ``zero_module`` must exist BEFORE the wrappers go on: the yaml's target strings
    # import some of them lazily (openaimodel, SR_modules), and a module first imported inside this block binds whatever ``zero_module``
    # its source module holds at that moment -- round 4's first build in a process left the UNet's zero-initialised convolutions
    # untagged (drawn uniform), every later build tagged them (normal): found in round 5 by a weights fingerprint that depended on
    # which test had built a model before.
    for info in pkgutil.walk_packages(rsvld_amd.__path__, "rsvld_amd."):
        spec = importlib.util.find_spec(info.name)
        if spec is not None and spec.origin and spec.origin.endswith(".py") and info.name not in sys.modules:
            try:
                with open(spec.origin) as fh:
                    if "zero_module" in fh.read():
                        importlib.import_module(info.name)
            except OSError:
                pass
    skipped = ("uniform_", "normal_", "kaiming_uniform_", "kaiming_normal_", "xavier_uniform_", "xavier_normal_", "trunc_normal_")
    saved = {n: getattr(init, n) for n in skipped}
    zsaved = []
    try:
        for n in skipped:
            setattr(init, n, lambda t, *a, **k: t)
        for name, mod in list(sys.modules.items()):
            if name.startswith("rsvld_amd") and callable(getattr(mod, "zero_module", None)) and not hasattr(mod.zero_module, "_orig"):
                orig = mod.zero_module

                def tagging(module, _orig=orig):
                    for p_ in module.parameters():
                        p_._zero_init = True
                    return _orig(module)
                tagging._orig = orig
                zsaved.append((mod, orig))
                mod.zero_module = tagging
        yield
    finally:
        for n, f in saved.items():
            setattr(init, n, f)
        for mod, orig in zsaved:
            mod.zero_module = orig


def _seed_stage2_on_device(m, dev, seed=1):
    """torch's default init of every Linear / Conv2d (weight and bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in))) drawn with a seeded
    DEVICE generator; ``zero_module`` weights ~ N(0, 0.02) (biases 0), as the host version of rounds 1-3 re-drew them.  Norm gains /
    offsets and directly constructed parameters keep their host values.  Same seed on every rank -> identical replicas."""
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        for mod in m.modules():
            if not isinstance(mod, (torch.nn.Linear, torch.nn.Conv2d)) or mod.weight.device.type != "cuda":
                continue
            w = mod.weight
            bound = 1.0 / (w[0].numel() ** 0.5)
            if getattr(w, "_zero_init", False):
                w.normal_(0.0, 0.02, generator=g)
            else:
                w.uniform_(-bound, bound, generator=g)
            if mod.bias is not None:
                if getattr(mod.bias, "_zero_init", False):
                    mod.bias.zero_()
                else:
                    mod.bias.uniform_(-bound, bound, generator=g)


def build_stage2(dev, tile_vae, live_conditioner=False):
    """Full juggernautXL.yaml networks (UNet 2.6 B + ControlNet 1.2 B params, seeded random init ON THE DEVICE with the
    zero-initialised tensors re-drawn); cached text embeddings (the reference's PreparedConditioner) or the live conditioner."""
    import rsvld_amd.models.SR_model  # noqa: F401  (the modules that define zero_module must be imported before they are wrapped)
    from rsvld_amd.sgm.util import instantiate_from_config
    torch.manual_seed(0)
    params = stage2_params(dev if live_conditioner else None)     # (the live text towers seed themselves on the device)
    with _host_init_skipped():
        m = instantiate_from_config({"target": "rsvld_amd.models.SR_model.SR_backbone", "params": params})
    live_ids = set()
    if live_conditioner:      # already seeded on the device by tools/synthetic_models.py: leave them alone
        live_ids = {id(x) for x in m.conditioner.modules()}
    m.to(dev).eval()
    keep = [mod for mod in m.modules() if id(mod) not in live_ids]
    _seed_stage2_on_device(torch.nn.ModuleList([mod for mod in keep if isinstance(mod, (torch.nn.Linear, torch.nn.Conv2d))]), dev)
    apply_precision(None, m, PRECISION)   # ("vae-split" = model_configs/juggernautXL_vae_split.yaml: the four VAE passes in the split precision only)
    if tile_vae:   # SR_model.py:95-125: encoder tiles of 512 px, decoder tiles of 64 latent px, cross-tile GroupNorm
        m.init_tile_vae(512, 64)
    return m


# --------------------------------------------------------------------------------------------- helpers
def schedulable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


class Phases:
    """Device-synchronised wall-clock stamps: ``stamp(name)`` closes the phase ``name``."""

    def __init__(self):
        self.t, self.acc = None, {}

    def start(self):
        torch.cuda.synchronize()
        self.t = time.perf_counter()

    def __call__(self, name):
        torch.cuda.synchronize()
        now = time.perf_counter()
        self.acc[name] = self.acc.get(name, 0.0) + now - self.t
        self.t = now


def dist_max(vals, dev, world):
    if world == 1:
        return [float(v) for v in vals]
    on_host = torch.distributed.get_backend() == "gloo"
    t = torch.tensor(vals, device="cpu" if on_host else dev, dtype=torch.float64)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return [float(v) for v in t.tolist()]


def dist_gather_rows(vals, dev, world):
    """Every rank's row of figures on every rank (one small all-gather outside the timed region) -> [world][len(vals)]."""
    if world == 1:
        return [[float(v) for v in vals]]
    on_host = torch.distributed.get_backend() == "gloo"
    t = torch.tensor(vals, device="cpu" if on_host else dev, dtype=torch.float64)
    out = torch.empty(world * len(vals), device=t.device, dtype=torch.float64)    # (the flat, concatenated form: the one every backend takes)
    torch.distributed.all_gather_into_tensor(out, t)
    return [[float(v) for v in row] for row in out.view(world, len(vals)).tolist()]


def barrier(world):
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()


def kernel_table(summ):
    whole = all("ms_whole_image" in r for r in summ.values())
    # (tflops = algorithmic FLOPs of the product; mfma_tflops = what the matrix pipe executes: x 2 in the weight-pair form, x 3 in the split precision)
    # (_q8: two fp16 MFMAs + one e4m3 MFMA over twice the K per 32 channels and tap = 2 sixteen-bit equivalents at the e4m3 form's nominal 2x rate)
    seg = lambda k: 2 if k.split(" [")[0].endswith(("_w2", "_q8")) else (3 if "_split" in k.split(" [")[0] else 1)
    return {k: dict({"ms": round(v["ms"], 2), "n": v["n"],
                     "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None,
                     "mfma_tflops": round(seg(k) * v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None,
                     "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] and not v["flops"] else None},
                    **({"s_per_image": round(v["ms_whole_image"] * 1e-3, 2)} if whole else {}))
            for k, v in sorted(summ.items(), key=lambda kv: -(kv[1]["ms_whole_image"] if whole else kv[1]["ms"]))}


# LaunchProfiler group -> kernel name in the PMC summary.  attention_d64: the headline's self-attention launches (99 % of the
# group's time) run the ping-pong kernel attn_d64c; the short cross-attention launches (group attention_d64_cross) stay on
# attn_d64b (row "attn_d64")
PMC_ALIAS = {"attention_d64": ("attn_d64c", "attn_d64"), "attention_d64_cross": ("attn_d64",), "attention_d512": ("attn_d512",),
             "gemm_256x256": ("gemm256",),
             # the split-operand pass (profiles/r04_c4_split_pmc_traffic.json): the RSVLD_SPLIT instantiations keep their kernels' names
             "attention_split_d64": ("attn_split_d64",), "attention_split_d64_cross": ("attn_split_d64",),
             "attention_split_d512": ("attn_split_d512",), "gemm_256x256_split": ("gemm256",), "conv_halo_128_split": ("conv_halo_128",),
             "conv_halo_64_split": ("conv_halo_64",), "conv_igemm_split": ("conv_igemm_64x128",), "groupnorm_apply_split": ("gn_apply_split",),
             "groupnorm_stats_split": ("gn_partial_f32",), "layernorm_split": ("layernorm_split",), "split_planes": ("split_planes",),
             # the weight-pair form (round 5): the RSVLD_F16W2 instantiations of the same kernels
             "gemm_256x256_w1": ("gemm256_w1", "gemm256"),
             "gemm_256x256_w2": ("gemm256_w2", "gemm256"), "conv_halo_128_w2": ("conv_halo_128_w2", "conv_halo_128"),
             "conv_halo_64_w2": ("conv_halo_64_w2", "conv_halo_64"), "conv_igemm_64x128_w2": ("conv_igemm_64x128",),
             "conv_igemm_128x128_w2": ("conv_igemm_128x128",),
             # round 6: the e4m3 cross-term form of the ResBlock convolutions and its GroupNorm apply pass
             "conv_halo_128_q8": ("conv_halo_128_q8", "conv_halo_128"), "groupnorm_apply_q8": ("gn_apply_split",)}


def _pmc_row(kern, name):
    if name in kern:
        return kern[name]
    for alias in PMC_ALIAS.get(name, ()):
        if alias in kern:
            return kern[alias]
    return None


def mfma_busy_of_pass(summ, pmc_file):
    """Time-weighted MFMA-busy fraction of the instrumented pass: every kernel group's busy fraction from the committed PMC
    passes of this workload (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)) weighted by its HIP-event time
    in THIS run; kernels without matrix work (or without a PMC entry) count as 0."""
    path = os.path.join(ROOT, "profiles", pmc_file)
    if not os.path.exists(path):
        return None
    kern = json.load(open(path)).get("kernels", {})
    key = "ms_whole_image" if all("ms_whole_image" in r for r in summ.values()) else "ms"    # (headline: weighted over a whole image)
    tot = sum(r[key] for r in summ.values())
    acc = 0.0
    for name, r in summ.items():
        k = _pmc_row(kern, name) or {}
        acc += r[key] * float(k.get("mfma_busy_frac_est") or 0.0)
    return round(acc / tot, 3) if tot > 0 else None


def roofline_of(summ, pmc_file):
    """The dominant kernel = the group with the largest share of a WHOLE image's time (``ms_whole_image``, where the caller measured
    it: the headline) or of the instrumented pass."""
    whole = all("ms_whole_image" in r for r in summ.values())
    dom = max(summ.values(), key=lambda r: r["ms_whole_image"] if whole else r["ms"])
    tf = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    traffic, src, busy = None, None, None
    path = os.path.join(ROOT, "profiles", pmc_file)
    if os.path.exists(path):
        kern = json.load(open(path)).get("kernels", {})
        k = _pmc_row(kern, dom["name"])
        if k is not None:
            traffic, src = round(k["hbm_bytes_per_launch"]), f"profiles/{pmc_file}"
            busy = None if k.get("mfma_busy_frac_est") is None else round(k["mfma_busy_frac_est"], 3)
    return {"bound": "mfma", "kernel": dom["name"], "achieved": round(tf, 2), "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s",
            "frac": round(tf / PEAK_TFLOPS_F16, 4), "traffic": traffic, "traffic_source": src, "mfma_busy_pmc": busy,
            "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["n"]),
            "algorithmic_flops_per_launch": round(dom["flops"] / dom["n"]),
            "launches": dom["n"], "avg_launch_us": round(dom["ms"] * 1e3 / dom["n"], 2),
            "kernel_time_share": round(dom["ms"] / sum(r["ms"] for r in summ.values()), 3),
            "kernel_share_of_whole_image": (round(dom["ms_whole_image"] / sum(r["ms_whole_image"] for r in summ.values()), 3) if whole else None),
            "by_kernel": kernel_table(summ)}


# --------------------------------------------------------------------------------------------- CPU baseline
def _time_threads(fn, counts):
    """Run ``fn`` once per thread count; -> (best seconds, best count, {count: seconds})."""
    res = {}
    for n in counts:
        torch.set_num_threads(n)
        t0 = time.perf_counter()
        fn()
        res[n] = time.perf_counter() - t0
    best = min(res, key=res.get)
    torch.set_num_threads(best)
    return res[best], best, res


def cpu_baseline_s1(side, T, steps_sampled=2, sweep=True):
    """The CPU oracle (oracle/sr3_oracle.py) on ancestral steps of ONE image at ``side``; -> (s/step, threads, note).
    ``sweep``: time one step at each candidate thread count first and keep the fastest (else: the current thread count)."""
    from oracle import sr3_oracle as O
    from rsvld_amd.sr3_model.sr3_modules.unet import UNet
    avail = schedulable_cores()
    c = O.SR3_CFG
    torch.manual_seed(0)
    unet = UNet(in_channel=c["in_channel"], out_channel=c["out_channel"], inner_channel=c["inner_channel"],
                norm_groups=c["norm_groups"], channel_mults=c["channel_mults"], attn_res=list(c["attn_res"]),
                res_blocks=c["res_blocks"], dropout=0.2, image_size=c["image_size"])
    sd = {"denoise_fn." + k: v.detach() for k, v in unet.state_dict().items()}
    sch = O.schedule(dict(schedule="linear", n_timestep=T, linear_start=1e-6, linear_end=1e-2))
    cond = synthetic_image((1, 3, side, side), seed=1, smooth=4)
    x = torch.randn(1, 3, side, side)
    with torch.no_grad():
        def one():
            O.p_sample(sd, c, sch, x, T - 1, cond, torch.randn_like(x))
        best, sweep_res = torch.get_num_threads(), {}
        if sweep:
            one()                                                 # warm-up (thread pool, allocator)
            cands = sorted({min(avail, n) for n in (16, 32, 64)})  # (all 256 hardware threads of a GPU box: 30x slower, measured)
            _, best, sweep_res = _time_threads(one, cands)        # which thread count is fastest on this host
        dt = None
        if steps_sampled > 0:
            t0 = time.perf_counter()
            for i in range(steps_sampled):
                O.p_sample(sd, c, sch, x, T - 1 - i, cond, torch.randn_like(x))
            dt = (time.perf_counter() - t0) / steps_sampled
    return dt, best, {str(k): round(v, 2) for k, v in sweep_res.items()}


def cpu_baseline_s2(m_state, params, latent, threads, steps_sampled=1):
    """The CPU oracle (oracle/s2_oracle.py) on un-cached EDM sampler steps of ONE image at latent side ``latent``
    with the FULL-size networks (the fp32 state dict of the benchmarked model); -> s/step."""
    from oracle import s2_oracle as O
    torch.set_num_threads(threads)
    cp = params["conditioner_config"]["params"]
    z = torch.randn(1, 4, latent, latent)
    c = {"crossattn": cp["cond_pth"]["crossattn"], "vector": cp["cond_pth"]["vector"], "control": z}
    uc = {"crossattn": cp["un_cond_pth"]["crossattn"], "vector": cp["un_cond_pth"]["vector"], "control": z}
    table = O.legacy_ddpm_sigmas(1000, append_zero=False, flip=True)
    sigmas = O.legacy_ddpm_sigmas(50)
    sopt = dict(s_churn=5, s_noise=1.003, restore_cfg=-1, scale=4.0, scale_min=7.5, control_scale=1.0)
    x = torch.randn(1, 4, latent, latent) * float(sigmas[0])
    with torch.no_grad():
        t0 = time.perf_counter()
        for i in range(steps_sampled):
            x, _ = O.restore_edm_step(m_state, table, O.Cache(), x, i, sigmas, c, uc, x, sopt, 0.0, torch.randn)
        dt = (time.perf_counter() - t0) / steps_sampled
    return dt


def cpu_baseline_headline(m, params, T, side, latent):
    """Both stages on the host cores: the CPU oracle timed at the LARGEST size a bounded
    sample allows -- 2 Stage-1 ancestral steps at 1024^2 (5.77 TFLOP each) and 1 un-cached Stage-2 step with the full-size
    networks at latent 128 (20.3 TFLOP) -- and reported two ways: extrapolated in STEPS ONLY at that size (``at_sample_size``),
    and scaled to the headline shape by the algorithmic-FLOP ratios of BASELINE.md section 2 (``value``; the ratio under-states the
    CPU time: the reference's Stage-1 attention materialises an N^2 score tensor, unet.py:133-141, whose share grows with size)."""
    avail = schedulable_cores()
    s1_side, s2_lat = 1024, 128
    _, thr, sweep = cpu_baseline_s1(512, T, steps_sampled=0)             # which thread count is fastest here (cheap: 512^2)
    torch.set_num_threads(thr)
    dt1, _, _ = cpu_baseline_s1(s1_side, T, steps_sampled=2, sweep=False)
    state = {k: v.detach().to("cpu", torch.float32) for k, v in m.state_dict().items()}
    dt2 = cpu_baseline_s2(state, params, s2_lat, thr, steps_sampled=1)
    del state
    s1_full = dt1 * S1_TF_PER_IMAGE_STEP[side] / S1_TF_PER_IMAGE_STEP[s1_side]
    s2_full = dt2 * S2_TF_PER_IMAGE_STEP[latent] / S2_TF_PER_IMAGE_STEP[s2_lat]
    per_image = T * (s1_full + s2_full)
    return {"value": 1.0 / per_image, "unit": "img/s", "cores": thr, "kind": "port",
            "sample": f"fp32 CPU oracle: 2 Stage-1 ancestral steps of 1 image at {s1_side}^2 ({dt1:.2f} s/step) and 1 un-cached "
                      f"Stage-2 EDM step (ControlNet + UNet, CFG pair, full-size networks) at latent {s2_lat} ({dt2:.2f} s/step), "
                      f"{thr} threads = the fastest of the sweep {sweep} s/step (512^2 Stage-1 steps) on {avail} schedulable cores; "
                      f"`value` scales them by the algorithmic-FLOP ratios {side}^2/{s1_side}^2 = "
                      f"{S1_TF_PER_IMAGE_STEP[side] / S1_TF_PER_IMAGE_STEP[s1_side]:.0f}x and latent {latent}/{s2_lat} = "
                      f"{S2_TF_PER_IMAGE_STEP[latent] / S2_TF_PER_IMAGE_STEP[s2_lat]:.0f}x, times {T} steps per stage; VAE and colour fix "
                      f"not counted (an upper bound on the CPU rate); `at_sample_size` is the same sample extrapolated in steps only",
            "s1_s_per_step_extrapolated": round(s1_full, 1), "s2_s_per_step_extrapolated": round(s2_full, 1),
            "at_sample_size": {"stage1_side": s1_side, "stage2_latent": s2_lat, "s1_s_per_step": round(dt1, 2), "s2_s_per_step": round(dt2, 2),
                               "seconds_per_image_steps_only": round(T * (dt1 + dt2), 1),
                               "what": f"a {s1_side}^2 Stage-1 image followed by a latent-{s2_lat} Stage-2 image, {T} + {T} steps"}}


# --------------------------------------------------------------------------------------------- headline: c4 / c4full
def bench_headline(args, dev, rank, world):
    from rsvld_amd import measure, ops, parallel
    T, K, W = args.ddpm_steps, args.steps, args.warmup
    full = args.workload == "c4full"
    side = args.lr_side * args.scale
    latent = side // 8
    is_metric_cfg = (args.lr_side, args.scale, T) == (512, 8, 50)      # the configuration BASELINE.json's metric is quoted on
    t0 = time.perf_counter()
    net, _ = build_stage1(T)
    net.use_graph = False            # launches are ms-long at 4096^2; eager keeps the per-launch HIP events usable
    live = not args.cached_cond      # configs[3]: "incl. live LLaVA-Next prompt" -> caption pass + live text towers on the clock
    params = stage2_params()         # (cached-embedding parameters: what the CPU baseline's oracle is fed)
    m = build_stage2(dev, True, live_conditioner=live)
    captioner = Captioner(dev) if live else None
    build_s = time.perf_counter() - t0
    thr = args.s2_threshold

    # A work unit = the BPG images one GPU holds per pass (--batch-per-gpu; BASELINE configs[3] says 16, the default line runs 1: a
    # 4096^2 image saturates the chip by itself, so the per-image time is the same and 16 x 25 iterations do not fit the driver's
    # clock).  Images are independent (infer_dir.py:196-201 runs them one at a time): inside a unit Stage 1 and Stage 2 run in
    # sub-batches (--s1-chunk / --s2-chunk) sized so that the 4096^2 activations of a sub-batch stay resident; with the feature cache
    # on, every image of a Stage-2 sub-batch takes its own decision.
    side_stream = torch.cuda.Stream(device=dev)
    BPG = max(1, args.batch_per_gpu)
    c1, c2 = max(1, min(args.s1_chunk, BPG)), max(1, min(args.s2_chunk, BPG))

    def image_inputs(unit):
        return stage1_input([unit * BPG + i for i in range(BPG)], args.lr_side, args.scale).to(dev)

    def one_image(unit, cond, n_iter, ph, gather=True):
        """One unit: Stage 1 -> hand-off -> caption -> Stage 2 -> uint8 -> all-gather; ``n_iter`` sampler iterations per stage (None = all)."""
        torch.manual_seed(42 + unit)                        # per-unit RNG streams: results do not depend on the GPU count
        ph.start()
        srs = []
        for i0 in range(0, BPG, c1):
            nb = min(c1, BPG - i0)
            with measure.hooks(net, stamp=ph, max_steps=n_iter):
                srs.append(net.super_resolution(cond[i0:i0 + nb], continous=True)[-nb:])
        sr = srs[0] if len(srs) == 1 else torch.cat(srs)
        u8 = parallel.to_uint8(sr)                          # utils/tensor2img.py:4-21: the 8-bit hand-off
        lq = u8.float() / 127.5 - 1.0                       # models/util.py:132-156 (4096 is a multiple of 64)
        del srs, sr
        ph("handoff")
        # Stage 2's three opening VAE passes depend on the image only: with a live caption pass they are issued on a second HIP stream
        # and run BESIDE the token loop (weight streaming: the matrix pipes idle), as rsvld_amd.infer.SuperResolutionPipeline.process
        # does (--serial-caption restores the reference's serial order; same kernels, same random draws, same images)
        fronts = [None] * ((BPG + c2 - 1) // c2)
        if live and not args.serial_caption:
            main_s = torch.cuda.current_stream()
            side_stream.wait_stream(main_s)
            with torch.cuda.stream(side_stream):
                fronts = [m.vae_front(lq[i0:i0 + c2], restoration_scale=S2_KW["restoration_scale"]) for i0 in range(0, BPG, c2)]
        captions = [captioner(u8[i:i + 1], seed=42 + unit * BPG + i) if live else "" for i in range(BPG)]   # infer.py:145-166
        if fronts[0] is not None:
            main_s.wait_stream(side_stream)
            for f in fronts:
                for t in f:
                    if t is not None:      # (z_stage1 is None while the restoration pull is off)
                        t.record_stream(main_s)
        ph("caption")
        outs, traces = [], []
        for j, i0 in enumerate(range(0, BPG, c2)):
            nb = min(c2, BPG - i0)
            with measure.hooks(m, stamp=ph, max_steps=n_iter):
                outs.append(parallel.to_uint8(m.just_sampling(lq[i0:i0 + nb], captions[i0:i0 + nb], vae_front=fronts[j],
                                                              **dict(S2_KW, img_threshold=thr, num_steps=T))))
            traces.append(m.cache_trace)
        one_image.cache_traces = traces
        res = outs[0] if len(outs) == 1 else torch.cat(outs)
        if gather:
            res = parallel.gather_images(res, world)        # one RCCL all-gather of finished uint8 images
        ph("gather")
        return res

    cond = image_inputs(rank)
    if not args.pmc_pass:
        # one-time work out of the timed region: lazy 16-bit weight packing of the Stage-2 networks (a tiny image, one step)
        # and the capture of the captioner's decode-step hipGraph
        small = synthetic_image((1, 3, 512, 512), seed=7, smooth=4).to(dev)
        if live:
            captioner(parallel.to_uint8(small), seed=0)
        m.just_sampling(small, [""], **dict(S2_KW, img_threshold=0.0, num_steps=1))
        del small
    torch.cuda.synchronize()

    if args.pmc_pass:            # under rocprofv3 --pmc: exactly ONE iteration of each stage + the fixed part, nothing else
        one_image(rank, cond, 1, Phases())
        torch.cuda.synchronize()
        if rank == 0:
            print(json.dumps({"pmc_pass": True, "workload": args.workload}), flush=True)
        return

    n_timed = None if full else K
    if full:
        for w in range(W):
            one_image(rank + world * w, cond, None, Phases())
    elif W > 0:
        one_image(rank, cond, W, Phases())                  # W untimed iterations of each stage (+ the fixed part once)
    ph = Phases()
    # one line per rank BEFORE the timed region: a hang behind it is attributable to a rank, a device and a phase
    rccl = None
    if world > 1 and torch.distributed.get_backend() == "nccl":
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:   # noqa: BLE001 -- informational only
            rccl = "unknown"
    print(f"[bench] rank {rank}/{world} on {torch.cuda.get_device_name(dev)} (cuda:{dev.index}), pid {os.getpid()}: world size reported "
          f"{torch.distributed.get_world_size() if world > 1 else 1}, backend {torch.distributed.get_backend() if world > 1 else None}, "
          f"RCCL {rccl}; precision {PRECISION}; models built in {build_s:.1f} s, warm-up done, entering the barrier of the timed region",
          file=sys.stderr, flush=True)
    barrier(world)
    t0 = time.perf_counter()
    if full:
        for k in range(K):
            out = one_image(rank + world * (W + k), cond, None, ph)
    else:
        out = one_image(rank, cond, n_timed, ph)
    barrier(world)
    dt = time.perf_counter() - t0
    a = ph.acc
    s1_loop, s2_loop = a["s1_loop"], a["edm_sampler_loop"]
    fixed = sum(v for k, v in a.items() if k not in ("s1_loop", "edm_sampler_loop"))
    per_rank = dist_gather_rows([s1_loop / K * 1e3 / (T if full else 1), s2_loop / K * 1e3 / (T if full else 1), fixed * 1e3 / (K if full else 1),
                                 a.get("caption", 0.0) * 1e3 / (K if full else 1), a.get("handoff", 0.0) * 1e3 / (K if full else 1)], dev, world)
    dt, s1_loop, s2_loop, fixed = dist_max([dt, s1_loop, s2_loop, fixed], dev, world)
    if full:
        value = world * BPG * K / dt
        it1, it2, fx = s1_loop / (K * T), s2_loop / (K * T), fixed / K
    else:
        it1, it2, fx = s1_loop / K, s2_loop / K, fixed     # (per unit of BPG images)
        value = world * BPG / (T * it1 + T * it2 + fx)
    finite = bool(torch.isfinite(out.float()).all())

    # ---- data-parallel self-check (outside the timed region): every rank runs ONE iteration per stage + the fixed part on
    # its image and the results are gathered; rank 0 then computes EVERY other rank's image by itself (world <= 8; else the last
    # rank's only), the way a 1-rank job would, and compares bit for bit with what that rank contributed (per-image seeds:
    # infer_dir.py:198-200 sharding must not change an image)
    dp_check = None
    if world > 1 and not full:
        chk = one_image(rank, cond, 1, Phases())
        if rank == 0:
            probes = list(range(1, world)) if world <= 8 else [world - 1]
            same = {}
            for probe in probes:
                ref = one_image(probe, image_inputs(probe), 1, Phases(), gather=False)
                same[str(probe)] = bool(torch.equal(chk[probe * BPG:(probe + 1) * BPG].cpu(), ref.cpu()))
            dp_check = {"images": probes, "iterations_per_stage": 1, "bit_identical_to_single_rank_run": all(same.values()),
                        "per_image": same}
        barrier(world)

    headline_prec = PRECISION
    extras = rank == 0 and world == 1 and not full and not args.no_extras and BPG == 1
    # per-image feature-cache decisions of the timed unit (threshold > 0): hits per image over the iterations that ran
    cache_per_image = None
    if thr > 0 and getattr(one_image, "cache_traces", None):
        cache_per_image = []
        for tr in one_image.cache_traces:                      # one trace per Stage-2 sub-batch: [step][image] = (threshold, diff, hit)
            n_img = len(tr[0]) if tr else 0
            cache_per_image += [{"hits": int(sum(bool(step[i][2]) for step in tr)), "decisions": len(tr)} for i in range(n_img)]

    def guarded(fn):
        """An extra (outside `value`) must never cost the line its headline figure: an exception becomes {"error": ...}."""
        try:
            return fn()
        except Exception as e:   # noqa: BLE001 -- reported in the line
            torch.cuda.synchronize()
            return {"error": f"{type(e).__name__}: {e}"[:500]}

    # ---- roofline: one extra instrumented pass in the HEADLINE precision (2 iterations per stage + the fixed part), not part of `value`
    # The pass is run twice, with 1 and with 3 iterations per stage: (B - A) / 2 is a kernel group's time per sampler iteration, the
    # rest of A its share of the per-image fixed part, and fixed + 50 x per-iteration its time in a WHOLE image -- the dominant kernel
    # is chosen on that (in 2 iterations + the fixed part the VAE's convolutions would outweigh the attention that is 28 % of an image).
    summ = None
    if rank == 0:
        passes = []
        for n_it in (1, 3):
            prof = ops.LaunchProfiler()
            ops.set_profiler(prof)
            one_image(rank, cond, n_it, Phases(), gather=False)
            torch.cuda.synchronize()
            ops.set_profiler(None)
            passes.append(prof.summary())
        sa, summ = passes
        for name, r in summ.items():
            ra = sa.get(name, {"ms": 0.0, "n": 0})
            per_it = max(r["ms"] - ra["ms"], 0.0) / 2.0
            r["ms_per_iteration"] = per_it
            r["ms_fixed_part"] = max(ra["ms"] - per_it, 0.0)
            r["ms_whole_image"] = r["ms_fixed_part"] + T * per_it

    # ---- the reference's default operating point, img_threshold 0.3 (infer.py:47-53), all 50 Stage-2 steps once
    cache_on = None
    if extras and is_metric_cfg and thr <= 0:
        def run_cache_on():
            torch.manual_seed(42)
            phc = Phases()
            phc.start()
            lq = synthetic_image((1, 3, side, side), seed=4321, smooth=4).to(dev)
            with measure.hooks(m, stamp=phc, max_steps=None):
                m.just_sampling(lq, [""], **dict(S2_KW, img_threshold=0.3, num_steps=T))
            phc("rest")
            tr = [h for step in m.cache_trace for (_, _, h) in step]
            t_s2 = phc.acc["edm_sampler_loop"]
            return {"img_threshold": 0.3, "hits": int(sum(tr)), "decisions": len(tr), "t_s2_total_s": round(t_s2, 2),
                    "seconds_per_image": round(T * it1 + t_s2 + fx, 2),
                    "note": "Stage 2 run over all 50 steps with the feature cache at the reference's default threshold; Stage 1 and the "
                            "fixed part as timed above; the hit rate is a property of the seeded random weights, not of the method"}
        cache_on = guarded(run_cache_on)

    # ---- north_star's third clause in the same line: the distance of the precision `value` was timed in from the REFERENCE's own CPU
    # runs after 50 + 50 steps (tools/tolerance_check.py against the committed reference-generated goldens, re-run on this device)
    tolerance = None
    if extras:
        def run_tolerance():
            from tools import tolerance_check as TC
            s1, ae, df = PRECISIONS[headline_prec]
            err = TC.errors_after_50_steps(dev, s1, ae, df)
            worst = max(err["stage1_T50_256px"]["max"], err["stage2_50_steps_64px"]["max"])
            return {"bar": 1e-3, "max_abs_err_50_steps": err, "inside_bar": bool(worst < 1e-3),
                    "full_depth": "tests/test_gpu_fulldepth.py: the same composition over 50 steps with the FULL juggernautXL networks at latent "
                                  "64 and Stage 1 at 512^2 batch 4, against the fp32-operand family on the device (DESIGN.md section 4)"}
        tolerance = guarded(run_tolerance)

    # ---- beside it: the reference's own GPU policy (fp16 UNets under autocast, bf16 VAE -- what rounds 1-4 quoted as `value`), timed at
    # the same shapes in the same process over 2 iterations per stage, with ITS distance from the reference's CPU path
    ref_gpu = None
    if extras and headline_prec == "tolerance":
        def run_ref_gpu():
            from tools import tolerance_check as TC
            apply_precision(net, m, "reference-gpu")
            try:
                one_image(rank, cond, 1, Phases(), gather=False)       # packs the 16-bit weights (one time)
                pht = Phases()
                one_image(rank, cond, 2, pht, gather=False)
                ta = pht.acc
                tfx = sum(v for k, v in ta.items() if k not in ("s1_loop", "edm_sampler_loop"))
                t1, t2 = ta["s1_loop"] / 2, ta["edm_sampler_loop"] / 2
                s1, ae, df = PRECISIONS["reference-gpu"]
                return {"dtype": "f16 (UNets, fp32 accumulate), bf16 (VAE): SR_model.py:28-33, wrappers.py:90",
                        "t_s1_iter_ms": round(t1 * 1e3, 1), "t_s2_iter_ms": round(t2 * 1e3, 1), "t_fixed_ms": round(tfx * 1e3, 1),
                        "seconds_per_image": round(T * t1 + T * t2 + tfx, 2), "img_per_s": round(1.0 / (T * t1 + T * t2 + tfx), 6),
                        "iterations_timed_per_stage": 2,
                        "headline_time_over_this": round((T * it1 + T * it2 + fx) / (T * t1 + T * t2 + tfx), 3),
                        "max_abs_err_50_steps": TC.errors_after_50_steps(dev, s1, ae, df)}
            finally:
                apply_precision(net, m, headline_prec)
        ref_gpu = guarded(run_ref_gpu)

    # ---- an EXECUTED image beside the projection: all 50 + 50 iterations of this unit, warm, after everything else of the line was
    # measured (models/SR_model.py:265-296 and sr3_modules/diffusion.py:177-201 run to the end; the same unit, seeds and shapes as `value`)
    whole = None
    if extras and not args.no_whole_image:
        def run_whole():
            phw = Phases()
            barrier(world)
            tw = time.perf_counter()
            res = one_image(rank, cond, None, phw, gather=False)
            torch.cuda.synchronize()
            secs = time.perf_counter() - tw
            aw = phw.acc
            proj = T * it1 + T * it2 + fx
            return {"seconds": round(secs, 2), "projected_seconds": round(proj, 2), "executed_over_projected": round(secs / proj, 4),
                    "s1_loop_s": round(aw["s1_loop"], 2), "s2_loop_s": round(aw["edm_sampler_loop"], 2),
                    "fixed_s": round(sum(v for k, v in aw.items() if k not in ("s1_loop", "edm_sampler_loop")), 2),
                    "iterations_executed": [T, T], "finite": bool(torch.isfinite(res.float()).all()),
                    "what": f"one whole unit ({BPG} image) executed warm in this process after the timed region: Stage 1 all {T} ancestral steps, "
                            f"hand-off, caption, Stage 2 all {T} EDM steps (feature cache as in `value`), tiled VAE, colour fix, uint8"}
        whole = guarded(run_whole)

    line = None
    if rank == 0:
        tol_pmc = "r06_c4_pmc_traffic.json" if os.path.exists(os.path.join(ROOT, "profiles", "r06_c4_pmc_traffic.json")) else "r05_c4_pmc_traffic.json"
        pmc = ({"default": "r04_c4_pmc_traffic.json", "reference-gpu": "r04_c4_pmc_traffic.json", "tolerance": tol_pmc,
                "split": "r04_c4_split_pmc_traffic.json"}.get(PRECISION, "none (no PMC pass for this mode)")
               if is_metric_cfg else "none (PMC passes exist for the metric's configuration only)")
        roof = roofline_of(summ, pmc)
        tf_img = (S1_TF_PER_IMAGE_STEP.get(side, 0) + S2_TF_PER_IMAGE_STEP.get(latent, 0)) * T
        line = {
            "metric": METRIC if is_metric_cfg else f"two-stage SR images/sec @{T} steps, {args.lr_side}px x{args.scale} (secondary workload)",
            "value": round(value, 6), "unit": "img/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"tolerance": "f16 / bf16 MFMA operands as hi + lo pairs and triples, f32 accumulation and f32 residual streams (Stage 1: "
                                   "f16 tensors x f16 weight pairs; Stage 2: rsvld_amd.ops.UNET_POLICY; config.precision) -- the composition "
                                   "inside 1e-3 of the reference's CPU path (config.tolerance)",
                      "fp32": "f32", "split": "f32 tensors, bf16 hi+lo split operands (3 MFMAs per product), fp16 layer inputs per ops.UNET_POLICY",
                      "vae-split": "f16 (UNets, fp32 accumulate); VAE: f32 tensors, bf16 hi+lo split operands"}.get(
                PRECISION, "f16 (UNets, fp32 accumulate), bf16 (VAE)"), "data": "synthetic",
            "config": {
                "workload": ((f"BASELINE configs[3]/[4] shape = the metric's configuration: " if is_metric_cfg else
                              "BASELINE configs[2] shape (Stage 1 + Stage 2, cached text embeds): " if (args.lr_side, args.scale) == (512, 4)
                              else "two-stage pipeline: ") + f"{args.lr_side}x{args.lr_side} -> {side}x{side} "
                             f"x{args.scale} SR, {BPG} image{'s' if BPG > 1 else ''} per GPU per pass: Stage 1 SR3 {T} ancestral DDPM steps at {side}^2 + 8-bit "
                             f"hand-off + Stage 2 {T} EDM steps at latent {latent} (ControlNet + UNet, CFG pair, feature cache "
                             f"{'OFF (threshold 0): uniform work per iteration' if thr <= 0 else thr}), tiled VAE 512/64, Wavelet colour "
                             f"fix, uint8 all-gather; " + ("live LLaVA-NeXT caption pass (Llama-3-8B + CLIP-L/336 architecture, fp16, "
                             "256 sampled tokens) and live CLIP-L / OpenCLIP-bigG text towers in the conditioner, all inside the "
                             "per-image fixed part; " if live else "cached text embeddings (PreparedConditioner, --cached-cond) "
                             "instead of the live LLaVA prompt; ") + "seeded random-init weights of the shipped architectures"),
                "step_definition": ("one complete image per step" if full else
                                    f"one sampler iteration of EACH stage at the real shapes; the timed region = the per-image fixed "
                                    f"part once + exactly {K} iterations per stage; value = n_gpus / ({T}*t_S1_iter + {T}*t_S2_iter + "
                                    f"t_fixed)"),
                "global_batch": world * BPG, "batch_per_gpu": BPG,
                "sub_batches": None if BPG == 1 else {"stage1_images_per_launch": c1, "stage2_images_per_launch": c2,
                                                      "why": "images are independent units (infer_dir.py:196-201); a sub-batch keeps its 4096^2 activations resident"},
                "max_memory_allocated_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                "parallelism": f"dp{world}",
                "precision_mode": PRECISION, "precision": precision_report(net, m),
                "t_s1_iter_ms": round(it1 * 1e3, 1), "t_s2_iter_ms": round(it2 * 1e3, 1), "t_fixed_ms": round(fx * 1e3, 1),
                "t_caption_ms": round(a.get("caption", 0.0) / (K if full else 1) * 1e3, 1) if live else None,
                "caption_overlaps_vae_front": bool(live and not args.serial_caption),
                "t_conditioner_ms": round(a.get("conditioner", 0.0) / (K if full else 1) * 1e3, 1),
                "caption_new_tokens": getattr(captioner, "last_tokens", None) if live else None,
                "caption_breakdown": getattr(captioner, "breakdown", None) if live else None,
                "seconds_per_image": round((T * it1 + T * it2 + fx) / BPG, 2),
                "whole_image_executed_s": None if not isinstance(whole, dict) else whole.get("seconds"), "whole_image_executed": whole,
                "per_rank_ms": [{"rank": r, "first_image": r * BPG, "t_s1_iter": round(row[0], 1), "t_s2_iter": round(row[1], 1), "t_fixed": round(row[2], 1),
                                 "caption": round(row[3], 1), "handoff": round(row[4], 1)} for r, row in enumerate(per_rank)],
                "phases_ms": {k: round(v * 1e3, 1) for k, v in a.items()},
                "timed_region_s": round(dt, 2), "model_build_s": round(build_s, 1), "finite": finite,
                "algorithmic_tflops_whole_image": round(tf_img * BPG / (T * it1 + T * it2 + fx), 1),
                "mfma_busy_whole_image_pmc": mfma_busy_of_pass(summ, pmc),
                "feature_cache": "off" if thr <= 0 else thr, "cache_decisions_per_image": cache_per_image, "dp_self_check": dp_check,
                "tolerance": tolerance, "reference_gpu_policy": ref_gpu, "cache_on": cache_on,
                "collective": {"backend": (torch.distributed.get_backend() if world > 1 else None), "world_size_reported":
                               (torch.distributed.get_world_size() if world > 1 else 1), "rccl_version": rccl,
                               "uint8_and_all_gather_ms": round(a.get("gather", 0.0) / (K if full else 1) * 1e3, 3),
                               "payload_bytes_per_rank": int(out[0].numel()) if out.dim() == 4 else None}},
            "roofline": roof}
    if rank == 0:
        line["cpu_baseline"] = None if (args.no_cpu_baseline or world > 1) else guarded(lambda: cpu_baseline_headline(m, params, T, side, latent))
        print(json.dumps(line), flush=True)


class Captioner:
    """The caption pass of infer.py:145-166 on the device the pipeline runs on: full-size LLaVA-NeXT architecture (Llama-3-8B +
    CLIP ViT-L/14-336, seeded random weights, fp16, SDPA; tools/synthetic_models.py), the product's own anyres image path and
    token loop (rsvld_amd.llava_next: process_images, multimodal_embeds, FastDecoder), 256 new tokens sampled at temperature 0.2
    inside a forked generator (models/util.py:17-66 with infer.py's max_new_tokens).  There is no tokenizer offline: the prompt is
    120 synthetic token ids around the image placeholder, and the caption handed to Stage 2 is the token ids written out as words
    (random weights produce no language anyway); it still drives the live text towers of the conditioner."""

    def __init__(self, dev):
        from tools import synthetic_models as SM
        self.dev = dev
        self.model, self.proc, self.prompt_ids = SM.llava_full(dev)
        self.prompt_ids = self.prompt_ids.to(dev)

    def __call__(self, u8_image, seed, max_new_tokens=256):
        from PIL import Image
        from rsvld_amd import llava_next as LN
        t0 = time.perf_counter()
        pil = Image.fromarray(u8_image[0].permute(1, 2, 0).contiguous().cpu().numpy())
        views = [v.to(device=self.dev, dtype=torch.float16) for v in LN.process_images([pil], self.proc, self.model.config)]
        torch.cuda.current_stream().synchronize()    # (this stream only: Stage 2's opening VAE passes may be running on the second one)
        t1 = time.perf_counter()
        with torch.random.fork_rng(devices=[self.dev]):
            torch.manual_seed(seed)
            with torch.no_grad():
                if getattr(self.model, "_fast_decoder", None) is not None:
                    self.model._fast_decoder.profile = self.breakdown = {}
                toks = LN.caption_tokens_fast(self.model, self.prompt_ids, views, [pil.size], max_new_tokens, True, 0.2, None)
        self.last_tokens = int(toks.numel())
        b = getattr(self, "breakdown", {})
        if b:   # host image path | CLIP tower + projector + splice | prefill | token loop, in ms
            b.update(image_preprocess_ms=round((t1 - t0) * 1e3, 1),
                     vision_and_splice_ms=round((time.perf_counter() - t1 - b["prefill_s"] - b["decode_s"]) * 1e3, 1),
                     prefill_ms=round(b.pop("prefill_s") * 1e3, 1), decode_ms=round(b.pop("decode_s") * 1e3, 1))
        return " ".join(f"w{t}" for t in toks.tolist())


# --------------------------------------------------------------------------------------------- Stage 2 only: s2 / c3
def bench_stage2(args, dev, rank, world):
    """Secondary workloads (not the contract line).  ``c3`` = BASELINE configs[2]: Stage 2 on 2048^2 inputs (latent 256),
    batch 8 per GPU, 50 EDM steps, per-image feature cache 0.3, Wavelet, tiled VAE.  ``s2``: any size / batch."""
    from rsvld_amd import measure, ops
    if args.workload == "c3":
        args.s2_side, args.batch, args.tile_vae = 2048, (8 if "--batch" not in sys.argv else args.batch), True
        if "--s2-threshold" not in sys.argv:
            args.s2_threshold = 0.3
    t0 = time.perf_counter()
    m = build_stage2(dev, args.tile_vae)
    side = args.s2_side
    img = torch.cat([synthetic_image((1, 3, side, side), seed=1234 + rank * args.batch + i, smooth=4)
                     for i in range(args.batch)]).to(dev)
    thr = args.s2_threshold
    kw = dict(S2_KW, img_threshold=thr, num_steps=args.ddpm_steps)
    build_s = time.perf_counter() - t0

    def one_pass():
        torch.manual_seed(42 + rank)
        return m.just_sampling(img, [""] * args.batch, **kw)

    for _ in range(args.warmup):
        one_pass()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_pass()
    barrier(world)
    dt = dist_max([time.perf_counter() - t0], dev, world)[0]
    trace = getattr(m, "cache_trace", None)
    hits = None
    if trace:
        flat = [h for step in trace for (_, _, h) in step]
        hits = round(sum(flat) / max(len(flat), 1), 3)
    if rank != 0:
        return
    prof = ops.LaunchProfiler()
    ops.set_profiler(prof)
    with measure.hooks(m, max_steps=2):
        m.just_sampling(img, [""] * args.batch, **kw)
    torch.cuda.synchronize()
    ops.set_profiler(None)
    summ = prof.summary()
    L = side // 8
    tf_step = S2_TF_PER_IMAGE_STEP.get(L)
    line = {"metric": "Stage-2 images/sec (secondary workload)", "value": round(args.batch * world * args.steps / dt, 4),
            "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if PRECISION == "fp32" else "f16 (UNet/ControlNet), bf16 (VAE)", "data": "synthetic",
            "config": {"workload": f"Stage 2 only, {side}x{side} input (latent {L}), batch {args.batch}/GPU, {args.ddpm_steps} EDM steps, "
                                   f"per-image feature-cache threshold {thr}, Wavelet, "
                                   f"{'tiled VAE (512 / 64)' if args.tile_vae else 'untiled VAE'}, full juggernautXL.yaml sizes",
                       "model_build_s": round(build_s, 1), "finite": bool(torch.isfinite(out).all()),
                       "cache_hit_rate": hits,
                       "algorithmic_tflops_no_cache": None if tf_step is None else round(
                           tf_step * args.ddpm_steps * args.batch * world * args.steps / dt, 1)},
            "roofline": roofline_of(summ, f"r02_{args.workload}_pmc_traffic.json"), "cpu_baseline": None}
    print(json.dumps(line), flush=True)


# --------------------------------------------------------------------------------------------- Stage 1 only: c2
def bench_stage1(args, dev, rank, world):
    """BASELINE configs[1]: Stage 1 only, 128 -> 512 x4, batch 4 per GPU, 50 ancestral steps (hipGraph replay)."""
    from rsvld_amd import ops, parallel
    side, T = args.lr_side * args.scale, args.ddpm_steps
    net, _ = build_stage1(T)
    net.use_graph = not args.no_graph
    ids = [rank * args.batch + i for i in range(args.batch)]
    cond = stage1_input(ids, args.lr_side, args.scale).to(dev)

    def one_pass(gather=True):
        # continous=False returns ret_img[-1], i.e. ONE image (diffusion.py:198-201): take the last
        # B rows of the 11-frame stack instead, as infer.py does for its single image (infer.py:133-135)
        sr = net.super_resolution(cond, continous=True)[-args.batch:]
        u8 = parallel.to_uint8(sr)
        return parallel.gather_images(u8, world) if gather else u8

    for _ in range(args.warmup):
        one_pass()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    barrier(world)
    dt = dist_max([time.perf_counter() - t0], dev, world)[0]
    if rank != 0:
        return
    prof = ops.LaunchProfiler()
    ops.set_profiler(prof)
    saved, net.use_graph = net.use_graph, False
    one_pass(gather=False)   # rank 0 only: no collective in here, the other ranks have moved on
    torch.cuda.synchronize()
    net.use_graph = saved
    ops.set_profiler(None)
    step_tf = S1_TF_PER_IMAGE_STEP.get(side)
    line = {"metric": f"Stage-1 (SR3) {args.lr_side}->{side} x{args.scale} images/sec @{T} steps (BASELINE configs[1], secondary workload)",
            "value": round(args.batch * world * args.steps / dt, 4), "unit": "img/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if PRECISION == "fp32" else "f16", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: Stage-1 (SR3) only, {args.lr_side}->{side} x{args.scale} SR, "
                                   f"batch {args.batch}/GPU, {T} ancestral DDPM steps, seeded random-init weights",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "hipgraph": bool(net.use_graph),
                       "algorithmic_tflops_whole_step": None if step_tf is None else round(
                           step_tf * T * args.batch * world * args.steps / dt, 2)},
            "roofline": roofline_of(prof.summary(), "r01_c2_pmc_traffic.json")}
    if not args.no_cpu_baseline and world == 1:
        dt1, thr, sweep = cpu_baseline_s1(side, T)
        line["cpu_baseline"] = {"value": 1.0 / (dt1 * T), "unit": "img/s", "cores": thr, "kind": "port",
                                "sample": f"2 ancestral steps of 1 image at {side}x{side} on the fp32 CPU oracle ({dt1:.2f} s/step, "
                                          f"{thr} threads: fastest of {sweep} on {schedulable_cores()} schedulable cores), "
                                          f"extrapolated linearly to {T} steps"}
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line), flush=True)


# --------------------------------------------------------------------------------------------- launcher
def self_launch(n):
    """``bench.py --gpus N`` started as a plain process: spawn N fresh ranks through torch.distributed.run as a CHILD
    process (this parent has not touched the GPU and never does) and exit with its code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4", choices=["c4", "c4full", "c2", "c3", "s2"],
                    help="c4 = the metric's configuration, timed per sampler iteration (default); c4full = the same, whole images; "
                         "c2 = Stage 1 only (BASELINE configs[1]); c3 = Stage 2 at 2048^2 batch 8 (configs[2]); s2 = Stage 2 only")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (c2: 4, s2: 1, c3: 8)")
    ap.add_argument("--batch-per-gpu", type=int, default=1, help="c4 / c4full: images per GPU per pass (BASELINE configs[3] says 16; "
                                                                "default 1: the per-image time is the same and the driver's clock is kept)")
    ap.add_argument("--s1-chunk", type=int, default=4, help="c4 with --batch-per-gpu: images per Stage-1 launch")
    ap.add_argument("--s2-chunk", type=int, default=2, help="c4 with --batch-per-gpu: images per Stage-2 launch (per-image cache decisions inside)")
    ap.add_argument("--lr-side", type=int, default=None)
    ap.add_argument("--scale", type=int, default=None)
    ap.add_argument("--ddpm-steps", type=int, default=50)
    ap.add_argument("--s2-side", type=int, default=1024)
    ap.add_argument("--s2-threshold", type=float, default=None, help="feature-cache threshold (headline: 0 = off; c3/s2: 0.3)")
    ap.add_argument("--tile-vae", action="store_true", help="s2: VAEHook tiling (needed from 2048x2048 up)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="c4: skip the cache-on pass, the 50-step tolerance check and the reference-GPU-policy pass")
    ap.add_argument("--no-whole-image", action="store_true", help="c4: skip the executed whole image (all 50 + 50 iterations, ~76 s) printed beside the projection")
    ap.add_argument("--cached-cond", action="store_true", help="c4 / c4full: cached text embeddings (PreparedConditioner) and no "
                                                                "caption pass, as rounds 1-2 measured (default: live LLaVA-NeXT "
                                                                "caption + live text towers, BASELINE configs[3])")
    ap.add_argument("--precision", default=None, choices=sorted(PRECISIONS),
                    help="tolerance (c4 / c4full default): the composition inside north_star's 1e-3 of the reference's CPU path -- `value` is "
                         "timed in it, the reference's GPU policy beside it (config.reference_gpu_policy); reference-gpu = default: fp16 "
                         "UNets, bf16 VAE (what rounds 1-4 quoted; the secondary workloads' default); split / fp32 / vae-split: secondary")
    ap.add_argument("--serial-caption", action="store_true", help="c4: the reference's serial order (caption pass, THEN Stage 2's opening VAE "
                                                                  "passes) instead of the VAE passes on a second HIP stream beside the caption")
    ap.add_argument("--no-graph", action="store_true", help="c2: launch kernels eagerly instead of replaying a hipGraph")
    ap.add_argument("--unet-f16-groups", default=None, help="experiment: the ops.SplitPolicy.f16_inputs of the Stage-2 UNets (comma list) "
                                                           "instead of ops.UNET_POLICY's")
    ap.add_argument("--unet-f16-weights", default=None, help="experiment: the ops.SplitPolicy.f16_weights of the Stage-2 UNets (comma list; an empty "
                                                            "string = every weight of the fp16-input GEMMs as a pair) instead of ops.UNET_POLICY's")
    ap.add_argument("--unet-q8-convs", default=None, help="experiment: the ops.SplitPolicy.q8_convs of the Stage-2 UNets (comma list; an empty string = the ResBlock "
                                                         "convolutions as three bf16 MFMAs per product, round 5's composition) instead of ops.UNET_POLICY's")
    ap.add_argument("--profile-detail", action="store_true", help="append every matrix layer's shape to its group in roofline.by_kernel")
    ap.add_argument("--dev-env", action="store_true", help="apply the developer A/B switches of the environment (rsvld_amd.devtools.apply_env)")
    ap.add_argument("--pmc-pass", action="store_true", help="c4: one iteration of each stage + the fixed part and nothing "
                                                             "else (the process rocprofv3 --pmc counts)")
    args = ap.parse_args()
    global PRECISION
    PRECISION = args.precision or ("tolerance" if args.workload in ("c4", "c4full") else "default")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    headline = args.workload in ("c4", "c4full")
    dflt = {"c4": (20, 5), "c4full": (1, 0), "c2": (3, 1), "c3": (1, 0), "s2": (1, 1)}[args.workload]
    args.steps = dflt[0] if args.steps is None else args.steps
    args.warmup = dflt[1] if args.warmup is None else args.warmup
    if args.lr_side is None:
        args.lr_side, args.scale = (512, 8) if headline else (128, 4)
    elif args.scale is None:
        args.scale = 8 if headline else 4
    if args.batch is None:
        args.batch = {"c2": 4, "c3": 8}.get(args.workload, 1)
    if args.s2_threshold is None:
        args.s2_threshold = 0.0 if headline else 0.3
    if args.workload == "c4" and max(args.steps, args.warmup) > args.ddpm_steps:
        raise SystemExit("--steps / --warmup must not exceed --ddpm-steps on the per-iteration workload")

    from rsvld_amd import devtools, ops, parallel
    if args.dev_env:
        devtools.apply_env()
    if args.profile_detail:
        ops.set_defaults(profile_detail=True)
    if args.unet_f16_groups is not None or args.unet_f16_weights is not None or args.unet_q8_convs is not None:     # an explicit policy ARGUMENT of set_precision (apply_precision)
        global UNET_POLICY
        gi = ops.UNET_POLICY.f16_inputs if args.unet_f16_groups is None else tuple(g for g in args.unet_f16_groups.split(",") if g)
        gw = None if args.unet_f16_weights is None else tuple(g for g in args.unet_f16_weights.split(",") if g)
        gq = None if args.unet_q8_convs is None else tuple(g for g in args.unet_q8_convs.split(",") if g)
        UNET_POLICY = ops.SplitPolicy(f16_inputs=tuple(gi), f16_weights=gw, q8_convs=gq)
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("RSVLD_DEVICE_OVERRIDE") is not None:   # debugging aid: several ranks on one GPU (with gloo)
        local = int(os.environ["RSVLD_DEVICE_OVERRIDE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:   # N ranks initialise 3.9 B parameters each on the host at the same time: share the cores instead of fighting
        torch.set_num_threads(max(4, schedulable_cores() // world))
    if headline:
        bench_headline(args, dev, rank, world)
    elif args.workload in ("s2", "c3"):
        bench_stage2(args, dev, rank, world)
    else:
        bench_stage1(args, dev, rank, world)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
