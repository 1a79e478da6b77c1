#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the Stage-1 sampler over one batch: BASELINE.json configs[1]
(Stage-1 only, 128 -> 512 x4 SR, batch 4 per GPU, 50 ancestral DDPM steps, synthetic inputs and
seeded random-init weights of the shipped architecture).  value = images / second over the whole
job (all ranks), inputs resident in HBM when the timed region starts.  Weak scaling: every rank
processes its own 4 images; the only collective is the all-gather of finished uint8 images.

Besides the contract fields the JSON line carries
  roofline     : the dominant kernel (implicit-GEMM conv, MFMA-bound): algorithmic FLOPs of its
                 launches / their summed HIP-event durations, from one extra instrumented pass;
  cpu_baseline : the CPU oracle (oracle/sr3_oracle.py, a port) timed on this host's cores on a
                 bounded sample and extrapolated linearly in step count.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

METRIC = "512px x8 SR images/sec @50 steps"
PEAK_TFLOPS_F16 = 2500.0   # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
UNET_TF_PER_IMAGE_STEP = {256: 0.262, 512: 1.126, 1024: 5.77, 2048: 43.3, 4096: 496.3}


def build_model(dev, T):
    from oracle import sr3_oracle as O
    from rsvld_amd.sr3_model.sr3_modules.diffusion import GaussianDiffusion
    from rsvld_amd.sr3_model.sr3_modules.unet import UNet
    c = O.SR3_CFG
    torch.manual_seed(0)
    unet = UNet(in_channel=c["in_channel"], out_channel=c["out_channel"], inner_channel=c["inner_channel"],
                norm_groups=c["norm_groups"], channel_mults=c["channel_mults"], attn_res=list(c["attn_res"]),
                res_blocks=c["res_blocks"], dropout=0.2, image_size=c["image_size"])
    net = GaussianDiffusion(unet, image_size=c["image_size"], channels=3, conditional=True)
    net.to(dev).eval()
    net.set_new_noise_schedule(dict(schedule="linear", n_timestep=T, linear_start=1e-6, linear_end=1e-2), dev)
    return net


def synthetic_batch(batch, lr_side, scale, rank):
    """LR images -> the Stage-1 input contract (data/dataset.py:16-21,30-42): bicubic x scale, [-1,1]."""
    from oracle import seeded
    lr = torch.cat([seeded.synthetic_image((1, 3, lr_side, lr_side), seed=1234 + rank * batch + i, smooth=4)
                    for i in range(batch)], 0)
    return torch.nn.functional.interpolate(lr, scale_factor=scale, mode="bicubic", align_corners=False).clamp(-1, 1)


def cpu_baseline(side, T, steps_sampled=2):
    """Time the CPU oracle on `steps_sampled` ancestral steps of ONE image at the real size."""
    from oracle import seeded, sr3_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    ncores = max(1, min(avail, 32))   # beyond ~32 threads the fp32 conv/GEMM oracle stops scaling (NUMA, sync)
    torch.set_num_threads(ncores)
    c = O.SR3_CFG
    names = []
    torch.manual_seed(0)
    from rsvld_amd.sr3_model.sr3_modules.unet import UNet
    unet = UNet(in_channel=c["in_channel"], out_channel=c["out_channel"], inner_channel=c["inner_channel"],
                norm_groups=c["norm_groups"], channel_mults=c["channel_mults"], attn_res=list(c["attn_res"]),
                res_blocks=c["res_blocks"], dropout=0.2, image_size=c["image_size"])
    sd = {"denoise_fn." + k: v.detach() for k, v in unet.state_dict().items()}
    sch = O.schedule(dict(schedule="linear", n_timestep=T, linear_start=1e-6, linear_end=1e-2))
    cond = seeded.synthetic_image((1, 3, side, side), seed=1, smooth=4)
    x = torch.randn(1, 3, side, side)
    with torch.no_grad():
        O.p_sample(sd, c, sch, x, T - 1, cond, torch.randn_like(x))  # warm-up (thread pool, allocator)
        t0 = time.perf_counter()
        for i in range(steps_sampled):
            x = O.p_sample(sd, c, sch, x, T - 1 - i, cond, torch.randn_like(x))
        dt = (time.perf_counter() - t0) / steps_sampled
    return {"value": 1.0 / (dt * T), "unit": "img/s", "cores": ncores, "kind": "port",
            "sample": f"{steps_sampled} ancestral steps of 1 image at {side}x{side} on the fp32 CPU oracle "
                      f"({dt:.2f} s/step, {ncores} threads of {avail} schedulable cores), extrapolated linearly to {T} steps"}


def build_stage2(dev, tile_vae):
    """Full juggernautXL.yaml networks (UNet 2.6 B + ControlNet 1.2 B params, seeded random init with the
    zero-initialised tensors re-drawn), cached text embeddings (PreparedConditioner)."""
    import yaml
    from rsvld_amd.sgm.util import instantiate_from_config
    cfg = yaml.safe_load(open(os.path.join(ROOT, "remote-sensing-vision-language-diffusion-model_amd", "model_configs",
                                           "juggernautXL.yaml")))["model"]["params"]
    g2, g3 = torch.Generator().manual_seed(2), torch.Generator().manual_seed(3)
    cfg["conditioner_config"]["params"] = {
        "cond_pth": {"crossattn": torch.randn(1, 77, 2048, generator=g2), "vector": torch.randn(1, 2816, generator=g2)},
        "un_cond_pth": {"crossattn": torch.randn(1, 77, 2048, generator=g3), "vector": torch.randn(1, 2816, generator=g3)}}
    torch.manual_seed(0)
    m = instantiate_from_config({"target": "rsvld_amd.models.SR_model.SR_backbone", "params": cfg})
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p_ in m.parameters():                  # zero_module() outputs would make the network output 0
            if p_.dim() >= 2 and float(p_.abs().max()) == 0.0:
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.02)
    m.to(dev).eval()
    if tile_vae:   # SR_model.py:95-125: encoder tiles of 512 px, decoder tiles of 64 latent px, cross-tile GroupNorm
        m.init_tile_vae(512, 64)
    return m


S2_KW = dict(p_p="", n_p="", dec_img=1.0, restoration_scale=-1, s_churn=5, s_noise=1.003, cfg_scale=7.5, control_scale=1.0,
             color_fix_type="Wavelet", use_linear_CFG=True, cfg_scale_start=4.0)


def bench_stage2(args, dev, rank, world):
    """Secondary workload (not the contract line): Stage 2 only — BASELINE configs[2] shape family:
    50 EDM steps, CFG 4.0->7.5 linear, s_churn 5, feature cache threshold 0.3, Wavelet colour fix."""
    from oracle import seeded
    t0 = time.perf_counter()
    m = build_stage2(dev, args.tile_vae)
    side = args.s2_side
    img = torch.cat([seeded.synthetic_image((1, 3, side, side), seed=1234 + rank * args.batch + i, smooth=4)
                     for i in range(args.batch)]).to(dev)
    thr = args.s2_threshold if args.batch == 1 else 0.0
    kw = dict(S2_KW, img_threshold=thr, num_steps=args.ddpm_steps)
    build_s = time.perf_counter() - t0

    def one_pass():
        return m.just_sampling(img, [""] * args.batch, **kw)

    for _ in range(args.warmup):
        one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_pass()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    from rsvld_amd import ops
    prof = ops.LaunchProfiler()
    ops.set_profiler(prof)
    one_pass()
    torch.cuda.synchronize()
    ops.set_profiler(None)
    summ = prof.summary()
    L = side // 8
    tf_step = {64: 4.76, 128: 20.3, 256: 107.8, 512: 865.9}.get(L)
    line = {"metric": "Stage-2 images/sec (secondary workload)", "value": round(args.batch * args.steps / dt, 4), "unit": "img/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 1),
            "dtype": "f16 (UNet/ControlNet), bf16 (VAE)", "data": "synthetic",
            "config": {"workload": f"Stage 2 only, {side}x{side} input (latent {L}), batch {args.batch}, {args.ddpm_steps} EDM steps, "
                                   f"cache threshold {thr}, Wavelet, {'tiled VAE (512 / 64)' if args.tile_vae else 'untiled VAE'}, "
                                   f"full juggernautXL.yaml sizes",
                       "model_build_s": round(build_s, 1), "finite": bool(torch.isfinite(out).all()),
                       "algorithmic_tflops_no_cache": None if tf_step is None else round(
                           tf_step * args.ddpm_steps * args.batch * args.steps / dt, 1)},
            "by_kernel": {k: {"ms": round(v["ms"], 1), "n": v["n"],
                              "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None}
                          for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}}
    print(json.dumps(line), flush=True)


def bench_pipeline(args, dev, rank, world):
    """The headline shape (BASELINE configs[3], one image per pass): 512 -> 4096 x8, Stage 1 (SR3, --ddpm-steps
    ancestral steps at 4096^2) -> 8-bit hand-off -> Stage 2 (--ddpm-steps EDM steps, ControlNet, feature cache 0.3,
    tiled VAE 512 / 64, Wavelet colour fix).  The caption comes from cached text embeddings (the LLaVA pass needs
    checkpoints that are not available offline and stays on PyTorch-ROCm by north_star).  Kernel statistics are taken
    in the timed pass itself (two HIP events per launch)."""
    from oracle import seeded
    from rsvld_amd import ops, parallel
    T = args.ddpm_steps
    t0 = time.perf_counter()
    net = build_model(dev, T)
    net.use_graph = False   # per-launch HIP events cannot be recorded inside a graph capture; launches are ms-long here
    m = build_stage2(dev, True)
    build_s = time.perf_counter() - t0
    cond = synthetic_batch(1, args.lr_side, args.scale, rank).to(dev)
    side = args.lr_side * args.scale
    # one-time work out of the timed region: lazy 16-bit weight packing of the Stage-2 networks (a tiny image, one step)
    small = seeded.synthetic_image((1, 3, 512, 512), seed=7, smooth=4).to(dev)
    m.just_sampling(small, [""], **dict(S2_KW, img_threshold=0.0, num_steps=1))
    torch.cuda.synchronize()
    stage = {}

    def one_pass():
        torch.cuda.synchronize()
        a = time.perf_counter()
        sr = net.super_resolution(cond, continous=True)[-1:]
        u8 = parallel.to_uint8(sr)                                   # utils/tensor2img.py:4-21: the 8-bit hand-off
        torch.cuda.synchronize()
        b = time.perf_counter()
        lq = u8.float() / 127.5 - 1.0                                # models/util.py:132-156 (4096 is a multiple of 64)
        out = m.just_sampling(lq, [""], **dict(S2_KW, img_threshold=args.s2_threshold, num_steps=T))
        torch.cuda.synchronize()
        c = time.perf_counter()
        stage["stage1_s"], stage["stage2_s"] = round(b - a, 2), round(c - b, 2)
        print(f"pipeline pass: stage 1 {b - a:.2f} s, stage 2 {c - b:.2f} s", flush=True)
        return out

    for _ in range(args.warmup):
        one_pass()
    prof = ops.LaunchProfiler()
    ops.set_profiler(prof)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_pass()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.set_profiler(None)
    summ = prof.summary()
    tf_img = UNET_TF_PER_IMAGE_STEP[side] * T + 865.9 * T   # BASELINE.md section 2, VAE and cache hits not counted
    dom = max(summ.values(), key=lambda r: r["ms"])
    tf = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    line = {"metric": METRIC, "value": round(args.steps / dt, 5), "unit": "img/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 1), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 (UNets), bf16 (VAE)", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3] shape, one image per pass: {args.lr_side}->{side} x{args.scale}, Stage 1 {T} DDPM "
                                   f"steps + Stage 2 {T} EDM steps (ControlNet, cache {args.s2_threshold}, tiled VAE 512/64, Wavelet), "
                                   f"cached text embeddings instead of the live LLaVA prompt, seeded random-init weights",
                       "global_batch": 1, "parallelism": "dp1", "model_build_s": round(build_s, 1),
                       "finite": bool(torch.isfinite(out).all()), "out_shape": list(out.shape), **stage,
                       "algorithmic_tflops_no_cache": round(tf_img * args.steps / dt, 1),
                       "stage2_trace": getattr(m, "last_trace", None)},
            "roofline": {"bound": "mfma", "kernel": dom["name"], "achieved": round(tf, 2), "peak": PEAK_TFLOPS_F16,
                         "unit": "TFLOP/s", "frac": round(tf / PEAK_TFLOPS_F16, 4), "traffic": None,
                         "launches": dom["n"], "avg_launch_us": round(dom["ms"] * 1e3 / dom["n"], 2),
                         "kernel_time_share": round(dom["ms"] / sum(r["ms"] for r in summ.values()), 3),
                         "by_kernel": {k: {"ms": round(v["ms"], 1), "n": v["n"],
                                           "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None}
                                       for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}},
            "cpu_baseline": None}
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2", choices=["c2", "s2", "c4"],
                    help="c2 = the contract line (default); s2 = Stage 2 only; c4 = the 512->4096 two-stage pipeline, one image")
    ap.add_argument("--s2-side", type=int, default=1024)
    ap.add_argument("--s2-threshold", type=float, default=0.3)
    ap.add_argument("--tile-vae", action="store_true", help="Stage 2: VAEHook tiling (needed from 2048x2048 up: the VAE's "
                                                               "single-head attention is quadratic in the pixel count)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU")
    ap.add_argument("--lr-side", type=int, default=128)
    ap.add_argument("--scale", type=int, default=4)
    ap.add_argument("--ddpm-steps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a hipGraph")
    args = ap.parse_args()

    from rsvld_amd import ops, parallel
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        if args.gpus != 1 and world == 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus}`")
    if os.environ.get("RSVLD_DEVICE_OVERRIDE") is not None:   # debugging aid: several ranks on one GPU (with gloo)
        local = int(os.environ["RSVLD_DEVICE_OVERRIDE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.workload == "c4":
        if "--lr-side" not in " ".join(sys.argv):
            args.lr_side, args.scale = 512, 8
        return bench_pipeline(args, dev, rank, world)
    if args.workload == "s2":
        if args.batch == 4 and "--batch" not in " ".join(sys.argv):
            args.batch = 1
        return bench_stage2(args, dev, rank, world)
    side = args.lr_side * args.scale
    T = args.ddpm_steps

    net = build_model(dev, T)
    net.use_graph = not args.no_graph
    cond = synthetic_batch(args.batch, args.lr_side, args.scale, rank).to(dev)

    def one_pass(gather=True):
        # continous=False returns ret_img[-1], i.e. ONE image (diffusion.py:198-201): take the last
        # B rows of the 11-frame stack instead, as infer.py does for its single image (infer.py:133-135)
        sr = net.super_resolution(cond, continous=True)[-args.batch:]
        u8 = parallel.to_uint8(sr)
        return parallel.gather_images(u8, world) if gather else u8

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_pass()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        on_host = torch.distributed.get_backend() == "gloo"
        tmax = torch.tensor([dt], device="cpu" if on_host else dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    n_img = args.batch * world * args.steps
    value = n_img / dt

    # ---- roofline of the dominant kernel: one extra, instrumented pass (not part of `value`)
    roof = None
    if rank == 0:
        prof = ops.LaunchProfiler()
        ops.set_profiler(prof)
        saved = net.use_graph
        net.use_graph = False
        one_pass(gather=False)   # rank 0 only: no collective in here, the other ranks have moved on
        torch.cuda.synchronize()
        net.use_graph = saved
        ops.set_profiler(None)
        summ = prof.summary()
        dom = max(summ.values(), key=lambda r: r["ms"])
        tf = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        traffic = None   # HBM-side bytes per launch from the committed rocprofv3 PMC passes of this workload
        pmc = os.path.join(ROOT, "profiles", "r01_c2_pmc_traffic.json")
        if os.path.exists(pmc) and side == 512 and args.batch == 4:
            k = json.load(open(pmc))["kernels"].get(dom["name"])
            traffic = None if k is None else round(k["hbm_bytes_per_launch"])
        roof = {"bound": "mfma", "kernel": dom["name"], "achieved": round(tf, 2), "peak": PEAK_TFLOPS_F16,
                "unit": "TFLOP/s", "frac": round(tf / PEAK_TFLOPS_F16, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["n"]),
                "launches": dom["n"], "avg_launch_us": round(dom["ms"] * 1e3 / dom["n"], 2),
                "kernel_time_share": round(dom["ms"] / sum(r["ms"] for r in summ.values()), 3),
                "by_kernel": {k: {"ms": round(v["ms"], 3), "n": v["n"],
                                  "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None}
                              for k, v in summ.items()}}

    if rank == 0:
        step_tf = UNET_TF_PER_IMAGE_STEP.get(side)
        line = {
            "metric": METRIC, "value": round(value, 4), "unit": "img/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: Stage-1 (SR3) only, {args.lr_side}->{side} x{args.scale} SR, "
                                   f"batch {args.batch}/GPU, {T} ancestral DDPM steps, seeded random-init weights",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "hipgraph": bool(net.use_graph),
                       "algorithmic_tflops_whole_step": None if step_tf is None else round(
                           step_tf * T * args.batch * world * args.steps / dt, 2)},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(side, T)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
