"""tests/test_gpu_fulldepth.py's Stage-2 measurement for arbitrary ops.SplitPolicy compositions: the FULL juggernautXL networks over all 50 EDM
steps at latent 64 against the fp32-operand family on the device (cache off and at threshold 0.3), max / mean |delta| per pixel.
    python tools/fulldepth_check.py [--only name,name]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rsvld_amd import ops

dev = torch.device("cuda:0")
P = ops.SplitPolicy
POLICIES = {"default": ops.UNET_POLICY, "pairs": P(f16_weights=()), "w1_qkv": P(f16_weights=("qkv",)), "w1_geglu": P(f16_weights=("geglu",)),
            "w1_all": P(f16_weights=("qkv", "geglu", "attn_out", "ff_out")), "w1_ao": P(f16_weights=("qkv", "geglu", "attn_out")),
            "w1_fo": P(f16_weights=("qkv", "geglu", "ff_out"))}
if "--only" in sys.argv:
    keep = sys.argv[sys.argv.index("--only") + 1].split(",")
    POLICIES = {k: v for k, v in POLICIES.items() if k in keep}
m = bench.build_stage2(dev, True)
img = bench.synthetic_image((1, 3, 512, 512), seed=4321, smooth=4).to(dev)


def run(ae, diff, policy, thr):
    m.noise_source = "cpu"
    m.set_precision(ae, diff, policy=policy)
    try:
        torch.manual_seed(7)
        out = m.just_sampling(img, [""], **dict(bench.S2_KW, img_threshold=thr, num_steps=50))
        return out.cpu(), [bool(step[0][2]) for step in m.cache_trace]
    finally:
        m.noise_source = "device"
        m.set_precision("bf16", "fp16")


for thr in (0.0, 0.3):
    want, wtrace = run("fp32", "fp32", None, thr)
    for name, pol in POLICIES.items():
        got, trace = run("split", "split", pol, thr)
        d = (got - want).abs()
        flips = [i for i, (a, b) in enumerate(zip(trace, wtrace)) if a != b]
        print(f"{name:8s} cache {thr}: max|d| = {float(d.max()):.3e}, mean|d| = {float(d.mean()):.3e}; cache decisions that differ: {flips}", flush=True)
