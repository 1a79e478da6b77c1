"""EXPERIMENT (numerics only, not a product path): which of Stage 2's 3x3 convolutions tolerate WEIGHTS rounded to fp16?  The split kernels run
unchanged on the triples of fp16(W) instead of W -- what a two-MFMA form on fp16 planes x fp16 weights would compute -- for a chosen subset of
the UNets' / ControlNet's convolutions: "res" = the ResBlocks' two convolutions (tagged conv1 / conv2 at their call sites), "other" = every
other 3x3 convolution (conv_in, Down / Upsample, ZeroSFT, the output convolution), "all".  Prints the 50-step distance from the reference's CPU
run on the goldens and from the fp32 family at full depth.
    python tools/experiment_conv_w16.py res|other|all"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsvld_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else "res"
orig = ops._conv2d_split
cache = {}
on = [False]


def patched(x, pc, **kw):
    tagged = kw.get("norm_group") in ("conv1", "conv2")
    take = on[0] and pc.kh == 3 and ops.context().policy is not None and ops.context().policy.f16_inputs and (
        which == "all" or (which == "res" and tagged) or (which == "other" and not tagged))
    if not take:
        return orig(x, pc, **kw)
    if id(pc) not in cache:
        w, w3 = pc.w, pc.w3
        pc.w, pc.w3 = pc.w.half().float(), None
        cache[id(pc)] = ops._w3(pc)
        pc.w, pc.w3 = w, w3
    keep = pc.w3
    ops._w3(pc)                      # (the real triple exists before it is swapped)
    keep = pc.w3
    pc.w3 = cache[id(pc)]
    try:
        return orig(x, pc, **kw)
    finally:
        pc.w3 = keep


ops._conv2d_split = patched
dev = torch.device("cuda:0")
import tolerance_check as TC
on[0] = True
print(which, "goldens:", TC.stage2_50(dev, "split", "split"), flush=True)
import bench
m = bench.build_stage2(dev, True)
img = bench.synthetic_image((1, 3, 512, 512), seed=4321, smooth=4).to(dev)


def run(ae, diff, thr):
    m.noise_source = "cpu"
    m.set_precision(ae, diff)
    try:
        torch.manual_seed(7)
        return m.just_sampling(img, [""], **dict(bench.S2_KW, img_threshold=thr, num_steps=50)).cpu()
    finally:
        m.noise_source = "device"
        m.set_precision("bf16", "fp16")


for thr in (0.0, 0.3):
    on[0] = False
    want = run("fp32", "fp32", thr)
    on[0] = True
    d = (run("split", "split", thr) - want).abs()
    print(f"{which} full depth, cache {thr}: max|d| = {float(d.max()):.3e}, mean|d| = {float(d.mean()):.3e}", flush=True)
