#!/usr/bin/env python3
"""Per-layer table of the headline shapes (one iteration of each stage + the fixed part), split-operand product path by default:
RSVLD_PROFILE_DETAIL=1 makes rsvld_amd.ops append every matrix layer's shape to its profiler group.  Usage (GPU box):
    python3 tools/profile_headline_layers.py [--top 40] [--precision default|split|vae-split]"""
import os
import sys

os.environ.setdefault("RSVLD_PROFILE_DETAIL", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    from rsvld_amd import measure, ops, parallel
    bench.PRECISION = sys.argv[sys.argv.index("--precision") + 1] if "--precision" in sys.argv else "split"
    dev = torch.device("cuda:0")
    net, _ = bench.build_stage1(50)
    net.use_graph = False
    m = bench.build_stage2(dev, True)
    cond = bench.stage1_input([0], 512, 8).to(dev)
    lq = bench.synthetic_image((1, 3, 4096, 4096), seed=3, smooth=4).to(dev)
    kw = dict(bench.S2_KW, img_threshold=0.0, num_steps=50)

    def one():
        with measure.hooks(net, max_steps=1):
            net.super_resolution(cond, continous=True)
        with measure.hooks(m, max_steps=1):
            m.just_sampling(lq, [""], **kw)
    one()                                   # packs weights
    prof = ops.LaunchProfiler()
    ops.set_profiler(prof)
    one()
    torch.cuda.synchronize()
    ops.set_profiler(None)
    rows = sorted(prof.summary().values(), key=lambda r: -r["ms"])
    tot = sum(r["ms"] for r in rows)
    print(f"total {tot:.1f} ms in {sum(r['n'] for r in rows)} launches")
    for r in rows[:top]:
        tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["flops"] else 0.0
        print(f"{r['ms']:9.2f} ms {r['n']:5d}x  {tf:7.1f} TF/s eff  {r['name']}")


if __name__ == "__main__":
    main()
