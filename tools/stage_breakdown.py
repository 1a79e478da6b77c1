"""Where one sampler iteration of each stage goes at the headline shapes (512 -> 4096, latent 512): per C-ABI entry
(HIP events around every launch of this library, ops.LaunchProfiler) next to the un-instrumented wall time of the same
iterations; "outside" = wall time - launches of this library = PyTorch glue kernels + launch gaps.
    python tools/stage_breakdown.py [iterations]      (GPU box, repo root)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from rsvld_amd import measure, ops, parallel

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
T = 50
net, _ = bench.build_stage1(T)
net.use_graph = False
m = bench.build_stage2(dev, True, live_conditioner=False)
cond = bench.stage1_input([0], 512, 8).to(dev)
small = bench.synthetic_image((1, 3, 512, 512), seed=7, smooth=4).to(dev)
m.just_sampling(small, [""], **dict(bench.S2_KW, img_threshold=0.0, num_steps=1))
lq = bench.synthetic_image((1, 3, 4096, 4096), seed=3, smooth=8).to(dev)


class LoopOnly:
    """stamp callable: switches the launch profiler on for the sampler loop only (phase names of bench.Phases)."""

    def __init__(self, before, loop, prof):
        self.before, self.loop, self.prof, self.t0, self.dt = before, loop, prof, None, None

    def __call__(self, name):
        torch.cuda.synchronize()
        if name == self.before:
            ops.set_profiler(self.prof)
            self.t0 = time.perf_counter()
        elif name == self.loop:
            self.dt = time.perf_counter() - self.t0
            ops.set_profiler(None)


def run(stage, prof):
    st = LoopOnly("s1_setup" if stage == 1 else "sampler_init", "s1_loop" if stage == 1 else "edm_sampler_loop", prof)
    torch.manual_seed(1)
    if stage == 1:
        with measure.hooks(net, stamp=st, max_steps=N):
            net.super_resolution(cond, continous=True)
    else:
        with measure.hooks(m, stamp=st, max_steps=N):
            m.just_sampling(lq, [""], **dict(bench.S2_KW, img_threshold=0.0, num_steps=T))
    return st.dt


for stage in (1, 2):
    run(stage, None)                       # warm
    wall = run(stage, None) / N
    prof = ops.LaunchProfiler()
    wall_i = run(stage, prof) / N
    summ = prof.summary()
    rows = sorted(summ.values(), key=lambda r: -r["ms"])
    tot = sum(r["ms"] for r in rows) / N
    print(f"== Stage {stage}: {wall*1e3:.1f} ms per iteration (instrumented: {wall_i*1e3:.1f}); launches of this library "
          f"{tot:.1f} ms; outside {wall*1e3 - tot:.1f} ms", flush=True)
    for r in rows:
        tf = r["flops"] / r["ms"] / 1e9 if r["flops"] else 0
        gb = r["bytes"] / r["ms"] / 1e6 if r["bytes"] else 0
        print(f"  {r['name']:34s} {r['ms']/N:8.2f} ms  n={r['n']//N:5d}  {tf:7.1f} TF/s {gb:8.1f} GB/s", flush=True)
