"""Which kernels of a Stage-2 iteration at latent 512 are NOT launches of this library (PyTorch glue), by device time.
    python tools/s2_outside.py       (GPU box, repo root)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from rsvld_amd import measure

dev = torch.device("cuda:0")
m = bench.build_stage2(dev, True, live_conditioner=False)
small = bench.synthetic_image((1, 3, 512, 512), seed=7, smooth=4).to(dev)
m.just_sampling(small, [""], **dict(bench.S2_KW, img_threshold=0.0, num_steps=1))
lq = bench.synthetic_image((1, 3, 4096, 4096), seed=3, smooth=8).to(dev)


class Gate:
    def __init__(self):
        self.prof = None

    def __call__(self, name):
        torch.cuda.synchronize()
        if name == "sampler_init":
            self.prof = profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU])
            self.prof.__enter__()
        elif name == "edm_sampler_loop":
            self.prof.__exit__(None, None, None)


for rep in range(2):
    g = Gate()
    torch.manual_seed(1)
    with measure.hooks(m, stamp=g, max_steps=2):
        m.just_sampling(lq, [""], **dict(bench.S2_KW, img_threshold=0.0, num_steps=50))
rows = []
for e in g.prof.key_averages():
    dt = getattr(e, "device_time_total", None)
    if dt is None:
        dt = getattr(e, "cuda_time_total", 0)
    if e.device_type.name != "CPU" or dt <= 0:
        pass
    rows.append((dt, e.count, e.key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"two iterations: {tot/1e3:.1f} ms of device time in {sum(r[1] for r in rows)} profiler rows")
for dt, n, k in rows[:45]:
    print(f"{dt/1e3/2:9.3f} ms/iter  n={n//2:5d}  {k[:110]}")
