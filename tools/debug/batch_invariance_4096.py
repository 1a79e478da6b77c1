"""Where does image 0 of a Stage-2 sub-batch of two 4096^2 images leave its batch-of-1 run?  (debug aid for tests/test_gpu_batch16.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from rsvld_amd import measure, ops, parallel
from rsvld_amd.sgm.modules.diffusionmodules import sampling as SMP

dev = torch.device("cuda:0")
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pol = sys.argv[2] if len(sys.argv) > 2 else "default"
m = bench.build_stage2(dev, True)
m.set_precision("split", "split", policy=ops.SplitPolicy(q8_convs=()) if pol == "noq8" else None)
lq = torch.cat([bench.synthetic_image((1, 3, side, side), seed=10 + i, smooth=4) for i in range(2)]).to(dev)


def per_image(shape, ids, tag, device):
    return torch.cat([torch.randn((1,) + tuple(shape[1:]), generator=torch.Generator(device=device).manual_seed(100_000 * tag + i), device=device)
                      for i in ids])


def run(ids):
    taps = {}
    draws = iter(range(200, 300))
    m._posterior_noise = lambda shape: per_image(shape, ids, 150, torch.device("cpu"))
    m._randn_like = lambda t: per_image(t.shape, ids, next(draws), t.device)
    x = lq[ids]
    with torch.no_grad():
        front = m.vae_front(x, restoration_scale=-1)
        taps["z_denoise"], taps["x_stage1"] = front[0].clone(), front[1].clone()
        orig = SMP.RestoreEDMSampler.step

        def spy(self, xx, i, *a, **k):
            out = orig(self, xx, i, *a, **k)
            taps[f"z_step{i}"] = out[0].clone()
            return out
        SMP.RestoreEDMSampler.step = spy
        try:
            with measure.hooks(m, max_steps=1):
                out = m.just_sampling(x, [""] * len(ids), vae_front=front, **dict(bench.S2_KW, img_threshold=0.3, num_steps=50))
        finally:
            SMP.RestoreEDMSampler.step = orig
        taps["final"] = out
    return taps


a, b = run([0, 1]), run([0])
for k in a:
    d = (a[k][:1].float() - b[k][:1].float()).abs()
    print(f"{k:12s} image 0 in a pair vs alone: equal {bool(torch.equal(a[k][:1], b[k][:1]))}  max|d| = {float(d.max()):.3e}  differing elements {int((d > 0).sum())} / {d.numel()}", flush=True)
