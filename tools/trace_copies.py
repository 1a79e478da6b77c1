"""Which Python call sites issue device copies (torch.cat / stack / clone / contiguous / copy_ / to / index ops) during ONE sampler
iteration of each stage and the per-image fixed part of the headline workload?  rocprofv3 counts ~2 500 __amd_rocclr_copyBuffer
dispatches per iteration pair; this lists who makes them (call site, count, bytes).  Run on the GPU box: python tools/trace_copies.py"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from rsvld_amd import measure, parallel

PKG = "remote-sensing-vision-language-diffusion-model_amd"
log = collections.Counter()
size = collections.Counter()
on = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if PKG in fr.filename or fr.filename.endswith("bench.py"):
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "?"


def nbytes(t):
    return t.numel() * t.element_size() if torch.is_tensor(t) else 0


def wrap_fn(mod, name, pick):
    orig = getattr(mod, name)

    def f(*a, **k):
        out = orig(*a, **k)
        if on[0]:
            t = pick(out, a)
            if torch.is_tensor(t) and t.is_cuda:
                key = (name, site())
                log[key] += 1
                size[key] += nbytes(t)
        return out

    setattr(mod, name, f)


for nm in ("cat", "stack"):
    wrap_fn(torch, nm, lambda out, a: out)
for nm in ("clone", "contiguous", "copy_", "to", "index_select", "index_copy", "index_copy_", "repeat", "float", "half"):
    orig = getattr(torch.Tensor, nm)

    def make(nm, orig):
        def f(self, *a, **k):
            out = orig(self, *a, **k)
            if on[0] and torch.is_tensor(out) and out.is_cuda and (out.data_ptr() != self.data_ptr() or nm in ("copy_", "index_copy_")):
                key = (nm, site())
                log[key] += 1
                size[key] += nbytes(out)
            return out
        return f

    setattr(torch.Tensor, nm, make(nm, orig))

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
net, _ = bench.build_stage1(50)
net.use_graph = False
m = bench.build_stage2(dev, True)
cond = bench.stage1_input([0], 512, 8).to(dev)
small = bench.synthetic_image((1, 3, 512, 512), seed=7, smooth=4).to(dev)
m.just_sampling(small, [""], **dict(bench.S2_KW, img_threshold=0.0, num_steps=1))


def report(title):
    print(f"== {title}: {sum(log.values())} copies, {sum(size.values()) / 1e6:.1f} MB")
    for key, n in log.most_common(25):
        print(f"   {n:6d} x  {size[key] / 1e6:10.2f} MB  {key[0]:12s} {key[1]}")
    log.clear()
    size.clear()


torch.manual_seed(42)
on[0] = True
with measure.hooks(net, max_steps=1):
    sr = net.super_resolution(cond, continous=True)[-1:]
on[0] = False
report("Stage 1: set-up + ONE iteration at 4096^2")
lq = parallel.to_uint8(sr).float() / 127.5 - 1.0
for steps in (1, 2):
    on[0] = True
    with measure.hooks(m, max_steps=steps):
        m.just_sampling(lq, [""], **dict(bench.S2_KW, img_threshold=0.0, num_steps=50))
    on[0] = False
    report(f"Stage 2: fixed part + {steps} iteration(s) at latent 512")
