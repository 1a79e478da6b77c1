"""Locate hidden device copies in one Stage-2 denoiser call (torch profiler, aten::copy_ / cat with stacks)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
import s2_common as S
from oracle import seeded
from rsvld_amd.sgm.util import instantiate_from_config
dev = torch.device("cuda:0")
m = instantiate_from_config({"target": "rsvld_amd.models.SR_model.SR_backbone", "params": S.product_params()})
seeded.seed_module(m, 1)
m.to(dev).eval()
t, y, ctx = torch.tensor([999.0, 999.0]).to(dev), S.rnd((2, 32), 62).to(dev), S.rnd((2, 77, 64), 51).to(dev)
xt, xc = S.rnd((2, 4, 32, 32), 70).to(dev), S.rnd((2, 4, 32, 32), 71, 0.5).to(dev)
c = {"crossattn": ctx, "vector": y, "control": xc}
for _ in range(2):
    m.model(xt, t, c, 1.0, "none", None)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    m.model(xt, t, c, 1.0, "none", None)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::cat", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::fill_", "aten::zero_"):
        st = [s for s in ev.stack if "rsvld" in s or "remote-sensing" in s][:2]
        cnt[(ev.name, str(ev.input_shapes)[:60], " <- ".join(x.split("/")[-1] for x in st))] += 1
for k, v in cnt.most_common(25):
    print(v, k)
