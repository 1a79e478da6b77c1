"""Which process state changes bench.build_stage2's seeded weights?  Round 5: a weights fingerprint printed by tests/test_gpu_fulldepth.py
depended on whether another test had built a model before.  This script showed that the FIRST build of a process differed from every later
one (from the first zero_module tensor on): the yaml's target strings import openaimodel / SR_modules lazily, inside
bench._host_init_skipped, after its zero_module wrappers had gone on, so the first build tagged no tensor as zero-initialised.  Fixed there
(every module that mentions zero_module is imported before the wrappers; tests/test_bench_seed.py); kept as the tool that found it: all five
fingerprints must be equal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")


def fp(tag):
    m = bench.build_stage2(dev, False)
    s = sum(float(p.detach().double().sum()) for p in m.parameters())
    names = {n: float(p.detach().double().sum()) for n, p in m.named_parameters()}
    print(f"{tag}: {s!r}", flush=True)
    del m
    torch.cuda.empty_cache()
    return names


a = fp("fresh")
b = fp("second build, nothing in between")
torch.manual_seed(3)
c = fp("after torch.manual_seed(3)")
from rsvld_amd.hipnn import HipNet
from rsvld_amd.sgm.modules.attention import BasicTransformerBlock


class Net(HipNet):
    def __init__(self):
        super().__init__()
        self.blk = BasicTransformerBlock(640, 10, 64, context_dim=2048)


net = Net()
d = fp("after constructing a HipNet with a BasicTransformerBlock on the host")
net = net.to(dev).eval()
e = fp("after moving it to the device")
for tag, x in (("b", b), ("c", c), ("d", d), ("e", e)):
    diff = [n for n in a if a[n] != x[n]]
    print(tag, "parameters that differ from the fresh build:", len(diff), diff[:5])
