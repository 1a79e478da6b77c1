"""Bisect: which launch of the tiled VAE encoder gives image 0 different bits when a second image shares the batch?  (debug aid)"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from rsvld_amd import ops

dev = torch.device("cuda:0")
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = bench.build_stage2(dev, True)
m.set_precision("split", "split")
lq = torch.cat([bench.synthetic_image((1, 3, side, side), seed=10 + i, smooth=4) for i in range(2)]).to(dev)
log, B_now = [], [1]


def digest(o):
    t = o.t if isinstance(o, ops.Planes) else o
    if not torch.is_tensor(t):
        return None
    rows = t[0::B_now[0]] if t.shape[0] % B_now[0] == 0 else t
    r = rows.contiguous()
    r = r.view(torch.int32) if (r.element_size() * r.shape[-1]) % 4 == 0 else r.view(torch.int16)
    return int(r.sum(dtype=torch.int64)) ^ int((r[..., ::3].sum(dtype=torch.int64)) << 1), tuple(t.shape)


def wrap(name):
    orig = getattr(ops, name)

    def f(*a, **k):
        out = orig(*a, **k)
        ctx = ops.context()
        log.append((name, digest(out), ctx.plan_div))
        return out
    setattr(ops, name, f)


for n in ("conv2d", "group_norm_stats", "group_norm_apply", "attention", "group_norm", "nchw_to_nhwc"):
    wrap(n)
runs = {}
for ids in ([0, 1], [0]):
    log.clear()
    B_now[0] = len(ids)
    with torch.no_grad():
        m.encode_first_stage_with_denoise(lq[ids], use_sample=False)
    runs[len(ids)] = list(log)
a, b = runs[2], runs[1]
print(len(a), len(b))
for i, (x, y) in enumerate(zip(a, b)):
    if x[0] != y[0] or x[1] is None or y[1] is None or x[1][0] != y[1][0]:
        print("first mismatch at call", i, x, y)
        for j in range(max(0, i - 3), min(len(a), i + 2)):
            print("   ", j, a[j], b[j])
        break
else:
    print("all equal")
