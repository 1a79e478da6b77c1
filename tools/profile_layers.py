"""Per-layer-shape timing of one Stage-1 UNet forward (HIP events around every C-ABI launch)."""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsvld_amd import ops
import bench

dev = torch.device("cuda:0")
net = bench.build_model(dev, 50)
B, S = int(os.environ.get("B", 4)), int(os.environ.get("S", 512))
x = torch.randn(B, 6, S, S, device=dev)
lv = torch.full((B, 1), 0.5, device=dev)
unet = net.denoise_fn
xin = ops.nchw_to_nhwc(x, unet.compute_dtype)
for _ in range(2):
    unet.forward_nhwc(xin, lv)
torch.cuda.synchronize()

recs = []
orig_conv = ops.conv2d
def conv2d(x, pc, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig_conv(x, pc, **kw); e1.record()
    Bq, Ho, Wo, _ = out.shape
    x2 = kw.get("x2")
    key = f"conv {pc.kh}x{pc.kw} s{kw.get('stride',1)} {'up ' if kw.get('upsample') else ''}Cin{pc.cin_p} Cout{pc.cout_p} M{Bq*Ho*Wo}"
    recs.append((key, 2.0 * Bq * Ho * Wo * pc.cout * pc.cin * pc.kh * pc.kw, e0, e1))
    return out
ops.conv2d = conv2d
import rsvld_amd.sr3_model.sr3_modules.unet as U
reps = 5
for _ in range(reps):
    unet.forward_nhwc(xin, lv)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for k, f, e0, e1 in recs:
    a = agg.setdefault(k, [0.0, 0.0, 0])
    a[0] += e0.elapsed_time(e1); a[1] += f; a[2] += 1
tot = sum(a[0] for a in agg.values())
print(f"conv total per forward: {tot/reps:.2f} ms")
for k, (ms, f, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{ms/reps:8.3f} ms  n={n//reps:2d}  {f/ms/1e9:7.1f} TF/s  {k}")
