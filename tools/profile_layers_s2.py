"""Per-layer-shape timing of the Stage-2 denoiser (ControlNet + UNet, CFG pair) at one latent size:
HIP events around every conv2d / linear call, grouped by shape.  S2_SIDE (image side, default 2048), STEPS (default 2)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsvld_amd import ops
import bench

dev = torch.device("cuda:0")
side = int(os.environ.get("S2_SIDE", 2048))
steps = int(os.environ.get("STEPS", 2))
m = bench.build_stage2(dev, True)
from oracle import seeded
img = seeded.synthetic_image((1, 3, side, side), seed=5, smooth=4).to(dev)
kw = dict(bench.S2_KW, img_threshold=0.0, num_steps=steps)
m.just_sampling(img, [""], **kw)          # packs weights
torch.cuda.synchronize()
recs = []
orig = ops.conv2d
def conv2d(x, pc, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(x, pc, **k); e1.record()
    B, Ho, Wo, _ = out.shape
    key = f"{pc.kh}x{pc.kw} s{k.get('stride',1)} {'up ' if k.get('upsample') else ''}Cin{pc.cin_p} Cout{pc.cout_p} M{B*Ho*Wo} act{k.get('act',0)}{' res' if k.get('residual') is not None else ''}{' norm' if k.get('norm') is not None else ''}"
    recs.append((key, 2.0 * B * Ho * Wo * pc.cout * pc.cin * pc.kh * pc.kw, e0, e1))
    return out
ops.conv2d = conv2d
m.just_sampling(img, [""], **kw)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for k, f, e0, e1 in recs:
    a = agg.setdefault(k, [0.0, 0.0, 0]); a[0] += e0.elapsed_time(e1); a[1] += f; a[2] += 1
tot = sum(a[0] for a in agg.values())
print(f"conv/linear total: {tot:.1f} ms over {steps} steps")
for k, (ms, f, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("TOP", 40))]:
    print(f"{ms:9.2f} ms  n={n:4d}  {f/ms/1e9:7.1f} TF/s  {k}")
