"""Times the 3x3 halo convolution in the two multi-segment forms at the headline's layer shapes: fp16 x weight pairs (RSVLD_F16W2,
Stage 1 at 4096^2, fused GroupNorm + SiLU prologue) and the split precision (RSVLD_SPLIT, Stage 2 at latent 512, planes in, fp32 out).
RSVLD_LIB=<another build> selects a different library for A/B runs.  TF/s = MFMA rate (2 or 3 MFMAs per product)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsvld_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
W2 = [(1, 4096, 4096, 64, 0, 64), (1, 2048, 2048, 128, 0, 128), (1, 2048, 2048, 192, 0, 128), (1, 1024, 1024, 256, 0, 256), (1, 1024, 1024, 384, 0, 256),
      (1, 512, 512, 512, 0, 512), (1, 512, 512, 768, 0, 512)]
SPLIT = [(2, 512, 512, 320, 0, 320), (2, 512, 512, 320, 320, 320), (2, 256, 256, 640, 0, 640), (2, 256, 256, 640, 640, 640), (2, 128, 128, 1280, 0, 1280)]
reps = int(os.environ.get("REPS", 10))
print("library:", os.environ.get("RSVLD_LIB", "in-tree build"))


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    prof = ops.LaunchProfiler()
    ops.set_profiler(prof)
    for _ in range(reps):
        fn()
    ops.set_profiler(None)
    agg = prof.summary()
    return sum(r["ms"] for k, r in agg.items() if k.startswith("conv_")) / reps, [k for k in agg if k.startswith("conv_")]


for (B, H, W, C1, C2, Co) in W2:
    x = torch.randn(B, H, W, C1, device=dev, dtype=torch.float16)
    w = torch.randn(Co, C1, 3, 3) / (3 * C1 ** 0.5)
    pc = ops.pack_conv(w, torch.zeros(Co), torch.float32, dev)
    gamma, beta = torch.ones(C1, device=dev), torch.zeros(C1, device=dev)
    ms, names = timed(lambda: ops.conv2d(x, pc, norm=(gamma, beta, 32, 1e-5, True), stats=True))
    fl = 2.0 * 2.0 * B * H * W * Co * C1 * 9
    print(f"w2    B{B} {H}x{W} Cin{C1} Cout{Co}: {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} MFMA-TF/s  {names}", flush=True)
    del x
for (B, H, W, C1, C2, Co) in SPLIT:
    x = ops.to_planes(torch.randn(B, H, W, C1, device=dev))
    x2 = ops.to_planes(torch.randn(B, H, W, C2, device=dev)) if C2 else None
    w = torch.randn(Co, C1 + C2, 3, 3) / (3 * (C1 + C2) ** 0.5)
    pc = ops.pack_conv(w, torch.zeros(Co), torch.float32, dev, cin_split=(C1, C2) if C2 else None)
    with ops.f32_split(ops.ALL_SPLIT):
        ms, names = timed(lambda: ops.conv2d(x, pc, x2=x2, stats=True))
    fl = 3.0 * 2.0 * B * H * W * Co * (C1 + C2) * 9
    print(f"split B{B} {H}x{W} Cin{C1}+{C2} Cout{Co}: {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} MFMA-TF/s  {names}", flush=True)

# round 6: the same Stage-2 layers as a ResBlock runs them (GroupNorm + SiLU in front): planes + three bf16 MFMAs vs RSVLD_F16Q8 rows + e4m3 cross terms
def timed_all(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    prof = ops.LaunchProfiler()
    with ops.tuning(profiler=prof):
        for _ in range(reps):
            fn()
    return {k: r["ms"] / reps for k, r in prof.summary().items()}


for (B, H, W, C1, C2, Co) in SPLIT:
    x = torch.randn(B, H, W, C1, device=dev)
    x2 = torch.randn(B, H, W, C2, device=dev) if C2 else None
    w = torch.randn(Co, C1 + C2, 3, 3) / (3 * (C1 + C2) ** 0.5)
    pc = ops.pack_conv(w, torch.zeros(Co), torch.float32, dev, cin_split=(C1, C2) if C2 else None)
    gamma, beta = torch.ones(C1 + C2, device=dev), torch.zeros(C1 + C2, device=dev)
    rv = torch.randn(B, Co, device=dev)
    res = {}
    for name, pol in (("split", ops.SplitPolicy(q8_convs=())), ("q8", ops.UNET_POLICY)):
        with ops.f32_split(pol):
            res[name] = timed_all(lambda: ops.conv2d(x, pc, x2=x2, norm=(gamma, beta, 32, 1e-5, True), norm_group="conv1", stats=True, rowvec=rv))
    tot = {k: sum(v.values()) for k, v in res.items()}
    conv = {k: sum(t for n, t in v.items() if n.startswith("conv_")) for k, v in res.items()}
    fl = 2.0 * B * H * W * Co * (C1 + C2) * 9
    print(f"resblock conv B{B} {H}x{W} Cin{C1}+{C2} Cout{Co}: split {tot['split']*1e3:7.0f} us (conv {conv['split']*1e3:7.0f}) | q8 {tot['q8']*1e3:7.0f} us "
          f"(conv {conv['q8']*1e3:7.0f} = {fl/conv['q8']/1e9:6.0f} algorithmic TF/s) | conv x{conv['split']/conv['q8']:.2f}, layer x{tot['split']/tot['q8']:.2f}  "
          f"{sorted(res['q8'])}", flush=True)
    del x, x2
