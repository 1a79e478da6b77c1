"""Times the 3x3 halo convolution on the Stage-1 (BASELINE configs[1]) layer shapes, with and without the fused
GroupNorm+SiLU prologue.  RSVLD_LIB=<another build of librsvld_hip.so> selects a different library for A/B runs; ONLY64=1
restricts the list to the Cout = 64 layers (the single-buffered 64-channel kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsvld_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = [  # B, H, W, Cin, Cin2, Cout
    (4, 256, 256, 128, 0, 128), (4, 256, 256, 384, 0, 128), (4, 256, 256, 192, 0, 128),
    (4, 128, 128, 256, 0, 256), (4, 128, 128, 768, 0, 256), (4, 128, 128, 384, 0, 256),
    (4, 64, 64, 512, 0, 512), (4, 64, 64, 1024, 0, 512), (4, 64, 64, 768, 0, 512),
    (4, 512, 512, 64, 0, 64), (4, 512, 512, 128, 64, 64),
]
if os.environ.get("ONLY64"):
    SHAPES = [s for s in SHAPES if s[5] <= 64] + [(4, 512, 512, 64, 64, 64), (1, 2048, 2048, 64, 0, 64)]
reps = int(os.environ.get("REPS", 20))
print("library:", os.environ.get("RSVLD_LIB", "in-tree build"))
for (B, H, W, C1, C2, Co) in SHAPES:
    x = torch.randn(B, H, W, C1, device=dev, dtype=torch.float16)
    x2 = torch.randn(B, H, W, C2, device=dev, dtype=torch.float16) if C2 else None
    w = torch.randn(Co, C1 + C2, 3, 3) / (3 * (C1 + C2) ** 0.5)
    pc = ops.pack_conv(w, torch.zeros(Co), torch.float16, dev)
    gamma, beta = torch.ones(C1 + C2, device=dev), torch.zeros(C1 + C2, device=dev)
    res_t = torch.randn(B, H, W, Co, device=dev, dtype=torch.float16) if os.environ.get("RESIDUAL") else None   # ResBlock second conv
    flops = 2.0 * B * H * W * Co * (C1 + C2) * 9
    row = f"B{B} {H}x{W} Cin{C1}+{C2} Cout{Co}: "
    for label, norm in (("plain", None), ("gn+silu", (gamma, beta, 32, 1e-5, True))):
        if norm is not None:
            # statistics out of the timed region: the partials come from a producing conv in the real network
            lib = ops.L.load()
            ab = None
        for _ in range(3):
            y = ops.conv2d(x, pc, x2=x2, pad=1, norm=norm, residual=res_t)
        torch.cuda.synchronize()
        # time only the conv kernel launches via the launch profiler
        prof = ops.LaunchProfiler()
        ops.set_profiler(prof)
        for _ in range(reps):
            y = ops.conv2d(x, pc, x2=x2, pad=1, norm=norm, residual=res_t)
        ops.set_profiler(None)
        agg = prof.summary()
        ms = sum(r["ms"] for k, r in agg.items() if k.startswith("conv_")) / reps
        row += f"{label} {ms*1e3:7.1f} us {flops/ms/1e9:7.1f} TF/s   "
    print(row, flush=True)
