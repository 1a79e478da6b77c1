"""Times rsvld_layernorm_split (fp32 rows -> fp16 / planes) on the Stage-2 token tensors of the headline (latent 512)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsvld_amd import ops

dev = torch.device("cuda:0")
for rows, C in ((32768, 1280), (131072, 640), (8192, 1280)):
    x = torch.randn(rows, C, device=dev)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    for group, name, nbytes in (("ff", "fp16 out", 6), ("proj", "planes out", 8)):
        with ops.f32_split(ops.UNET_POLICY):
            for _ in range(3):
                ops.layer_norm(x, g, b, 1e-5, planes=True, group=group)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.layer_norm(x, g, b, 1e-5, planes=True, group=group)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"layernorm_split {rows} x {C} -> {name}: {ms * 1e3:7.1f} us  {rows * C * nbytes / ms / 1e6:7.1f} GB/s", flush=True)
