"""Times ops.linear (conv_igemm) on the Stage-2 transformer GEMM shapes, beside torch.matmul (rocBLAS / hipBLASLt) on the
same shapes as a calibration of what the hardware allows (not used by the product)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsvld_amd import ops
from rsvld_amd import _lib as L

dev = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = [(8192, 1280, 1280, 0), (8192, 1280, 3840, 0), (8192, 1280, 10240, 2), (8192, 5120, 1280, 0),
          (32768, 640, 640, 0), (32768, 640, 5120, 2), (32768, 2560, 640, 0), (131072, 640, 1920, 0), (131072, 640, 5120, 2)]
if os.environ.get("HEADLINE"):   # Stage-2 transformer GEMMs at latent 512 (CFG pair): level 3 (1280 ch, 32 768 tokens), level 2 (640 ch, 131 072)
    SHAPES = [(32768, 1280, 1280, 0), (32768, 1280, 3840, 0), (32768, 1280, 10240, 2), (32768, 5120, 1280, 0), (32768, 2048, 2560, 0),
              (131072, 640, 640, 0), (131072, 640, 1920, 0), (131072, 640, 5120, 2), (131072, 2560, 640, 0)]
reps = int(os.environ.get("REPS", 10))
for (M, K, N, act) in SHAPES:
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K) / K ** 0.5
    pc = ops.pack_conv(w, torch.zeros(N), torch.float16, dev, geglu=(act == 2))
    wt = w.to(dev, torch.float16)
    res_t = torch.randn(M, N, device=dev, dtype=torch.float16) if (os.environ.get("RESIDUAL") and act == 0) else None   # to_out / FF2 / proj_out
    def ours():
        return ops.linear(x, pc, act=act, residual=res_t)
    def lib():
        return x @ wt.t()
    res = []
    for fn in (ours, lib):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / reps)
    fl = 2.0 * M * K * N
    print(f"M{M} K{K} N{N} act{act}: ours {res[0]*1e3:8.1f} us {fl/res[0]/1e9:7.1f} TF/s | torch.matmul {res[1]*1e3:8.1f} us {fl/res[1]/1e9:7.1f} TF/s", flush=True)
