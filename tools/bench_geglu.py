import os, sys
sys.path.insert(0, os.getcwd())
import torch
from rsvld_amd import ops, _lib as L
dev = torch.device("cuda:0"); torch.manual_seed(0)
for (M, K, N) in [(32768, 1280, 10240), (131072, 640, 5120)]:
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K) / K ** 0.5
    pc = ops.pack_conv(w, torch.zeros(N), torch.float16, dev, geglu=True)
    for _ in range(3): ops.linear(x, pc, act=2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.linear(x, pc, act=2)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"M{M} K{K} N{N} GEGLU: {t*1e3:8.1f} us {2.0*M*K*N/t/1e9:7.1f} TF/s", flush=True)
