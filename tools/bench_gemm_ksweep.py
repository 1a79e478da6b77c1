import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsvld_amd import ops
dev = torch.device("cuda:0")
for (M, N) in [(8192, 3840), (32768, 2560)]:
  for K in (320, 640, 1280, 2560, 5120):
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K) / K ** 0.5
    pc = ops.pack_conv(w, torch.zeros(N), torch.float16, dev)
    for _ in range(3): ops.linear(x, pc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.linear(x, pc)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"M{M} N{N} K{K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
