import os, sys
sys.path.insert(0, os.getcwd())
import torch
from rsvld_amd import ops
dev = torch.device("cuda:0")
for rows, C in [(131072, 640), (32768, 1280), (8192, 1280), (524288, 320)]:
    x = torch.randn(rows, C, device=dev, dtype=torch.float16)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    for _ in range(3): ops.layer_norm(x, g, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.layer_norm(x, g, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"rows {rows} C {C}: {ms*1e3:7.1f} us  {2*rows*C*2/ms/1e9:6.2f} TB/s")
