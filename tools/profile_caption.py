"""The caption pass of the headline bench alone (full-size LLaVA-NeXT architecture, seeded random weights): two captions of a
synthetic 4096^2 image; run under ``rocprofv3 --kernel-trace --stats`` to see which kernels the 256-token decode loop spends
its time in (weight-streaming GEMVs vs the ~40 small kernels per layer)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cap = bench.Captioner(dev)
img = ((bench.synthetic_image((1, 3, 4096, 4096), seed=1234, smooth=4) + 1) * 127.5).round().to(torch.uint8).to(dev)
for i in range(int(os.environ.get("REPS", 2))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cap(img, seed=42 + i)
    torch.cuda.synchronize()
    print(f"caption {i}: {time.perf_counter() - t0:.3f} s  {getattr(cap, 'breakdown', None)}", flush=True)
