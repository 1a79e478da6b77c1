"""d = 64 self-attention over a scan of (batch, heads, tokens): time per 64-key tile and workgroup round (what is fixed per workgroup?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsvld_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = [(2, 20, 16384), (2, 40, 16384), (4, 20, 16384), (1, 20, 16384), (2, 20, 8192), (2, 20, 32768), (2, 10, 65536), (1, 10, 65536)]
if os.environ.get('SCAN2'):
    SHAPES = [(2, 20, 16384), (1, 40, 16384), (3, 20, 16384), (2, 30, 16384), (2, 16, 16384), (2, 24, 16384), (2, 20, 16384)]
for (B, heads, N) in SHAPES:
    D = 64
    qkv = torch.randn(B, N, 3 * heads * D, device=dev, dtype=torch.float16)
    q, k, v = qkv[..., :heads * D], qkv[..., heads * D:2 * heads * D], qkv[..., 2 * heads * D:]
    for _ in range(2):
        ops.attention(q, k, v, heads)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = int(os.environ.get('REPS', 5))
    e0.record()
    for _ in range(reps):
        ops.attention(q, k, v, heads)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    wgs = (N // 512) * heads * B
    rounds = -(-wgs // 256)
    tiles = N // 64
    print(f"B{B} h{heads} N{N}: {ms*1e3:9.1f} us  {4.0*B*heads*N*N*D/ms/1e9:7.1f} TF/s  workgroups {wgs} = {wgs/256:.2f} rounds; {ms*1e3/(rounds*tiles):.3f} us per tile and round", flush=True)
