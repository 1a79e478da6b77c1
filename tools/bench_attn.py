"""Times rsvld_attention on the Stage-1 (d = 512, one head) and Stage-2 (d = 64, multi-head) shapes.
RSVLD_LIB=<another build of librsvld_hip.so> selects a different library for A/B runs (ONLY512 / ONLY64 filter the list)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsvld_amd import _lib, ops

if os.environ.get("RSVLD_LIB"):      # an OLDER build may lack entry points added since: bind only what it exports (this tool calls rsvld_attention only)
    import ctypes
    _old = ctypes.CDLL(os.environ["RSVLD_LIB"])
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if hasattr(_old, k)}

dev = torch.device("cuda:0")
torch.manual_seed(0)
reps = int(os.environ.get("REPS", 10))
CASES = [(4, 1, 4096, 512), (4, 1, 1024, 512), (1, 1, 16384, 512), (1, 1, 65536, 512),
         (2, 10, 4096, 64), (2, 20, 1024, 64), (2, 10, 16384, 64), (2, 20, 16384, 64), (1, 10, 32768, 64)]
if os.environ.get("HEADLINE"):   # the attention shapes of the 512 -> 4096 headline workload (S1 at 4096^2, S2 at latent 512)
    CASES = [(1, 1, 65536, 512), (1, 1, 262144, 512), (2, 20, 16384, 64), (2, 10, 65536, 64)]
    reps = int(os.environ.get("REPS", 3))
if os.environ.get("ONLY512"):
    CASES = [c for c in CASES if c[3] == 512]
if os.environ.get("ONLY64"):
    CASES = [c for c in CASES if c[3] == 64]
print("library:", os.environ.get("RSVLD_LIB", "in-tree build"))
NK = int(os.environ.get("CROSS", 0))     # CROSS=77: cross-attention against that many keys (the text tokens) instead of self-attention
for (B, heads, N, D) in CASES:
    qkv = torch.randn(B, N, 3 * heads * D, device=dev, dtype=torch.float16)
    q, k, v = qkv[..., :heads * D], qkv[..., heads * D:2 * heads * D], qkv[..., 2 * heads * D:]
    if NK:
        k, v = k[:, :NK], v[:, :NK]
    if os.environ.get("SHARED") and D == 512:     # keys = values = one tensor: the shared-tile instantiation (Stage 1)
        v = k
    for _ in range(2):
        o = ops.attention(q, k, v, heads)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        o = ops.attention(q, k, v, heads)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 4.0 * B * heads * N * (NK or N) * D
    print(f"B{B} heads{heads} N{N} D{D}: {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
