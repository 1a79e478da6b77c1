import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from rsvld_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
g = torch.Generator(device="cuda").manual_seed(1)
with ops.f32_split(True):
    for N in (65536, 262144):
        q = ops.to_planes(torch.randn(1, N, 512, device=dev, generator=g) * 0.5)
        x = ops.to_planes(torch.randn(1, N, 512, device=dev, generator=g) * 0.5)
        ms = t(lambda: ops.attention(q, x, x, heads=1, scale=512 ** -0.5), 2)
        print(f"split d512 fused N={N}: {ms:.1f} ms, {4.0 * N * N * 512 / ms / 1e9:.1f} TFLOP/s effective ({3 * 4.0 * N * N * 512 / ms / 1e9:.0f} MFMA)")
    for (B, h, N) in ((2, 20, 16384), (2, 10, 65536)):
        qkv = ops.to_planes(torch.randn(B, N, 3 * h * 64, device=dev, generator=g))
        HD = h * 64
        ms = t(lambda: ops.attention(qkv[..., :HD], qkv[..., HD:2 * HD], qkv[..., 2 * HD:], heads=h, scale=0.125), 3)
        print(f"split d64 B{B} h{h} N={N}: {ms:.1f} ms, {4.0 * B * h * N * N * 64 / ms / 1e9:.1f} TFLOP/s effective")
