"""Go / no-go for Winograd F(2x2, 3x3) on the DEEP 3x3 layers of the Stage-2 UNets in the split precision, measured with the
kernels the product already has -- no Winograd kernel is written unless this says it can pay.

Direct form (today): conv_halo_128_split, 3 bf16 MFMAs per product, 9 taps.
Winograd form: V = B^T d B (16 positions per 2x2 output tile, 4x the input elements, fp32 transform then hi | lo planes), 16 position-GEMMs
[tiles x Cin] x [Cin x Cout] in the split precision (gemm256 SEG = 3), M = A^T (.) A with bias / row vector / residual / statistics.
MFMA work: 16 / (4 * 9) = 1 / 2.25 of the direct form.  Extra HBM passes per layer (none of them exists in the direct form):
    V written (planes, 4 B / element) by the input transform, read by the GEMMs:   2 x 4 x (4 x B H W Cin)  bytes
    M written (fp32) by the GEMMs, read by the output transform:                  2 x 4 x (4 x B H W Cout) bytes
The tool times, per headline shape: (a) the direct kernel; (b) the 16 position-GEMMs -- as 16 launches of the product's split GEMM at
M = B H W / 4 rows, and as ONE launch over 16 x as many rows (an upper bound for a batched launch: same tiles, one weight matrix); (c) a
streaming pass (rsvld_split_planes, 8 B per element) moving the bytes of the two passes the GEMMs' time does not hold: V written, M read.
Winograd time >= (b) + (c); go only if direct / that >= 1.3 (VERDICT's bar) -- and only then would the 50-step full-depth test be due."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rsvld_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
REPS = int(os.environ.get("REPS", 10))
# (B = the CFG pair, H, W, Cin, Cin2, Cout): the deep 3x3 layers of juggernautXL at latent 512 (openaimodel.py:207-350)
SHAPES = [(2, 256, 256, 640, 0, 640), (2, 256, 256, 640, 640, 640), (2, 128, 128, 1280, 0, 1280), (2, 128, 128, 1280, 1280, 1280)]


def timed(fn, prefix):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    prof = ops.LaunchProfiler()
    with ops.tuning(profiler=prof):
        for _ in range(REPS):
            fn()
    agg = prof.summary()
    return sum(r["ms"] for k, r in agg.items() if k.startswith(prefix)) / REPS, sorted(k for k in agg if k.startswith(prefix))


for (B, H, W, C1, C2, Co) in SHAPES:
    C = C1 + C2
    x = ops.to_planes(torch.randn(B, H, W, C1, device=dev))
    x2 = ops.to_planes(torch.randn(B, H, W, C2, device=dev)) if C2 else None
    pc = ops.pack_conv(torch.randn(Co, C, 3, 3) / (3 * C ** 0.5), torch.zeros(Co), torch.float32, dev, cin_split=(C1, C2) if C2 else None)
    with ops.f32_split(ops.ALL_SPLIT):
        t_dir, names = timed(lambda: ops.conv2d(x, pc, x2=x2, stats=True), "conv_")
    tiles = B * (H // 2) * (W // 2)
    pg = ops.pack_conv(torch.randn(Co, C) / C ** 0.5, None, torch.float32, dev)
    v1 = ops.to_planes(torch.randn(tiles, C, device=dev))
    v16 = ops.to_planes(torch.randn(16 * tiles, C, device=dev))
    with ops.f32_split(ops.ALL_SPLIT):
        t_g1, gn1 = timed(lambda: ops.linear(v1, pg), "gemm_")
        t_g16, gn16 = timed(lambda: ops.linear(v16, pg), "gemm_")
    if not gn1:      # below gemm256's grid threshold the layer runs on the implicit-GEMM kernel
        with ops.f32_split(ops.ALL_SPLIT):
            t_g1, gn1 = timed(lambda: ops.linear(v1, pg), "conv_")
    # V is written by the input transform and M is read by the output transform (their other sides -- V read, M written -- are inside the
    # GEMMs' time): 4 B x 4 B H W (Cin + Cout).  A split_planes pass moves 8 B per element, so it runs over half as many elements.
    extra_elems = 4 * B * H * W * (C + Co)
    src = torch.randn(extra_elems // 2, device=dev).view(-1, 64)
    t_pass, _ = timed(lambda: ops.to_planes(src), "split_planes")
    wino_16 = 16 * t_g1 + t_pass
    wino_1 = t_g16 + t_pass
    fl = 2.0 * B * H * W * Co * C * 9
    print(f"B{B} {H}x{W} Cin {C1}+{C2} Cout {Co}: direct {t_dir * 1e3:7.0f} us ({3 * fl / t_dir / 1e9:6.0f} MFMA-TF/s, {names})", flush=True)
    print(f"    16 position-GEMMs [{tiles} x {C}] x [{C} x {Co}]: 16 launches {16 * t_g1 * 1e3:7.0f} us ({gn1}); one launch over 16x the rows "
          f"{t_g16 * 1e3:7.0f} us ({3 * 2.0 * 16 * tiles * C * Co / t_g16 / 1e9:6.0f} MFMA-TF/s)")
    print(f"    the four extra tensor passes (V written, M read: {4 * extra_elems / 2 ** 30:.2f} GiB): {t_pass * 1e3:7.0f} us ({4 * extra_elems / (t_pass * 1e-3) / 1e9:6.0f} GB/s)")
    print(f"    Winograd >= {wino_1 * 1e3:7.0f} us (batched bound) / {wino_16 * 1e3:7.0f} us (16 launches)  ->  direct / Winograd <= "
          f"{t_dir / wino_1:.2f} / {t_dir / wino_16:.2f}   (bar: 1.3)", flush=True)
    del x, x2, v1, v16, src
    torch.cuda.empty_cache()
