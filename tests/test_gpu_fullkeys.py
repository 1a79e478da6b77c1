"""An INDEPENDENT fp32 reference at the metric's full key length: the code paths that only exist at size --
the range-major split-KV of the d = 512 attention at 262 144 keys (+ its combine pass), the >= 1 024-workgroup grid of the ping-pong
d = 64 kernel, the 16-row halo tiles (NW = 8) on a 4096-pixel-wide map -- compared with plain torch fp32 on the host, not with
another kernel of this library.  The device runs the FULL launch (all query rows, the un-forced dispatch); the host computes a
256-row subset of the queries against ALL keys (0.14 TFLOP) / a cropped band of the convolution.
Replaces the einsum + softmax of sr3_modules/unet.py:133-141, xformers.ops.memory_efficient_attention at
sgm/modules/attention.py:357-359 and the 3x3 convolutions of unet.py:81-92.  Bounds: the kernel-test tolerance, 4e-3 x range (fp16)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 4e-3


def _host_attention_rows(q_rows, k, v, scale):
    """q_rows [R, D], k / v [N, D] (fp32 on the host) -> softmax(scale q k^T) v, in chunks of keys (online merge in fp64 weights)"""
    s = (q_rows @ k.t()) * scale                     # [R, N] fp32: 256 x 262 144 = 268 MB
    p = torch.softmax(s, dim=-1)
    return p @ v


@pytest.mark.parametrize("shared", [False, True])
def test_attention_d512_262144_keys_vs_host_fp32(cuda, shared):
    """Stage 1 at 4096^2, level 3: 262 144 queries x 262 144 keys, single head d = 512 -- range-major split-KV (64 Ki-key ranges)
    + combine; two-tensor and shared-tile instantiations."""
    from rsvld_amd import ops
    N, D = 262144, 512
    g = torch.Generator(device="cuda").manual_seed(17 + int(shared))
    # (amplitude 1.5: scores of standard deviation 2.25, i.e. a softmax that a few hundred of the 262 144 keys dominate -- with
    #  near-uniform weights every output would be the mean of V and any kernel would pass)
    q = (torch.randn(1, N, D, device=cuda, generator=g) * 1.5).half()
    k = (torch.randn(1, N, D, device=cuda, generator=g) * 1.5).half()
    v = k if shared else torch.randn(1, N, D, device=cuda, generator=g).half()
    out = ops.attention(q, k, v, heads=1, scale=D ** -0.5)
    rows = torch.randint(0, N, (256,), device=cuda, generator=g)
    want = _host_attention_rows(q[0, rows].float().cpu(), k[0].float().cpu(), v[0].float().cpu(), D ** -0.5)
    got = out[0, rows].float().cpu()
    e, r = float((got - want).abs().max()), float(want.abs().max())
    print(f"d512 attention, 262 144 keys, shared={shared}: 256 sampled rows vs host fp32: max|d| = {e:.3e} (range {r:.3f})")
    assert r > 0.3 and e <= TOL * r


def test_attention_d64_65536_tokens_pingpong_grid_vs_host_fp32(cuda):
    """Stage 2 at latent 512, first transformer level: 65 536 tokens; 8 heads x 128 workgroups = the 1 024-workgroup grid from which
    the un-forced dispatch runs the ping-pong kernel attn_d64c (rsvld_attention: tune = 0)."""
    from rsvld_amd import ops
    N, heads, D = 65536, 8, 64
    g = torch.Generator(device="cuda").manual_seed(23)
    qkv = (torch.randn(1, N, 3 * heads * D, device=cuda, generator=g) * 1.7).half()    # scores of standard deviation ~2.9: a peaked softmax
    HD = heads * D
    q, k, v = qkv[..., :HD], qkv[..., HD:2 * HD], qkv[..., 2 * HD:]
    out = ops.attention(q, k, v, heads=heads, scale=D ** -0.5)
    rows = torch.randint(0, N, (256,), device=cuda, generator=g)
    qc, kc, vc = q[0, rows].float().cpu(), k[0].float().cpu(), v[0].float().cpu()
    worst, rng = 0.0, 0.0
    for h in range(heads):
        sl = slice(h * D, (h + 1) * D)
        want = _host_attention_rows(qc[:, sl].contiguous(), kc[:, sl].contiguous(), vc[:, sl].contiguous(), D ** -0.5)
        got = out[0, rows][:, sl].float().cpu()
        worst, rng = max(worst, float((got - want).abs().max())), max(rng, float(want.abs().max()))
    print(f"d64 attention, 65 536 tokens x 8 heads (ping-pong grid): 256 sampled rows vs host fp32: max|d| = {worst:.3e} (range {rng:.3f})")
    assert rng > 0.5 and worst <= TOL * rng


@pytest.mark.parametrize("cin,cout,H,W,norm", [(192, 128, 512, 4096, False),   # conv_halo32, NW = 8: 16 x 32-pixel tiles (>= 192 of them, K >= 192)
                                               (64, 64, 256, 4096, True)])    # conv_halo_64 with the fused GroupNorm + SiLU prologue
def test_conv_halo_4096_wide_band_vs_host_fp32(cuda, cin, cout, H, W, norm):
    """A full-resolution-width map: the device convolves the whole tensor, the host a band of 16 output rows (one NW = 8 tile row
    that does not start on the image border) across all 4096 columns."""
    from rsvld_amd import ops
    g = torch.Generator(device="cuda").manual_seed(cin + H)
    x = torch.randn(1, H, W, cin, device=cuda, generator=g).half()
    w = (torch.randn(cout, cin, 3, 3, device=cuda, generator=g) / math.sqrt(9 * cin))
    b = torch.randn(cout, device=cuda, generator=g) * 0.1
    pc = ops.pack_conv(w, b, torch.float16, cuda)
    gamma, beta = torch.randn(cin, device=cuda, generator=g), torch.randn(cin, device=cuda, generator=g)
    out = ops.conv2d(x, pc, pad=1, norm=(gamma, beta, 32, 1e-5, True) if norm else None)
    y0 = 48                                                        # band of output rows [48, 64): input rows [47, 65)
    xin = x[0].float()
    if norm:                                                       # GroupNorm statistics are over the WHOLE map: computed on the device in fp32
        xg = xin.reshape(H * W, 32, cin // 32)
        mean = xg.mean(dim=(0, 2), keepdim=True)
        var = xg.var(dim=(0, 2), unbiased=False, keepdim=True)
        xin = F.silu(((xg - mean) / torch.sqrt(var + 1e-5)).reshape(H, W, cin) * gamma + beta)
    band = xin[y0 - 1:y0 + 17].permute(2, 0, 1)[None].cpu()        # [1, cin, 18, W]
    want = F.conv2d(band, w.float().cpu(), b.float().cpu(), padding=(0, 1))   # rows padded by the band itself, columns by zeros
    got = out[0, y0:y0 + 16].permute(2, 0, 1)[None].float().cpu()
    e, r = float((got - want).abs().max()), float(want.abs().max())
    print(f"conv 3x3 {cin}->{cout} on {H}x{W}{' + fused GN/SiLU' if norm else ''}: 16-row band vs host fp32: max|d| = {e:.3e} (range {r:.2f})")
    assert e <= TOL * max(r, 1.0)
