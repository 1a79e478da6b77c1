"""GPU parity of every kernel behind the C ABI against a plain torch fp32 CPU reference of the
same op (inputs pre-rounded to the 16-bit storage type, so the tolerance covers fp32-accumulate
MFMA arithmetic + one output rounding)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16]


def _tol(dtype):
    return (4e-3, 4e-3) if dtype == torch.float16 else (3e-2, 3e-2)


def _close(got, want, dtype, scale=None):
    rt, at = _tol(dtype)
    want = want.float()
    got = got.float().cpu()
    s = float(want.abs().max()) if scale is None else scale
    err = float((got - want).abs().max())
    assert err <= at * max(s, 1e-6) + 1e-6, f"max|d|={err:.3e} scale={s:.3e}"
    return err


def _nhwc(x, dtype, dev):
    return x.permute(0, 2, 3, 1).contiguous().to(dev, dtype)


def _rt(x, dtype):  # round-trip through the storage type
    return x.to(dtype).float()


CONV_CASES = [
    # B, Cin, Cout, H, W, k, stride, pad, upsample
    (2, 64, 64, 24, 20, 3, 1, 1, False),
    (1, 8, 64, 16, 16, 3, 1, 1, False),      # first conv: 6 channels padded to 8
    (2, 64, 128, 17, 13, 3, 2, 1, False),    # stride 2, ragged M
    (1, 128, 256, 9, 11, 1, 1, 0, False),    # 1x1
    (1, 64, 8, 32, 32, 3, 1, 1, False),      # tiny Cout (final conv 64 -> 3 padded)
    (1, 256, 256, 6, 5, 3, 1, 1, True),      # fused nearest x2
    (1, 320, 640, 8, 8, 3, 1, 1, False),     # K = 2880 (not a multiple of 64 per tap)
    (1, 72, 40, 10, 10, 3, 1, 1, False),     # Cin, Cout multiples of 8 only
    (1, 128, 512, 64, 64, 3, 1, 1, False),   # 256 tiles of 64x128: two K groups per workgroup, 18 K-steps
    (1, 192, 512, 64, 64, 3, 1, 1, False),   # same, 27 K-steps (the second K group runs one step fewer)
    (1, 512, 128, 31, 33, 1, 1, 0, False),   # 64x64 tiles with two K groups, ragged M
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(cuda, dtype, case):
    from rsvld_amd import ops
    B, Cin, Cout, H, W, k, stride, pad, up = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = _rt(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = _rt(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), dtype)
    b = torch.randn(Cout, generator=g) * 0.1
    pc = ops.pack_conv(w, b, dtype, cuda)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    want = F.conv2d(xin, w, b, stride=stride, padding=pad)
    got = ops.conv2d(_nhwc(x, dtype, cuda), pc, stride=stride, pad=pad, upsample=up)
    assert got.shape == (B, want.shape[2], want.shape[3], Cout)
    _close(got.permute(0, 3, 1, 2), want, dtype)


HALO_CASES = [
    # B, C1, C2, Cout, H, W, fuse_norm
    (2, 64, 0, 64, 16, 32, True),       # exact 8x32 tiles, BN = 64
    (1, 128, 0, 128, 19, 45, True),     # ragged H and W, BN = 128
    (2, 128, 64, 256, 9, 33, True),     # two-source (skip concat) + fused GN whose groups straddle the sources
    (1, 320, 0, 320, 8, 16, True),      # W = 16 (half-filled tile), 5 channel chunks
    (1, 64, 0, 8, 24, 40, True),        # final conv: Cout 3 padded to 8
    (1, 192, 0, 64, 12, 64, False),     # plain 3x3 conv through the halo kernel (no norm)
    (2, 192, 0, 128, 10, 35, False),    # no norm, BN = 128 (double-buffered 32-channel patch pipeline), 3 bodies, batch 2
    (2, 512, 512, 128, 8, 32, True),    # 16 bodies across two sources: every steady-state prefetch slot is exercised
    (3, 64, 0, 128, 9, 31, True),       # single body (no prefetch beyond the prologue), batch 3
    (2, 64, 0, 512, 96, 128, True),     # >= 192 workgroups of 16x32 pixels: the 8-wave variant, exact tiles
    (1, 128, 64, 1024, 88, 100, True),  # 8-wave variant, ragged H (half a tile) and W, two sources, 3 bodies
    (1, 64, 0, 1024, 90, 97, False),    # 8-wave variant without a fused norm
    (2, 64, 64, 64, 17, 40, True),      # BN = 64, skip concat 128 -> 64 (Stage-1 level 0), 2 chunks
    (1, 256, 0, 48, 9, 70, True),       # BN = 64, Cout 48 (clamped weight rows), 4 chunks, three tile columns
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", HALO_CASES)
def test_conv3x3_halo_fused_groupnorm(cuda, dtype, case):
    """conv_halo.hip: GN(32)+SiLU applied while the input patch is staged, 9 taps from LDS; vs torch fp32."""
    from rsvld_amd import ops
    B, C1, C2, Cout, H, W, fuse = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    x1 = _rt(torch.randn(B, C1, H, W, generator=g) * 1.5 + 0.3, dtype)
    x2 = _rt(torch.randn(B, C2, H, W, generator=g), dtype) if C2 else None
    Ct = C1 + C2
    w = _rt(torch.randn(Cout, Ct, 3, 3, generator=g) / math.sqrt(Ct * 9), dtype)
    b = torch.randn(Cout, generator=g) * 0.1
    gamma, beta = 1 + 0.1 * torch.randn(Ct, generator=g), 0.1 * torch.randn(Ct, generator=g)
    rv = torch.randn(B, pad8 := (Cout + 7) // 8 * 8, generator=g)
    res = _rt(torch.randn(B, Cout, H, W, generator=g), dtype)
    xc = x1 if x2 is None else torch.cat([x1, x2], 1)
    xin = F.silu(F.group_norm(xc, 32, gamma, beta, eps=1e-5)) if fuse else xc
    want = F.conv2d(xin, w, b, padding=1) + rv[:, :Cout, None, None] + res
    pc = ops.pack_conv(w, b, dtype, cuda)
    resp = torch.zeros(B, H, W, pad8, dtype=dtype)
    resp[..., :Cout] = res.permute(0, 2, 3, 1)
    norm = (gamma.to(cuda), beta.to(cuda), 32, 1e-5, True) if fuse else None
    with ops.tuning(halo_min_wgs=0):   # force the halo kernel at test sizes
        got = ops.conv2d(_nhwc(x1, dtype, cuda), pc, x2=None if x2 is None else _nhwc(x2, dtype, cuda), pad=1,
                         rowvec=rv.to(cuda), residual=resp.to(cuda), norm=norm)
    _close(got[..., :Cout].permute(0, 3, 1, 2), want, dtype)
    # the same call through the gather kernel (unfused norm) must agree with the fused path
    with ops.tuning(use_halo=False):
        ref = ops.conv2d(_nhwc(x1, dtype, cuda), pc, x2=None if x2 is None else _nhwc(x2, dtype, cuda), pad=1,
                         rowvec=rv.to(cuda), residual=resp.to(cuda), norm=norm)
    _close(got.float(), ref.float().cpu(), dtype, scale=float(want.abs().max()))


@pytest.mark.parametrize("dtype", DTYPES)
def test_groupnorm_statistics_from_conv_epilogue(cuda, dtype):
    """Producer convs write per-tile per-channel (sum, sumsq) from their epilogue; the consumer's fused GroupNorm
    uses them instead of a pass over the tensor.  Two producers feed one consumer (skip concat, groups straddle)."""
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(21)
    B, H, W = 2, 19, 40
    xa = _rt(torch.randn(B, 64, H, W, generator=g), dtype)
    xb = _rt(torch.randn(B, 128, H, W, generator=g), dtype)
    wa = _rt(torch.randn(128, 64, 3, 3, generator=g) / 24, dtype)
    wb = _rt(torch.randn(64, 128, 3, 3, generator=g) / 34, dtype)
    wc = _rt(torch.randn(128, 192, 3, 3, generator=g) / 41, dtype)
    res = _rt(torch.randn(B, 128, H, W, generator=g), dtype)
    gamma, beta = 1 + 0.1 * torch.randn(192, generator=g), 0.1 * torch.randn(192, generator=g)
    with ops.tuning(halo_min_wgs=0):
        ya = ops.conv2d(_nhwc(xa, dtype, cuda), ops.pack_conv(wa, None, dtype, cuda), pad=1, residual=_nhwc(res, dtype, cuda), stats=True)
        yb = ops.conv2d(_nhwc(xb, dtype, cuda), ops.pack_conv(wb, None, dtype, cuda), pad=1, stats=True)
        assert hasattr(ya, "_gn_part") and hasattr(yb, "_gn_part")
        norm = (gamma.to(cuda), beta.to(cuda), 32, 1e-5, True)
        pcc = ops.pack_conv(wc, None, dtype, cuda)
        got = ops.conv2d(ya, pcc, x2=yb, pad=1, norm=norm)                       # statistics from the epilogues
        ya2, yb2 = ya.clone(), yb.clone()                                          # clones carry no partials
        ref = ops.conv2d(ya2, pcc, x2=yb2, pad=1, norm=norm)                     # statistics pass over the tensors
    cat = torch.cat([ya.float().cpu().permute(0, 3, 1, 2), yb.float().cpu().permute(0, 3, 1, 2)], 1)
    want = F.conv2d(F.silu(F.group_norm(cat, 32, gamma, beta, eps=1e-5)), wc, None, padding=1)
    _close(got.permute(0, 3, 1, 2), want, dtype)
    _close(got.float(), ref.float().cpu(), dtype, scale=float(want.abs().max()))


def test_groupnorm_statistics_from_8wave_conv_epilogue(cuda):
    """The 16x32-pixel (8-wave) halo variant writes its statistics on the same 8x32 partial grid as the 4-wave one
    (two sub-tiles per workgroup, the second one absent on a ragged last tile row): a consumer conv fed by it must
    agree with the statistics pass over the tensor."""
    from rsvld_amd import ops
    dtype = torch.float16
    g = torch.Generator().manual_seed(33)
    B, H, W = 2, 88, 100
    xa = _rt(torch.randn(B, 64, H, W, generator=g), dtype)
    wa = _rt(torch.randn(512, 64, 3, 3, generator=g) / 24, dtype)      # 192 workgroups of 16x32 -> 8-wave kernel
    wc = _rt(torch.randn(128, 512, 3, 3, generator=g) / 68, dtype)
    gamma, beta = 1 + 0.1 * torch.randn(512, generator=g), 0.1 * torch.randn(512, generator=g)
    with ops.tuning(halo_min_wgs=0):
        ya = ops.conv2d(_nhwc(xa, dtype, cuda), ops.pack_conv(wa, None, dtype, cuda), pad=1, stats=True)
        assert hasattr(ya, "_gn_part")
        norm = (gamma.to(cuda), beta.to(cuda), 32, 1e-5, True)
        pcc = ops.pack_conv(wc, None, dtype, cuda)
        got = ops.conv2d(ya, pcc, pad=1, norm=norm)            # statistics from the producer's epilogue
        ref = ops.conv2d(ya.clone(), pcc, pad=1, norm=norm)    # statistics pass over the tensor
    want = F.conv2d(F.silu(F.group_norm(ya.float().cpu().permute(0, 3, 1, 2), 32, gamma, beta, eps=1e-5)), wc, None, padding=1)
    _close(got.permute(0, 3, 1, 2), want, dtype)
    _close(got.float(), ref.float().cpu(), dtype, scale=float(want.abs().max()))


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_halo_fused_upsample(cuda, dtype):
    """nearest x2 folded into the halo patch staging (Upsample modules of all three networks)."""
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(9)
    x = _rt(torch.randn(2, 128, 9, 21, generator=g), dtype)
    w = _rt(torch.randn(128, 128, 3, 3, generator=g) / math.sqrt(128 * 9), dtype)
    b = torch.randn(128, generator=g) * 0.1
    want = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
    with ops.tuning(halo_min_wgs=0):
        got = ops.conv2d(_nhwc(x, dtype, cuda), ops.pack_conv(w, b, dtype, cuda), pad=1, upsample=True)
    assert got.shape == (2, 18, 42, 128)
    _close(got.permute(0, 3, 1, 2), want, dtype)


GEMM_CASES = [
    # M, K, N, act (0 none, 1 SiLU, 2 GEGLU), residual
    (4160, 320, 3840, 0, False),    # ragged M tile (64 rows), 10 K tiles, 255 workgroups
    (4096, 32, 3848, 0, True),      # a single K tile (prologue only), ragged N tile (8 columns), residual
    (8192, 64, 2560, 1, True),      # two K tiles, SiLU + residual
    (4224, 1280, 4096, 2, False),   # GEGLU pairs, 40 K tiles
    # the persistent form (K >= 128: one workgroup per CU walks its tiles)
    (4352, 256, 4104, 0, True),     # 17 x 17 = 289 tiles on 256 workgroups (33 of them take two), ragged N tile (8 columns), residual
    (8192, 128, 2560, 1, True),     # four K tiles (its minimum), SiLU + residual
    (65536, 640, 640, 0, True),     # 768 tiles = three per workgroup, N % 256 = 128, residual (Stage 2, level 2 to_out)
    (16640, 640, 1280, 2, False),   # GEGLU, 65 x 5 tiles, ragged M tile (Stage 2, level 2 feed-forward shape)
    (33024, 320, 768, 1, False),    # 129 x 3 = 387 tiles: runs of 48 / 49 tiles per XCD, SiLU without a residual
    # ... and its half tiles (the last round of an XCD's run cut into 128-row halves when it holds <= half as many tiles as the XCD has workgroups)
    (32768, 256, 1280, 0, True),    # 640 tiles = 2.5 rounds: every workgroup ends on a half tile; residual (Stage 2, level 3 to_out shape)
    (32896, 160, 1280, 2, False),   # 129 x 5 = 645 tiles: 80 / 81 per XCD = two rounds + 16 (halved) / + 17 (whole): both kinds in one launch; GEGLU;
                                    # the last row of tiles has 128 rows: its lower halves lie beyond M
    (9216, 192, 2560, 1, True),     # 360 tiles = 45 per XCD: one round + 13 halved tiles (26 of 32 workgroups get a half); SiLU + residual
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", GEMM_CASES)
def test_gemm256_linear(cuda, dtype, case):
    """csrc/gemm.hip: the 256x256 ping-pong GEMM that takes the large 1x1 / Linear layers; vs torch fp32 and vs the
    implicit-GEMM kernel on the same call (RSVLD_GEMM256_OFF is read once per process, so the cross-check is numeric)."""
    from rsvld_amd import ops, _lib as L
    M, K, N, act, use_res = case
    g = torch.Generator().manual_seed(M + K + N)
    x = _rt(torch.randn(M, K, generator=g), dtype)
    w = _rt(torch.randn(N, K, generator=g) / math.sqrt(K), dtype)
    b = torch.randn(N, generator=g) * 0.1
    res = _rt(torch.randn(M, N, generator=g), dtype) if use_res else None
    y = x @ w.t() + b
    if act == 1:
        y = F.silu(y)
    if act == 2:
        val, gate = y.chunk(2, dim=-1)
        y = val * F.gelu(gate)
    if use_res:
        y = y + res
    pc = ops.pack_conv(w, b, dtype, cuda, geglu=(act == 2))
    got = ops.linear(x.to(cuda, dtype), pc, residual=None if res is None else res.to(cuda, dtype),
                     act={0: L.ACT_NONE, 1: L.ACT_SILU, 2: L.ACT_GEGLU}[act])
    _close(got, y, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm256_persistent_vs_one_tile(cuda, dtype):
    """The persistent form against the one-tile form of the same kernel (RSVLD_TUNE_GEMM_ONE_TILE) and torch fp32: no bias, alpha / beta
    != 1 with a residual, GEGLU with alpha.  The two forms differ by one fp32 rounding (bias + sum against sum + bias): a 16-bit ulp
    on a few outputs at most."""
    from rsvld_amd import ops, _lib as L
    g = torch.Generator().manual_seed(77)
    for (M, K, N, act, use_res, use_bias, alpha, beta) in [(8448, 384, 2304, 0, True, True, 0.5, 2.0), (8192, 640, 2048, 0, False, False, 1.0, 1.0),
                                                            (8192, 320, 4096, 2, False, True, 0.75, 1.0)]:
        x = _rt(torch.randn(M, K, generator=g), dtype)
        w = _rt(torch.randn(N, K, generator=g) / math.sqrt(K), dtype)
        b = torch.randn(N, generator=g) * 0.1 if use_bias else None
        res = _rt(torch.randn(M, N, generator=g), dtype) if use_res else None
        y = x @ w.t() + (b if use_bias else 0.0)
        if act == 2:
            val, gate = y.chunk(2, dim=-1)
            y = val * F.gelu(gate)
        y = alpha * y + (beta * res if use_res else 0.0)
        pc = ops.pack_conv(w, b, dtype, cuda, geglu=(act == 2))
        kw = dict(residual=None if res is None else res.to(cuda, dtype), act={0: L.ACT_NONE, 2: L.ACT_GEGLU}[act], alpha=alpha, beta=beta)
        got = ops.linear(x.to(cuda, dtype), pc, **kw)
        with ops.tuning(tune=ops.context().tune | L.TUNE_GEMM_ONE_TILE):
            one = ops.linear(x.to(cuda, dtype), pc, **kw)
        _close(got, y, dtype)
        _close(one, y, dtype)
        d = (got.float() - one.float()).abs()
        # one 16-bit ulp of the larger of the output and the product term (with a residual both forms round alpha * (x W^T + b) to 16
        # bits before the fp32 add, as the reference's own 16-bit execution does: they may differ by an ulp of THAT term)
        mag = torch.maximum(y.abs(), (y - (beta * res if use_res else 0.0)).abs()).to(cuda)
        ulp = (2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7) * mag.clamp_min(2.0 ** -14)
        # (+ 2^-14: an fp16-subnormal output may be flushed by one form's conversion instruction (v_fma_mixlo_f16) and kept by the
        #  other's (v_cvt_f16_f32): measured 0 against 1.2e-5 in the GEGLU case; and an output that is the small difference of O(1)
        #  terms moves by the fp32 rounding of those terms, not of itself)
        assert bool((d <= (2.002 if use_res else 1.001) * ulp + 2.0 ** -14).all()), float((d / ulp).max())   # (term + sum roundings)
        assert float((d > 0).float().mean()) < 0.02


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv2d_epilogue_rowvec_residual_concat(cuda, dtype):
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(5)
    B, C1, C2, Cout, H, W = 3, 64, 32, 128, 12, 10
    x1 = _rt(torch.randn(B, C1, H, W, generator=g), dtype)
    x2 = _rt(torch.randn(B, C2, H, W, generator=g), dtype)
    w = _rt(torch.randn(Cout, C1 + C2, 3, 3, generator=g) / math.sqrt((C1 + C2) * 9), dtype)
    b = torch.randn(Cout, generator=g) * 0.1
    rv_full = torch.randn(B, 2 * Cout, generator=g)        # strided row vector (slice of a wider table)
    res = _rt(torch.randn(B, Cout, H, W, generator=g), dtype)
    pc = ops.pack_conv(w, b, dtype, cuda, cin_split=(C1, C2))
    rv_dev = rv_full.to(cuda)
    got = ops.conv2d(_nhwc(x1, dtype, cuda), pc, x2=_nhwc(x2, dtype, cuda), rowvec=rv_dev[:, Cout:],
                     residual=_nhwc(res, dtype, cuda), alpha=0.5, beta=2.0)
    want = 0.5 * (F.conv2d(torch.cat([x1, x2], 1), w, b, padding=1) + rv_full[:, Cout:, None, None]) + 2.0 * res
    _close(got.permute(0, 3, 1, 2), want, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv2d_asymmetric_pad_and_f32_out(cuda, dtype):
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(6)
    x = _rt(torch.randn(1, 64, 16, 16, generator=g), dtype)
    w = _rt(torch.randn(64, 64, 3, 3, generator=g) / 24.0, dtype)
    pc = ops.pack_conv(w, None, dtype, cuda)
    # VAE downsample: pad (0,1,0,1) then stride-2 conv (model.py:81-85)
    want = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, None, stride=2)
    got = ops.conv2d(_nhwc(x, dtype, cuda), pc, stride=2, pad=(0, 0, 1, 1))
    _close(got.permute(0, 3, 1, 2), want, dtype)
    w3 = _rt(torch.randn(3, 64, 3, 3, generator=g) / 24.0, dtype)
    b3 = torch.randn(3, generator=g)
    pc3 = ops.pack_conv(w3, b3, dtype, cuda)
    got = ops.conv2d(_nhwc(x, dtype, cuda), pc3, pad=1, out_f32=True)
    assert got.dtype == torch.float32 and got.shape[-1] == 8
    _close(got[..., :3].permute(0, 3, 1, 2), F.conv2d(x, w3, b3, padding=1), dtype)
    assert float(got[..., 3:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_and_geglu(cuda, dtype):
    from rsvld_amd import ops, _lib as L
    g = torch.Generator().manual_seed(7)
    rows, Cin, Cout = 200, 320, 640
    x = _rt(torch.randn(2, rows // 2, Cin, generator=g), dtype)
    w = _rt(torch.randn(Cout, Cin, generator=g) / math.sqrt(Cin), dtype)
    b = torch.randn(Cout, generator=g) * 0.1
    res = _rt(torch.randn(2, rows // 2, Cout, generator=g), dtype)
    got = ops.linear(x.to(cuda, dtype), ops.pack_conv(w, b, dtype, cuda), residual=res.to(cuda, dtype))
    _close(got, F.linear(x, w, b) + res, dtype)
    # GEGLU fused in the epilogue (attention.py:84-96)
    got = ops.linear(x.to(cuda, dtype), ops.pack_conv(w, b, dtype, cuda, geglu=True), act=L.ACT_GEGLU)
    y = F.linear(x, w, b)
    val, gate = y.chunk(2, dim=-1)
    _close(got, val * F.gelu(gate), dtype)
    # stand-alone geglu kernel
    got = ops.geglu(_rt(y, dtype).to(cuda, dtype))
    v2, g2 = _rt(y, dtype).chunk(2, dim=-1)
    _close(got, v2 * F.gelu(g2), dtype)


GN_CASES = [(2, 64, 0, 16, 12, 32, True), (1, 320, 0, 9, 7, 32, False), (3, 512, 256, 6, 6, 32, True),
            (1, 2560, 0, 4, 4, 32, True), (2, 128, 0, 40, 40, 32, True), (1, 960, 0, 5, 5, 32, True),
            # one-launch path for small tensors (one workgroup per image x group, slab held in registers)
            (4, 512, 0, 32, 32, 32, True), (2, 512, 512, 17, 19, 32, True), (4, 512, 0, 64, 64, 32, False),
            (1, 1024, 0, 9, 9, 32, True)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", GN_CASES)
def test_group_norm(cuda, dtype, case):
    from rsvld_amd import ops
    B, C1, C2, H, W, G, silu = case
    g = torch.Generator().manual_seed(C1 + C2)
    x1 = _rt(torch.randn(B, C1, H, W, generator=g) * 2 + 0.5, dtype)
    x2 = _rt(torch.randn(B, C2, H, W, generator=g), dtype) if C2 else None
    gamma = 1 + 0.1 * torch.randn(C1 + C2, generator=g)
    beta = 0.1 * torch.randn(C1 + C2, generator=g)
    xc = x1 if x2 is None else torch.cat([x1, x2], 1)
    want = F.group_norm(xc, G, gamma, beta, eps=1e-5)
    if silu:
        want = F.silu(want)
    got = ops.group_norm(_nhwc(x1, dtype, cuda), gamma.to(cuda), beta.to(cuda), G, 1e-5,
                         x2=None if x2 is None else _nhwc(x2, dtype, cuda), silu=silu)
    _close(got.permute(0, 3, 1, 2), want, dtype)
    st = ops.group_norm_stats(_nhwc(x1, dtype, cuda), G, x2=None if x2 is None else _nhwc(x2, dtype, cuda)).cpu()
    xg = xc.view(B, G, -1)
    assert torch.allclose(st[..., 0], xg.mean(-1), atol=1e-4, rtol=1e-4)
    assert torch.allclose(st[..., 1], xg.var(-1, unbiased=False), atol=1e-4, rtol=1e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_group_norm_zerosft_modulation(cuda, dtype):
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(3)
    B, Cc, H, W = 2, 320, 6, 6
    x = _rt(torch.randn(B, Cc, H, W, generator=g), dtype)
    sc = _rt(torch.randn(B, Cc, H, W, generator=g) * 0.3, dtype)
    sh = _rt(torch.randn(B, Cc, H, W, generator=g) * 0.3, dtype)
    want = F.group_norm(x, 32, None, None, eps=1e-5) * (1 + sc) + sh
    got = ops.group_norm(_nhwc(x, dtype, cuda), None, None, 32, 1e-5, mod_scale1p=_nhwc(sc, dtype, cuda),
                         mod_shift=_nhwc(sh, dtype, cuda))
    _close(got.permute(0, 3, 1, 2), want, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Cc", [640, 1280, 320])
def test_layer_norm(cuda, dtype, Cc):
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(Cc)
    x = _rt(torch.randn(3, 37, Cc, generator=g) * 1.5 + 0.3, dtype)
    gamma, beta = 1 + 0.1 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    got = ops.layer_norm(x.to(cuda, dtype), gamma.to(cuda), beta.to(cuda), 1e-5)
    _close(got, F.layer_norm(x, (Cc,), gamma, beta, 1e-5), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(1, 64, 64), (2, 144, 144), (1, 1, 1), (1, 100, 333), (2, 1024, 1024), (24, 1000, 160),
                                   (1, 300, 2500)])   # split-KV (4 and 9 key ranges), no split, ragged last range
def test_attention_d512(cuda, dtype, shape):
    from rsvld_amd import ops
    B, Nq, Nk = shape
    D = 512
    g = torch.Generator().manual_seed(Nq * 7 + Nk)
    qkv = _rt(torch.randn(B, max(Nq, Nk), 3 * D, generator=g), dtype)
    q, k, v = qkv[:, :Nq, :D], qkv[:, :Nk, D:2 * D], qkv[:, :Nk, 2 * D:]
    scale = 1.0 / math.sqrt(D)
    want = torch.softmax(q @ k.transpose(1, 2) * scale, -1) @ v
    dq = qkv.to(cuda, dtype)
    got = ops.attention(dq[:, :Nq, :D], dq[:, :Nk, D:2 * D], dq[:, :Nk, 2 * D:], heads=1, scale=scale)
    _close(got, want, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(1, 64, 64), (2, 144, 144), (1, 1, 1), (1, 100, 333), (2, 1024, 1024), (3, 1000, 160),
                                   (1, 300, 2500), (1, 2048, 4096), (1, 33, 95)])
def test_attention_d512_shared_kv_tile(cuda, dtype, shape):
    """Keys and values are ONE tensor (k is v): the shared-tile instantiation -- one LDS image read row-wise for S and
    transposed for P V, three-buffer ring (1, 2, 3+ tiles; ragged tails; split-KV ranges)."""
    from rsvld_amd import ops
    B, Nq, Nk = shape
    D = 512
    g = torch.Generator().manual_seed(Nq * 5 + Nk)
    q = _rt(torch.randn(B, Nq, D, generator=g), dtype)
    x = _rt(torch.randn(B, Nk, D, generator=g), dtype)
    scale = 1.0 / math.sqrt(D)
    want = torch.softmax(q @ x.transpose(1, 2) * scale, -1) @ x
    dx = x.to(cuda, dtype)
    got = ops.attention(q.to(cuda, dtype), dx, dx, heads=1, scale=scale)
    _close(got, want, dtype)
    # and it agrees with the two-tensor kernel on the same data (k and v as separate copies)
    ref = ops.attention(q.to(cuda, dtype), dx, dx.clone(), heads=1, scale=scale)
    assert float((got.float() - ref.float()).abs().max()) < (2e-3 if dtype == torch.float16 else 1.6e-2)


@pytest.fixture(params=["b", "c", "p"])
def d64_kernel(request):
    """Both d = 64 kernels on every case: ``b`` (four waves per SIMD) and ``c`` (ping-pong, 512 query rows per workgroup), which
    the library otherwise chooses between by the number of query rows (csrc/attention.hip; rsvld_amd.devtools.d64_kernel)."""
    from rsvld_amd import devtools
    devtools.d64_kernel(request.param)
    yield request.param
    devtools.d64_kernel("")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 10, 256, 256), (1, 20, 64, 77), (2, 5, 100, 333), (1, 10, 4096, 4096), (3, 2, 1, 1),
                                   (1, 2, 700, 64), (1, 1, 513, 129), (1, 2, 1100, 192)])
def test_attention_d64_multihead(cuda, dtype, shape, d64_kernel):
    """SDXL self / cross attention and ZeroCrossAttn: heads x d=64, keys 77 (text), ragged, 4096 tokens; one, two, three and
    four key tiles (the ping-pong kernel's prologue requests three and its ring holds four)."""
    from rsvld_amd import ops
    B, heads, Nq, Nk = shape
    D = 64
    g = torch.Generator().manual_seed(Nq + 3 * Nk + heads)
    q = _rt(torch.randn(B, Nq, heads * D, generator=g), dtype)
    kv = _rt(torch.randn(B, Nk, 2 * heads * D, generator=g), dtype)       # fused k|v projection output
    k, v = kv[..., :heads * D], kv[..., heads * D:]
    qh, kh, vh = (t.view(B, -1, heads, D).transpose(1, 2) for t in (q, k, v))
    want = (torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, -1) @ vh).transpose(1, 2).reshape(B, Nq, heads * D)
    dkv = kv.to(cuda, dtype)
    got = ops.attention(q.to(cuda, dtype), dkv[..., :heads * D], dkv[..., heads * D:], heads=heads)
    _close(got, want, dtype)


def test_attention_d64_rescale_path(cuda, d64_kernel):
    """A late key dominates one query row (running max jumps by >> 2^8 in a late tile) and another row's max creeps
    up by less than the deferral threshold: both branches of the deferred-max logic."""
    from rsvld_amd import ops
    dtype, heads, D, N = torch.float16, 2, 64, 320
    g = torch.Generator().manual_seed(8)
    q = torch.randn(1, N, heads * D, generator=g)
    k = torch.randn(1, N, heads * D, generator=g)
    v = torch.randn(1, N, heads * D, generator=g)
    k[0, 300, :D] = q[0, 17, :D] * 4.0
    k[0, 150, D:] = q[0, 40, D:] * 0.5
    q, k, v = _rt(q, dtype), _rt(k, dtype), _rt(v, dtype)
    qh, kh, vh = (t.view(1, N, heads, D).transpose(1, 2) for t in (q, k, v))
    want = (torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, -1) @ vh).transpose(1, 2).reshape(1, N, heads * D)
    got = ops.attention(q.to(cuda, dtype), k.to(cuda, dtype), v.to(cuda, dtype), heads=heads)
    _close(got, want, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("offset", [-40.0, 0.0, 60.0])
def test_attention_d64_bias_step_extremes(cuda, dtype, offset, d64_kernel):
    """The running maximum enters the score MFMA chain as a 16-bit bias operand (csrc/attention.hip, A6B_BIAS): scores far
    below zero on the FIRST tile (the bias starts at 0 and must move down), far above it, a maximum that keeps growing by
    less than the deferral threshold per tile and then jumps, and a ragged last tile -- all against an fp64 softmax."""
    from rsvld_amd import ops
    heads, D, Nq, Nk = 3, 64, 96, 200
    g = torch.Generator().manual_seed(int(offset) + 77)
    q = torch.randn(1, Nq, heads * D, generator=g)
    k = torch.randn(1, Nk, heads * D, generator=g)
    v = torch.randn(1, Nk, heads * D, generator=g)
    # a common component along one direction shifts every score of head 0 / 1 by ~offset (after the 1/8 scale)
    u = torch.zeros(heads * D)
    u[:D] = 1.0 / 8.0
    q = q + u * 8.0
    k = k + u * (offset * 8.0)
    k[0, 190, D:2 * D] = q[0, 5, D:2 * D] * 3.0          # late spike in the last (ragged) tile, head 1
    # head 1, query 9: a key in tile 1 scores ~7 nats (2^10) above everything before it -- below the 2^14 row-sum limit of the
    # sum-checked softmax, so the tile is NOT redone and P, O, l carry values up to 2^10 until a later rescale
    k[0, 100, D:2 * D] = q[0, 9, D:2 * D] * (7.0 * 8.0 / float((q[0, 9, D:2 * D] ** 2).sum()))
    for j in range(Nk):                                  # head 2: the maximum of query 7 creeps up tile by tile
        k[0, j, 2 * D:] += q[0, 7, 2 * D:] * (0.02 * j / 8.0)
    q, k, v = _rt(q, dtype), _rt(k, dtype), _rt(v, dtype)
    qh, kh, vh = (t.double().view(1, -1, heads, D).transpose(1, 2) for t in (q, k, v))
    want = (torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, -1) @ vh).transpose(1, 2).reshape(1, Nq, heads * D).float()
    got = ops.attention(q.to(cuda, dtype), k.to(cuda, dtype), v.to(cuda, dtype), heads=heads)
    assert bool(torch.isfinite(got).all())
    _close(got, want, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 3, 1500, 1000), (1, 2, 4096, 2048 + 17), (1, 1, 64, 640), (1, 2, 300, 128), (1, 1, 257, 191), (1, 2, 256, 70)])
def test_attention_d64_pingpong_equals_four_wave_kernel_bit_for_bit(cuda, dtype, shape, monkeypatch):
    """attn_d64c re-schedules attn_d64b's arithmetic (same MFMA chains, same exponentials, row-sum order, sum check and redo):
    the outputs are EQUAL, so which kernel a shape gets is not a numerics decision (batch-invariant by construction). A late
    spike forces the redo path in a late tile."""
    from rsvld_amd import ops
    B, heads, Nq, Nk = shape
    D = 64
    g = torch.Generator().manual_seed(Nq + Nk)
    q = torch.randn(B, Nq, heads * D, generator=g)
    k = torch.randn(B, Nk, heads * D, generator=g)
    v = torch.randn(B, Nk, heads * D, generator=g)
    k[0, Nk - 5, :D] = q[0, 33, :D] * 4.0
    q, k, v = (t.to(cuda, dtype) for t in (q, k, v))
    outs = {}
    from rsvld_amd import devtools
    try:
        for kern in ("b", "c", "p"):
            devtools.d64_kernel(kern)
            outs[kern] = ops.attention(q, k, v, heads=heads)
    finally:
        devtools.d64_kernel("")
    assert bool(torch.isfinite(outs["c"]).all()) and bool(torch.isfinite(outs["p"]).all())
    assert torch.equal(outs["b"], outs["c"])
    assert torch.equal(outs["b"], outs["p"])


def test_attention_online_softmax_rescale_path(cuda):
    """Force the running max to jump at a late key tile (cdna_hip_programming.md rule 26)."""
    from rsvld_amd import ops
    dtype, D, N = torch.float16, 512, 256
    g = torch.Generator().manual_seed(1)
    q = torch.randn(1, N, D, generator=g) * 0.5
    k = torch.randn(1, N, D, generator=g) * 0.5
    v = torch.randn(1, N, D, generator=g)
    k[0, 200] = q[0, 17] * 3.0      # one key spikes against one query, in tile 6
    k[0, 70] = q[0, 40] * 2.0
    q, k, v = _rt(q, dtype), _rt(k, dtype), _rt(v, dtype)
    scale = 1.0 / math.sqrt(D)
    want = torch.softmax(q @ k.transpose(1, 2) * scale, -1) @ v
    got = ops.attention(q.to(cuda, dtype), k.to(cuda, dtype), v.to(cuda, dtype), heads=1, scale=scale)
    _close(got, want, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(4096, 4096), (6144, 4096), (28672, 4096), (4096, 14336), (1003, 520), (17, 8)])
def test_gemv_weight_streaming(cuda, dtype, shape):
    """rsvld_gemv (the caption pass's decode-step products: Llama-3-8B q|k|v, o, gate|up, down shapes, ragged N and K) against
    an fp64 matrix-vector product of the same 16-bit operands."""
    from rsvld_amd import ops
    N, K = shape
    g = torch.Generator().manual_seed(N + K)
    w = _rt(torch.randn(N, K, generator=g) / K ** 0.5, dtype)
    x = _rt(torch.randn(K, generator=g), dtype)
    b = _rt(torch.randn(N, generator=g) * 0.1, dtype)
    want = (w.double() @ x.double()).float()
    got = ops.gemv(w.to(cuda, dtype), x.to(cuda, dtype))
    _close(got, want, dtype)
    got = ops.gemv(w.to(cuda, dtype), x.to(cuda, dtype), b.to(cuda, dtype))
    _close(got, want + b, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(6144, 4096), (4096, 14336), (1003, 520)])
def test_gemv_fused_decode_step_neighbours(cuda, dtype, shape):
    """rsvld_gemv_fused: the RMSNorm prologue, the SwiGLU prologue and the residual epilogue of a Llama decode step, each against the unfused
    torch sequence on the SAME 16-bit operands (F.rms_norm -> linear; silu(g) * u -> linear; h + linear): equal up to the order of one fp32 sum."""
    from rsvld_amd import ops
    N, K = shape
    g = torch.Generator().manual_seed(N + K)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(cuda, dtype)
    x = (torch.randn(K, generator=g) * 2).to(cuda, dtype)
    nw = (torch.randn(K, generator=g) * 0.2 + 1).to(cuda, dtype)
    res = torch.randn(N, generator=g).to(cuda, dtype)
    gu = torch.randn(2 * K, generator=g).to(cuda, dtype)
    lin = lambda v: (w.double() @ v.double()).to(dtype)        # the product's fp32 accumulation, rounded once
    want_n = lin(F.rms_norm(x, (K,), nw, 1e-5))
    want_r = (res + lin(x))
    want_g = (res + lin(F.silu(gu[:K]) * gu[K:]))
    _close(ops.gemv_fused(w, x, None, norm=(nw, 1e-5)), want_n.float().cpu(), dtype)
    _close(ops.gemv_fused(w, x, None, residual=res), want_r.float().cpu(), dtype)
    _close(ops.gemv_fused(w, gu, None, glu=True, residual=res), want_g.float().cpu(), dtype)
    assert torch.equal(ops.gemv_fused(w, x, None), ops.gemv(w, x))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("pos", [0, 5, 255, 256, 700])
def test_llama_decode_attention_vs_torch_sequence(cuda, dtype, pos):
    """rsvld_llama_decode_attention (rotary embedding of the new q / k, cache write at a DEVICE position, grouped-query attention over the
    filled prefix) against the torch-op sequence of FastDecoder.forward for ONE new token: 8 query heads on 2 kv heads, head_dim 128,
    positions on both sides of the 256-key chunk boundary."""
    from rsvld_amd import ops
    nq, nkv, hd, max_len = 8, 2, 128, 777
    g = torch.Generator().manual_seed(pos + 1)
    qkv = torch.randn(nq + 2 * nkv, hd, generator=g).to(cuda, dtype)
    kc = torch.randn(nkv, max_len, hd, generator=g).to(cuda, dtype)
    vc = torch.randn(nkv, max_len, hd, generator=g).to(cuda, dtype)
    ang = torch.rand(hd // 2, generator=g) * 6
    cos, sin = torch.cat([ang.cos(), ang.cos()]).to(cuda, dtype), torch.cat([ang.sin(), ang.sin()]).to(cuda, dtype)
    p = torch.tensor([pos], device=cuda)
    # the unfused sequence (llava_next.FastDecoder.forward, T = 1)
    k_ref, v_ref = kc.clone(), vc.clone()
    qk = qkv[:nq + nkv]
    half = hd // 2
    qk = qk * cos + torch.cat((-qk[..., half:], qk[..., :half]), dim=-1) * sin
    k_ref[:, pos] = qk[nq:]
    v_ref[:, pos] = qkv[nq + nkv:]
    gq = nq // nkv
    q = qk[:nq].reshape(nkv, gq, hd)
    sc = torch.matmul(q, k_ref.transpose(1, 2)).float() * hd ** -0.5
    sc[..., pos + 1:] = float("-inf")
    want = torch.matmul(torch.softmax(sc, dim=-1).to(dtype), v_ref).reshape(nq * hd)
    got = ops.llama_decode_attention(qkv.reshape(-1), cos, sin, p, kc, vc, nq, nkv, hd ** -0.5)
    assert torch.equal(kc[:, pos], k_ref[:, pos]) and torch.equal(vc[:, pos], v_ref[:, pos])      # the cache rows as the sequence writes them
    assert torch.equal(kc[:, :pos], k_ref[:, :pos]) and torch.equal(kc[:, pos + 1:], k_ref[:, pos + 1:])
    _close(got, want.float().cpu(), dtype)


def test_small_layers_and_embeddings(cuda):
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(4, 320, generator=g)
    w, b = torch.randn(1280, 320, generator=g) / 18, torch.randn(1280, generator=g)
    got = ops.linear_small(x.to(cuda), w.to(cuda), b.to(cuda), 1, 1).cpu()
    assert torch.allclose(got, F.silu(F.linear(F.silu(x), w, b)), atol=1e-4, rtol=1e-4)
    lv = torch.tensor([0.1, 0.7, 0.999])
    import oracle.sr3_oracle as O
    got = ops.sinusoidal(lv.to(cuda), 64, 0).cpu()
    assert torch.allclose(got, O.positional_encoding(lv, 64), atol=2e-6)
    t = torch.tensor([0.0, 19.0, 999.0])
    half = 160
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None] * freqs[None]
    want = torch.cat([torch.cos(args), torch.sin(args)], -1)
    got = ops.sinusoidal(t.to(cuda), 320, 1).cpu()
    assert torch.allclose(got, want, atol=2e-4)


def test_layout_and_ddpm_step(cuda):
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 10, 12, generator=g)
    c = torch.randn(2, 3, 10, 12, generator=g)
    buf = ops.nchw_to_nhwc(c.to(cuda), torch.float16)
    assert buf.shape == (2, 10, 12, 8) and float(buf[..., 3:].abs().max()) == 0.0
    buf = torch.zeros((2, 10, 12, 8), device=cuda, dtype=torch.float16)
    ops.nchw_to_nhwc(c.to(cuda), torch.float16, c_off=0, out=buf)
    ops.nchw_to_nhwc(x.to(cuda), torch.float16, c_off=3, out=buf)
    want = torch.cat([c, x, torch.zeros(2, 2, 10, 12)], 1).permute(0, 2, 3, 1).half()
    assert torch.equal(buf.cpu(), want)
    back = ops.nhwc_to_nchw(buf, channels=3, c_off=3).cpu()
    assert torch.equal(back, x.half().float())
    eps = torch.randn(2, 10, 12, 8, generator=g)
    nz = torch.randn(2, 3, 10, 12, generator=g)
    a, bq, c1, c2, s = 1.3, 0.8, 0.4, 0.6, 0.05
    x0 = (a * x - bq * eps[..., :3].permute(0, 3, 1, 2)).clamp(-1, 1)
    want = c1 * x0 + c2 * x + s * nz
    got = ops.ddpm_step(x.to(cuda), eps.to(cuda), nz.to(cuda), a, bq, c1, c2, s).cpu()
    assert torch.allclose(got, want, atol=1e-6)
    got = ops.ddpm_step(x.to(cuda), eps.to(cuda), None, a, bq, c1, c2, 0.0).cpu()
    assert torch.allclose(got, c1 * x0 + c2 * x, atol=1e-6)
    a16, b16 = torch.randn(64, generator=g).half(), torch.randn(64, generator=g).half()
    got = ops.axpby(a16.to(cuda), b16.to(cuda), 0.5, 2.0).cpu()
    assert torch.allclose(got.float(), (0.5 * a16.float() + 2 * b16.float()), atol=2e-3)


def test_errors_are_loud(cuda):
    from rsvld_amd import ops, _lib as L
    with pytest.raises(L.RsvldError):
        ops.group_norm(torch.zeros(1, 4, 4, 64, dtype=torch.float16), None, None, 32, 1e-5)  # CPU tensor
    x = torch.zeros(1, 4, 4, 64, dtype=torch.float16, device=cuda)
    with pytest.raises(L.RsvldError):
        ops.group_norm(x, None, None, 48, 1e-5)  # 64 % 48 != 0 -> RSVLD_EINVAL
    q = torch.zeros(1, 8, 96, dtype=torch.float16, device=cuda)
    with pytest.raises(L.RsvldError):
        ops.attention(q, q, q, heads=1)  # D = 96 unsupported


def test_hand_scheduled_kernels_are_deterministic(cuda):
    """Race screen for the kernels whose synchronisation is counted by hand (asm loads left in flight over raw barriers,
    LDS-DMA rings, ping-pong wave groups): the same launch repeated 8 times on a busy chip must give bit-identical
    output -- a missing wait or barrier shows up as run-to-run differences long before it fails a tolerance test."""
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(123)
    dt = torch.float16

    def repeat(fn, n=8):
        ref = fn().clone()
        for _ in range(n - 1):
            assert torch.equal(fn(), ref)

    # d = 512 attention: split-KV (2048 keys -> 4 ranges) and a long single-range run
    qkv = (torch.randn(2, 2048, 1536, generator=g) * 0.5).to(cuda, dt)
    repeat(lambda: ops.attention(qkv[..., :512], qkv[..., 512:1024], qkv[..., 1024:], heads=1))
    qkv2 = (torch.randn(26, 900, 1536, generator=g) * 0.5).to(cuda, dt)
    repeat(lambda: ops.attention(qkv2[..., :512], qkv2[..., 512:1024], qkv2[..., 1024:], heads=1))
    # d = 64 attention, ragged keys
    q = torch.randn(2, 3000, 640, generator=g).to(cuda, dt)
    kv = torch.randn(2, 2777, 1280, generator=g).to(cuda, dt)
    repeat(lambda: ops.attention(q, kv[..., :640], kv[..., 640:], heads=10))
    # 256x256 GEMM (ping-pong groups), ragged M
    x = torch.randn(8200, 1280, generator=g).to(cuda, dt)
    pc = ops.pack_conv(torch.randn(3840, 1280, generator=g) / 36, torch.zeros(3840), dt, cuda)
    repeat(lambda: ops.linear(x, pc))
    # double-buffered halo conv with the fused norm prologue, 8 bodies
    xi = (torch.randn(2, 40, 72, 512, generator=g)).to(cuda, dt)
    pcc = ops.pack_conv(torch.randn(256, 512, 3, 3, generator=g) / 68, None, dt, cuda)
    norm = (torch.ones(512, device=cuda), torch.zeros(512, device=cuda), 32, 1e-5, True)
    with ops.tuning(halo_min_wgs=0):
        repeat(lambda: ops.conv2d(xi, pcc, pad=1, norm=norm))
