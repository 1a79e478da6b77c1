"""GPU parity of the split-operand PRODUCT path (round 4: csrc/split.hip, the RSVLD_SPLIT instantiations of gemm256 / conv_halo /
conv_igemm, the fp32-input GroupNorm / LayerNorm that write planes) against plain torch on the host in fp64 -- every kernel through
the C ABI (rsvld_amd.ops).  What the mode must reproduce: the reference's CPU path, which runs without autocast
(models/SR_model.py:28-33,57-85; sgm/modules/diffusionmodules/wrappers.py:84-110).
Tolerance: three bf16 products per fp32 product, fp32 accumulation -> ~1e-5 relative per product; asserted at 1e-4 of the tensor's
range (measured figures are printed), the bound test_gpu_f32.py uses for round 3's on-the-fly split kernels."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
REL = 1e-4


def _all_split():
    """The kernel-level bounds of this file are those of the split KERNELS (three MFMAs per product, the split attention kernels):
    policy ``ops.ALL_SPLIT``.  The compositions that hand some layer inputs over in fp16 (``ops.UNET_POLICY``: the 16-bit attention
    kernels, the weight-pair form RSVLD_F16W2) have their own tests at the bottom."""
    from rsvld_amd import ops
    return ops.f32_split(ops.ALL_SPLIT)


def _cmp(got, want, rel, what):
    got = got.double().cpu()
    want = want.double()
    s = float(want.abs().max())
    e = float((got - want).abs().max())
    print(f"{what}: max|d| = {e:.3e} (range {s:.2f}, rel {e / max(s, 1e-30):.2e})")
    assert e <= rel * max(s, 1e-6), f"{what}: max|d| = {e:.3e}, range {s:.3e}"
    return e


def _nhwc(x, dev):
    return x.permute(0, 2, 3, 1).contiguous().to(dev)


def test_split_merge_round_trip_and_weight_triples(cuda):
    from rsvld_amd import ops, _lib as L
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(37, 5, 72, generator=g) * torch.logspace(-6, 4, 72)).to(cuda)
    pl = ops.to_planes(x)
    assert pl.shape == (37, 5, 72) and pl.t.shape == (37, 5, 2, 72) and pl.t.dtype == torch.bfloat16
    hi, lo = pl.t[..., 1, :].float(), pl.t[..., 0, :].float()
    assert torch.equal(hi, x.bfloat16().float()) and torch.equal(lo, (x - hi).bfloat16().float())
    back = pl.f32()
    assert float(((back - x).abs() / x.abs().clamp_min(1e-30)).max()) < 2.0 ** -16
    # weight triples: [Cout][taps][hi | lo | hi]
    w = torch.randn(24, 3 * 3 * 16, generator=g).to(cuda)
    w3 = torch.empty(24, 9 * 48, device=cuda, dtype=torch.bfloat16)
    L.check(L.load().rsvld_split_pack_weights(ops._ptr(w), ops._ptr(w3), 24, 9, 16, ops._stream()), "pack")
    w3 = w3.view(24, 9, 3, 16).float()
    wr = w.view(24, 9, 16)
    h = wr.bfloat16().float()
    assert torch.equal(w3[:, :, 0], h) and torch.equal(w3[:, :, 2], h) and torch.equal(w3[:, :, 1], (wr - h).bfloat16().float())


LINEAR_CASES = [
    # rows, K, N, residual, geglu, out_planes          (gemm256 needs rows >= 4096, N >= 256, >= 128 tiles, K % 32 == 0)
    (8192, 640, 1920, False, False, True),      # fused q|k|v -> planes
    (8192 + 77, 1280, 1280, True, False, False),  # to_out with the residual, ragged rows
    (16384, 320, 2560, False, True, True),      # GEGLU feed-forward -> planes
    (16384, 1280, 320, True, False, False),     # N = 320: two 256-wide tiles, the second mostly padding
    (4096, 5120, 1280, True, False, False),     # long K
    (300, 640, 640, True, False, False),        # small M: the implicit-GEMM kernel
    (1000, 72, 40, False, False, True),         # K, N multiples of 8 only
    (2048, 640, 5120, False, True, False),      # GEGLU on the implicit-GEMM kernel, fp32 out
]


@pytest.mark.parametrize("case", LINEAR_CASES)
def test_linear_split(cuda, case):
    from rsvld_amd import _lib as L, ops
    rows, K, N, use_res, geglu, out_planes = case
    g = torch.Generator().manual_seed(rows + K + N)
    x = torch.randn(rows, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    y = x.double() @ w.double().t() + b.double()
    if geglu:
        y = y[:, :N // 2] * F.gelu(y[:, N // 2:])
    res = torch.randn(y.shape, generator=g) if use_res else None
    if use_res:
        y = y * 0.5 + res.double()
    pc = ops.pack_conv(w, b, torch.float32, cuda, geglu=geglu)
    xd = x.to(cuda)
    with _all_split():
        got_p = ops.linear(ops.to_planes(xd), pc, residual=None if res is None else res.to(cuda), act=L.ACT_GEGLU if geglu else L.ACT_NONE,
                           alpha=0.5 if use_res else 1.0, out_planes=out_planes)
        got_f = ops.linear(xd, pc, residual=None if res is None else res.to(cuda), act=L.ACT_GEGLU if geglu else L.ACT_NONE,
                           alpha=0.5 if use_res else 1.0)
    if out_planes:
        assert isinstance(got_p, ops.Planes) and got_p.shape == tuple(y.shape)
        got_p = got_p.f32()
    _cmp(got_p, y, REL, f"linear split (planes in) {case}")
    _cmp(got_f, y, REL, f"linear split (fp32 in)   {case}")
    # the round-3 implementation of the same arithmetic (on-the-fly split inside the fp32 family): an independent check
    with ops.f32_split(ops.SplitPolicy(impl="f32", f16_inputs=())):
        ref = ops.linear(xd, pc, residual=None if res is None else res.to(cuda), act=L.ACT_GEGLU if geglu else L.ACT_NONE,
                         alpha=0.5 if use_res else 1.0)
    _cmp(got_f, ref.cpu(), REL, "  planes path vs the fp32 family's split kernel")


CONV_CASES = [
    # B, Cin, Cin2, Cout, H, W, k, stride, pad, upsample, residual, rowvec, stats
    # (the halo kernels take grids of >= 256 workgroups of 8 x 32 pixels x 128 channels; smaller maps run the gather kernel)
    (1, 64, 0, 64, 250, 260, 3, 1, 1, False, True, True, True),      # conv_halo_64 (full-resolution SR3 layers), ragged tiles
    (2, 128, 0, 128, 128, 250, 3, 1, 1, False, True, True, True),    # conv_halo32, NW = 4
    (1, 192, 64, 128, 512, 256, 3, 1, 1, False, False, False, True),  # two-source (skip concat), NW = 8 (K >= 192, >= 192 16-row tiles)
    (1, 256, 0, 256, 100, 128, 3, 1, 1, True, False, False, False),  # nearest x2 folded into the halo patch
    (1, 64, 0, 64, 96, 128, 3, 1, 1, False, True, True, False),      # small map: the gather kernel with the same epilogue
    (2, 128, 0, 128, 33, 41, 3, 2, (0, 0, 1, 1), False, False, False, False),   # Downsample: stride 2 (implicit GEMM), ragged
    (1, 8, 0, 128, 40, 40, 3, 1, 1, False, False, False, False),     # conv_in: 3 channels padded to 8
    (1, 128, 0, 8, 64, 64, 3, 1, 1, False, False, False, False),     # conv_out
    (1, 320, 0, 640, 24, 24, 1, 1, 0, False, True, False, False),    # 1x1 skip connection with residual
    (1, 72, 48, 40, 20, 20, 3, 1, 1, False, True, True, False),      # multiples of 8 only, two sources on the gather kernel
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_split(cuda, case):
    from rsvld_amd import ops
    B, Cin, Cin2, Cout, H, W, k, stride, pad, up, use_res, use_rv, stats = case
    g = torch.Generator().manual_seed(abs(hash(case)) % 1000)
    cin_real = 3 if Cin == 8 else Cin
    cout_real = 3 if Cout == 8 else Cout
    x = torch.randn(B, cin_real, H, W, generator=g)
    x2 = torch.randn(B, Cin2, H, W, generator=g) if Cin2 else None
    ct = cin_real + Cin2
    w = torch.randn(cout_real, ct, k, k, generator=g) / math.sqrt(ct * k * k)
    b = torch.randn(cout_real, generator=g) * 0.1
    pc = ops.pack_conv(w, b, torch.float32, cuda, cin_split=(cin_real, Cin2) if Cin2 else None)
    xin = x if x2 is None else torch.cat([x, x2], 1)
    if up:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    xin, wd, bd = xin.double(), w.double(), b.double()
    if isinstance(pad, tuple):
        pt, pl, pb, pr = pad
        want = F.conv2d(F.pad(xin, (pl, pr, pt, pb)), wd, bd, stride=stride)
    else:
        want = F.conv2d(xin, wd, bd, stride=stride, padding=pad)
    rv = torch.randn(B, cout_real, generator=g) if use_rv else None
    if use_rv:
        want = want + rv.double()[:, :, None, None]
    res = torch.randn(want.shape, generator=g) if use_res else None
    if use_res:
        want = want + res.double()
    xd = ops.nchw_to_nhwc(x.to(cuda), torch.float32)
    x2d = None if x2 is None else ops.nchw_to_nhwc(x2.to(cuda), torch.float32)
    rvd = None
    if use_rv:
        rvd = torch.zeros(B, pc.cout_p, device=cuda)
        rvd[:, :cout_real] = rv.to(cuda)
    with _all_split():
        got = ops.conv2d(xd, pc, x2=x2d, stride=stride, pad=pad, upsample=up, rowvec=rvd, residual=None if res is None else _nhwc(res, cuda),
                         stats=stats)
    assert got.dtype == torch.float32 and got.shape == (B, want.shape[2], want.shape[3], pc.cout_p)
    _cmp(got[..., :cout_real].permute(0, 3, 1, 2), want, REL, f"conv split {case}")
    if cout_real != pc.cout_p:
        assert float(got[..., cout_real:].abs().max()) == 0.0
    if stats:   # (grids large enough for the halo kernels by construction) the epilogue's per-tile (sum, sum of squares) of the stored tensor, merged: what the next GroupNorm consumes
        part, ntiles = got._gn_part
        assert part.shape == (B, ntiles, pc.cout_p, 2)
        tot = part.double().sum(1).cpu()
        _cmp(tot[..., 0], want.sum((2, 3)), 1e-5 * want[0, 0].numel() ** 0.5, "  epilogue partials: sum")
        _cmp(tot[..., 1], (want * want).sum((2, 3)), 1e-5, "  epilogue partials: sum of squares")


@pytest.mark.parametrize("shape,silu,two", [((2, 128, 19, 23), True, False), ((1, 64, 70, 66), True, True), ((1, 640, 32, 32), False, False),
                                            ((1, 1920, 16, 16), True, True)])
def test_group_norm_split(cuda, shape, silu, two):
    """fp32 NHWC in -> fp32 and planes out; two sources = the skip concat, never materialised; conv2d(norm=) through the same kernels"""
    from rsvld_amd import ops
    B, Cc, H, W = shape
    g = torch.Generator().manual_seed(Cc + H)
    x = torch.randn(shape, generator=g) * 3.0 + 0.7
    gamma, beta = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    want = F.group_norm(x.double(), 32, gamma.double(), beta.double(), eps=1e-6)
    if silu:
        want = F.silu(want)
    c1 = Cc // 2 if two else Cc
    xd = _nhwc(x[:, :c1], cuda)
    x2d = _nhwc(x[:, c1:], cuda) if two else None
    with _all_split():
        got = ops.group_norm(xd, gamma.to(cuda), beta.to(cuda), 32, 1e-6, x2=x2d, silu=silu)
        gp = ops.group_norm(xd, gamma.to(cuda), beta.to(cuda), 32, 1e-6, x2=x2d, silu=silu, planes=True)
    assert got.dtype == torch.float32 and isinstance(gp, ops.Planes)
    _cmp(got.permute(0, 3, 1, 2), want, 2e-5, f"group_norm split {shape} fp32 out")
    _cmp(gp.f32().permute(0, 3, 1, 2), want, 3e-5, f"group_norm split {shape} planes out")


def test_group_norm_split_modulated(cuda):
    """ZeroSFT: norm(x) * (1 + gamma) + beta with gamma | beta two channel slices of one stacked conv output (SR_modules.py:100-106)"""
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(7)
    B, Cc, H, W = 2, 320, 16, 24
    x = torch.randn(B, Cc, H, W, generator=g) * 2 + 0.3
    gb = torch.randn(B, H, W, 2 * Cc, generator=g) * 0.5
    want = F.group_norm(x.double(), 32, None, None, eps=1e-5) * (1 + gb[..., :Cc].permute(0, 3, 1, 2).double()) + gb[..., Cc:].permute(0, 3, 1, 2).double()
    gbd = gb.to(cuda)
    with _all_split():
        got = ops.group_norm(_nhwc(x, cuda), None, None, 32, 1e-5, mod_scale1p=gbd[..., :Cc], mod_shift=gbd[..., Cc:])
    _cmp(got.permute(0, 3, 1, 2), want, 2e-5, "group_norm split, modulated")


@pytest.mark.parametrize("rows,Cc", [(1000, 640), (4099, 1280), (77, 320), (50, 2048)])
def test_layer_norm_split(cuda, rows, Cc):
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, Cc, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    want = F.layer_norm(x.double(), (Cc,), gamma.double(), beta.double(), 1e-5)
    with _all_split():
        got = ops.layer_norm(x.to(cuda), gamma.to(cuda), beta.to(cuda), 1e-5)
        gp = ops.layer_norm(x.to(cuda), gamma.to(cuda), beta.to(cuda), 1e-5, planes=True)
    _cmp(got, want, 1e-5, f"layer_norm split {rows}x{Cc} fp32 out")
    _cmp(gp.f32(), want, 2e-5, f"layer_norm split {rows}x{Cc} planes out")


def _attn_ref(q, k, v, heads, scale):
    B, Nq, HD = q.shape
    D = HD // heads
    qh = q.double().view(B, Nq, heads, D).transpose(1, 2)
    kh = k.double().view(B, -1, heads, D).transpose(1, 2)
    vh = v.double().view(B, -1, heads, D).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1)
    return (p @ vh).transpose(1, 2).reshape(B, Nq, HD)


@pytest.mark.parametrize("B,heads,Nq,Nk,peaky", [(2, 5, 300, 300, False), (1, 10, 1024, 1024, True), (2, 20, 200, 77, False), (1, 3, 129, 65, True),
                                                 (1, 2, 4096, 4096, False)])
def test_attention_split_d64(cuda, B, heads, Nq, Nk, peaky):
    """the fused split flash kernel; q | k | v are column slices of ONE fused projection in planes (token stride 2 * 3 * heads * 64),
    the cross-attention form takes k | v from a second tensor; ``peaky``: large logits (one key dominates a row)"""
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(Nq + Nk)
    HD = heads * 64
    amp = 3.0 if peaky else 1.0
    q = torch.randn(B, Nq, HD, generator=g) * amp
    k = torch.randn(B, Nk, HD, generator=g) * amp
    v = torch.randn(B, Nk, HD, generator=g)
    want = _attn_ref(q, k, v, heads, 0.125)
    if Nq == Nk:
        qkv = ops.to_planes(torch.cat([q, k, v], -1).to(cuda))
        qp, kp, vp = qkv[..., :HD], qkv[..., HD:2 * HD], qkv[..., 2 * HD:]
    else:
        qp = ops.to_planes(q.to(cuda))
        kv = ops.to_planes(torch.cat([k, v], -1).to(cuda))
        kp, vp = kv[..., :HD], kv[..., HD:]
    with _all_split():
        got = ops.attention(qp, kp, vp, heads=heads, scale=0.125)
        got32 = ops.attention(q.to(cuda), k.to(cuda), v.to(cuda), heads=heads, scale=0.125)   # fp32 in: split on demand
    assert isinstance(got, ops.Planes) and got.shape == (B, Nq, HD)
    _cmp(got.f32(), want, REL, f"attention split d64 B{B} h{heads} {Nq}x{Nk}")
    _cmp(got32.f32(), want, REL, "  fp32 inputs")


@pytest.mark.parametrize("B,Nq,Nk,D,shared", [(1, 200, 200, 512, True), (2, 64, 64, 128, False), (1, 33, 1000, 512, False), (1, 4096, 4096, 512, True),
                                              (1, 5000, 333, 512, False)])
def test_attention_split_gemm_form(cuda, B, Nq, Nk, D, shared):
    """single-head attention (SR3 / VAE: d = 512) as S = Q K^T -> row softmax -> P V on the split GEMMs, keys padded to 8"""
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(Nq + Nk + D)
    q = torch.randn(B, Nq, D, generator=g) * 0.5
    k = torch.randn(B, Nk, D, generator=g) * 0.5
    v = k if shared else torch.randn(B, Nk, D, generator=g)
    want = _attn_ref(q, k, v, 1, D ** -0.5)
    kd = ops.to_planes(k.to(cuda))
    with _all_split():
        got = ops.attention(ops.to_planes(q.to(cuda)), kd, kd if shared else ops.to_planes(v.to(cuda)), heads=1, scale=D ** -0.5)
    _cmp(got.f32(), want, REL, f"attention split (GEMM form) d{D} {Nq}x{Nk} shared={shared}")


def test_transformer_block_split_planes_vs_fp32_family(cuda):
    """one BasicTransformerBlock-shaped chain through the product path (LayerNorm -> planes -> q|k|v -> fused attention -> to_out +
    residual -> LayerNorm -> GEGLU -> planes -> linear + residual) against the fp32-operand family on the same device"""
    from rsvld_amd import _lib as L, ops
    g = torch.Generator().manual_seed(11)
    N, Cc, heads = 4096, 640, 10
    x = (torch.randn(1, N, Cc, generator=g)).to(cuda)
    def lin(o, i):
        return ops.pack_conv(torch.randn(o, i, generator=g) / math.sqrt(i), torch.randn(o, generator=g) * 0.1, torch.float32, cuda)
    wqkv, wo, w2 = lin(3 * Cc, Cc), lin(Cc, Cc), lin(Cc, 4 * Cc)
    wg = ops.pack_conv(torch.randn(8 * Cc, Cc, generator=g) / math.sqrt(Cc), torch.randn(8 * Cc, generator=g) * 0.1, torch.float32, cuda, geglu=True)
    ga, be = torch.ones(Cc, device=cuda), torch.zeros(Cc, device=cuda)

    def run(planes):
        n = ops.layer_norm(x, ga, be, planes=planes)
        qkv = ops.linear(n, wqkv, out_planes=planes)
        o = ops.attention(qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:], heads=heads, scale=0.125)
        h = ops.linear(o, wo, residual=x)
        n = ops.layer_norm(h, ga, be, planes=planes)
        gg = ops.linear(n, wg, act=L.ACT_GEGLU, out_planes=planes)
        return ops.linear(gg, w2, residual=h)

    want = run(False).cpu()                      # fp32 family
    with _all_split():
        got = run(True)
    _cmp(got, want, REL, "transformer block: split product path vs fp32 family")


@pytest.mark.parametrize("B,Nq,Nk", [(1, 64, 64), (2, 200, 333), (1, 4096, 4096), (1, 1000, 8200)])
def test_attention_split_d512_fused(cuda, B, Nq, Nk):
    """the fused split kernel for ONE head of d = 512 with keys = values = the same planes tensor (SR3's re-associated SelfAttention):
    two waves share 32 query rows, each contracting half of the head dimension; against fp64 on the host and the GEMM form"""
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(Nq + Nk)
    q = torch.randn(B, Nq, 512, generator=g) * 0.7
    x = torch.randn(B, Nk, 512, generator=g) * 0.7
    want = _attn_ref(q, x, x, 1, 512 ** -0.5)
    qp, xp = ops.to_planes(q.to(cuda)), ops.to_planes(x.to(cuda))
    with _all_split():
        with ops.tuning(split_d512_fused_min=1):
            got = ops.attention(qp, xp, xp, heads=1, scale=512 ** -0.5)
        with ops.tuning(split_d512_fused_min=1 << 30):
            ref = ops.attention(qp, xp, xp, heads=1, scale=512 ** -0.5)
    _cmp(got.f32(), want, REL, f"attention split d512 fused B{B} {Nq}x{Nk}")
    _cmp(got.f32(), ref.f32().cpu(), REL, "  fused vs GEMM form")


@pytest.mark.parametrize("B,heads,Nq,Nk", [(2, 20, 4096, 4096), (1, 16, 16384 + 77, 16384 + 77), (1, 40, 2048, 300)])
def test_attention_split_d64_pingpong_vs_four_wave(cuda, B, heads, Nq, Nk):
    """the ping-pong form (8 waves, anti-phase groups, three-stage ring; an experiment behind ``out_f32`` bit 2, see csrc/split.hip)
    against the 4-wave kernel -- bit for bit -- and fp64 on the host for sampled rows; ragged query and key counts"""
    from rsvld_amd import ops, _lib as L
    g = torch.Generator(device="cuda").manual_seed(Nq + heads)
    HD = heads * 64
    qkv = torch.randn(B, max(Nq, Nk), 3 * HD, device=cuda, generator=g) * 1.3
    pl = ops.to_planes(qkv)
    q, k, v = pl[:, :Nq, :HD], pl[:, :Nk, HD:2 * HD], pl[:, :Nk, 2 * HD:]
    with _all_split():
        ref = ops.attention(q, k, v, heads=heads, scale=0.125)          # the library's choice: the 4-wave kernel
    got_t = torch.empty_like(ref.t)
    lib = L.load()
    L.check(lib.rsvld_attention_split_d64(ops._ptr(q.t), ops._ptr(k.t), ops._ptr(v.t), ops._ptr(got_t), B, heads, Nq, Nk,
                                          q.t.stride(0), q.t.stride(1), q.t.stride(2), k.t.stride(0), k.t.stride(1), k.t.stride(2),
                                          v.t.stride(0), v.t.stride(1), v.t.stride(2), got_t.stride(0), got_t.stride(1), got_t.stride(2), 0.125, 4,
                                          ops._stream()), "ping-pong kernel (developer override)")
    got = ops.Planes(got_t)
    assert torch.equal(got_t, ref.t), "the ping-pong form must agree with the 4-wave kernel bit for bit"
    rows = torch.randint(0, Nq, (64,), generator=torch.Generator().manual_seed(1))
    want = _attn_ref(qkv[:1, rows, :HD].cpu(), qkv[:1, :Nk, HD:2 * HD].cpu(), qkv[:1, :Nk, 2 * HD:].cpu(), heads, 0.125)
    _cmp(got.f32()[:1, rows.to(cuda)], want, REL, "  sampled rows vs fp64")


@pytest.mark.parametrize("B,heads,Nq,Nk,D,shared", [(2, 5, 1024, 1024, 64, False), (1, 20, 300, 77, 64, False), (1, 1, 2304, 2304, 512, True),
                                                    (1, 1, 1000, 1000, 512, False)])
def test_attention_split_mode_f16_composition(cuda, B, heads, Nq, Nk, D, shared):
    """Policy group "attn" (without "attn_out"): q | k | v leave the planes as fp16 (rsvld_planes_to_f16, channel slices of a fused
    planes tensor read in place), the 16-bit attention kernels run, the fp16 result returns as planes (rsvld_f16_to_planes, exact);
    with "attn_out" (the UNets' default policy) the fp16 result itself is returned for the weight-pair form of ``to_out``.
    Against fp64 on the host at the 16-bit kernels' tolerance, and against the same kernels called on fp16(fp32) operands."""
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(Nq + D)
    HD = heads * D
    q = torch.randn(B, Nq, HD, generator=g)
    k = torch.randn(B, Nk, HD, generator=g)
    v = k if shared else torch.randn(B, Nk, HD, generator=g)
    want = _attn_ref(q, k, v, heads, D ** -0.5)
    if Nq == Nk and not shared:
        qkv = ops.to_planes(torch.cat([q, k, v], -1).to(cuda))
        qp, kp, vp = qkv[..., :HD], qkv[..., HD:2 * HD], qkv[..., 2 * HD:]
    else:
        qp, kp = ops.to_planes(q.to(cuda)), ops.to_planes(k.to(cuda))
        vp = kp if shared else ops.to_planes(v.to(cuda))
    with ops.f32_split(ops.SplitPolicy(f16_inputs=("attn",))):
        got = ops.attention(qp, kp, vp, heads=heads, scale=D ** -0.5)
    with ops.f32_split(ops.UNET_POLICY):
        got16 = ops.attention(qp, kp, vp, heads=heads, scale=D ** -0.5)
    assert isinstance(got, ops.Planes) and got.shape == (B, Nq, HD)
    assert got16.dtype == torch.float16 and torch.equal(got16.float(), got.f32())
    _cmp(got.f32(), want, 4e-3, f"split mode, attention in fp16: B{B} h{heads} {Nq}x{Nk} d{D}")
    q16, k16 = q.to(cuda).half(), k.to(cuda).half()
    ref16 = ops.attention(q16, k16, k16 if shared else v.to(cuda).half(), heads=heads, scale=D ** -0.5)
    # (not bit-identical: fp16(bf16 hi + bf16 lo) rounds the 16 kept bits of an fp32 value, fp16(fp32) all 24 -- a different fp16
    #  neighbour where the kept bits sit on a tie)
    _cmp(got.f32(), ref16.float().cpu(), 2e-3, "  against the same kernels on fp16(fp32) operands")
    # the conversions alone: planes -> fp16 -> planes is the identity on fp16-representable values
    x16 = torch.randn(3, 70, 136, generator=g).half()
    back = ops._f16_to_planes(ops._planes_to_f16(ops.to_planes(x16.float().to(cuda))))
    assert torch.equal(back.f32().cpu(), x16.float())


def test_planes_to_f16_saturates_finite_and_keeps_nan(cuda):
    """The fp16 hand-over of the split precision saturates finite values at +-65504 and must NOT turn NaN into a finite number
    (fminf / fmaxf are minnum / maxnum on this target: fmaxf(NaN, -65504) = -65504 -- the clamp is a compare + select): a NaN operand
    has to reach the attention output as NaN so that the isfinite() guards of the tests and of bench.py can see it.  Same for the
    weight-pair packer's hi half."""
    from rsvld_amd import ops
    x = torch.zeros(1, 64, 64)
    x[0, 0, :8] = torch.tensor([1.0, 1e6, -1e6, float("inf"), float("-inf"), float("nan"), 65504.0, -70000.0])
    h = ops._planes_to_f16(ops.to_planes(x.to(cuda))).float().cpu()[0, 0, :8]
    assert h[:3].tolist() == [1.0, 65504.0, -65504.0] and h[6:].tolist() == [65504.0, -65504.0]
    # (an infinity has no planes form: lo = bf16(inf - inf) = NaN, so it arrives as NaN -- non-finite either way, never a finite number)
    assert not bool(torch.isfinite(h[3:6]).any()) and bool(torch.isnan(h[5])), f"non-finite inputs became {h[3:6].tolist()}"
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(1, 256, 64, generator=g) for _ in range(3))
    k[0, 17, 3] = float("nan")                                     # one poisoned key: every query row's softmax sees it
    with ops.f32_split(ops.UNET_POLICY):
        out = ops.attention(ops.to_planes(q.to(cuda)), ops.to_planes(k.to(cuda)), ops.to_planes(v.to(cuda)), heads=1, scale=0.125)
    assert bool(torch.isnan(out.float()).any()), "a NaN key vanished on the way through the fp16 hand-over"
    w = torch.randn(16, 64, generator=g)
    w[3, 5], w[4, 6] = float("nan"), 1e6
    pc = ops.pack_conv(w, None, torch.float32, cuda)
    with ops.f32_split(PAIRS_ONLY()):
        y = ops.linear(torch.ones(8, 64, device=cuda, dtype=torch.float16), pc)
    y = y.float().cpu()
    assert bool(torch.isnan(y[:, 3]).all()) and bool(torch.isfinite(y[:, [0, 1, 2, 5]]).all())


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 5: the weight-pair form (dtype RSVLD_F16W2: fp16 activation x fp16 [W_lo | W_hi], two MFMAs per product) and the fp16 hand-over
# of the split kernels.  Reference = fp64 on the host over the SAME fp16-rounded activation (the rounding of the input is the policy's
# decision, measured end to end in DESIGN.md section 4; here the KERNEL must add nothing beyond the weights' ~2^-22).
def PAIRS_ONLY():
    """The UNets' policy with every weight of the fp16-input GEMMs kept as a pair (the default rounds the transformer GEMMs' weights to fp16)."""
    from rsvld_amd import ops
    return ops.SplitPolicy(f16_weights=())


W2_LINEAR_CASES = [
    # rows, K, N, residual (fp32), geglu, f16 out
    (8192, 640, 5120, False, True, True),        # GEGLU feed-forward, fp16 out: the persistent gemm256 (SEG = 2)
    (8192 + 77, 2560, 640, True, False, False),  # ff.net.2 with the fp32 residual: one-tile form, fp32 epilogue, ragged rows
    (16384, 1280, 1280, True, False, False),     # to_out on the fp16 attention output: the persistent form's fp32 + residual epilogue, half tiles
    (20000, 640, 640, True, False, False),       # the same, ragged rows and a ragged last column tile (N = 640 = 2.5 tiles)
    (66000, 320, 256, True, False, False),       # one column tile, 258 row tiles over 256 workgroups
    (4096, 64, 320, False, False, True),         # K' = 128: four K tiles, the persistent form's minimum
    (300, 640, 640, True, False, False),         # small M: the implicit-GEMM kernel (SEG = 2)
    (1000, 72, 80, False, True, True),           # K, N multiples of 8 only, GEGLU, fp16 out
    (2048, 640, 5120, False, True, False),       # GEGLU on the implicit-GEMM kernel, fp32 out
]


@pytest.mark.parametrize("case", W2_LINEAR_CASES)
def test_linear_weight_pairs(cuda, case):
    from rsvld_amd import _lib as L, ops
    rows, K, N, use_res, geglu, f16_out = case
    g = torch.Generator().manual_seed(rows + K + N)
    x16 = torch.randn(rows, K, generator=g).half()
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    y = x16.double() @ w.double().t() + b.double()
    if geglu:
        y = y[:, :N // 2] * F.gelu(y[:, N // 2:])
    res = torch.randn(y.shape, generator=g) if use_res else None
    if use_res:
        y = y * 0.5 + res.double()
    pc = ops.pack_conv(w, b, torch.float32, cuda, geglu=geglu)
    with ops.f32_split(PAIRS_ONLY()):         # inside a split-precision network: fp32 out unless the consumer takes fp16 (out_planes); every weight as a pair
        got = ops.linear(x16.to(cuda), pc, residual=None if res is None else res.to(cuda), act=L.ACT_GEGLU if geglu else L.ACT_NONE,
                         alpha=0.5 if use_res else 1.0, out_planes=f16_out, out_group="ff")
    assert got.dtype == (torch.float16 if f16_out else torch.float32) and got.shape == tuple(y.shape)
    # fp16 out: one rounding of the result (2^-11 of its magnitude); fp32 out: the weights' lo half keeps the product at ~2^-21
    _cmp(got, y, 6e-4 if f16_out else 3e-6, f"linear, weight pairs {case}")
    # the pair itself: [W_lo | W_hi] with W_hi = fp16(W), W_lo = fp16(W - W_hi)
    w2 = ops._w2(pc).view(pc.cout_p, 2, pc.cin_p).float().cpu()
    wp = pc.w.cpu()
    assert torch.equal(w2[:, 1], wp.half().float()) and torch.equal(w2[:, 0], (wp - wp.half().float()).half().float())


@pytest.mark.parametrize("rows,K,N,geglu", [(8192, 640, 1920, False), (4096 + 13, 1280, 5120, True), (300, 320, 960, False)])
def test_linear_fp16_weights_route(cuda, rows, K, N, geglu):
    """``SplitPolicy.f16_weights`` ("qkv": the consumer of the fp16 output is the attention; "geglu": a FeedForward): the fp32-packed weights are
    rounded to fp16 once and the PLAIN fp16 kernels run -- one MFMA per product.  The route must be exactly that: bit for bit the result of the
    same layer packed in fp16; a policy without the group keeps the pair (and differs)."""
    from rsvld_amd import _lib as L, ops
    g = torch.Generator().manual_seed(rows + N)
    x16 = torch.randn(rows, K, generator=g).half().to(cuda)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    act = L.ACT_GEGLU if geglu else L.ACT_NONE
    pc32 = ops.pack_conv(w, b, torch.float32, cuda, geglu=geglu)
    pc16 = ops.pack_conv(w, b, torch.float16, cuda, geglu=geglu)
    want = ops.linear(x16, pc16, act=act)                                   # the reference's GPU policy for this layer
    og = "ff" if geglu else "attn"
    with ops.f32_split(ops.UNET_POLICY):
        got = ops.linear(x16, pc32, act=act, out_planes=True, out_group=og)
    with ops.f32_split(PAIRS_ONLY()):
        pair = ops.linear(x16, pc32, act=act, out_planes=True, out_group=og)
    assert got.dtype == torch.float16 and torch.equal(got, want)
    assert pair.dtype == torch.float16 and not torch.equal(pair, want)
    assert ("geglu" if geglu else "qkv") in ops.UNET_POLICY.f16_weights and ops.UNET_POLICY.key() != PAIRS_ONLY().key()
    with pytest.raises(ValueError):
        ops.SplitPolicy(f16_inputs=("attn",), f16_weights=("qkv",))       # a weight is rounded only where its input is


@pytest.mark.parametrize("rows,K,N", [(16384, 1280, 1280), (8192 + 77, 2560, 640), (20000, 640, 640), (66000, 320, 256), (300, 640, 640), (4096, 128, 320)])
def test_linear_fp16_weights_fp32_residual(cuda, rows, K, N):
    """dtype RSVLD_F16W1 (``SplitPolicy.f16_weights`` "attn_out" / "ff_out"): fp16 activation x the weights rounded to fp16, ONE MFMA per
    product, fp32 accumulation, fp32 out + fp32 residual -- through the persistent gemm256 (SEG = 4), its one-tile form and the implicit-GEMM
    kernel.  Against fp64 over the SAME rounded operands: the kernel adds nothing but the fp32 summation."""
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(rows + K + N)
    x16 = torch.randn(rows, K, generator=g).half()
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    res = torch.randn(rows, N, generator=g)
    y = (x16.double() @ w.half().double().t() + b.double()) * 0.5 + res.double()
    pc = ops.pack_conv(w, b, torch.float32, cuda)
    assert {"attn_out", "ff_out"} <= ops.UNET_POLICY.f16_weights
    with ops.f32_split(ops.UNET_POLICY):
        got = ops.linear(x16.to(cuda), pc, residual=res.to(cuda), alpha=0.5, group="ff_out")
        none = ops.linear(x16.to(cuda), pc, alpha=0.5, group="attn_out")             # fp32 out without a residual: the one-tile / implicit-GEMM forms
    with ops.f32_split(PAIRS_ONLY()):
        pair = ops.linear(x16.to(cuda), pc, residual=res.to(cuda), alpha=0.5, group="ff_out")
    assert got.dtype == torch.float32 and none.dtype == torch.float32
    _cmp(got, y, 3e-6, f"linear, fp16 weights, fp32 out + residual ({rows}, {K}, {N})")
    _cmp(none, y - res.double(), 3e-6, "  without a residual")
    assert not torch.equal(pair, got)      # (a policy without the groups keeps these layers' weights as pairs)
    with pytest.raises(ValueError):
        ops.SplitPolicy(f16_weights=("conv",))


@pytest.mark.parametrize("rows,K,N,geglu", [(8192, 640, 1920, False), (4096 + 13, 1280, 2560, True), (500, 320, 960, False)])
def test_linear_split_fp16_hand_over(cuda, rows, K, N, geglu):
    """RSVLD_SPLIT with out_f32 = 2: planes in, three MFMAs per product, fp16 OUT (q | k | v on their way to the 16-bit attention kernels):
    equal to the planes output rounded to fp16 up to the double rounding of a tie."""
    from rsvld_amd import _lib as L, ops
    g = torch.Generator().manual_seed(rows + N)
    x = torch.randn(rows, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    y = x.double() @ w.double().t()
    if geglu:
        y = y[:, :N // 2] * F.gelu(y[:, N // 2:])
    pc = ops.pack_conv(w, None, torch.float32, cuda, geglu=geglu)
    with ops.f32_split(ops.UNET_POLICY):
        got = ops.linear(ops.to_planes(x.to(cuda)), pc, act=L.ACT_GEGLU if geglu else L.ACT_NONE, out_planes=True, out_group="attn")
        pl = ops.linear(ops.to_planes(x.to(cuda)), pc, act=L.ACT_GEGLU if geglu else L.ACT_NONE, out_planes=True, out_group=None)
    assert got.dtype == torch.float16 and isinstance(pl, ops.Planes)
    _cmp(got, y, 6e-4, f"linear split -> fp16 ({rows}, {K}, {N})")
    _cmp(got, pl.f32().cpu(), 6e-4, "  against the planes output")


W2_CONV_CASES = [
    # B, Cin, Cin2, Cout, H, W, upsample, residual, rowvec, norm, fp32 out
    (1, 64, 0, 64, 250, 260, False, True, True, True, False),        # conv_halo_64 with the fused GroupNorm + SiLU, fp16 out + fp16 residual
    (2, 128, 0, 128, 128, 250, False, True, True, True, False),      # conv_halo32 NW = 4, fused norm
    (1, 192, 64, 128, 512, 256, False, False, False, True, False),   # two sources, NW = 8, fused norm over the concat
    (1, 256, 0, 256, 100, 128, True, False, False, False, False),    # nearest x2
    (2, 128, 0, 128, 128, 250, False, True, False, False, True),     # fp32 out + fp32 residual (a split-precision network's conv on an fp16 input)
    (1, 64, 0, 64, 96, 128, False, True, True, True, False),         # small map: norm unfused, the gather kernel (SEG = 2)
]


@pytest.mark.parametrize("case", W2_CONV_CASES)
def test_conv3x3_weight_pairs(cuda, case):
    """SR3's compute dtype "w2": fp16 tensors, every weight as the pair -- the fp16 path's launches (fused GroupNorm prologue, epilogue
    statistics) with dtype RSVLD_F16W2.  Against fp64 over the fp16 inputs; the fused prologue rounds the normalised input to fp16 once."""
    from rsvld_amd import ops
    B, Cin, Cin2, Cout, H, W, up, use_res, use_rv, use_norm, f32_out = case
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g).half()
    x2 = torch.randn(B, Cin2, H, W, generator=g).half() if Cin2 else None
    w = torch.randn(Cout, Cin + Cin2, 3, 3, generator=g) / math.sqrt(9 * (Cin + Cin2))
    b = torch.randn(Cout, generator=g) * 0.1
    xin = x.double() if x2 is None else torch.cat([x.double(), x2.double()], 1)
    gamma = torch.randn(Cin + Cin2, generator=g) * 0.2 + 1.0
    beta = torch.randn(Cin + Cin2, generator=g) * 0.1
    if use_norm:
        xin = F.silu(F.group_norm(xin, 32, gamma.double(), beta.double(), 1e-5)).half().double()   # (the prologue's one rounding)
    if up:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    y = F.conv2d(xin, w.double(), b.double(), padding=1)
    rv = torch.randn(B, Cout, generator=g) if use_rv else None
    if use_rv:
        y = y + rv.double()[:, :, None, None]
    Ho, Wo = y.shape[-2:]
    res = torch.randn(B, Cout, Ho, Wo, generator=g) if use_res else None
    if use_res:
        res = res if f32_out else res.half().float()
        y = y + res.double()
    pc = ops.pack_conv(w, b, torch.float32, cuda, cin_split=None if x2 is None else (Cin, Cin2))
    kw = dict(x2=None if x2 is None else _nhwc(x2, cuda), upsample=up, rowvec=None if rv is None else rv.to(cuda),
              norm=(gamma.to(cuda), beta.to(cuda), 32, 1e-5, True) if use_norm else None, stats=True, out_f32=f32_out)
    if use_res:
        kw["residual"] = _nhwc(res, cuda) if f32_out else _nhwc(res, cuda).half()
    got = ops.conv2d(_nhwc(x, cuda), pc, **kw)
    assert got.dtype == (torch.float32 if f32_out else torch.float16)
    want = y.permute(0, 2, 3, 1)
    # fused norm: the device normalises with fp32 statistics of the fp16 input, the host rounds ITS normalised tensor: neighbouring fp16
    # values where the two differ in the last bit -> 2^-11 of one input per tap, averaged over 9 C terms
    _cmp(got, want, 2e-3 if (use_norm or not f32_out) else 3e-6, f"conv3x3, weight pairs {case}")
    part = getattr(got, "_gn_part", None)
    if part is not None:       # the epilogue's statistics are those of the stored tensor
        s = part[0].double().sum(1).cpu()          # [B, Cout, 2]
        gd = got.double().cpu()
        assert torch.allclose(s[..., 0], gd.sum((1, 2)), rtol=1e-3, atol=1e-1) and torch.allclose(s[..., 1], (gd * gd).sum((1, 2)), rtol=1e-3, atol=1e-1)


def test_norms_fp16_hand_over(cuda):
    """LayerNorm / GroupNorm of fp32 -> fp16 (out mode 2): the input of a weight-pair layer; equal to the fp32 output rounded once."""
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(2, 40, 24, 640, generator=g) * 3 + 0.5).to(cuda)
    ga, be = (torch.randn(640, generator=g) * 0.2 + 1).to(cuda), (torch.randn(640, generator=g) * 0.1).to(cuda)
    with ops.f32_split(ops.UNET_POLICY):
        ln16 = ops.layer_norm(x, ga, be, 1e-5, planes=True, group="ff")
        ln32 = ops.layer_norm(x, ga, be, 1e-5)
        lnp = ops.layer_norm(x, ga, be, 1e-5, planes=True, group="proj")         # "proj" is not in the default policy: planes
        gn16 = ops.group_norm(x, ga, be, 32, 1e-6, silu=True, planes=True, group="ff")
        gn32 = ops.group_norm(x, ga, be, 32, 1e-6, silu=True)
    assert ln16.dtype == torch.float16 and gn16.dtype == torch.float16 and isinstance(lnp, ops.Planes)
    assert torch.equal(ln16, ln32.half()) and torch.equal(gn16, gn32.half())


def test_transformer_block_policy_composition(cuda):
    """One BasicTransformerBlock (sgm/modules/attention.py:376-486) at 640 channels x 4096 tokens under the UNets' default policy
    (attention operands, to_out and FeedForward inputs in fp16; everything else three MFMAs) against the all-split policy and the
    fp32 family: the composition is a numerics decision of ~1e-3 of the block's range per call, the plumbing must be exact."""
    from rsvld_amd import ops
    from rsvld_amd.hipnn import HipNet
    from rsvld_amd.sgm.modules.attention import BasicTransformerBlock

    class Net(HipNet):
        def __init__(self):
            super().__init__()
            self.blk = BasicTransformerBlock(640, 10, 64, context_dim=2048)

    torch.manual_seed(3)
    net = Net().to(cuda).eval()
    net.compute_dtype = torch.float32
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 4096, 640, generator=g).to(cuda)
    ctx = torch.randn(2, 77, 2048, generator=g).to(cuda)
    outs = {}
    for name, pol in (("fp32", None), ("all_split", ops.ALL_SPLIT), ("unet", ops.UNET_POLICY)):
        with ops.f32_split(pol):
            outs[name] = net.blk.run(net, x, ctx).float().cpu()
    rng = float(outs["fp32"].abs().max())
    e_all = float((outs["all_split"] - outs["fp32"]).abs().max()) / rng
    e_pol = float((outs["unet"] - outs["fp32"]).abs().max()) / rng
    print(f"transformer block vs the fp32 family: all-split {e_all:.2e}, UNet policy {e_pol:.2e} of the range {rng:.2f}")
    assert e_all < 1e-4 and e_pol < 2e-3


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: RSVLD_F16Q8 -- the ResBlock convolutions of a split-precision network with the two cross terms of the product in e4m3 on the
# block-scaled matrix instruction (SplitPolicy.q8_convs).  Two references: (a) the ARITHMETIC the kernel promises, restated on the host in
# fp64 from the same fp32 normalised tensor (torch's own e4m3 casts): must agree to fp32 accumulation noise -- this pins the row format,
# the 16-byte interleave, the lane-half pairing and the E8M0 scale bytes; (b) plain fp64 of the layer: must sit where the split precision
# sits (~2^-15 per product), an order of magnitude inside what an fp16-operand form would give.
def _q8_parts(v, s_hi, s_lo):
    """host restatement of st_hq8: (fp16 part, e4m3(v 2^s_hi) / 2^s_hi, e4m3((v - fp16 v) 2^s_lo) / 2^s_lo), all fp64"""
    h = v.half().float()
    q = lambda t, s: (t * 2.0 ** s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).double() / 2.0 ** s
    return h.double(), q(v, s_hi), q(v - h, s_lo)


Q8_CASES = [
    # B, H, W, C1, C2, Cout, rowvec, residual        which halo instantiation
    (2, 24, 40, 128, 0, 256, True, True),          # 4-wave tiles, one source, every epilogue input
    (1, 19, 37, 64, 64, 128, False, True),         # two sources (a decoder ResBlock's skip concat), ragged tile rows and columns
    (4, 128, 128, 192, 0, 256, True, False),       # 8-wave (16 x 32 pixel) tiles: >= 192 of them and K >= 192
    (2, 16, 32, 320, 0, 320, False, False),        # Cout = 2.5 column tiles, 10 bodies
]


@pytest.mark.parametrize("case", Q8_CASES)
def test_conv3x3_q8_cross_terms(cuda, case):
    from rsvld_amd import ops
    B, H, W, C1, C2, Co, use_rv, use_res = case
    C = C1 + C2
    g = torch.Generator().manual_seed(B * H + C + Co)
    x1 = torch.randn(B, C1, H, W, generator=g) * 1.7 + 0.3
    x2 = torch.randn(B, C2, H, W, generator=g) * 0.6 - 0.2 if C2 else None
    w = torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(Co, generator=g) * 0.1
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    rv = torch.randn(B, Co, generator=g) if use_rv else None
    res = torch.randn(B, Co, H, W, generator=g) if use_res else None
    pc = ops.pack_conv(w, b, torch.float32, cuda, cin_split=(C1, C2) if C2 else None)
    xd, x2d = _nhwc(x1, cuda), None if x2 is None else _nhwc(x2, cuda)
    norm = (gamma.to(cuda), beta.to(cuda), 32, 1e-5, True)
    kw = dict(x2=x2d, pad=1, rowvec=None if rv is None else rv.to(cuda), residual=None if res is None else _nhwc(res, cuda), norm=norm,
              stats=True, alpha=0.5 if use_res else 1.0)
    with ops.f32_split(ops.UNET_POLICY), ops.tuning(split_halo_min_wgs=0, profiler=(prof := ops.LaunchProfiler())):
        got = ops.conv2d(xd, pc, norm_group="conv1", **kw)
        ab = ops._gn_scale_shift_f32(xd, x2d, norm[0], norm[1], 32, 1e-5)
        xn = ops._gn_apply_split(xd, x2d, ab, True, planes=False).cpu()          # the fp32 tensor the q8 apply kernel quantises
    names = set(prof.summary())
    assert "conv_halo_128_q8" in names and "groupnorm_apply_q8" in names, names   # the route under test ran
    with ops.f32_split(ops.SplitPolicy(q8_convs=())), ops.tuning(split_halo_min_wgs=0):
        split = ops.conv2d(xd, pc, norm_group="conv1", **kw)                      # the same layer as three bf16 MFMAs per product
    assert got.dtype == torch.float32 and got.shape == (B, H, W, Co) and hasattr(got, "_gn_part")
    # (a) the promised arithmetic, in fp64 on the host
    from rsvld_amd import _lib as L
    wp = pc.w.cpu().view(Co, 3, 3, C).permute(0, 3, 1, 2)                         # the packed K order is [tap][x | x2] = the concat order
    xh, xq, xlq = _q8_parts(xn.permute(0, 3, 1, 2), 2, 14)
    wh, wq, wlq = _q8_parts(wp, 6, 18)
    conv = lambda a, ww: F.conv2d(a, ww, None, padding=1)
    y = conv(xh, wh) + conv(xlq, wq) + conv(xq, wlq) + b.double().view(1, -1, 1, 1)
    if rv is not None:
        y = y + rv.double()[:, :, None, None]
    y = y * kw["alpha"]
    if res is not None:
        y = y + res.double()
    _cmp(got.permute(0, 3, 1, 2), y, 5e-6, f"q8 convolution {case} vs its arithmetic restated in fp64")
    # (b) the layer itself
    xe = F.silu(F.group_norm(x1.double() if x2 is None else torch.cat([x1, x2], 1).double(), 32, gamma.double(), beta.double(), 1e-5))
    ye = F.conv2d(xe, w.double(), b.double(), padding=1)
    if rv is not None:
        ye = ye + rv.double()[:, :, None, None]
    ye = ye * kw["alpha"] + (0 if res is None else res.double())
    e_q8 = _cmp(got.permute(0, 3, 1, 2), ye, 4e-5, f"q8 convolution {case} vs fp64")
    e_sp = _cmp(split.permute(0, 3, 1, 2), ye, 4e-5, "   the same layer, three bf16 MFMAs per product")
    print(f"   q8 / split distance from fp64: {e_q8 / max(e_sp, 1e-30):.2f}")
    # epilogue statistics = sums over the stored tensor
    part, ntiles = got._gn_part
    s = part.double().sum(1).cpu()
    gd = got.double().cpu()
    _cmp(s[..., 0], gd.sum((1, 2)), 1e-5, "   epilogue sum")
    _cmp(s[..., 1], (gd * gd).sum((1, 2)), 1e-5, "   epilogue sum of squares")


def test_q8_rows_format_and_saturation(cuda):
    """rsvld_split_hq8 / rsvld_pack_weight_hq8: the row format bit for bit against torch's casts; saturation at the fp16 / e4m3 ranges; NaN kept."""
    from rsvld_amd import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(5, 7, 64, generator=g) * torch.logspace(-5, 2, 64)
    x[0, 0, :4] = torch.tensor([1e6, -70000.0, 300.0, float("nan")])
    q = ops.to_q8rows(x.to(cuda))
    assert q.shape == (5, 7, 64) and q.t.shape == (5, 7, 2, 64)
    t = q.t.cpu()
    h = t[..., 0, :].float()
    want_h = x.clone()
    want_h[0, 0, 0], want_h[0, 0, 1] = 65504.0, -65504.0
    assert torch.equal(h[0, 0, 4:], x.half().float()[0, 0, 4:]) and bool(torch.isnan(h[0, 0, 3])) and h[0, 0, :2].tolist() == [65504.0, -65504.0]
    blocks = t[..., 1, :].contiguous().view(torch.uint8).view(5, 7, 2, 4, 16)          # [block][16-byte piece][byte]; pieces P0a P1a P0b P1b
    p0 = torch.cat([blocks[..., 0, :], blocks[..., 2, :]], -1).reshape(5, 7, 64).view(torch.float8_e4m3fn).float()
    p1 = torch.cat([blocks[..., 1, :], blocks[..., 3, :]], -1).reshape(5, 7, 64).view(torch.float8_e4m3fn).float()
    hh = x.half().float()
    want0 = ((x - hh) * 2.0 ** 14).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    want1 = (x * 4.0).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    ok = torch.isfinite(x) & (x.abs() < 60000)
    assert torch.equal(p0[ok], want0[ok]) and torch.equal(p1[ok], want1[ok])
    assert p1[0, 0, 0] == 448.0 and p1[0, 0, 1] == -448.0 and p1[0, 0, 2] == 448.0      # 300 * 4 saturates
