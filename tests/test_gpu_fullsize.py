"""Size-independent properties of the hot-path kernels at BASELINE.json's FULL sizes (the 512 -> 4096 x8 shapes of
configs[3]/[4]), where no CPU oracle finishes in reasonable time: softmax weights sum to one, key-order invariance,
identity weights, linearity, translation equivariance across tile borders, unit statistics after GroupNorm.  Every
check goes through the C ABI (rsvld_amd.ops)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _gen(seed):
    return torch.Generator(device="cuda").manual_seed(seed)


@pytest.mark.parametrize("N", [65536, 262144])   # Stage-1 attention at 4096^2: mid block (S/16)^2 and level 3 (S/8)^2
def test_attention_d512_full_size_properties(cuda, N):
    from rsvld_amd import ops
    D, dt = 512, torch.float16
    g = _gen(N)
    q = (torch.randn(1, N, D, device=cuda, generator=g) * 0.5).to(dt)
    k = (torch.randn(1, N, D, device=cuda, generator=g) * 0.5).to(dt)
    v = torch.randn(1, N, D, device=cuda, generator=g).to(dt)
    rows = torch.randint(0, N, (512,), device=cuda, generator=g)      # sampled query rows (outputs are 268 MB)
    # (a) constant V: the output is that constant whatever the scores (the weights sum to one, every tile is normalised)
    c = torch.linspace(-2, 2, D, device=cuda).to(dt)
    out = ops.attention(q, k, c.expand(1, N, D).contiguous(), heads=1)
    assert torch.allclose(out[0, rows].float(), c.float().expand(512, D), atol=2e-3, rtol=2e-3)
    # (b) Q = 0: uniform weights, the output is the mean of V over all keys
    out = ops.attention(torch.zeros_like(q), k, v, heads=1)
    mean = v.float().mean(1)
    assert float((out[0, rows].float() - mean).abs().max()) < 2e-3
    # (c) key-order invariance: permuting (K, V) rows together changes tile contents, rescale points and the split-KV
    # ranges, not the result
    ref = ops.attention(q[:, :4096], k, v, heads=1)
    perm = torch.randperm(N, device=cuda, generator=g)
    got = ops.attention(q[:, :4096], k[:, perm].contiguous(), v[:, perm].contiguous(), heads=1)
    assert float((got.float() - ref.float()).abs().max()) < 4e-3 * max(1.0, float(ref.float().abs().max()))


@pytest.mark.parametrize("N", [65536, 262144])
def test_attention_d512_shared_tile_full_size_properties(cuda, N):
    """The shared-tile form (keys = values = one tensor, what Stage 1 runs at 4096^2): Q = 0 returns the mean of X, key-order
    invariance, and agreement with the two-tensor kernel on sampled rows."""
    from rsvld_amd import ops
    D, dt = 512, torch.float16
    g = _gen(N + 1)
    q = (torch.randn(1, 4096, D, device=cuda, generator=g) * 0.5).to(dt)
    x = torch.randn(1, N, D, device=cuda, generator=g).to(dt)
    out = ops.attention(torch.zeros_like(q), x, x, heads=1)
    assert float((out[0].float() - x.float().mean(1)).abs().max()) < 2e-3
    ref = ops.attention(q, x, x, heads=1)
    two = ops.attention(q, x, x.clone(), heads=1)                 # separate K and V tensors: the two-tensor instantiation
    assert float((ref.float() - two.float()).abs().max()) < 2e-3
    perm = torch.randperm(N, device=cuda, generator=g)
    xp = x[:, perm].contiguous()
    got = ops.attention(q, xp, xp, heads=1)
    assert float((got.float() - ref.float()).abs().max()) < 4e-3 * max(1.0, float(ref.float().abs().max()))


def test_attention_d64_full_size_properties(cuda):
    """Stage-2 self-attention at latent 512: (L/2)^2 = 65 536 tokens, 10 heads of 64."""
    from rsvld_amd import ops
    N, heads, D, dt = 65536, 10, 64, torch.float16
    g = _gen(7)
    q = torch.randn(1, N, heads * D, device=cuda, generator=g).to(dt)
    k = torch.randn(1, N, heads * D, device=cuda, generator=g).to(dt)
    v = torch.randn(1, N, heads * D, device=cuda, generator=g).to(dt)
    rows = torch.randint(0, N, (512,), device=cuda, generator=g)
    c = torch.linspace(-2, 2, heads * D, device=cuda).to(dt)
    out = ops.attention(q, k, c.expand(1, N, heads * D).contiguous(), heads=heads)
    assert torch.allclose(out[0, rows].float(), c.float().expand(512, heads * D), atol=2e-3, rtol=2e-3)
    out = ops.attention(torch.zeros_like(q), k, v, heads=heads)
    assert float((out[0, rows].float() - v.float().mean(1)).abs().max()) < 2e-3
    ref = ops.attention(q[:, :2048], k, v, heads=heads)
    perm = torch.randperm(N, device=cuda, generator=g)
    got = ops.attention(q[:, :2048], k[:, perm].contiguous(), v[:, perm].contiguous(), heads=heads)
    assert float((got.float() - ref.float()).abs().max()) < 4e-3 * max(1.0, float(ref.float().abs().max()))


def test_gemm_full_size_identity_and_linearity(cuda):
    """Stage-2 transformer linears at latent 512: 131 072 token rows.  Identity weights return the input bit for bit
    (every product is x*1 or x*0, fp32 accumulation of one non-zero term); two inputs superpose."""
    from rsvld_amd import ops
    M, C, dt = 131072, 640, torch.float16
    g = _gen(11)
    x = torch.randn(M, C, device=cuda, generator=g).to(dt)
    pc = ops.pack_conv(torch.eye(C), None, dt, cuda)
    assert torch.equal(ops.linear(x, pc), x)
    w = torch.randn(1920, C, generator=torch.Generator().manual_seed(3)) / math.sqrt(C)
    pcw = ops.pack_conv(w, None, dt, cuda)
    x2 = torch.randn(M, C, device=cuda, generator=g).to(dt)
    s = (x.float() + x2.float()).to(dt)          # the rounded sum is what the kernel sees
    lhs = ops.linear(s, pcw).float()
    rhs = ops.linear(x, pcw).float() + ops.linear(x2, pcw).float()
    assert float((lhs - rhs).abs().max()) < 2e-2   # three fp16 roundings of values of magnitude ~1.5


def test_conv3x3_full_resolution_translation_and_linearity(cuda):
    """Stage-1 level-0 convolution at the 4096 x 4096 output size (64 channels): shifting the input by a non-multiple of
    the 8x32 tile shifts the output (no tile-border artefacts); a bias-free conv is linear."""
    from rsvld_amd import ops
    H = W = 4096
    C, dt = 64, torch.float16
    g = _gen(13)
    x = torch.randn(1, H, W, C, device=cuda, generator=g).to(dt)
    w = torch.randn(C, C, 3, 3, generator=torch.Generator().manual_seed(5)) / math.sqrt(9 * C)
    pc = ops.pack_conv(w, None, dt, cuda)
    y = ops.conv2d(x, pc, pad=1)
    dy, dx = 5, 13
    xs = torch.zeros_like(x)
    xs[:, dy:, dx:] = x[:, :H - dy, :W - dx]
    ys = ops.conv2d(xs, pc, pad=1)
    # interior (away from the zero band the shift introduced and from the far border, where the shifted image has lost
    # rows / columns): identical arithmetic, identical bits
    assert torch.equal(ys[:, dy + 1:H - 1, dx + 1:W - 1], y[:, 1:H - dy - 1, 1:W - dx - 1])
    x2 = torch.randn(1, H, W, C, device=cuda, generator=g).to(dt)
    s = (x.float() + x2.float()).to(dt)
    lhs = ops.conv2d(s, pc, pad=1).float()
    rhs = y.float() + ops.conv2d(x2, pc, pad=1).float()
    assert float((lhs - rhs).abs().max()) < 3e-2


def test_groupnorm_full_resolution_unit_statistics(cuda):
    """GroupNorm(32) over a 2048 x 2048 x 128 Stage-1 tensor: every (image, group) of the output has mean 0 and variance 1
    (gamma = 1, beta = 0), also through the fused conv prologue's (scale, shift) rows."""
    from rsvld_amd import ops
    H = W = 2048
    C, G, dt = 128, 32, torch.float16
    g = _gen(17)
    x = (torch.randn(1, H, W, C, device=cuda, generator=g) * 3 + 1.5).to(dt)
    y = ops.group_norm(x, torch.ones(C, device=cuda), torch.zeros(C, device=cuda), G, 1e-5)
    yg = y.float().view(H * W, G, C // G)
    assert float(yg.mean((0, 2)).abs().max()) < 2e-3
    assert float((yg.var((0, 2), unbiased=False) - 1).abs().max()) < 5e-3
    st = ops.group_norm_stats(x, G)
    xg = x.float().view(H * W, G, C // G)
    assert torch.allclose(st[0, :, 0], xg.mean((0, 2)), atol=1e-3)
    assert torch.allclose(st[0, :, 1], xg.var((0, 2), unbiased=False), rtol=2e-3)
