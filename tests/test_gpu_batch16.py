"""BASELINE configs[3] says "batch = 16" per GPU: sixteen images through BOTH stages in one pass, in the precision the headline is
timed in (Stage 1: fp16 tensors x weight pairs; Stage 2: the split precision under ops.UNET_POLICY), with the feature cache at the
reference's default threshold 0.3 -- every image takes its own hit / miss decisions and the second UNet half runs on the sub-batch
that missed (reference: the per-image loop of infer_dir.py:196-201, the cache test of models/modules/DFBCache.py:98-112).
Images 0 and 15 of the batch must equal their batch-of-1 runs BIT FOR BIT (batch-invariant launch plans), traces included.
Reduced to 1024^2 (128 -> 1024 x8) and 2 + 3 sampler iterations: sixteen 1024^2 images in ONE Stage-2 launch hold the activations of
one 4096^2 image (at 2048^2 the fp32 residual streams of sixteen images plus the resident VAE tiles exceed the 288 GB: measured);
the 4096^2 run at batch 16 -- Stage 2 in sub-batches of two -- is ``python bench.py --batch-per-gpu 16`` (profiles/r05_bench_c4_b16.json)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
B, LR, SCALE = 16, 128, 8


def test_batch16_two_stage_images_equal_their_batch_of_one_runs(cuda, full_model):
    import bench
    from rsvld_amd import measure, parallel
    side = LR * SCALE
    m = full_model
    net, _ = bench.build_stage1(50)
    net.use_graph = False
    net.batch_invariant = True                # launch plans per image: a batch member = its batch-of-1 run
    net.denoise_fn.set_compute_dtype("w2")
    cond = bench.stage1_input(list(range(B)), LR, SCALE)
    g = torch.Generator().manual_seed(3)
    s1_noise = [torch.randn(B, 3, side, side, generator=g) for _ in range(3)]            # x_T, then one draw per step with t > 0
    lat = side // 8
    post, xt = torch.randn(B, 4, lat, lat, generator=g), torch.randn(B, 4, lat, lat, generator=g)
    s2_noise = [torch.randn(B, 4, lat, lat, generator=g) for _ in range(3)]
    kw = dict(bench.S2_KW, img_threshold=0.3, num_steps=50)

    def run(sl):
        it = iter(s1_noise)
        net._randn = lambda shape, device: next(it)[sl].to(device)
        draws = iter([xt] + s2_noise)
        m._posterior_noise = lambda shape: post[sl]
        m._randn_like = lambda t: next(draws)[sl].to(t.device)
        m.set_precision("split", "split")
        try:
            with measure.hooks(net, max_steps=2):
                sr = net.super_resolution(cond[sl].to(cuda), continous=True)
            u8 = parallel.to_uint8(sr[-len(range(B)[sl]):])       # the last frame of every image (continous=True stacks the kept frames)
            lq = u8.float() / 127.5 - 1.0
            with measure.hooks(m, max_steps=3):
                out = m.just_sampling(lq, [""] * lq.shape[0], **kw)
            return u8.cpu(), parallel.to_uint8(out).cpu(), [list(s) for s in m.cache_trace]
        finally:
            del net._randn, m._posterior_noise, m._randn_like
            m.set_precision("bf16", "fp16")

    torch.cuda.reset_peak_memory_stats(cuda)
    s1_all, out_all, trace = run(slice(0, B))
    peak = torch.cuda.max_memory_allocated(cuda) / 2 ** 30
    print(f"batch {B} at {side}^2, both stages: peak memory {peak:.1f} GiB")
    assert s1_all.shape == (B, 3, side, side) and out_all.shape == (B, 3, side, side)
    assert len(trace) == 3 and all(len(s) == B for s in trace)          # one decision per image and step
    assert not torch.equal(out_all[0], out_all[1])
    for b in (0, B - 1):
        s1_one, out_one, tr1 = run(slice(b, b + 1))
        assert torch.equal(s1_all[b:b + 1], s1_one), f"image {b}: the Stage-1 hand-off differs from its batch-of-1 run"
        assert [s[b] for s in trace] == [s[0] for s in tr1], f"image {b}: cache trace differs"
        assert torch.equal(out_all[b:b + 1], out_one), f"image {b}: the final image differs from its batch-of-1 run"
        print(f"image {b} of {B}: Stage-1 hand-off, cache trace {[bool(s[b][2]) for s in trace]} and final uint8 image equal the batch-of-1 run")


def test_batch16_at_4096_as_configs3_is_written(cuda, full_model):
    """BASELINE configs[3] AS WRITTEN: 512 -> 4096 x8, tiled VAE, batch = 16 on one MI355X -- sixteen 4096^2 images held by one GPU in one
    pass, the way ``bench.py --batch-per-gpu 16`` runs them (images are independent units, infer_dir.py:196-201: Stage 1 in sub-batches of
    four launches-wise, Stage 2 in sub-batches of two with per-image cache decisions), in the precision the headline is timed in, ONE
    sampler iteration per stage + the whole fixed part (four 64-tile VAE passes per image, colour fix).  Images 0 and 15 must equal their
    batch-of-1 runs bit for bit; the pass must fit the 288 GB (bench.py measured 226 GB with the caption model resident)."""
    import bench
    from rsvld_amd import measure, parallel
    Bn, lr, scale, c1, c2 = 16, 512, 8, 4, 2
    side, lat = lr * scale, lr * scale // 8
    m = full_model
    net, _ = bench.build_stage1(50)
    net.use_graph = False
    net.batch_invariant = True
    net.denoise_fn.set_compute_dtype("w2")
    kw = dict(bench.S2_KW, img_threshold=0.3, num_steps=50)

    def per_image(shape, ids, tag, device):
        """one seeded draw per IMAGE (not per batch): image i sees the same noise whatever shares its launch"""
        outs = []
        for i in ids:
            g = torch.Generator(device=device).manual_seed(100_000 * tag + i)
            outs.append(torch.randn((1,) + tuple(shape[1:]), generator=g, device=device))
        return torch.cat(outs)

    def run(ids):
        m.set_precision("split", "split")
        try:
            srs = []
            for i0 in range(0, len(ids), c1):
                chunk, draws = ids[i0:i0 + c1], iter(range(1, 100))
                net._randn = lambda shape, device, _c=chunk, _d=draws: per_image(shape, _c, next(_d), device)
                cond = bench.stage1_input(chunk, lr, scale).to(cuda)
                with measure.hooks(net, max_steps=1):
                    srs.append(parallel.to_uint8(net.super_resolution(cond, continous=True)[-len(chunk):]))
                del cond
            u8 = torch.cat(srs)
            outs, traces = [], []
            for i0 in range(0, len(ids), c2):
                chunk, draws = ids[i0:i0 + c2], iter(range(200, 300))
                m._posterior_noise = lambda shape, _c=chunk: per_image(shape, _c, 150, torch.device("cpu"))
                m._randn_like = lambda t, _c=chunk, _d=draws: per_image(t.shape, _c, next(_d), t.device)
                lq = u8[i0:i0 + len(chunk)].float() / 127.5 - 1.0
                with measure.hooks(m, max_steps=1):
                    outs.append(parallel.to_uint8(m.just_sampling(lq, [""] * len(chunk), **kw)).cpu())
                traces += [[step[j] for step in m.cache_trace] for j in range(len(chunk))]
            return u8.cpu(), torch.cat(outs), traces
        finally:
            del net._randn, m._posterior_noise, m._randn_like
            m.set_precision("bf16", "fp16")

    torch.cuda.reset_peak_memory_stats(cuda)
    s1_all, out_all, tr_all = run(list(range(Bn)))
    peak = torch.cuda.max_memory_allocated(cuda) / 2 ** 30
    print(f"configs[3] as written: batch {Bn} at {side}^2 (latent {lat}), both stages, one iteration each + the fixed part: peak memory {peak:.1f} GiB")
    assert s1_all.shape == (Bn, 3, side, side) and out_all.shape == (Bn, 3, side, side) and out_all.dtype == torch.uint8
    assert peak < 288 and len(tr_all) == Bn
    assert not torch.equal(out_all[0], out_all[1]) and float(out_all[0].float().std()) > 1.0
    for b in (0, Bn - 1):
        s1_one, out_one, tr1 = run([b])
        assert torch.equal(s1_all[b:b + 1], s1_one), f"image {b}: the Stage-1 hand-off differs from its batch-of-1 run"
        assert tr_all[b] == tr1[0], f"image {b}: cache trace differs"
        assert torch.equal(out_all[b:b + 1], out_one), f"image {b}: the final image differs from its batch-of-1 run"
        print(f"image {b} of {Bn} at {side}^2: Stage-1 hand-off and final uint8 image equal the batch-of-1 run")
