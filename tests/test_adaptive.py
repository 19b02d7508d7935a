"""Adaptive sampling: `Tile.GetNoisiestPixel` (tile.go:34-66) inside the worker loop (main.go:837-897).

The reference picks one pixel of a tile at a time; pathtrace_amd/adaptive.py renders the next batch of EVERY pixel that
is still eligible in one call per round.  The claim that the finished frames are equal -- a pixel's stopping point
depends on that pixel alone -- is checked here three ways: on synthetic samples (the sequential loop, restated in
oracle/orc_trace.c, against the per-pixel rule), on the oracle's own scene samples (the loop tracing on demand against
the loop on precomputed samples), and on the GPU (rounds through ptmi_render_tiles_stats_dev against the sequential loop
fed with the GPU's own per-sample radiance: sums, counts and luminance ranges bit for bit).
"""
import numpy as np
import pytest

from pathtrace_amd import adaptive, scenes
from pathtrace_amd.render import Renderer, make_params
import oracle_lib

f32 = np.float32


def test_pick_counts():
    assert adaptive.pick_counts(32, 32) == [32] and adaptive.pick_counts(8, 44) == [8, 16, 24, 32, 40, 45]
    assert adaptive.pick_counts(32, 33) == [32, 34] and adaptive.pick_counts(32, 128) == [32, 64, 96, 128]


def per_pixel_rule(samples, spp, max_samples):
    """Every pixel on its own: batches of spp samples until count >= max_samples (tile.go:39) or count > 32 and
    contrast < 0.01 (:51); a batch ends early once count > max_samples (main.go:880).  Pixel.AddSample (renderer.go:25-82)
    in numpy float32 scalars: stats = min / max luminance, Mean, M2, Variance."""
    P = samples.shape[0]
    sums = np.zeros((P, 3), f32); counts = np.zeros(P, np.uint32); stats = np.zeros((P, 9), f32)
    for p in range(P):
        n = 0
        st = stats[p]
        while n < max_samples and not (n >= 4 and n > 32 and f32(st[1] - st[0]) < f32(0.01)):
            for _ in range(adaptive.batch_size(n, spp, max_samples)):
                c = samples[p, n]
                n += 1
                fn = f32(n)
                d = c - st[2:5]
                st[2:5] += d / fn
                st[5:8] += d * (c - st[2:5])
                if n > 1:
                    st[8] = f32(f32(st[5] + st[6]) + st[7]) / f32(f32(3.0) * f32(fn - f32(1.0)))
                sums[p] += c
                l = f32(f32(f32(0.2126) * c[0] + f32(0.7152) * c[1]) + f32(0.0722) * c[2])
                if n == 1:
                    st[0] = st[1] = l
                else:
                    if l < st[0]: st[0] = l
                    if l > st[1]: st[1] = l
        counts[p] = n
    return sums, counts, stats


@pytest.mark.parametrize("spp,max_samples", [(8, 44), (32, 128), (16, 16), (5, 64), (32, 33)])
def test_sequential_picker_equals_the_per_pixel_rule(spp, max_samples):
    rng = np.random.default_rng(spp * 1000 + max_samples)
    P, S = 257, max_samples + 1
    base = rng.random((P, 1, 3), dtype=f32)
    noise = rng.random((P, S, 3), dtype=f32)
    # a third of the pixels constant (contrast 0), a third barely noisy (contrast around the 0.01 threshold), a third noisy
    amp = np.where(np.arange(P) % 3 == 0, 0.0, np.where(np.arange(P) % 3 == 1, 0.012, 1.0)).astype(f32)[:, None, None]
    samples = (base + amp * noise).astype(f32)
    s1, c1, l1, picks = oracle_lib.pick_simulate(samples, spp, max_samples)
    s2, c2, l2 = per_pixel_rule(samples, spp, max_samples)
    assert np.array_equal(c1, c2)
    assert np.array_equal(s1.view(np.uint32), s2.view(np.uint32))
    assert np.array_equal(l1.view(np.uint32), l2.view(np.uint32))
    assert picks == sum(-(-int(c) // spp) for c in c1)          # one pick per batch
    cap = 0
    while cap < max_samples:                                    # where a pixel that never converges ends
        cap += adaptive.batch_size(cap, spp, max_samples)
    assert c1.max() == cap
    if max_samples > 40:
        assert len(np.unique(c1)) >= 2                          # some pixels converge early, some run to the cap


def test_mask_runs():
    rng = np.random.default_rng(5)
    for shape in [(1, 1), (3, 7), (16, 33)]:
        for density in (0.0, 0.3, 1.0):
            m = rng.random(shape) < density
            runs = adaptive.mask_runs(m)
            back = np.zeros(shape, bool)
            for x, y, w, h in runs:
                assert h == 1 and w >= 1 and not back[y, x:x + w].any()
                back[y, x:x + w] = True
            assert np.array_equal(back, m)
            # maximal runs: no two runs of a row touch
            for (x, y, w, _), (x2, y2, _, _) in zip(runs[:-1], runs[1:]):
                assert y2 > y or x2 > x + w


def _oparams(p):
    return oracle_lib.OrcParams(p.width, p.height, p.bounces, p.scatter_rays, p.max_steps, p.ambient, p.step_size, p.seed)


def test_oracle_loop_tracing_on_demand_equals_the_loop_on_precomputed_samples():
    """orc_render_tile_adaptive (the worker loop calling TraceRay when it needs a sample) against orc_pick_simulate on
    orc_render_samples' output for the same tile: the k-th sample of a pixel is the same whenever it is taken."""
    sc = scenes.bunny_scene(small=True, glass=True, emissive=True)
    p = make_params(40, 32, bounces=2, seed=3)
    osc = oracle_lib.OracleScene(sc)
    cam = sc.camera.as_array()
    spp, mx = 8, 44
    rect = (8, 4, 24, 20)
    sums, counts, lum, picks = osc.render_tile_adaptive(cam, _oparams(p), rect, spp, mx)
    smp = osc.render_samples(cam, _oparams(p), rect=rect, spp_count=mx + 1)
    s2, c2, l2, picks2 = oracle_lib.pick_simulate(smp.reshape(-1, mx + 1, 3), spp, mx)
    assert picks == picks2
    assert np.array_equal(counts.ravel(), c2)
    assert np.array_equal(sums.reshape(-1, 3).view(np.uint32), s2.view(np.uint32))
    assert np.array_equal(lum.reshape(-1, 9).view(np.uint32), l2.view(np.uint32))
    assert counts.min() >= 40 and counts.max() == 45 and len(np.unique(counts)) >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("spp,max_samples", [(8, 44), (32, 96)])
def test_gpu_rounds_equal_the_sequential_picker(gpu_ctx, spp, max_samples):
    """render_adaptive (one ptmi_render_tiles_stats_dev call per round over the runs of eligible pixels; and its
    two-call form) against the reference's sequential picker loop (oracle/orc_trace.c) fed with the GPU's own per-sample radiance, tile by tile
    (2 x 2 tiles as main.go:454-482 cuts them): sums, counts and luminance ranges equal bit for bit."""
    sc = scenes.bunny_scene(small=True, glass=True, emissive=True)
    W, H = 48, 40
    p = make_params(W, H, bounces=2, seed=21)
    r = Renderer(sc, ctx=gpu_ctx)
    accum, counts, lum, rounds = adaptive.render_adaptive(r, p, sc.camera, samples_per_pixel=spp, max_samples=max_samples, merge=False)
    merged = adaptive.render_adaptive(r, p, sc.camera, samples_per_pixel=spp, max_samples=max_samples)        # two calls
    assert len(merged[3]) == 2 and len(rounds) == len(adaptive.pick_counts(spp, max_samples))
    import torch
    for a, b in zip((accum, counts, lum), merged[:3]):
        assert (a.view(dtype=torch.int32) == b.view(dtype=torch.int32)).all()
    # ... and the library's own two passes (ptmi_render_adaptive_dev)
    nat = [torch.zeros_like(t) for t in (accum, counts, lum)]
    r.render_adaptive_device(nat[0], nat[1], nat[2], p, sc.camera, samples_per_pixel=spp, max_samples=max_samples)
    for a, b in zip((accum, counts, lum), nat):
        assert (a.view(dtype=torch.int32) == b.view(dtype=torch.int32)).all()
    accum, counts, lum = accum.cpu().numpy(), counts.cpu().numpy(), lum.cpu().numpy()
    S = max_samples + 1
    smp = r.render_samples(p, sc.camera, spp=S)                   # [H, W, S, 3]
    tw, th = W // 2, H // 2
    for ty in range(2):
        for tx in range(2):
            ys, xs = slice(ty * th, ty * th + th), slice(tx * tw, tx * tw + tw)
            s2, c2, l2, _ = oracle_lib.pick_simulate(np.ascontiguousarray(smp[ys, xs]).reshape(-1, S, 3), spp, max_samples)
            assert np.array_equal(counts[ys, xs].ravel().astype(np.uint32), c2)
            assert np.array_equal(np.ascontiguousarray(accum[ys, xs]).reshape(-1, 3).view(np.uint32), s2.view(np.uint32))
            assert np.array_equal(np.ascontiguousarray(lum[ys, xs]).reshape(-1, 9).view(np.uint32), l2.view(np.uint32))   # all of AddSample's state
    assert len(np.unique(counts)) >= 2 and counts.min() > 32           # background pixels stop early, noisy ones go on
    assert sum(n for n, _ in rounds) == counts.max()
    assert rounds[0][1] == W * H and rounds[-1][1] < W * H


@pytest.mark.gpu
def test_luminance_range_across_passes_and_lanes(monkeypatch):
    """A call cut into several passes (PTMI_CHUNK_MSAMPLES=1: 1 M samples per pass) and run on two lanes keeps the sums
    and the per-pixel statistics of the same call in one pass: accumulate continues a pixel's range from the pass before."""
    import torch
    from pathtrace_amd import Context
    sc = scenes.bunny_scene(small=True, glass=True, emissive=True)
    W, H, spp = 256, 192, 72                     # 3.5 M samples
    p = make_params(W, H, bounces=2, seed=9)
    dev = torch.device("cuda", 0)
    got = []
    for env in ({}, {"PTMI_CHUNK_MSAMPLES": "1"}, {"PTMI_CHUNK_MSAMPLES": "1", "PTMI_RENDER_LANES": "1"}):
        for var in ("PTMI_CHUNK_MSAMPLES", "PTMI_RENDER_LANES"):
            monkeypatch.delenv(var, raising=False)
        for var, val in env.items():
            monkeypatch.setenv(var, val)
        ctx = Context(0)
        try:
            r = Renderer(sc, ctx=ctx)
            acc = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
            cnt = torch.zeros((H, W), dtype=torch.int32, device=dev)
            lum = torch.zeros((H, W, 9), dtype=torch.float32, device=dev)
            tiles = [(0, 0, W, H // 2), (0, H // 2, W // 2, H // 2), (W // 2, H // 2, W // 2, H // 2)]
            r.render_tiles_device(acc, cnt, p, sc.camera, tiles, spp=spp - 8, stats=lum)
            r.render_tiles_device(acc, cnt, p, sc.camera, tiles, spp=8, spp_begin=spp - 8, stats=lum)      # a later call continues too
            got.append((acc.cpu().numpy(), cnt.cpu().numpy(), lum.cpu().numpy()))
        finally:
            ctx.close()
    a0, c0, l0 = got[0]
    assert (c0 == spp).all() and (l0[..., 1] >= l0[..., 0]).all() and (l0[..., 1] > l0[..., 0]).any()
    for a, c, l in got[1:]:
        assert np.array_equal(a.view(np.uint32), a0.view(np.uint32)) and np.array_equal(c, c0)
        assert np.array_equal(l.view(np.uint32), l0.view(np.uint32))
    # and the range is that of the per-sample luminances
    ctx = Context(0)
    try:
        smp = Renderer(sc, ctx=ctx).render_samples(p, sc.camera, rect=(100, 60, 32, 8), spp=spp)
    finally:
        ctx.close()
    lumv = (f32(0.2126) * smp[..., 0] + f32(0.7152) * smp[..., 1]) + f32(0.0722) * smp[..., 2]
    assert np.array_equal(lumv.min(-1), l0[60:68, 100:132, 0]) and np.array_equal(lumv.max(-1), l0[60:68, 100:132, 1])
    # ... and mean / variance are Welford's: close to the two-pass values (float32 running sums)
    mean = smp.astype(np.float64).mean(2)
    assert np.allclose(l0[60:68, 100:132, 2:5], mean, rtol=1e-4, atol=1e-5)
    var = smp.astype(np.float64).var(2, ddof=1).sum(-1) / 3.0
    assert np.allclose(l0[60:68, 100:132, 8], var, rtol=1e-3, atol=1e-6)
