"""The reference's adaptive sampling (`Tile.GetNoisiestPixel`, tile.go:34-66, driven by the worker loop main.go:837-897)
as rounds over the whole frame.

The reference's worker picks ONE pixel of its tile at a time -- the one with the largest contrast / sqrt(count), pixels
with fewer than 4 samples first -- and gives it `samplesPerPixel` samples (main.go:853-883).  Which pixel comes next
depends on all pixels of the tile, but whether a pixel is ever picked AGAIN depends on that pixel alone: it is skipped
for good once `SampleCount >= maxSamples` (tile.go:39) or `SampleCount > 32 && Contrast < 0.01` (:51), and nothing else
removes it; a skipped pixel gets no samples, so its contrast never changes again.  With a sample's radiance a function
of (pixel, sample index) -- this library's counter-based RNG -- the frame a tile ends with is therefore the one in which
every pixel got batches of `samplesPerPixel` samples until one of the two tests stopped it, in any order.  That is data
parallel: round r renders the next batch of every pixel still eligible, in one `ptmi_render_tiles_stats_dev` call
whose tiles are the runs of eligible pixels.  (At the reference's committed constants, samplesPerPixel ==
maxSamplesPerPixel == 32, there is one round and every pixel gets 32 samples: SURVEY 3.2.)

The reference's sequential loop itself is restated on the CPU for the tests (`tests/test_adaptive.py` holds the two
against each other, bit for bit).
"""
from __future__ import annotations

import numpy as np

CONTRAST_DONE = np.float32(0.01)        # tile.go:51
ALWAYS_BELOW = 4                        # tile.go:45: fewer samples than this -> priority 1e9
CONTRAST_FROM = 32                      # tile.go:51: the contrast test applies from SampleCount > 32


def mask_runs(mask: np.ndarray) -> np.ndarray:
    """Runs of True in each row of a [H,W] mask as (x0, y, length, 1) tiles, int32 [n,4]."""
    h, w = mask.shape
    m = np.zeros((h, w + 2), np.int8)
    m[:, 1:-1] = mask
    d = np.diff(m, axis=1)
    ys, xs = np.nonzero(d == 1)
    ye, xe = np.nonzero(d == -1)           # same order: row-major, one end per start
    assert np.array_equal(ys, ye)
    return np.stack([xs, ys, xe - xs, np.ones_like(xs)], axis=1).astype(np.int32)


def batch_size(count: int, samples_per_pixel: int, max_samples: int) -> int:
    """Samples a pixel with `count` < max_samples gets when picked: main.go:861-883 adds samplesPerPixel samples but
    leaves the loop once SampleCount > maxSamplesPerPixel."""
    return min(samples_per_pixel, max_samples + 1 - count)


def pick_counts(samples_per_pixel: int, max_samples: int):
    """SampleCount of a pixel that never converges after each of its picks (the last one is where it stops)."""
    seq, c = [], 0
    while c < max_samples:
        c += batch_size(c, samples_per_pixel, max_samples)
        seq.append(c)
    return seq


def render_adaptive(renderer, params, camera, samples_per_pixel=32, max_samples=1024, device=None, log=None, merge=True):
    """Renders the frame until Tile.GetNoisiestPixel would return nil for every tile.  Returns (accum [H,W,3] f32,
    counts [H,W] i32, stats [H,W,9] f32: min / max luminance, Mean, M2, Variance) as torch CUDA tensors and the list of (samples in the call, eligible pixels).

    merge=False: one call per batch of samples_per_pixel, the eligible set re-read before each.  merge=True (same sums,
    bit for bit): TWO calls.  The contrast test (tile.go:51) first applies to a pixel holding more than 32 samples, so
    every pixel gets the batches up to that count whatever it shows; and a pixel's contrast is max - min luminance of
    its samples, which only grows with more samples, so a pixel that passes the test once passes it at every later
    count: all its remaining batches can be rendered at once."""
    import torch
    dev = device or torch.device("cuda", 0)
    H, W = params.height, params.width
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    counts = torch.zeros((H, W), dtype=torch.int32, device=dev)
    lum = torch.zeros((H, W, renderer.PIXEL_STATS), dtype=torch.float32, device=dev)
    seq = pick_counts(samples_per_pixel, max_samples)
    if merge:       # the count at which the contrast test first decides anything, then the stopping count
        tested = [c for c in seq[:-1] if c > CONTRAST_FROM]
        seq = ([tested[0]] if tested else []) + [seq[-1]]
    eligible = np.ones((H, W), bool)
    count, calls = 0, []
    for target in seq:
        if count > CONTRAST_FROM:           # every eligible pixel holds `count` samples: the :51 test decides who goes on
            contrast = (lum[..., 1] - lum[..., 0]).cpu().numpy()          # Pixel.Contrast, renderer.go:81 (float32)
            eligible &= ~(contrast < CONTRAST_DONE)
        if not eligible.any():
            break
        tiles = mask_runs(eligible)
        renderer.render_tiles_device(accum, counts, params, camera, tiles, spp=target - count, spp_begin=count, stats=lum)
        calls.append((target - count, int(eligible.sum())))
        if log:
            log(f"call {len(calls)}: samples {count}..{target} of {calls[-1][1]} pixels in {len(tiles)} runs")
        count = target
    return accum, counts, lum, calls
