"""Multi-GPU sharding of a frame: one process per GPU, BVH + scene replicated, RCCL reduce of the framebuffer.

The reference already renders a frame as independent tiles, one goroutine per tile (main.go:454-491, :935-937).
A GPU wants its whole share in ONE call (every wavefront iteration costs ~0.3 ms whatever the number of paths in
it, so a 64x64 tile never fills the machine): `mode="rows"` deals blocks of `row_block` rows round-robin to ranks
(`(y // row_block) % world == rank`, rendered by ptmi_render_rows_dev in one pass); `mode="tile"` keeps the
reference's granularity (`tile_id % world == rank`; all of a rank's tiles in one ptmi_render_tiles_dev call).  Every
rank adds its share into a zero-initialised full-frame accumulator, which is then summed onto `dst` with one
`torch.distributed.reduce` (backend "nccl" = RCCL over xGMI; 4K frame = 99.5 MB + 33 MB of counts).  Regions are
disjoint, so the reduced sums are bit-identical to a single-GPU render (x + 0 == x).  RNG streams are keyed by
(pixel, sample), so the image does not depend on the number of ranks.  `mode="sample"` splits the sample range
instead (for small frames): partial sums are then added in rank order, deterministic but not bit-identical to
one GPU's sequential sum.
"""
from __future__ import annotations


def tile_rects(width, height, tile=64):
    """All tiles of the frame, row-major, as (x0, y0, w, h)."""
    out = []
    for y0 in range(0, height, tile):
        for x0 in range(0, width, tile):
            out.append((x0, y0, min(tile, width - x0), min(tile, height - y0)))
    return out


def my_tiles(width, height, world, rank, tile=64):
    """Tile-interleaved shard of `rank`."""
    return [r for i, r in enumerate(tile_rects(width, height, tile)) if i % world == rank]


def my_rows(height, world, rank, row_block=8):
    """Rows of `rank` under the row-block interleave of ptmi_render_rows_dev."""
    return [y for y in range(height) if (y // row_block) % world == rank]


def sample_range(spp, world, rank):
    """Contiguous sample sub-range of `rank` (first ranks take the remainder)."""
    base, rem = divmod(spp, world)
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def merge_rects(rects):
    """Merge horizontally adjacent tiles of one row into wider rectangles (fewer render calls)."""
    out = []
    for r in sorted(rects, key=lambda r: (r[1], r[0])):
        if out and out[-1][1] == r[1] and out[-1][3] == r[3] and out[-1][0] + out[-1][2] == r[0]:
            out[-1] = (out[-1][0], r[1], out[-1][2] + r[2], r[3])
        else:
            out.append(r)
    return out


def render_sharded(render_rect, accum, counts, width, height, spp, rank=0, world=1, dist=None, dst=0, mode="tile",
                   tile=64, group=None, render_rows=None, row_block=8, render_tiles=None):
    """Render this rank's shard with `render_rect(rect, spp_begin, spp)` (which must ADD into the rows/cols of the
    full-frame tensors `accum` [H,W,3] / `counts` [H,W] it is given views of) and reduce onto `dst`.

    render_rect(accum_view, counts_view, rect, spp_begin, spp_count) -> None
    """
    accum.zero_()
    counts.zero_()
    if mode == "tile":
        mine = my_tiles(width, height, world, rank, tile)
        if render_tiles is not None:           # all of this rank's tiles in one call (ptmi_render_tiles_dev)
            if mine:
                render_tiles(accum, counts, mine, 0, spp)
        else:
            for (x0, y0, w, h) in mine:
                render_rect(accum, counts, (x0, y0, w, h), 0, spp)
    elif mode == "rows":
        # render_rows(accum, counts, shard, nshards, row_block, spp_begin, spp_count): one call for the whole share
        if render_rows is None:
            raise ValueError("mode='rows' needs render_rows")
        render_rows(accum, counts, rank, world, row_block, 0, spp)
    elif mode == "sample":
        b, n = sample_range(spp, world, rank)
        if n:
            render_rect(accum, counts, (0, 0, width, height), b, n)
    else:
        raise ValueError(mode)
    if dist is not None and world > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
        dist.reduce(counts, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return accum, counts


def make_device_render_rect(renderer, params, camera, stream=None):
    """render_rect callable for the real backend: full-frame CUDA tensors, rectangle rendered into a contiguous
    staging tensor by ptmi_render_dev and added into the frame."""
    import torch

    def render_rect(accum, counts, rect, spp_begin, spp_count):
        x0, y0, w, h = rect
        if x0 == 0 and w == accum.shape[1]:
            a, c = accum[y0:y0 + h], counts[y0:y0 + h]          # full-width rows are contiguous
            renderer.render_device(a, c, params, camera, rect, spp_count, spp_begin, stream)
            return
        a = torch.zeros((h, w, 3), dtype=torch.float32, device=accum.device)
        c = torch.zeros((h, w), dtype=counts.dtype, device=accum.device)
        renderer.render_device(a, c, params, camera, rect, spp_count, spp_begin, stream)
        accum[y0:y0 + h, x0:x0 + w] += a
        counts[y0:y0 + h, x0:x0 + w] += c

    return render_rect


def make_device_render_tiles(renderer, params, camera, stream=None):
    """render_tiles callable for the real backend (ptmi_render_tiles_dev on full-frame CUDA tensors)."""
    def render_tiles(accum, counts, tiles, spp_begin, spp_count):
        renderer.render_tiles_device(accum, counts, params, camera, tiles, spp_count, spp_begin, stream)

    return render_tiles


def make_device_render_rows(renderer, params, camera, stream=None):
    """render_rows callable for the real backend (ptmi_render_rows_dev on full-frame CUDA tensors)."""
    def render_rows(accum, counts, shard, nshards, row_block, spp_begin, spp_count):
        renderer.render_rows_device(accum, counts, params, camera, shard, nshards, row_block, spp_count, spp_begin, stream)

    return render_rows
