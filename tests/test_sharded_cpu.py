"""N>1 path on CPU: tile / sample sharding + framebuffer reduce under torch.distributed (gloo, world_size 2).
The tile renderer is a deterministic stand-in (the real one needs a GPU); what is tested is the host logic:
coverage, disjointness, reduce, and independence of the result from the number of ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pathtrace_amd import sharded


def _fake_render_rect(accum, counts, rect, spp_begin, spp_count):
    x0, y0, w, h = rect
    ys, xs = torch.meshgrid(torch.arange(y0, y0 + h), torch.arange(x0, x0 + w), indexing="ij")
    for s in range(spp_begin, spp_begin + spp_count):
        val = ((xs * 31 + ys * 17 + s * 7) % 251).to(torch.float32) / 8.0          # exactly representable
        accum[y0:y0 + h, x0:x0 + w, 0] += val
        accum[y0:y0 + h, x0:x0 + w, 1] += val * 2
        accum[y0:y0 + h, x0:x0 + w, 2] += 1.0
        counts[y0:y0 + h, x0:x0 + w] += 1


def _fake_render_rows(accum, counts, shard, nshards, row_block, spp_begin, spp_count):
    H, W = counts.shape
    for y in sharded.my_rows(H, nshards, shard, row_block):
        _fake_render_rect(accum, counts, (0, y, W, 1), spp_begin, spp_count)


def _single(width, height, spp):
    a = torch.zeros((height, width, 3)); c = torch.zeros((height, width), dtype=torch.int32)
    _fake_render_rect(a, c, (0, 0, width, height), 0, spp)
    return a, c


def test_tile_plan_covers_frame_exactly_once():
    for (W, H, tile, world) in [(512, 512, 64, 4), (100, 70, 32, 3), (3840, 2160, 64, 8), (17, 5, 64, 2)]:
        cover = np.zeros((H, W), np.int32)
        per_rank = []
        for r in range(world):
            rects = sharded.my_tiles(W, H, world, r, tile)
            per_rank.append(sum(w * h for _, _, w, h in rects))
            for (x0, y0, w, h) in sharded.merge_rects(rects):
                cover[y0:y0 + h, x0:x0 + w] += 1
        assert (cover == 1).all()
        if W * H >= 512 * 512:
            assert max(per_rank) <= 1.2 * min(per_rank)          # interleaving balances the load
    for (H, world, block) in [(1080, 8, 8), (70, 3, 8), (5, 2, 8), (2160, 8, 16), (9, 4, 1)]:
        rows = [sharded.my_rows(H, world, r, block) for r in range(world)]
        assert sorted(y for rr in rows for y in rr) == list(range(H))
        if H >= 1000:
            assert max(map(len, rows)) - min(map(len, rows)) <= block
    assert [sharded.sample_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert sum(n for _, n in (sharded.sample_range(7, 8, r) for r in range(8))) == 7


def _worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, spp = 96, 80, 5
    a = torch.zeros((H, W, 3)); c = torch.zeros((H, W), dtype=torch.int32)
    sharded.render_sharded(_fake_render_rect, a, c, W, H, spp, rank=rank, world=world, dist=dist, dst=0, mode=mode,
                           tile=32, render_rows=_fake_render_rows, row_block=8)
    if rank == 0:
        ra, rc = _single(W, H, spp)
        q.put((bool(torch.equal(a, ra)), bool(torch.equal(c, rc))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["tile", "rows", "sample"])
def test_reduce_matches_single_process(mode):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok == (True, True)
