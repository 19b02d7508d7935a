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
    render_tiles = None
    if mode == "tile_one_call":         # bench.py's render_c4 leg: a rank's tiles go down in ONE call (ptmi_render_tiles_dev)
        mode = "tile"
        calls = []

        def render_tiles(accum, counts, tiles, spp_begin, spp_count):
            calls.append(len(tiles))
            for rect in tiles:
                _fake_render_rect(accum, counts, rect, spp_begin, spp_count)
    sharded.render_sharded(_fake_render_rect, a, c, W, H, spp, rank=rank, world=world, dist=dist, dst=0, mode=mode,
                           tile=32, render_rows=_fake_render_rows, row_block=8, render_tiles=render_tiles)
    if render_tiles is not None:
        assert calls == [len(sharded.my_tiles(W, H, world, rank, 32))]
    if rank == 0:
        ra, rc = _single(W, H, spp)
        q.put((bool(torch.equal(a, ra)), bool(torch.equal(c, rc))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["tile", "tile_one_call", "rows", "sample"])
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


class _FakeRenderer:
    """Stands in for render.Renderer: only the call render_frame_sharded makes without an RCCL communicator."""

    def render_rows_device(self, accum, counts, params, camera, shard=0, nshards=1, row_block=8, spp=32, spp_begin=0, stream=None):
        _fake_render_rows(accum, counts, shard, nshards, row_block, spp_begin, spp)

    def render_sharded_device(self, *a, **k):
        raise AssertionError("the C-ABI RCCL path needs a GPU: the CPU test must go through the stub communicator")


def _frame_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, spp = 96, 83, 4
    a = torch.full((H, W, 3), 5.0); c = torch.full((H, W), 5, dtype=torch.int32)        # stale contents must be cleared
    comm = sharded.make_comm(None, rank, world, dist, None, rehearse=True)           # the gloo stub of the communicator
    assert comm.kind == "dist" and (comm.rank, comm.world) == (rank, world)
    sharded.render_frame_sharded(_FakeRenderer(), comm, a, c, None, None, spp, row_block=8, root=0)
    if rank == 0:
        ra, rc = _single(W, H, spp)
        q.put((bool(torch.equal(a, ra)), bool(torch.equal(c, rc))))
    dist.barrier()
    dist.destroy_process_group()


def test_render_frame_sharded_world_2_with_stub_comm():
    """render_frame_sharded -- what bench.py's render leg and a multi-GPU host run -- with two ranks: the per-rank row
    shards and one reduce through the communicator interface (PtmiComm on GPUs, the gloo stub here) give the frame."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_frame_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok == (True, True)


def _agree_worker(rank, world, port, case, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class _Comm:                                             # what PtmiComm(ctx, rank, world, id) would be
        kind, note, closed = "rccl", "", False

        def __init__(self, ctx, r, w, uid):
            if case == "init_fails_on_rank_1" and r == 1:
                raise RuntimeError("ncclCommInitRank: unhandled system error")
            assert uid == bytes(range(128)), "every rank must receive rank 0's id"

        def close(self):
            type(self).closed = True

        @staticmethod
        def unique_id():
            if case == "load_fails_on_rank_1" and rank == 1:
                raise RuntimeError("RCCL unavailable: librccl.so not found")
            return bytes(range(128)) if rank == 0 else bytes(128)          # only rank 0's id may be used

    sharded.PtmiComm = _Comm
    renderer = type("R", (), {"ctx": None})()
    comm = sharded.make_comm(renderer, rank, world, dist, torch.device("cpu"))
    q.put((rank, comm.kind, comm.note, _Comm.closed))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["all_fine", "load_fails_on_rank_1", "init_fails_on_rank_1"])
def test_ranks_agree_on_the_communicator(case):
    """make_comm's outcome is collective: if one rank cannot load RCCL through the C ABI, or its communicator fails to
    initialise, EVERY rank falls back to the launcher's process group (a rank that went ahead alone would hang in the
    next collective); the rank that got a communicator closes it again."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    kinds = [g[1] for g in got]
    if case == "all_fine":
        assert kinds == ["rccl", "rccl"]
    else:
        assert kinds == ["dist", "dist"]
        assert "rank 1" in got[1][2] and got[0][2]                 # rank 1 knows why, rank 0 knows it was another rank
        if case == "init_fails_on_rank_1":
            assert got[0][3] is True                               # rank 0's communicator was closed again


def test_comm_entry_points_fail_cleanly_without_rccl():
    """The comm entry points bind RCCL at run time; when it cannot be loaded they return PTMI_ERR_COMM with a reason,
    and bad arguments are refused before that.  (A child process: the binding is made once per process.)"""
    import subprocess
    import sys
    code = (
        "import ctypes as C\n"
        "from pathtrace_amd import _lib\n"
        "lib = _lib.load()\n"
        "assert lib.ptmi_comm_unique_id(None) == -1\n"
        "assert lib.ptmi_comm_init_rank(None, 1, 0, None) == -1\n"
        "assert lib.ptmi_comm_init_all(None, 0) == -1\n"
        "assert lib.ptmi_reduce_frame(None, None, None, None, None, 4, 4, 0, None) == -1\n"
        "buf = (C.c_uint8 * 128)()\n"
        "rc = lib.ptmi_comm_unique_id(buf)\n"
        "msg = lib.ptmi_last_error(None).decode()\n"
        "assert rc == -8 and 'RCCL unavailable' in msg and '/nonexistent/librccl.so' in msg, (rc, msg)\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PTMI_RCCL_LIB="/nonexistent/librccl.so", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr
