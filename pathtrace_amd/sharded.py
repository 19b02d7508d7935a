"""Multi-GPU sharding of a frame: one process per GPU, BVH + scene replicated, RCCL reduce of the framebuffer.

The reference already renders a frame as independent tiles, one goroutine per tile (main.go:454-491, :935-937).
A GPU wants its whole share in ONE call (every wavefront iteration costs ~0.3 ms whatever the number of paths in
it, so a 64x64 tile never fills the machine): `mode="rows"` deals blocks of `row_block` rows round-robin to ranks
(`(y // row_block) % world == rank`, rendered by ptmi_render_rows_dev in one pass); `mode="tile"` keeps the
reference's granularity (`tile_id % world == rank`; all of a rank's tiles in one ptmi_render_tiles_dev call).  Every
rank adds its share into a zero-initialised full-frame accumulator, which is then summed onto `dst` with one
reduce (RCCL over xGMI; 4K frame = 99.5 MB + 33 MB of counts).  The reduce is the library's own: `PtmiComm` wraps the
C ABI's communicator (`ptmi_comm_init_rank`, `ptmi_reduce_frame`, `ptmi_render_sharded`: ncclReduce behind
include/ptmi.h, what a Go host calls too); `DistComm` (torch.distributed) only serves the CPU tests (gloo) and the
one-GPU rehearsal.  Regions are disjoint, so the reduced sums are bit-identical to a single-GPU render (x + 0 == x).  RNG streams are keyed by
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


class DistComm:
    """Framebuffer reduce over torch.distributed -- gloo on the CPU (tests) or any object with the same `reduce`
    (bench.py's host-staged rehearsal).  The product's GPU path uses PtmiComm."""
    kind = "dist"
    note = ""

    def __init__(self, dist, rank, world, group=None):
        self.dist, self.rank, self.world, self.group = dist, rank, world, group

    def reduce_frame(self, accum, counts, root=0, stream=None):
        if self.world > 1:
            self.dist.reduce(accum, dst=root, op=self.dist.ReduceOp.SUM, group=self.group)
            self.dist.reduce(counts, dst=root, op=self.dist.ReduceOp.SUM, group=self.group)

    def close(self):
        pass


class PtmiComm:
    """The C ABI's RCCL communicator of a context (include/ptmi.h "multi-GPU"): one rank per process.  `unique_id` is
    the 128 bytes rank 0 got from PtmiComm.unique_id(), handed over by any channel the host has."""
    kind = "rccl"
    note = ""

    def __init__(self, ctx, rank, world, unique_id: bytes):
        import ctypes as C
        from . import _lib
        from ._lib import check
        self._lib, self._check, self._C = _lib.load(), check, C
        self.ctx, self.rank, self.world = ctx, rank, world
        buf = (C.c_uint8 * _lib.UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        check(self._lib.ptmi_comm_init_rank(ctx.handle, world, rank, buf), ctx.handle)

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib
        from ._lib import check
        buf = (C.c_uint8 * _lib.UNIQUE_ID_BYTES)()
        check(_lib.load().ptmi_comm_unique_id(buf))
        return bytes(buf)

    def reduce_frame(self, accum, counts, root=0, stream=None):
        """In-place ncclReduce(sum) of FULL-FRAME torch CUDA tensors onto `root`, enqueued on `stream`."""
        C = self._C
        h, w = counts.shape
        s = C.c_void_p(stream) if stream else None
        self._check(self._lib.ptmi_reduce_frame(self.ctx.handle, accum.data_ptr(), counts.data_ptr(), accum.data_ptr(),
                                                counts.data_ptr(), w, h, root, s), self.ctx.handle)

    def close(self):
        if self.ctx is not None and self.ctx.handle:
            self._lib.ptmi_comm_destroy(self.ctx.handle)
        self.ctx = None


def make_comm(renderer, rank, world, dist, dev, rehearse=False):
    """The communicator of this rank's context.  The RCCL unique id travels over the process group the launcher set up
    (torch.distributed: a 128-byte broadcast); with rehearse=True (all ranks on one GPU, which RCCL refuses) the reduce
    goes through the host instead.

    The ranks agree on the outcome: if any of them cannot reach RCCL through the C ABI (library not found, init refused),
    ALL of them fall back to a DistComm over the launcher's process group -- a rank that went ahead alone would wait in
    ncclCommInitRank for ever.  The reason is kept in `.note`."""
    if world <= 1:
        return None
    if rehearse:
        return DistComm(dist, rank, world)
    import torch

    def all_ok(flag):
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def fallback(why):
        c = DistComm(dist, rank, world)
        c.note = why
        return c

    why, my_id = "", None
    try:                                   # loads and binds RCCL in this process; only rank 0's id is used
        my_id = PtmiComm.unique_id()
    except Exception as e:                 # noqa: BLE001 -- whatever it is, the ranks have to agree on it
        why = f"rank {rank}: {e}"
    if not all_ok(my_id is not None):
        return fallback(why or "another rank could not load RCCL through the C ABI")
    uid = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(my_id), dtype=torch.uint8))
    dist.broadcast(uid, src=0)
    comm = None
    try:
        comm = PtmiComm(renderer.ctx, rank, world, bytes(uid.cpu().numpy().tobytes()))
    except Exception as e:                 # noqa: BLE001
        why = f"rank {rank}: {e}"
    if not all_ok(comm is not None):
        if comm is not None:
            comm.close()
        return fallback(why or "another rank could not initialise its RCCL communicator")
    return comm


def render_frame_sharded(renderer, comm, accum, counts, params, camera, spp, row_block=8, root=0, stream=None, spp_begin=0):
    """One frame over all ranks: this rank's rows into the cleared full-frame CUDA tensors, then the reduce onto
    `root`.  With a PtmiComm (or no communicator) that is ONE C-ABI call, ptmi_render_sharded."""
    if comm is None or comm.kind == "rccl":
        renderer.render_sharded_device(accum, counts, params, camera, row_block=row_block, spp=spp, spp_begin=spp_begin,
                                       root=root, stream=stream)
        return accum, counts
    accum.zero_()
    counts.zero_()
    renderer.render_rows_device(accum, counts, params, camera, shard=comm.rank, nshards=comm.world, row_block=row_block,
                                spp=spp, spp_begin=spp_begin, stream=stream)
    comm.reduce_frame(accum, counts, root, stream)
    return accum, counts


def render_sharded(render_rect, accum, counts, width, height, spp, rank=0, world=1, dist=None, dst=0, mode="tile",
                   tile=64, group=None, render_rows=None, row_block=8, render_tiles=None, comm=None, stream=None):
    """Render this rank's shard with `render_rect(rect, spp_begin, spp)` (which must ADD into the rows/cols of the
    full-frame tensors `accum` [H,W,3] / `counts` [H,W] it is given views of) and reduce onto `dst` through `comm`
    (PtmiComm on GPUs; `dist=` wraps a torch.distributed module in a DistComm for the CPU tests).

    render_rect(accum_view, counts_view, rect, spp_begin, spp_count) -> None
    """
    if comm is None and dist is not None and world > 1:
        comm = DistComm(dist, rank, world, group)
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
    if comm is not None and world > 1:
        comm.reduce_frame(accum, counts, dst, stream)
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
