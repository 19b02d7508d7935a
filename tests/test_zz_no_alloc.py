"""ptmi_closest_hit_dev_noalloc -- the allocation-free form of ptmi_closest_hit_dev (static __device__ scratch in the
library, no hipMalloc in the call) -- against the existing entry point, through the C ABI, bit for bit.

The file sorts last on purpose: the round's end runs `pytest -x`, and this entry point is an extra, so a failure here
must not hide the parity tests.  It passes alone in a fresh process and after the other tests.
"""
import ctypes as C

import numpy as np
import pytest

from pathtrace_amd import BuildBVH, ConstructLinearBVH, scenes, _lib
import oracle_lib
from helpers import random_rays, adversarial_rays, bits_equal

pytestmark = pytest.mark.gpu

NOALLOC_MAX_RAYS = 1 << 32          # PTMI_NOALLOC_MAX_RAYS (include/ptmi.h)


def _dev(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_noalloc_matches_the_existing_entry_point_twice_in_a_row(gpu_ctx):
    import torch
    dev = torch.device("cuda", 0)
    v, f = scenes.make_soup(120_000, seed=77)
    box = BuildBVH(v, f)
    bvh = ConstructLinearBVH(box, ctx=gpu_ctx)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream

    def both(o, d, tmax):
        n = o.shape[0]
        O, D, T = _dev(o, dev), _dev(d, dev), _dev(np.broadcast_to(np.float32(tmax), (n,)).copy(), dev)
        P0 = torch.full((n,), -5, dtype=torch.int32, device=dev); t0 = torch.full((n,), -5.0, dtype=torch.float32, device=dev)
        P1 = torch.full((n,), -6, dtype=torch.int32, device=dev); t1 = torch.full((n,), -6.0, dtype=torch.float32, device=dev)
        rc = lib.ptmi_closest_hit_dev(gpu_ctx.handle, O.data_ptr(), D.data_ptr(), T.data_ptr(), n, P0.data_ptr(), t0.data_ptr(), C.c_void_p(st))
        assert rc == 0, lib.ptmi_last_error(gpu_ctx.handle)
        rc = lib.ptmi_closest_hit_dev_noalloc(gpu_ctx.handle, O.data_ptr(), D.data_ptr(), T.data_ptr(), n, P1.data_ptr(), t1.data_ptr(), C.c_void_p(st))
        assert rc == 0, lib.ptmi_last_error(gpu_ctx.handle)
        torch.cuda.synchronize()
        return P0.cpu().numpy(), t0.cpu().numpy(), P1.cpu().numpy(), t1.cpu().numpy()

    # first call: a batch large enough for every persistent wave to draw several chunks from the cursor
    o, d = random_rays(v, 3_000_000, 5)
    p0, t0, p1, t1 = both(o, d, 1000.0)
    assert np.array_equal(p0, p1) and bits_equal(t0, t1)
    assert 0.05 < (p1 >= 0).mean() < 1.0
    # second call, straight after, with a smaller input (the scratch still holds the first call's cursor): ragged size,
    # adversarial rays (NaN slab path, ties), per-ray tmax
    o2, d2, tm2 = adversarial_rays(v, f, 70_001, 9)
    q0, u0, q1, u1 = both(o2, d2, tm2)
    assert np.array_equal(q0, q1) and bits_equal(u0, u1)
    # ... and both equal the oracle
    orc = oracle_lib.OracleBVH(v, f, linear=box.linear())
    op, ot, _ = orc.closest_hit(o2, d2, tm2, counters=False)
    assert np.array_equal(q1, op) and bits_equal(u1, ot)
    idx = np.arange(0, o.shape[0], 37)
    op, ot, _ = orc.closest_hit(np.ascontiguousarray(o[idx]), np.ascontiguousarray(d[idx]), 1000.0, counters=False)
    assert np.array_equal(p1[idx], op) and bits_equal(t1[idx], ot)
    # a third, tiny call (fewer rays than one wave) and an empty one
    r0, s0, r1, s1 = both(o[:7], d[:7], 1000.0)
    assert np.array_equal(r0, r1) and bits_equal(s0, s1) and np.array_equal(r1, p1[:7])
    assert lib.ptmi_closest_hit_dev_noalloc(gpu_ctx.handle, None, None, None, 0, None, None, None) == 0


def test_noalloc_rejects_sizes_beyond_its_maximum_and_bad_arguments(gpu_ctx):
    from pathtrace_amd import Context
    lib = _lib.load()
    v, f = scenes.make_soup(2000, seed=3)
    ConstructLinearBVH(BuildBVH(v, f), ctx=gpu_ctx)
    # the size is checked before any buffer is looked at
    rc = lib.ptmi_closest_hit_dev_noalloc(gpu_ctx.handle, None, None, None, NOALLOC_MAX_RAYS + 1, None, None, None)
    assert rc == -1
    msg = lib.ptmi_last_error(gpu_ctx.handle).decode()
    assert "maximum" in msg and str(NOALLOC_MAX_RAYS) in msg
    assert lib.ptmi_closest_hit_dev_noalloc(gpu_ctx.handle, None, None, None, 5, None, None, None) == -1      # null buffers
    assert lib.ptmi_closest_hit_dev_noalloc(None, None, None, None, 5, None, None, None) == -1
    fresh = Context(0)
    try:
        assert lib.ptmi_closest_hit_dev_noalloc(fresh.handle, None, None, None, 5, None, None, None) == -4     # no BVH
    finally:
        fresh.close()


def test_noalloc_runs_under_stream_capture(gpu_ctx):
    """What the entry point is for: a caller that records a HIP graph.  torch captures in "global" mode, in which an
    allocation made by this thread during the capture invalidates it; the replayed graph gives the existing entry point's
    results."""
    import torch
    dev = torch.device("cuda", 0)
    v, f = scenes.make_soup(50_000, seed=19)
    bvh = ConstructLinearBVH(BuildBVH(v, f), ctx=gpu_ctx)
    n = 400_000
    o, d = random_rays(v, n, 23)
    O, D = _dev(o, dev), _dev(d, dev)
    T = torch.full((n,), 1000.0, dtype=torch.float32, device=dev)
    P0 = torch.empty(n, dtype=torch.int32, device=dev); t0 = torch.empty(n, dtype=torch.float32, device=dev)
    P1 = torch.full((n,), -9, dtype=torch.int32, device=dev); t1 = torch.full((n,), -9.0, dtype=torch.float32, device=dev)
    bvh.CheckIntersectionDevice(O, D, T, P0, t0, torch.cuda.current_stream().cuda_stream)
    bvh.CheckIntersectionDeviceNoAlloc(O, D, T, P1, t1, torch.cuda.current_stream().cuda_stream)     # warm: code object loaded
    torch.cuda.synchronize()
    P1.fill_(-9); t1.fill_(-9.0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        bvh.CheckIntersectionDeviceNoAlloc(O, D, T, P1, t1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for _ in range(2):
        P1.fill_(-9); t1.fill_(-9.0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(P0, P1) and torch.equal(t0.view(torch.int32), t1.view(torch.int32))
