"""GPU parity: HIP extend/connect kernels (through the C ABI) vs the CPU oracle.

Bar (BASELINE.json north_star): primitive id bit-exact, t within 1e-5 relative -- we require t
bit-exact too, since the kernel keeps the reference's operation order with FMA contraction off.
"""
import numpy as np
import pytest

from pathtrace_amd import BuildBVH, ConstructLinearBVH, scenes
import oracle_lib
from helpers import random_rays, adversarial_rays, bits_equal

pytestmark = pytest.mark.gpu


def _check(bvh, orc, o, d, tmax, label):
    hit, t, prim = bvh.CheckIntersection((o, d), tmax)
    oprim, ot, _ = orc.closest_hit(o, d, tmax)
    nbad = int((prim != oprim).sum())
    assert nbad == 0, f"{label}: {nbad}/{len(prim)} closest-hit prim ids differ"
    assert bits_equal(t, ot), f"{label}: t differs in {(t.view(np.uint32) != ot.view(np.uint32)).sum()} rays"
    occ = bvh.QuickCheckIntersection((o, d), tmax)
    oocc, _ = orc.any_hit(o, d, tmax)
    assert np.array_equal(occ, oocc), f"{label}: {(occ != oocc).sum()} any-hit results differ"
    return hit


@pytest.mark.parametrize("ntris,seed", [(0, 1), (1, 2), (3, 3), (4, 4), (5, 5), (37, 6), (2000, 7), (70000, 8)])
def test_soup_parity(gpu_ctx, ntris, seed):
    if ntris == 0:
        v = np.zeros((3, 3), np.float32); f = np.zeros((0, 3), np.uint32)
    else:
        v, f = scenes.make_soup(ntris, seed)
    box = BuildBVH(v, f)
    bvh = ConstructLinearBVH(box, ctx=gpu_ctx)
    orc = oracle_lib.OracleBVH(v, f, linear=box.linear())
    nr = 20000 if ntris < 70000 else 200000
    vv = v if ntris else np.array([[-1, -1, -1], [1, 1, 1]], np.float32)
    o, d = random_rays(vv, nr, seed + 100)
    hit = _check(bvh, orc, o, d, 1000.0, f"soup{ntris}/random")
    if ntris >= 2000:
        assert hit.mean() > 0.05
    if ntris:
        o, d, tmax = adversarial_rays(v, f, 4096, seed + 200)
        _check(bvh, orc, o, d, tmax, f"soup{ntris}/adversarial")


def test_scene_parity_with_camera_rays(gpu_ctx):
    sc = scenes.bunny_scene(small=False)
    box = BuildBVH(sc.verts, sc.tris)
    bvh = ConstructLinearBVH(box, ctx=gpu_ctx)
    orc = oracle_lib.OracleBVH(sc.verts, sc.tris, linear=box.linear())
    o, d = scenes.camera_rays(sc.camera, 256, 256)
    hit = _check(bvh, orc, o, d, 1000.0, "bunny/primary")
    assert hit.mean() > 0.3
    prim = bvh.CheckIntersection((o, d), 1000.0)[2]
    t = bvh.CheckIntersection((o, d), 1000.0)[1]
    o2, d2 = scenes.cosine_bounce_rays(sc.verts, sc.tris, o, d, prim, t, seed=42)
    _check(bvh, orc, o2, d2, 1000.0, "bunny/secondary")
    # short segments (the reference marches in stepSize segments, tracer.go:34-35)
    _check(bvh, orc, o, d, 2.5, "bunny/short-step")


def test_counters_match_oracle(gpu_ctx):
    v, f = scenes.make_soup(20000, 11)
    box = BuildBVH(v, f)
    bvh = ConstructLinearBVH(box, ctx=gpu_ctx)
    orc = oracle_lib.OracleBVH(v, f, linear=box.linear())
    o, d = random_rays(v, 50000, 5)
    bvh.set_counting(True)
    bvh.reset_counters()
    bvh.CheckIntersection((o, d), 1000.0)
    c = bvh.counters()
    _, _, oc = orc.closest_hit(o, d, 1000.0)
    assert (c.closest_queries, c.inner_visits, c.tri_tests, c.leaf_visits) == \
        (50000, oc["inner_visits"], oc["tri_tests"], oc["leaf_visits"])
    bvh.reset_counters()
    bvh.QuickCheckIntersection((o, d), 1000.0)
    c = bvh.counters()
    _, oc = orc.any_hit(o, d, 1000.0)
    assert (c.any_queries, c.inner_visits, c.tri_tests, c.leaf_visits) == \
        (50000, oc["inner_visits"], oc["tri_tests"], oc["leaf_visits"])
    bvh.set_counting(False)


def test_upload_prebuilt_tree_and_errors(gpu_ctx):
    from pathtrace_amd import UploadLinearBVH, PtmiError, Context, LinearBVH
    v, f = scenes.make_soup(3000, 21)
    box = BuildBVH(v, f)
    bvh = UploadLinearBVH(v, f, box.linear(), ctx=gpu_ctx)
    orc = oracle_lib.OracleBVH(v, f)        # literal oracle builder this time
    o, d = random_rays(v, 10000, 6)
    _check(bvh, orc, o, d, 1000.0, "uploaded")
    lin = box.linear()
    lin.second_child = lin.second_child.copy(); lin.second_child[0] = 10 ** 9
    with pytest.raises(PtmiError):
        UploadLinearBVH(v, f, lin, ctx=gpu_ctx)
    fresh = Context(0)
    with pytest.raises(PtmiError) as e:
        LinearBVH(fresh).CheckIntersection((o, d), 1.0)
    assert e.value.code == -4
    fresh.close()


def test_dev_calls_overlapping_on_two_streams(gpu_ctx):
    """ptmi_closest_hit_dev / ptmi_any_hit_dev are asynchronous on caller-chosen streams, so launches of one context can
    overlap on the device: each launch has a ray cursor of its own (a ring of counter lines), and every result equals the
    oracle's.  Several rounds of closest-hit on stream A against any-hit (and a second closest-hit batch) on stream B,
    both non-blocking streams, nothing synchronised in between."""
    import torch
    v, f = scenes.make_soup(200_000, seed=31)
    box = BuildBVH(v, f)
    bvh = ConstructLinearBVH(box, ctx=gpu_ctx)
    orc = oracle_lib.OracleBVH(v, f, linear=box.linear())
    dev = torch.device("cuda", 0)
    n = 1 << 20
    oa, da = random_rays(v, n, 71)
    ob, db = random_rays(v, n // 2, 72)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    OA, DA, OB, DB = (torch.from_numpy(x).to(dev) for x in (oa, da, ob, db))
    TA = torch.full((n,), 1000.0, dtype=torch.float32, device=dev); TB = torch.full((n // 2,), 1000.0, dtype=torch.float32, device=dev)
    rounds = 4
    PA = [torch.full((n,), -7, dtype=torch.int32, device=dev) for _ in range(rounds)]
    tA = [torch.full((n,), -7.0, dtype=torch.float32, device=dev) for _ in range(rounds)]
    OCC = [torch.full((n // 2,), 9, dtype=torch.uint8, device=dev) for _ in range(rounds)]
    PB = [torch.full((n // 2,), -7, dtype=torch.int32, device=dev) for _ in range(rounds)]
    tB = [torch.full((n // 2,), -7.0, dtype=torch.float32, device=dev) for _ in range(rounds)]
    torch.cuda.synchronize()
    for r in range(rounds):
        bvh.CheckIntersectionDevice(OA, DA, TA, PA[r], tA[r], sa.cuda_stream)
        bvh.QuickCheckIntersectionDevice(OB, DB, TB, OCC[r], sb.cuda_stream)
        bvh.CheckIntersectionDevice(OB, DB, TB, PB[r], tB[r], sb.cuda_stream)
    torch.cuda.synchronize()
    pa, ta, _ = orc.closest_hit(oa, da, 1000.0, counters=False)
    pb, tb, _ = orc.closest_hit(ob, db, 1000.0, counters=False)
    occ, _ = orc.any_hit(ob, db, 1000.0, counters=False)
    for r in range(rounds):
        assert np.array_equal(PA[r].cpu().numpy(), pa) and bits_equal(tA[r].cpu().numpy(), ta), f"round {r}: stream A"
        assert np.array_equal(PB[r].cpu().numpy(), pb) and bits_equal(tB[r].cpu().numpy(), tb), f"round {r}: stream B closest"
        assert np.array_equal(OCC[r].cpu().numpy().astype(bool), occ), f"round {r}: stream B any-hit"


import glob
import os

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_vectors_on_gpu(gpu_ctx, path):
    """HIP path vs the committed golden vectors (inputs + expected outputs, tests/golden/)."""
    g = np.load(path)
    bvh = ConstructLinearBVH(BuildBVH(g["verts"], g["tris"]), ctx=gpu_ctx)
    hit, t, prim = bvh.CheckIntersection((g["o"], g["d"]), g["tmax"])
    assert np.array_equal(prim, g["prim"]) and bits_equal(t, g["t"])
    occ = bvh.QuickCheckIntersection((g["o"], g["d"]), g["tmax"])
    assert np.array_equal(occ, g["occluded"].astype(bool))
    bvh.set_counting(True); bvh.reset_counters()
    bvh.CheckIntersection((g["o"], g["d"]), g["tmax"])
    c = bvh.counters()
    bvh.set_counting(False)
    assert [c.inner_visits, c.tri_tests, c.leaf_visits] == g["closest_counters"].tolist()


def test_full_size_batch_properties(gpu_ctx, monkeypatch):
    """BASELINE.json configs[1] at full size (1 M triangles, 16.7 M primary + secondary rays), checked through
    properties that need no full CPU pass: the vote-scheduled and the one-shot kernels (two independent
    implementations) agree bit for bit on every ray, results do not depend on batch order or on a re-run, every hit
    lies in (0, tmax], and a 1-in-64 sample equals the oracle."""
    import torch
    from pathtrace_amd import Context
    v, f = scenes.make_soup(1_000_000, seed=42)
    box = BuildBVH(v, f)
    bvh = ConstructLinearBVH(box, ctx=gpu_ctx)
    monkeypatch.setenv("PTMI_EXTEND_VARIANT", "0")
    ctx0 = Context(0)                                     # a second context that runs the one-shot kernels
    monkeypatch.delenv("PTMI_EXTEND_VARIANT")
    bvh0 = ConstructLinearBVH(box, ctx=ctx0)
    cam = scenes.soup_camera(v)
    o1, d1 = scenes.camera_rays(cam, 4096, 2048)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream

    def closest(b, o, d, tmax):
        n = o.shape[0]
        O, D = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        T = torch.full((n,), tmax, dtype=torch.float32, device=dev)
        P = torch.empty(n, dtype=torch.int32, device=dev); t = torch.empty(n, dtype=torch.float32, device=dev)
        occ = torch.empty(n, dtype=torch.uint8, device=dev)
        b.CheckIntersectionDevice(O, D, T, P, t, st)
        b.QuickCheckIntersectionDevice(O, D, T, occ, st)
        torch.cuda.synchronize()
        return P.cpu().numpy(), t.cpu().numpy(), occ.cpu().numpy()

    p1, t1, _ = closest(bvh, o1, d1, 1000.0)
    o2, d2 = scenes.cosine_bounce_rays(v, f, o1, d1, p1, t1, seed=42)
    o = np.concatenate([o1, o2], 0); d = np.concatenate([d1, d2], 0)
    n = o.shape[0]
    assert n == 1 << 24
    P, T, OCC = closest(bvh, o, d, 1000.0)
    # (1) two kernels, one answer
    P0, T0, OCC0 = closest(bvh0, o, d, 1000.0)
    assert np.array_equal(P, P0) and bits_equal(T, T0) and np.array_equal(OCC, OCC0)
    # (2) idempotent, and independent of the order of the batch (scheduling, refill and chunking play no part)
    Pb, Tb, OCCb = closest(bvh, o, d, 1000.0)
    assert np.array_equal(P, Pb) and bits_equal(T, Tb) and np.array_equal(OCC, OCCb)
    perm = np.random.default_rng(5).permutation(n)[: 1 << 22]
    Pp, Tp, OCCp = closest(bvh, np.ascontiguousarray(o[perm]), np.ascontiguousarray(d[perm]), 1000.0)
    assert np.array_equal(Pp, P[perm]) and bits_equal(Tp, T[perm]) and np.array_equal(OCCp, OCC[perm])
    # (3) range: hits lie in (0, tmax], misses report t = 0 and prim = -1; ids are valid triangles
    hit = P >= 0
    assert 0.3 < hit.mean() < 0.95
    assert (T[hit] > 0).all() and (T[hit] <= 1000.0).all() and (T[~hit] == 0).all() and (P[~hit] == -1).all()
    assert P.max() < f.shape[0]
    # (4) a shorter segment can only lose hits beyond it: same answer where t <= tmax', a miss or a nearer hit otherwise
    sub = np.arange(0, n, 16)
    Ps, Ts, _ = closest(bvh, np.ascontiguousarray(o[sub]), np.ascontiguousarray(d[sub]), 3.0)
    near = hit[sub] & (T[sub] <= 3.0)
    assert np.array_equal(Ps[near], P[sub][near]) and bits_equal(Ts[near], T[sub][near])
    assert (Ps[~near] == -1).all()
    # (5) the oracle on every 64th ray
    idx = np.arange(0, n, 64)
    orc = oracle_lib.OracleBVH(v, f, linear=box.linear())
    op, ot, _ = orc.closest_hit(np.ascontiguousarray(o[idx]), np.ascontiguousarray(d[idx]), 1000.0)
    oocc, _ = orc.any_hit(np.ascontiguousarray(o[idx]), np.ascontiguousarray(d[idx]), 1000.0)
    assert np.array_equal(P[idx], op) and bits_equal(T[idx], ot) and np.array_equal(OCC[idx].astype(bool), oocc)


def test_host_batches_from_page_locked_buffers_go_through_in_chunks(gpu_ctx):
    """ptmi_closest_hit / ptmi_any_hit with page-locked caller buffers run as 2 M-ray chunks on two streams (copies
    overlapped with the walk); 2.5 chunks here.  The answers equal the one-chunk path (pageable buffers) bit for bit and
    the oracle on a sample."""
    import ctypes as C
    import torch
    from pathtrace_amd import _lib
    lib = _lib.load()
    v, f = scenes.make_soup(20000, 31)
    box = BuildBVH(v, f)
    bvh = ConstructLinearBVH(box, ctx=gpu_ctx)
    n = (5 << 20) + 12345
    o, d = random_rays(v, n, 9)
    tmax = np.full(n, 1000.0, np.float32)
    tmax[::7] = 3.0                                              # a mix of ranges
    hit, t_ref, prim_ref = bvh.CheckIntersection((o, d), tmax)   # numpy arrays: pageable, one chunk
    occ_ref = bvh.QuickCheckIntersection((o, d), tmax)
    pin = lambda x: torch.from_numpy(np.ascontiguousarray(x)).pin_memory()
    O, D, T = pin(o), pin(d), pin(tmax)
    P = torch.full((n,), -7, dtype=torch.int32).pin_memory()
    Tt = torch.full((n,), -7.0, dtype=torch.float32).pin_memory()
    Oc = torch.full((n,), 9, dtype=torch.uint8).pin_memory()
    fp = lambda x: C.cast(x.data_ptr(), C.POINTER(C.c_float))
    h = gpu_ctx.handle
    _lib.check(lib.ptmi_closest_hit(h, fp(O), fp(D), fp(T), n, C.cast(P.data_ptr(), C.POINTER(C.c_int32)), fp(Tt)), h)
    _lib.check(lib.ptmi_any_hit(h, fp(O), fp(D), fp(T), n, C.cast(Oc.data_ptr(), C.POINTER(C.c_uint8))), h)
    assert np.array_equal(P.numpy(), prim_ref) and bits_equal(Tt.numpy(), t_ref)
    assert np.array_equal(Oc.numpy().astype(bool), np.asarray(occ_ref).astype(bool))
    idx = np.arange(0, n, 257)
    orc = oracle_lib.OracleBVH(v, f, linear=box.linear())
    op, ot, _ = orc.closest_hit(np.ascontiguousarray(o[idx]), np.ascontiguousarray(d[idx]), np.ascontiguousarray(tmax[idx]))
    assert np.array_equal(P.numpy()[idx], op) and bits_equal(Tt.numpy()[idx], ot)


def _caterpillar(depth):
    """A tree in the reference's LinearBVH form whose every inner node has an inner child and a one-triangle leaf child:
    max depth `depth` (root = 1), and a ray along +z through all of it defers one leaf per level, so the traversal stack
    holds depth - 1 entries -- the most a tree of that depth can ask for."""
    from pathtrace_amd.api import LinearArrays
    D = depth
    ntri = D                                              # leaf_k for k = 0 .. D-2, and the bottom leaf
    # triangle k (k = 0 .. D-2) sits at z = k + 1, the bottom triangle (number D-1) at z = 0; all cross the z axis
    v = np.zeros((3 * ntri, 3), np.float32)
    for k in range(ntri):
        z = float(k + 1) if k < D - 1 else 0.0
        v[3 * k:3 * k + 3] = [[-2.0, -2.0, z], [2.0, -2.0, z], [0.0, 3.0, z]]
    f = np.arange(3 * ntri, dtype=np.uint32).reshape(-1, 3)
    nn = 2 * D - 1
    bounds = np.zeros((nn, 6), np.float32)
    is_leaf = np.zeros(nn, np.uint8); off = np.zeros(nn, np.uint32); cnt = np.zeros(nn, np.uint32); second = np.zeros(nn, np.uint32)
    tri_ids = np.zeros(ntri, np.uint32)

    def tri_box(k):
        p = v[3 * k:3 * k + 3]
        return np.concatenate([p.min(0), p.max(0)])
    # pre-order: inner 0 .. D-2, the bottom leaf, then leaf D-2 .. leaf 0
    leaf_index = {k: 2 * D - 2 - k for k in range(D - 1)}
    bottom = D - 1
    order = [D - 1] + list(range(D - 2, -1, -1))          # triangles in leaf (pre-order) order
    for pos, k in enumerate(order):
        tri_ids[pos] = k
        node = bottom if k == D - 1 else leaf_index[k]
        is_leaf[node] = 1; off[node] = pos; cnt[node] = 1; bounds[node] = tri_box(k)
    for k in range(D - 2, -1, -1):                         # inner k = its first child (k + 1: inner or bottom leaf) + leaf k
        a, b = bounds[k + 1], bounds[leaf_index[k]]
        bounds[k] = np.concatenate([np.minimum(a[:3], b[:3]), np.maximum(a[3:], b[3:])])
        second[k] = leaf_index[k]
    return v, f, LinearArrays(bounds, is_leaf, off, cnt, second, tri_ids)


@pytest.mark.parametrize("depth", [2, 17, 18, 21, 25, 29, 33, 49])
def test_deepest_stack_a_tree_can_ask_for(gpu_ctx, depth, monkeypatch):
    """The kernels' LDS stacks are sized by the tree: depth - 1 entries (ptmi_ctx::stack_entries), in classes of 16 / 20 /
    24 / 28 / 32 / 48 / 64.  Depths 17, 21, 25, 29, 33 and 49 fill their class to the last entry with rays that defer a child at
    every level; closest hit and any hit equal the oracle on them and on random rays, for the vote-scheduled and the
    one-shot kernels."""
    from pathtrace_amd import UploadLinearBVH, Context
    v, f, arrays = _caterpillar(depth)
    orc = oracle_lib.OracleBVH(v, f, linear=arrays)
    n = 4096
    rng = np.random.default_rng(depth)
    o = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1.5, 1.5, n), np.full(n, -5.0)], 1).astype(np.float32)
    d = np.tile(np.array([0.0, 0.0, 1.0], np.float32), (n, 1))
    d[n // 2:] = [0.0, 0.0, -1.0]; o[n // 2:, 2] = depth + 5.0                 # and from the other end
    d[::5] += rng.uniform(-0.05, 0.05, (len(d[::5]), 3)).astype(np.float32)    # slightly off axis
    o2, d2 = random_rays(v, n, depth + 1)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    bvh = UploadLinearBVH(v, f, arrays, ctx=gpu_ctx)
    assert bvh.GetStats().max_depth == depth
    hit = _check(bvh, orc, o, d, 1000.0, f"caterpillar{depth}/vote")
    assert hit[:n].mean() > 0.5
    monkeypatch.setenv("PTMI_EXTEND_VARIANT", "0")
    ctx = Context(0)
    try:
        _check(UploadLinearBVH(v, f, arrays, ctx=ctx), orc, o, d, 1000.0, f"caterpillar{depth}/one-shot")
    finally:
        ctx.close()


def test_small_host_batches_after_a_large_one(gpu_ctx):
    """ptmi_closest_hit stages prim ids and t in two 256-byte-aligned halves of its output area: batches whose 4 n is
    not a multiple of 256 (n = 1, 32, 65, 4097) must get the area the layout uses, whatever an earlier call sized
    the staging buffer for."""
    v, f = scenes.make_soup(3000, 21)
    box = BuildBVH(v, f)
    bvh = ConstructLinearBVH(box, ctx=gpu_ctx)
    orc = oracle_lib.OracleBVH(v, f, linear=box.linear())
    o, d = random_rays(v, 70000, 9)
    _check(bvh, orc, o, d, 1000.0, "host batch 70000")
    for n in (1, 32, 65, 4097):
        _check(bvh, orc, o[:n], d[:n], 1000.0, f"host batch {n}")


def test_more_launches_in_flight_than_cursor_lines(gpu_ctx):
    """The persistent batch kernels take their ray cursor from a ring of 64 lines per context.  150 asynchronous launches
    on three streams: the ring wraps twice, a line is only reused once its previous launch has finished (an event per
    line), and every launch's answers equal the first launch's."""
    import torch
    v, f = scenes.make_soup(20000, 12)
    box = BuildBVH(v, f)
    bvh = ConstructLinearBVH(box, ctx=gpu_ctx)
    o, d = random_rays(v, 40000, 3)
    dev = torch.device("cuda", 0)
    O, D = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    T = torch.full((o.shape[0],), 1000.0, dtype=torch.float32, device=dev)
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    outs = [(torch.full((o.shape[0],), -7, dtype=torch.int32, device=dev), torch.full((o.shape[0],), -1.0, dtype=torch.float32, device=dev))
            for _ in range(150)]
    torch.cuda.synchronize()                    # the fills ran on torch's stream, the launches go to three others
    for k, (P, t) in enumerate(outs):
        bvh.CheckIntersectionDevice(O, D, T, P, t, streams[k % 3].cuda_stream)
    torch.cuda.synchronize()
    for P, t in outs[1:]:
        assert torch.equal(P, outs[0][0]) and torch.equal(t.view(torch.int32), outs[0][1].view(torch.int32))
    assert int((outs[0][0] >= 0).sum()) > 1000
