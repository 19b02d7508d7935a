"""Host-side mirror of the reference's interface for the hot path, over the C ABI.

Names and argument meaning follow the Go code so that parity tests read like the
reference's call sites (main.go:804-810, :875):

    box  = BuildBVH(verts, tris, -1000, 1000, -1000, 1000, -1000, 1000, 4, 42)   # bvh.go:248
    bvh  = ConstructLinearBVH(box, device=0)                                    # linear_bvh.go:108
    hit, t, tri = bvh.CheckIntersection((origins, dirs), step)                   # linear_bvh.go:116
    occluded    = bvh.QuickCheckIntersection((origins, dirs), step)              # linear_bvh.go:202

The only difference is that queries are batched (arrays of rays instead of one Ray) and the
returned triangle is its number (tri.Index/3), -1 for the reference's nil.  Everything
computes on the GPU through libptmi.so; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import PtBvhStats, PtCounters, PtmiError, check


def _f32(a, shape_last=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape_last is not None and (a.ndim == 0 or a.shape[-1] != shape_last):
        raise ValueError(f"expected [...,{shape_last}] float32 array, got {a.shape}")
    return a


def _ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


@dataclass
class LinearArrays:
    """LinearBVH in the reference's own form (linear_bvh.go:11-21, 72-75)."""
    node_bounds: np.ndarray   # [n,6] min xyz, max xyz
    is_leaf: np.ndarray       # [n] u8
    tri_offset: np.ndarray    # [n] u32
    tri_count: np.ndarray     # [n] u32
    second_child: np.ndarray  # [n] u32
    tri_ids: np.ndarray       # [m] u32 triangle numbers in leaf order

    def digest(self) -> str:
        import hashlib
        h = hashlib.sha256()
        for a in (self.node_bounds, self.is_leaf, self.tri_offset, self.tri_count, self.second_child, self.tri_ids):
            h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()


class Box:
    """Result of BuildBVH (the reference returns the root *Box of a pointer tree, bvh.go:72-80;
    here it is an opaque host handle holding the same tree already flattened)."""

    def __init__(self, handle, verts, tris):
        self._h = handle
        self.verts = verts
        self.tris = tris

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        try:
            if h:
                _lib.load().ptmi_host_bvh_free(h)
        except Exception:       # interpreter shutdown: module globals may already be gone
            pass

    def GetStats(self) -> PtBvhStats:
        """bvh.go:117-146 (called as GetStats(1) at main.go:809)."""
        st = PtBvhStats()
        check(_lib.load().ptmi_host_bvh_stats(self._h, C.byref(st)))
        return st

    def linear(self) -> LinearArrays:
        lib = _lib.load()
        nn, nt = C.c_uint64(), C.c_uint64()
        check(lib.ptmi_host_bvh_sizes(self._h, C.byref(nn), C.byref(nt)))
        arr = _alloc_linear(nn.value, nt.value)
        check(lib.ptmi_host_bvh_get(self._h, _ptr(arr.node_bounds, C.c_float), _ptr(arr.is_leaf, C.c_uint8),
                                    _ptr(arr.tri_offset, C.c_uint32), _ptr(arr.tri_count, C.c_uint32),
                                    _ptr(arr.second_child, C.c_uint32), _ptr(arr.tri_ids, C.c_uint32)))
        return arr


def _alloc_linear(nn, nt):
    return LinearArrays(np.zeros((nn, 6), np.float32), np.zeros(nn, np.uint8), np.zeros(nn, np.uint32),
                        np.zeros(nn, np.uint32), np.zeros(nn, np.uint32), np.zeros(max(nt, 1), np.uint32)[:nt])


def BuildBVH(verts, tris, x1=-1000.0, x2=1000.0, y1=-1000.0, y2=1000.0, z1=-1000.0, z2=1000.0, threshold=4,
             depth=42) -> Box:
    """bvh.go:248.  verts [nv,3] f32, tris [nt,3] vertex indices.  Host-side, no GPU needed."""
    v = _f32(verts, 3)
    t = np.ascontiguousarray(tris, dtype=np.uint32)
    if t.ndim != 2 or t.shape[1] != 3:
        raise ValueError("tris must be [nt,3]")
    bounds = np.array([x1, x2, y1, y2, z1, z2], dtype=np.float32)
    h = C.c_void_p()
    check(_lib.load().ptmi_host_bvh_build(_ptr(v, C.c_float), v.shape[0], _ptr(t, C.c_uint32), t.shape[0],
                                          _ptr(bounds, C.c_float), int(threshold), int(depth), C.byref(h)))
    return Box(h, v, t)


class Context:
    """One per GPU (ptmi_create)."""

    def __init__(self, device=0):
        self._ctx = C.c_void_p()
        check(_lib.load().ptmi_create(int(device), C.byref(self._ctx)))
        self.device = device

    def close(self):
        c, self._ctx = getattr(self, "_ctx", None), None
        if c:
            _lib.load().ptmi_destroy(c)

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass

    @property
    def handle(self):
        return self._ctx

    def selftest_math(self):
        """ptmi_selftest_math: floats on which the march's short square root / reciprocal differ from sqrtf / the
        division, over every float of the ranges they are used on (0 on a correct installation)."""
        n = C.c_uint64(0)
        check(_lib.load().ptmi_selftest_math(self._ctx, C.byref(n)), self._ctx)
        return int(n.value)


class LinearBVH:
    """linear_bvh.go:72-75, resident on one GPU."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._lib = _lib.load()

    # ---- queries -------------------------------------------------------------------------
    def CheckIntersection(self, ray, stepSize):
        """linear_bvh.go:116.  ray = (origins [n,3], directions [n,3]); stepSize scalar or [n].
        Returns (hit bool[n], t f32[n], tri i32[n])."""
        o, d = _f32(ray[0], 3), _f32(ray[1], 3)
        n = o.shape[0]
        step = np.ascontiguousarray(np.broadcast_to(np.asarray(stepSize, dtype=np.float32), (n,)))
        prim = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        check(self._lib.ptmi_closest_hit(self.ctx.handle, _ptr(o, C.c_float), _ptr(d, C.c_float),
                                         _ptr(step, C.c_float), n, _ptr(prim, C.c_int32), _ptr(t, C.c_float)),
              self.ctx.handle)
        return prim >= 0, t, prim

    def QuickCheckIntersection(self, ray, stepSize):
        """linear_bvh.go:202.  Returns occluded bool[n]."""
        o, d = _f32(ray[0], 3), _f32(ray[1], 3)
        n = o.shape[0]
        step = np.ascontiguousarray(np.broadcast_to(np.asarray(stepSize, dtype=np.float32), (n,)))
        occ = np.empty(n, np.uint8)
        check(self._lib.ptmi_any_hit(self.ctx.handle, _ptr(o, C.c_float), _ptr(d, C.c_float), _ptr(step, C.c_float),
                                     n, _ptr(occ, C.c_uint8)), self.ctx.handle)
        return occ.astype(bool)

    def CheckIntersectionDevice(self, o, d, step, prim, t, stream=None):
        """Device-resident form: torch CUDA tensors (float32 [n,3], [n,3], [n]; int32 [n], float32 [n])."""
        n = o.shape[0]
        s = C.c_void_p(stream) if stream else None
        check(self._lib.ptmi_closest_hit_dev(self.ctx.handle, o.data_ptr(), d.data_ptr(), step.data_ptr(), n,
                                             prim.data_ptr(), t.data_ptr(), s), self.ctx.handle)

    def CheckIntersectionDeviceNoAlloc(self, o, d, step, prim, t, stream=None, n=None):
        """ptmi_closest_hit_dev_noalloc: the same query without any device allocation (static scratch in the library);
        calls on one GPU must not overlap."""
        n = o.shape[0] if n is None else n
        s = C.c_void_p(stream) if stream else None
        check(self._lib.ptmi_closest_hit_dev_noalloc(self.ctx.handle, o.data_ptr(), d.data_ptr(), step.data_ptr(), n,
                                                     prim.data_ptr(), t.data_ptr(), s), self.ctx.handle)

    def QuickCheckIntersectionDevice(self, o, d, step, occ, stream=None):
        n = o.shape[0]
        s = C.c_void_p(stream) if stream else None
        check(self._lib.ptmi_any_hit_dev(self.ctx.handle, o.data_ptr(), d.data_ptr(), step.data_ptr(), n,
                                         occ.data_ptr(), s), self.ctx.handle)

    # ---- introspection -------------------------------------------------------------------
    def GetStats(self) -> PtBvhStats:
        st = PtBvhStats()
        check(self._lib.ptmi_bvh_stats(self.ctx.handle, C.byref(st)), self.ctx.handle)
        return st

    def linear(self) -> LinearArrays:
        nn, nt = C.c_uint64(), C.c_uint64()
        check(self._lib.ptmi_bvh_sizes(self.ctx.handle, C.byref(nn), C.byref(nt)), self.ctx.handle)
        arr = _alloc_linear(nn.value, nt.value)
        check(self._lib.ptmi_get_linear_bvh(self.ctx.handle, _ptr(arr.node_bounds, C.c_float),
                                            _ptr(arr.is_leaf, C.c_uint8), _ptr(arr.tri_offset, C.c_uint32),
                                            _ptr(arr.tri_count, C.c_uint32), _ptr(arr.second_child, C.c_uint32),
                                            _ptr(arr.tri_ids, C.c_uint32)), self.ctx.handle)
        return arr

    def set_counting(self, on: bool):
        check(self._lib.ptmi_set_counting(self.ctx.handle, 1 if on else 0), self.ctx.handle)

    def reset_counters(self):
        check(self._lib.ptmi_reset_counters(self.ctx.handle), self.ctx.handle)

    def counters(self) -> PtCounters:
        c = PtCounters()
        check(self._lib.ptmi_counters(self.ctx.handle, C.byref(c)), self.ctx.handle)
        return c


def ConstructLinearBVH(box: Box, device=0, ctx: Context | None = None) -> LinearBVH:
    """linear_bvh.go:108: flatten + (here) pack into 64-byte nodes and upload to the GPU."""
    ctx = ctx or Context(device)
    check(_lib.load().ptmi_upload_host_bvh(ctx.handle, box._h), ctx.handle)
    return LinearBVH(ctx)


def UploadLinearBVH(verts, tris, arrays: LinearArrays, device=0, ctx: Context | None = None) -> LinearBVH:
    """Upload a tree built elsewhere (e.g. by the Go host) in the reference's LinearBVH form."""
    ctx = ctx or Context(device)
    v = _f32(verts, 3)
    t = np.ascontiguousarray(tris, dtype=np.uint32)
    a = arrays
    nb = np.ascontiguousarray(a.node_bounds, np.float32)
    il = np.ascontiguousarray(a.is_leaf, np.uint8)
    to = np.ascontiguousarray(a.tri_offset, np.uint32)
    tc = np.ascontiguousarray(a.tri_count, np.uint32)
    sc = np.ascontiguousarray(a.second_child, np.uint32)
    ti = np.ascontiguousarray(a.tri_ids, np.uint32)
    check(_lib.load().ptmi_upload_bvh(ctx.handle, _ptr(v, C.c_float), v.shape[0], _ptr(t, C.c_uint32), t.shape[0],
                                      nb.shape[0], _ptr(nb, C.c_float), _ptr(il, C.c_uint8), _ptr(to, C.c_uint32),
                                      _ptr(tc, C.c_uint32), _ptr(sc, C.c_uint32), ti.shape[0], _ptr(ti, C.c_uint32)),
          ctx.handle)
    return LinearBVH(ctx)


__all__ = ["BuildBVH", "ConstructLinearBVH", "UploadLinearBVH", "LinearBVH", "Box", "Context", "LinearArrays",
           "PtmiError"]
