"""ctypes binding of oracle/liborc.so -- the CPU restatement of the Go reference.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
ORC_PATH = os.path.join(ORC_DIR, "liborc.so")

_F = C.POINTER(C.c_float)
_U32 = C.POINTER(C.c_uint32)
_U8 = C.POINTER(C.c_uint8)
_I32 = C.POINTER(C.c_int32)
_U64P = C.POINTER(C.c_uint64)
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORC_PATH):
        subprocess.check_call(["make", "-C", ORC_DIR], stdout=subprocess.DEVNULL)
    lib = C.CDLL(ORC_PATH)
    lib.orc_build_bvh.restype = C.c_void_p
    lib.orc_build_bvh.argtypes = [_F, C.c_uint64, _U32, C.c_uint64, _F, C.c_int, C.c_int]
    lib.orc_bvh_import.restype = C.c_void_p
    lib.orc_bvh_import.argtypes = [_F, C.c_uint64, _U32, C.c_uint64, C.c_uint64, _F, _U8, _U32, _U32, _U32,
                                   C.c_uint64, _U32]
    lib.orc_free_bvh.restype = None
    lib.orc_free_bvh.argtypes = [C.c_void_p]
    lib.orc_bvh_num_nodes.restype = C.c_uint64
    lib.orc_bvh_num_nodes.argtypes = [C.c_void_p]
    lib.orc_bvh_num_tri_ids.restype = C.c_uint64
    lib.orc_bvh_num_tri_ids.argtypes = [C.c_void_p]
    lib.orc_bvh_export.restype = None
    lib.orc_bvh_export.argtypes = [C.c_void_p, _F, _U8, _U32, _U32, _U32, _U32]
    lib.orc_closest_hit_batch.restype = None
    lib.orc_closest_hit_batch.argtypes = [C.c_void_p, _F, _F, _F, C.c_uint64, _I32, _F, _U64P, C.c_int]
    lib.orc_any_hit_batch.restype = None
    lib.orc_any_hit_batch.argtypes = [C.c_void_p, _F, _F, _F, C.c_uint64, _U8, _U64P, C.c_int]
    lib.orc_probe_segment_triangle.restype = C.c_int
    lib.orc_probe_segment_triangle.argtypes = [_F, _F, C.c_float, _F, _F]
    lib.orc_probe_shadow_triangle.restype = C.c_int
    lib.orc_probe_shadow_triangle.argtypes = [_F, _F, C.c_float, _F]
    lib.orc_probe_aabb.restype = C.c_float
    lib.orc_probe_aabb.argtypes = [_F, _F, _F, _F, C.c_float]
    _lib = lib
    return lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


class OracleBVH:
    """BuildBVH + ConstructLinearBVH of the restatement (or an imported LinearBVH)."""

    def __init__(self, verts, tris, bounds=(-1000, 1000, -1000, 1000, -1000, 1000), threshold=4, depth=42,
                 linear=None):
        self.lib = load()
        self.verts = np.ascontiguousarray(verts, np.float32)
        self.tris = np.ascontiguousarray(tris, np.uint32)
        if linear is None:
            b = np.array(bounds, np.float32)
            self.h = self.lib.orc_build_bvh(_p(self.verts, C.c_float), self.verts.shape[0], _p(self.tris, C.c_uint32),
                                            self.tris.shape[0], _p(b, C.c_float), threshold, depth)
        else:
            a = linear
            nb = np.ascontiguousarray(a.node_bounds, np.float32)
            il = np.ascontiguousarray(a.is_leaf, np.uint8)
            to = np.ascontiguousarray(a.tri_offset, np.uint32)
            tc = np.ascontiguousarray(a.tri_count, np.uint32)
            sc = np.ascontiguousarray(a.second_child, np.uint32)
            ti = np.ascontiguousarray(a.tri_ids, np.uint32)
            self.h = self.lib.orc_bvh_import(_p(self.verts, C.c_float), self.verts.shape[0],
                                             _p(self.tris, C.c_uint32), self.tris.shape[0], nb.shape[0],
                                             _p(nb, C.c_float), _p(il, C.c_uint8), _p(to, C.c_uint32),
                                             _p(tc, C.c_uint32), _p(sc, C.c_uint32), ti.shape[0], _p(ti, C.c_uint32))

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self.lib.orc_free_bvh(h)

    def linear(self):
        from pathtrace_amd.api import LinearArrays
        nn = self.lib.orc_bvh_num_nodes(self.h)
        nt = self.lib.orc_bvh_num_tri_ids(self.h)
        a = LinearArrays(np.zeros((nn, 6), np.float32), np.zeros(nn, np.uint8), np.zeros(nn, np.uint32),
                         np.zeros(nn, np.uint32), np.zeros(nn, np.uint32), np.zeros(max(nt, 1), np.uint32)[:nt])
        self.lib.orc_bvh_export(self.h, _p(a.node_bounds, C.c_float), _p(a.is_leaf, C.c_uint8),
                                _p(a.tri_offset, C.c_uint32), _p(a.tri_count, C.c_uint32),
                                _p(a.second_child, C.c_uint32), _p(a.tri_ids, C.c_uint32))
        return a

    def closest_hit(self, o, d, tmax, nthreads=8, counters=True):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        tm = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)))
        prim = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        cnt = np.zeros(4, np.uint64)
        self.lib.orc_closest_hit_batch(self.h, _p(o, C.c_float), _p(d, C.c_float), _p(tm, C.c_float), n,
                                       _p(prim, C.c_int32), _p(t, C.c_float),
                                       _p(cnt, C.c_uint64) if counters else None, nthreads)
        return prim, t, dict(inner_visits=int(cnt[0]), box_tests=int(cnt[1]), tri_tests=int(cnt[2]),
                             leaf_visits=int(cnt[3]))

    def any_hit(self, o, d, tmax, nthreads=8, counters=True):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        tm = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)))
        occ = np.empty(n, np.uint8)
        cnt = np.zeros(4, np.uint64)
        self.lib.orc_any_hit_batch(self.h, _p(o, C.c_float), _p(d, C.c_float), _p(tm, C.c_float), n,
                                   _p(occ, C.c_uint8), _p(cnt, C.c_uint64) if counters else None, nthreads)
        return occ.astype(bool), dict(inner_visits=int(cnt[0]), box_tests=int(cnt[1]), tri_tests=int(cnt[2]),
                                      leaf_visits=int(cnt[3]))


def probe_segment_triangle(o, d, step, abc):
    o = np.asarray(o, np.float32); d = np.asarray(d, np.float32); abc = np.ascontiguousarray(abc, np.float32)
    t = C.c_float()
    h = load().orc_probe_segment_triangle(_p(o, C.c_float), _p(d, C.c_float), np.float32(step), _p(abc, C.c_float),
                                          C.byref(t))
    return bool(h), np.float32(t.value)


def probe_shadow_triangle(o, d, tmax, abc):
    o = np.asarray(o, np.float32); d = np.asarray(d, np.float32); abc = np.ascontiguousarray(abc, np.float32)
    return bool(load().orc_probe_shadow_triangle(_p(o, C.c_float), _p(d, C.c_float), np.float32(tmax),
                                                 _p(abc, C.c_float)))


def probe_aabb(minb, maxb, o, d, step):
    minb = np.asarray(minb, np.float32); maxb = np.asarray(maxb, np.float32)
    o = np.asarray(o, np.float32); d = np.asarray(d, np.float32)
    return np.float32(load().orc_probe_aabb(_p(minb, C.c_float), _p(maxb, C.c_float), _p(o, C.c_float),
                                            _p(d, C.c_float), np.float32(step)))
