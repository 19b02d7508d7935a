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


# ----------------------------------------------------------------------------- integrator (orc_trace.c)
class OrcParams(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("bounces", C.c_int), ("scatter_rays", C.c_int),
                ("max_steps", C.c_int), ("ambient", C.c_float), ("step_size", C.c_float), ("seed", C.c_uint32)]


def _load_trace():
    lib = load()
    if getattr(lib, "_trace_ready", False):
        return lib
    U16 = C.POINTER(C.c_uint16)
    lib.orc_scene_create.restype = C.c_void_p
    lib.orc_scene_create.argtypes = [C.c_void_p, _F, _U32, C.c_uint64, _F, _U32, _F, C.c_uint64, _F, C.c_uint64, _F,
                                     U16, C.c_int, C.c_int, _F]
    lib.orc_scene_free.restype = None
    lib.orc_scene_free.argtypes = [C.c_void_p]
    lib.orc_scene_set_textures.restype = None
    lib.orc_scene_set_textures.argtypes = [C.c_void_p, _F, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _U32]
    lib.orc_probe_tile.restype = C.c_float
    lib.orc_probe_tile.argtypes = [C.c_float]
    lib.orc_probe_texel.restype = C.c_uint32
    lib.orc_probe_texel.argtypes = [_U32, C.c_int, C.c_int, C.c_float, C.c_float]
    lib.orc_probe_bump.restype = None
    lib.orc_probe_bump.argtypes = [_U32, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, _F]
    lib.orc_probe_tbn.restype = None
    lib.orc_probe_tbn.argtypes = [_F, _F, _F, _F, _F]
    lib.orc_render.restype = None
    lib.orc_render.argtypes = [C.c_void_p, _F, C.POINTER(OrcParams), C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32,
                               C.c_uint32, _F, _U32, _U64P, C.c_int]
    lib.orc_render_samples.restype = None
    lib.orc_render_samples.argtypes = [C.c_void_p, _F, C.POINTER(OrcParams), C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_uint32, C.c_uint32, _F, C.c_int]
    lib.orc_pick_simulate.restype = C.c_int64
    lib.orc_pick_simulate.argtypes = [_F, C.c_int64, C.c_int64, C.c_int, C.c_int, _F, _U32, _F]
    lib.orc_render_tile_adaptive.restype = C.c_int64
    lib.orc_render_tile_adaptive.argtypes = [C.c_void_p, _F, C.POINTER(OrcParams), C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, _F, _U32, _F]
    lib.orc_render_tile_workers.restype = C.c_double
    lib.orc_render_tile_workers.argtypes = [C.c_void_p, _F, C.POINTER(OrcParams), C.c_int, C.c_int, C.c_uint32, _F, _U32, _U64P]
    lib.orc_to_rgba.restype = None
    lib.orc_to_rgba.argtypes = [_F, _U32, C.c_int, C.c_int, _U8]
    lib._trace_ready = True
    return lib


class OracleScene:
    """Scene + BVH for the restated integrator (TraceRay & co)."""

    def __init__(self, sc, bvh: OracleBVH | None = None):
        self.lib = _load_trace()
        self.sc = sc
        self.bvh = bvh or OracleBVH(sc.verts, sc.tris)
        self.verts = self.bvh.verts
        self.tris = self.bvh.tris
        self.normals = np.ascontiguousarray(sc.normals, np.float32)
        self.tri_material = np.ascontiguousarray(sc.tri_material, np.uint32)
        self.materials = np.ascontiguousarray(sc.materials, np.float32)
        self.lights = np.ascontiguousarray(sc.lights, np.float32).reshape(-1, 12)
        self.sky = np.ascontiguousarray(sc.sky, np.float32)
        self.sky_img = None if getattr(sc, "sky_image", None) is None else np.ascontiguousarray(sc.sky_image, np.uint16)
        self.bh = None if sc.blackhole is None else np.ascontiguousarray(sc.blackhole, np.float32)
        img_p, iw, ih = None, 0, 0
        if self.sky_img is not None:
            img_p, ih, iw = _p(self.sky_img, C.c_uint16), self.sky_img.shape[0], self.sky_img.shape[1]
        self.h = self.lib.orc_scene_create(self.bvh.h, _p(self.verts, C.c_float), _p(self.tris, C.c_uint32),
                                           self.tris.shape[0], _p(self.normals, C.c_float),
                                           _p(self.tri_material, C.c_uint32), _p(self.materials, C.c_float),
                                           self.materials.shape[0],
                                           _p(self.lights, C.c_float) if self.lights.size else None,
                                           self.lights.shape[0], _p(self.sky, C.c_float), img_p, iw, ih,
                                           _p(self.bh, C.c_float) if self.bh is not None else None)
        tex = getattr(sc, "textures", None)
        if tex and getattr(sc, "uvs", None) is not None:
            self.uvs = np.ascontiguousarray(sc.uvs, np.float32).reshape(-1, 6)
            self.tex_w = np.array([t.shape[1] for t in tex], np.int32)
            self.tex_h = np.array([t.shape[0] for t in tex], np.int32)
            self.tex_px = np.ascontiguousarray(np.concatenate([np.asarray(t, np.uint32).reshape(-1) for t in tex]))
            self.lib.orc_scene_set_textures(self.h, _p(self.uvs, C.c_float), len(tex), _p(self.tex_w, C.c_int32),
                                            _p(self.tex_h, C.c_int32), _p(self.tex_px, C.c_uint32))

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self.lib.orc_scene_free(h)

    @staticmethod
    def params(width, height, bounces=2, scatter_rays=1, max_steps=1, ambient=0.0, step_size=1000.0, seed=1):
        return OrcParams(width, height, bounces, scatter_rays, max_steps, ambient, step_size, seed)

    def render(self, cam, params, rect=None, spp_begin=0, spp_count=1, nthreads=8, accum=None, counts=None):
        x0, y0, w, h = rect or (0, 0, params.width, params.height)
        cam = np.ascontiguousarray(cam, np.float32)
        accum = np.zeros((h, w, 3), np.float32) if accum is None else accum
        counts = np.zeros((h, w), np.uint32) if counts is None else counts
        cnt = np.zeros(4, np.uint64)
        self.lib.orc_render(self.h, _p(cam, C.c_float), C.byref(params), x0, y0, w, h, spp_begin, spp_count,
                            _p(accum, C.c_float), _p(counts, C.c_uint32), _p(cnt, C.c_uint64), nthreads)
        return accum, counts, dict(closest=int(cnt[0]), any=int(cnt[1]), diffuse_events=int(cnt[2]),
                                   samples=int(cnt[3]))

    def render_tile_workers(self, cam, params, splits=(4, 4), spp=16):
        """The goroutine tile renderer (main.go:435, :454-491, :832-899): one worker thread per tile.  Returns
        (accum [H,W,3], counts [H,W], counters, wall seconds of the render loop)."""
        cam = np.ascontiguousarray(cam, np.float32)
        accum = np.zeros((params.height, params.width, 3), np.float32)
        counts = np.zeros((params.height, params.width), np.uint32)
        cnt = np.zeros(4, np.uint64)
        secs = self.lib.orc_render_tile_workers(self.h, _p(cam, C.c_float), C.byref(params), int(splits[0]), int(splits[1]),
                                                int(spp), _p(accum, C.c_float), _p(counts, C.c_uint32), _p(cnt, C.c_uint64))
        if secs < 0:
            raise RuntimeError("orc_render_tile_workers: could not start the workers")
        return accum, counts, dict(closest=int(cnt[0]), any=int(cnt[1]), diffuse_events=int(cnt[2]), samples=int(cnt[3])), secs

    def render_tile_adaptive(self, cam, params, rect, spp, max_samples):
        """The reference's adaptive worker loop on one tile (main.go:837-897 + tile.go:34-66): (sums [h,w,3], counts
        [h,w], stats [h,w,9] = min / max luminance, Mean, M2, Variance, picks)."""
        x0, y0, w, h = rect
        cam = np.ascontiguousarray(cam, np.float32)
        sums = np.zeros((h, w, 3), np.float32); counts = np.zeros((h, w), np.uint32); lum = np.zeros((h, w, 9), np.float32)
        picks = self.lib.orc_render_tile_adaptive(self.h, _p(cam, C.c_float), C.byref(params), x0, y0, w, h, int(spp),
                                                  int(max_samples), _p(sums, C.c_float), _p(counts, C.c_uint32), _p(lum, C.c_float))
        if picks < 0:
            raise RuntimeError("orc_render_tile_adaptive: bad arguments")
        return sums, counts, lum, int(picks)

    def render_samples(self, cam, params, rect=None, spp_begin=0, spp_count=1, nthreads=8):
        x0, y0, w, h = rect or (0, 0, params.width, params.height)
        cam = np.ascontiguousarray(cam, np.float32)
        out = np.zeros((h, w, spp_count, 3), np.float32)
        self.lib.orc_render_samples(self.h, _p(cam, C.c_float), C.byref(params), x0, y0, w, h, spp_begin, spp_count,
                                    _p(out, C.c_float), nthreads)
        return out


def pick_simulate(samples, spp, max_samples):
    """The reference's adaptive worker loop on one tile whose samples are given ([P, S, 3] float32, S > max_samples):
    (sums [P,3], counts [P], stats [P,9], picks)."""
    lib = _load_trace()
    smp = np.ascontiguousarray(samples, np.float32)
    P, S = smp.shape[0], smp.shape[1]
    sums = np.zeros((P, 3), np.float32); counts = np.zeros(P, np.uint32); lum = np.zeros((P, 9), np.float32)
    picks = lib.orc_pick_simulate(_p(smp, C.c_float), P, S, int(spp), int(max_samples), _p(sums, C.c_float),
                                  _p(counts, C.c_uint32), _p(lum, C.c_float))
    if picks < 0:
        raise RuntimeError("orc_pick_simulate: need S > max_samples and spp >= 1")
    return sums, counts, lum, int(picks)


def to_rgba(accum, counts):
    lib = _load_trace()
    h, w = counts.shape
    rgba = np.zeros((h, w, 4), np.uint8)
    rgba[..., 3] = 255
    lib.orc_to_rgba(_p(np.ascontiguousarray(accum, np.float32), C.c_float),
                    _p(np.ascontiguousarray(counts, np.uint32), C.c_uint32), w, h, _p(rgba, C.c_uint8))
    return rgba
