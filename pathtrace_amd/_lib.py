"""ctypes binding of libptmi.so (the C ABI declared in include/ptmi.h).

The product path has no CPU fallback: if the HIP library is missing this module raises, and
every query needs a GPU context.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTMI_LIB: another build of the same library (A/B and ablation builds of tools/build_variant.sh); never a fallback
LIB_PATH = os.environ.get("PTMI_LIB") or os.path.join(_HERE, "libptmi.so")

PTMI_OK = 0
ERR_NAMES = {-1: "PTMI_ERR_INVALID", -2: "PTMI_ERR_NO_DEVICE", -3: "PTMI_ERR_HIP", -4: "PTMI_ERR_NO_BVH",
             -5: "PTMI_ERR_DEPTH", -6: "PTMI_ERR_NO_SCENE", -7: "PTMI_ERR_OOM", -8: "PTMI_ERR_COMM"}
UNIQUE_ID_BYTES = 128


class PtmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class PtBvhStats(C.Structure):
    _fields_ = [("max_depth", C.c_int64), ("max_tris", C.c_int64), ("min_tris", C.c_int64),
                ("total_leafs", C.c_int64), ("total_nodes", C.c_int64), ("total_triangles", C.c_int64),
                ("linear_nodes", C.c_int64), ("packed_inner_nodes", C.c_int64), ("build_ms", C.c_double)]


class PtCounters(C.Structure):
    _fields_ = [("closest_queries", C.c_uint64), ("any_queries", C.c_uint64), ("inner_visits", C.c_uint64),
                ("tri_tests", C.c_uint64), ("leaf_visits", C.c_uint64), ("diffuse_events", C.c_uint64),
                ("samples", C.c_uint64), ("any_inner_visits", C.c_uint64), ("any_tri_tests", C.c_uint64),
                ("shade_ret_visits", C.c_uint64), ("shade_diff_visits", C.c_uint64), ("shade_general_visits", C.c_uint64),
                ("frames_read", C.c_uint64), ("extend_entries", C.c_uint64)]


class PtRenderProfile(C.Structure):
    """Per-kernel time of the render calls since ptmi_set_profiling(1), in milliseconds."""
    _fields_ = [("prepare_ms", C.c_double), ("generate_ms", C.c_double), ("extend_ms", C.c_double), ("connect_ms", C.c_double),
                ("shade_ms", C.c_double), ("accumulate_ms", C.c_double), ("iterations", C.c_uint64), ("launches", C.c_uint64)]


class PtCamera(C.Structure):
    """camera.go:9-15"""
    _fields_ = [("position", C.c_float * 3), ("forward", C.c_float * 3), ("right", C.c_float * 3),
                ("up", C.c_float * 3), ("frustum_distance", C.c_float)]


class PtParams(C.Structure):
    """The literals of main.go:441-452 that TraceRay receives positionally (main.go:875)."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("bounces", C.c_int32), ("scatter_rays", C.c_int32),
                ("max_steps", C.c_int32), ("ambient", C.c_float), ("step_size", C.c_float), ("seed", C.c_uint32)]


_P = C.c_void_p
_F = C.POINTER(C.c_float)
_U32 = C.POINTER(C.c_uint32)
_U8 = C.POINTER(C.c_uint8)
_I32 = C.POINTER(C.c_int32)
_U64 = C.c_uint64

# name -> (restype, argtypes).  tests/test_abi.py checks that this table and include/ptmi.h agree.
SIGNATURES = {
    "ptmi_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "ptmi_destroy": (None, [_P]),
    "ptmi_last_error": (C.c_char_p, [_P]),
    "ptmi_version": (C.c_char_p, []),
    "ptmi_build_bvh": (C.c_int, [_P, _F, _U64, _U32, _U64, _F, C.c_int, C.c_int]),
    "ptmi_host_bvh_build": (C.c_int, [_F, _U64, _U32, _U64, _F, C.c_int, C.c_int, C.POINTER(_P)]),
    "ptmi_host_bvh_free": (None, [_P]),
    "ptmi_host_bvh_sizes": (C.c_int, [_P, C.POINTER(_U64), C.POINTER(_U64)]),
    "ptmi_host_bvh_get": (C.c_int, [_P, _F, _U8, _U32, _U32, _U32, _U32]),
    "ptmi_host_bvh_stats": (C.c_int, [_P, C.POINTER(PtBvhStats)]),
    "ptmi_upload_host_bvh": (C.c_int, [_P, _P]),
    "ptmi_upload_bvh": (C.c_int, [_P, _F, _U64, _U32, _U64, _U64, _F, _U8, _U32, _U32, _U32, _U64, _U32]),
    "ptmi_bvh_sizes": (C.c_int, [_P, C.POINTER(_U64), C.POINTER(_U64)]),
    "ptmi_get_linear_bvh": (C.c_int, [_P, _F, _U8, _U32, _U32, _U32, _U32]),
    "ptmi_bvh_stats": (C.c_int, [_P, C.POINTER(PtBvhStats)]),
    "ptmi_closest_hit": (C.c_int, [_P, _F, _F, _F, _U64, _I32, _F]),
    "ptmi_closest_hit_dev": (C.c_int, [_P, _P, _P, _P, _U64, _P, _P, _P]),
    "ptmi_closest_hit_dev_noalloc": (C.c_int, [_P, _P, _P, _P, _U64, _P, _P, _P]),
    "ptmi_any_hit": (C.c_int, [_P, _F, _F, _F, _U64, _U8]),
    "ptmi_any_hit_dev": (C.c_int, [_P, _P, _P, _P, _U64, _P, _P]),
    "ptmi_set_counting": (C.c_int, [_P, C.c_int]),
    "ptmi_reset_counters": (C.c_int, [_P]),
    "ptmi_counters": (C.c_int, [_P, C.POINTER(PtCounters)]),
    "ptmi_set_pool_limit": (C.c_int, [_P, C.c_uint64, C.c_uint64]),
    "ptmi_pool_info": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ptmi_selftest_math": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "ptmi_set_profiling": (C.c_int, [_P, C.c_int]),
    "ptmi_render_profile": (C.c_int, [_P, C.POINTER(PtRenderProfile)]),
    "ptmi_upload_scene": (C.c_int, [_P, _P, _U64, _U32, _F, _P, _U64, _P, C.POINTER(C.c_uint16), _P]),
    "ptmi_upload_textures": (C.c_int, [_P, _F, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _U32]),
    "ptmi_render": (C.c_int, [_P, C.POINTER(PtCamera), C.POINTER(PtParams), C.c_int32, C.c_int32, C.c_int32,
                              C.c_int32, C.c_uint32, C.c_uint32, _F, _U32]),
    "ptmi_render_dev": (C.c_int, [_P, C.POINTER(PtCamera), C.POINTER(PtParams), C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_uint32, C.c_uint32, _P, _P, _P]),
    "ptmi_render_rows_dev": (C.c_int, [_P, C.POINTER(PtCamera), C.POINTER(PtParams), C.c_int32, C.c_int32, C.c_int32,
                                      C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ptmi_render_tiles_dev": (C.c_int, [_P, C.POINTER(PtCamera), C.POINTER(PtParams), C.c_uint32, C.POINTER(C.c_int32),
                                       C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ptmi_render_tiles_stats_dev": (C.c_int, [_P, C.POINTER(PtCamera), C.POINTER(PtParams), C.c_uint32, C.POINTER(C.c_int32),
                                             C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ptmi_render_adaptive_dev": (C.c_int, [_P, C.POINTER(PtCamera), C.POINTER(PtParams), C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ptmi_render_frames_dev": (C.c_int, [_P, C.POINTER(PtCamera), C.c_uint32, C.POINTER(PtParams), C.c_uint32, C.c_uint32,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "ptmi_render_samples": (C.c_int, [_P, C.POINTER(PtCamera), C.POINTER(PtParams), C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_uint32, C.c_uint32, _F]),
    "ptmi_to_rgba": (C.c_int, [_P, _F, _U32, C.c_int32, C.c_int32, _U8]),
    "ptmi_comm_unique_id": (C.c_int, [_U8]),
    "ptmi_comm_init_rank": (C.c_int, [_P, C.c_int, C.c_int, _U8]),
    "ptmi_comm_init_all": (C.c_int, [C.POINTER(_P), C.c_int]),
    "ptmi_comm_rank": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ptmi_comm_destroy": (C.c_int, [_P]),
    "ptmi_reduce_frame": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int, _P]),
    "ptmi_render_sharded": (C.c_int, [_P, C.POINTER(PtCamera), C.POINTER(PtParams), C.c_int32, C.c_uint32, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ptmi_obj_load": (C.c_int, [C.c_char_p, C.c_float, _F, C.POINTER(C.c_void_p)]),
    "ptmi_obj_free": (None, [_P]),
    "ptmi_obj_last_error": (C.c_char_p, []),
    "ptmi_obj_sizes": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    "ptmi_obj_get": (C.c_int, [_P, _F, _U32, _F, _U32, _P, _F]),
    "ptmi_obj_string": (C.c_char_p, [_P, C.c_int, C.c_uint64]),
}

_lib = None


def load():
    """Load libptmi.so; raises (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C pathtrace_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64, libptmi.so asks for the ROCm installation's.
    # Whichever is loaded first serves both (same SONAME) -- but if libptmi comes first and torch later brings its own
    # copy, the second runtime finds no GPU ("No HIP GPUs are available").  The host side uses torch for device
    # buffers anyway, so it goes first when it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, ctx=None):
    if rc != PTMI_OK:
        msg = load().ptmi_last_error(ctx)
        raise PtmiError(rc, msg.decode() if msg else "")
