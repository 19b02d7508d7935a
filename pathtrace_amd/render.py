"""Host-side mirror of the reference's render seam (main.go:784-899, tile.go, renderer.go) over the C ABI.

    r = Renderer(scene_data, device=0)                      # DecomposeObjects + BuildBVH + ConstructLinearBVH
    fb = r.new_framebuffer(W, H)                            # the Pixel buffer: R,G,B sums + SampleCount
    r.render_tile(fb, params, camera, (x0, y0, w, h), spp)  # one `calculate` worker's tile (main.go:832-899)
    img = r.to_rgba(fb)                                     # Vec3.ToRGBA + the mirrored img.Set (main.go:886-896)

The integrator runs entirely in libptmi.so (generate / extend / shade / connect / accumulate kernels).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import PtCamera, PtParams, check
from .api import BuildBVH, ConstructLinearBVH, Context, LinearBVH, _ptr
from .scenes import Camera, SceneData


def make_params(width, height, bounces=2, scatter_rays=1, max_steps=1, ambient=0.0, step_size=1000.0, seed=1):
    """main.go:441-452 (the committed literals are bounces=2, scatterRays=1, ambient=0, and for mesh scenes
    maxSteps=1, stepSize=1000: main.go:34-38; the black-hole scene uses maxSteps=3000, stepSize=1)."""
    return PtParams(int(width), int(height), int(bounces), int(scatter_rays), int(max_steps), float(ambient),
                    float(step_size), int(seed) & 0xFFFFFFFF)


def make_camera(cam: Camera) -> PtCamera:
    c = PtCamera()
    c.position[:] = cam.position
    c.forward[:] = cam.forward
    c.right[:] = cam.right
    c.up[:] = cam.up
    c.frustum_distance = cam.frustum
    return c


@dataclass
class Framebuffer:
    """The reference's [][]Pixel reduced to what AddSample keeps for output: sums and counts."""
    accum: np.ndarray      # [H,W,3] f32  (Pixel.R,G,B)
    counts: np.ndarray     # [H,W] u32    (Pixel.SampleCount)

    @property
    def mean(self):
        return self.accum / np.maximum(self.counts, 1)[..., None].astype(np.float32)

    # Checkpoint / resume.  The reference has none (its only outputs are PNGs, main.go:770-778, :916-924); the sums and
    # counts ARE the whole state of a frame in progress, and because a sample's RNG stream is keyed by (pixel, sample
    # number), rendering samples [n, m) into a reloaded framebuffer gives the same sums as never having stopped.
    def save(self, path, tag=None):
        """Write the accumulation state (f32 sums, u32 counts) as an .npz file; `tag` = a string that names the frame
        (driver.checkpoint_tag).  The file is written beside `path` and renamed over it, so a kill during the write
        leaves the previous checkpoint whole."""
        import os
        tmp = f"{path}.tmp"
        with open(tmp, "wb") as f:
            np.savez(f, accum=self.accum, counts=self.counts, tag=np.frombuffer((tag or "").encode(), np.uint8))
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, path)

    @staticmethod
    def load(path) -> "Framebuffer":
        with np.load(path, allow_pickle=False) as z:
            accum, counts = np.ascontiguousarray(z["accum"], np.float32), np.ascontiguousarray(z["counts"], np.uint32)
        if accum.ndim != 3 or accum.shape[2] != 3 or counts.shape != accum.shape[:2]:
            raise ValueError(f"{path}: not a framebuffer checkpoint (accum {accum.shape}, counts {counts.shape})")
        return Framebuffer(accum, counts)

    @staticmethod
    def load_tag(path):
        """The tag a checkpoint was saved with ("" for one saved without)."""
        with np.load(path, allow_pickle=False) as z:
            return bytes(z["tag"]).decode() if "tag" in z.files else ""

    @property
    def samples_done(self) -> int:
        """Samples every pixel has (the next sample number to render when resuming a uniformly sampled frame)."""
        return int(self.counts.min()) if self.counts.size else 0


class Renderer:
    def __init__(self, sc: SceneData, device=0, ctx: Context | None = None, threshold=4, depth=42):
        self.sc = sc
        self.ctx = ctx or Context(device)
        self._lib = _lib.load()
        # main.go:804: BuildBVH(vertices, tris, -1000, 1000, ..., 4, 42); :810 ConstructLinearBVH
        self.box = BuildBVH(sc.verts, sc.tris, threshold=threshold, depth=depth)
        self.bvh: LinearBVH = ConstructLinearBVH(self.box, ctx=self.ctx)
        self.upload_scene(sc)

    def upload_scene(self, sc: SceneData):
        mats = np.ascontiguousarray(sc.materials, np.float32)
        tm = np.ascontiguousarray(sc.tri_material, np.uint32)
        nrm = np.ascontiguousarray(sc.normals, np.float32)
        lights = np.ascontiguousarray(sc.lights, np.float32).reshape(-1, 12)
        sky = np.ascontiguousarray(sc.sky, np.float32)
        img = None if sc.sky_image is None else np.ascontiguousarray(sc.sky_image, np.uint16)
        bh = None if sc.blackhole is None else np.ascontiguousarray(sc.blackhole, np.float32)
        assert mats.shape[1] == 16 and nrm.shape[0] == 3 * tm.shape[0] == 3 * sc.tris.shape[0]
        check(self._lib.ptmi_upload_scene(
            self.ctx.handle, mats.ctypes.data_as(C.c_void_p), mats.shape[0], _ptr(tm, C.c_uint32), _ptr(nrm, C.c_float),
            lights.ctypes.data_as(C.c_void_p) if lights.size else None, lights.shape[0],
            sky.ctypes.data_as(C.c_void_p), _ptr(img, C.c_uint16) if img is not None else None,
            bh.ctypes.data_as(C.c_void_p) if bh is not None else None), self.ctx.handle)
        self.upload_textures(getattr(sc, "uvs", None), getattr(sc, "textures", None))

    def upload_textures(self, uvs, textures):
        """vnmu.UVs + the CachedImage table (obj_loader.go:29-57): textures = list of [h,w] packed-RGBA u32 images."""
        if not textures or uvs is None:
            check(self._lib.ptmi_upload_textures(self.ctx.handle, None, 0, None, None, None), self.ctx.handle)
            return
        uv = np.ascontiguousarray(uvs, np.float32).reshape(-1, 6)
        assert uv.shape[0] == self.bvh.ntris if hasattr(self.bvh, "ntris") else True
        w = np.array([t.shape[1] for t in textures], np.int32)
        h = np.array([t.shape[0] for t in textures], np.int32)
        px = np.ascontiguousarray(np.concatenate([np.asarray(t, np.uint32).reshape(-1) for t in textures]))
        check(self._lib.ptmi_upload_textures(self.ctx.handle, _ptr(uv, C.c_float), len(textures), _ptr(w, C.c_int32),
                                             _ptr(h, C.c_int32), _ptr(px, C.c_uint32)), self.ctx.handle)

    def set_pool_limit(self, max_paths=0, max_bytes=0):
        """Ceiling of the integrator's path pool (ptmi_set_pool_limit; 0 keeps a value).  The image does not depend on it."""
        check(self._lib.ptmi_set_pool_limit(self.ctx.handle, int(max_paths), int(max_bytes)), self.ctx.handle)

    def pool_info(self):
        """(paths, bytes) of the pool in place."""
        n, b = C.c_uint64(0), C.c_uint64(0)
        check(self._lib.ptmi_pool_info(self.ctx.handle, C.byref(n), C.byref(b)), self.ctx.handle)
        return int(n.value), int(b.value)

    def set_profiling(self, enabled=True):
        """Per-kernel HIP-event timing of the render calls (the reference's pprof hook, main.go:797-822); resets the totals."""
        check(self._lib.ptmi_set_profiling(self.ctx.handle, 1 if enabled else 0), self.ctx.handle)

    def profile(self) -> dict:
        """Milliseconds per kernel class since set_profiling(True), plus the iteration and launch counts."""
        p = _lib.PtRenderProfile()
        check(self._lib.ptmi_render_profile(self.ctx.handle, C.byref(p)), self.ctx.handle)
        return {name: getattr(p, name) for name, _ in p._fields_}

    @staticmethod
    def new_framebuffer(width, height) -> Framebuffer:
        return Framebuffer(np.zeros((height, width, 3), np.float32), np.zeros((height, width), np.uint32))

    def render_tile(self, fb: Framebuffer, params: PtParams, camera: Camera | PtCamera, rect=None, spp=32,
                    spp_begin=0):
        """One tile of the reference's schedule: every pixel of `rect` gets samples
        [spp_begin, spp_begin+spp) (main.go:861-879), added to the framebuffer."""
        cam = camera if isinstance(camera, PtCamera) else make_camera(camera)
        x0, y0, w, h = rect or (0, 0, params.width, params.height)
        acc = np.ascontiguousarray(fb.accum[y0:y0 + h, x0:x0 + w])
        cnt = np.ascontiguousarray(fb.counts[y0:y0 + h, x0:x0 + w])
        check(self._lib.ptmi_render(self.ctx.handle, C.byref(cam), C.byref(params), x0, y0, w, h, spp_begin, spp,
                                    _ptr(acc, C.c_float), _ptr(cnt, C.c_uint32)), self.ctx.handle)
        fb.accum[y0:y0 + h, x0:x0 + w] = acc
        fb.counts[y0:y0 + h, x0:x0 + w] = cnt

    def render(self, params: PtParams, camera, spp=32, splits=(4, 4)) -> Framebuffer:
        """The reference's frame: splitsX x splitsY tiles (main.go:454-491), each rendered to spp samples."""
        fb = self.new_framebuffer(params.width, params.height)
        sx, sy = splits
        tw, th = params.width // sx, params.height // sy
        for i in range(sx * sy):
            self.render_tile(fb, params, camera, ((i % sx) * tw, (i // sx) * th, tw, th), spp)
        return fb

    def render_device(self, accum, counts, params: PtParams, camera, rect=None, spp=32, spp_begin=0, stream=None):
        """Device-resident accumulators (torch CUDA tensors: f32 [h,w,3], i32/u32 [h,w] of the rectangle)."""
        cam = camera if isinstance(camera, PtCamera) else make_camera(camera)
        x0, y0, w, h = rect or (0, 0, params.width, params.height)
        s = C.c_void_p(stream) if stream else None
        check(self._lib.ptmi_render_dev(self.ctx.handle, C.byref(cam), C.byref(params), x0, y0, w, h, spp_begin, spp,
                                        accum.data_ptr(), counts.data_ptr(), s), self.ctx.handle)

    def render_rows_device(self, accum, counts, params: PtParams, camera, shard=0, nshards=1, row_block=8, spp=32,
                           spp_begin=0, stream=None):
        """One GPU's row-interleaved share of the frame in one call, added into FULL-FRAME torch CUDA tensors
        (f32 [H,W,3], i32/u32 [H,W]): rows y with (y // row_block) % nshards == shard."""
        cam = camera if isinstance(camera, PtCamera) else make_camera(camera)
        assert tuple(accum.shape) == (params.height, params.width, 3) and accum.is_contiguous()
        assert tuple(counts.shape) == (params.height, params.width) and counts.is_contiguous()
        s = C.c_void_p(stream) if stream else None
        check(self._lib.ptmi_render_rows_dev(self.ctx.handle, C.byref(cam), C.byref(params), row_block, shard, nshards,
                                             spp_begin, spp, accum.data_ptr(), counts.data_ptr(), s), self.ctx.handle)

    def render_sharded_device(self, accum, counts, params: PtParams, camera, row_block=8, spp=32, spp_begin=0, root=0,
                              stream=None):
        """ptmi_render_sharded: this rank's rows of the frame + the RCCL reduce onto `root`, in one call (the context's
        communicator decides rank and world; without one it is a single-GPU frame).  FULL-FRAME torch CUDA tensors,
        cleared by the call."""
        cam = camera if isinstance(camera, PtCamera) else make_camera(camera)
        assert tuple(accum.shape) == (params.height, params.width, 3) and accum.is_contiguous()
        assert tuple(counts.shape) == (params.height, params.width) and counts.is_contiguous()
        s = C.c_void_p(stream) if stream else None
        check(self._lib.ptmi_render_sharded(self.ctx.handle, C.byref(cam), C.byref(params), row_block, spp_begin, spp,
                                            accum.data_ptr(), counts.data_ptr(), root, s), self.ctx.handle)

    PIXEL_STATS = 9      # PTMI_PIXEL_STATS: min / max luminance, Mean[3], M2[3], Variance

    def render_tiles_device(self, accum, counts, params: PtParams, camera, tiles, spp=32, spp_begin=0, stream=None, stats=None):
        """Several disjoint tiles [(x0, y0, w, h), ...] in one call, added into FULL-FRAME torch CUDA tensors.
        stats: optional [H,W,9] float32 CUDA tensor that keeps the rest of Pixel.AddSample's state per pixel (min / max
        sample luminance, Mean, M2, Variance: renderer.go:25-82; ptmi_render_tiles_stats_dev)."""
        cam = camera if isinstance(camera, PtCamera) else make_camera(camera)
        assert tuple(accum.shape) == (params.height, params.width, 3) and accum.is_contiguous()
        assert tuple(counts.shape) == (params.height, params.width) and counts.is_contiguous()
        t = np.ascontiguousarray(np.asarray(tiles, np.int32).reshape(-1, 4))
        s = C.c_void_p(stream) if stream else None
        if stats is not None:
            assert tuple(stats.shape) == (params.height, params.width, self.PIXEL_STATS) and stats.is_contiguous() and stats.element_size() == 4
            check(self._lib.ptmi_render_tiles_stats_dev(self.ctx.handle, C.byref(cam), C.byref(params), t.shape[0], _ptr(t, C.c_int32),
                                                        spp_begin, spp, accum.data_ptr(), counts.data_ptr(), stats.data_ptr(), s),
                  self.ctx.handle)
            return
        check(self._lib.ptmi_render_tiles_dev(self.ctx.handle, C.byref(cam), C.byref(params), t.shape[0], _ptr(t, C.c_int32),
                                              spp_begin, spp, accum.data_ptr(), counts.data_ptr(), s), self.ctx.handle)

    def render_adaptive_device(self, accum, counts, stats, params: PtParams, camera, samples_per_pixel=32, max_samples=1024, stream=None):
        """The frame the reference's adaptive worker loop ends with (main.go:837-897, tile.go:34-66), in one call
        (ptmi_render_adaptive_dev): zeroed full-frame torch CUDA tensors accum [H,W,3], counts [H,W], stats [H,W,9]."""
        cam = camera if isinstance(camera, PtCamera) else make_camera(camera)
        assert tuple(accum.shape) == (params.height, params.width, 3) and accum.is_contiguous()
        assert tuple(counts.shape) == (params.height, params.width) and counts.is_contiguous()
        assert tuple(stats.shape) == (params.height, params.width, self.PIXEL_STATS) and stats.is_contiguous() and stats.element_size() == 4
        s = C.c_void_p(stream) if stream else None
        check(self._lib.ptmi_render_adaptive_dev(self.ctx.handle, C.byref(cam), C.byref(params), int(samples_per_pixel), int(max_samples),
                                                 accum.data_ptr(), counts.data_ptr(), stats.data_ptr(), s), self.ctx.handle)

    def render_frames_device(self, accum, counts, params: PtParams, cameras, spp=32, spp_begin=0, stream=None):
        """Several frames of the scene in one call (the reference's orbit, main.go:903-937, a batch of cameras at a time):
        frame f through cameras[f], added into accum[f] / counts[f] of contiguous torch CUDA tensors [F,H,W,3] / [F,H,W].
        The sums equal one render_device call per frame bit for bit; the pool's drain is paid once per call."""
        cams = (PtCamera * len(cameras))(*[c if isinstance(c, PtCamera) else make_camera(c) for c in cameras])
        n = len(cameras)
        assert tuple(accum.shape) == (n, params.height, params.width, 3) and accum.is_contiguous()
        assert tuple(counts.shape) == (n, params.height, params.width) and counts.is_contiguous()
        s = C.c_void_p(stream) if stream else None
        check(self._lib.ptmi_render_frames_dev(self.ctx.handle, cams, n, C.byref(params), spp_begin, spp, accum.data_ptr(),
                                               counts.data_ptr(), s), self.ctx.handle)

    def render_samples(self, params: PtParams, camera, rect=None, spp=1, spp_begin=0):
        """Per-sample radiance [h,w,spp,3] (no accumulation) -- for fine-grained parity checks."""
        cam = camera if isinstance(camera, PtCamera) else make_camera(camera)
        x0, y0, w, h = rect or (0, 0, params.width, params.height)
        out = np.zeros((h, w, spp, 3), np.float32)
        check(self._lib.ptmi_render_samples(self.ctx.handle, C.byref(cam), C.byref(params), x0, y0, w, h, spp_begin,
                                            spp, _ptr(out, C.c_float)), self.ctx.handle)
        return out

    def to_rgba(self, fb: Framebuffer) -> np.ndarray:
        h, w = fb.counts.shape
        rgba = np.zeros((h, w, 4), np.uint8)
        rgba[..., 3] = 255
        acc = np.ascontiguousarray(fb.accum, np.float32)
        cnt = np.ascontiguousarray(fb.counts, np.uint32)
        check(self._lib.ptmi_to_rgba(self.ctx.handle, _ptr(acc, C.c_float), _ptr(cnt, C.c_uint32), w, h,
                                     _ptr(rgba, C.c_uint8)), self.ctx.handle)
        return rgba
