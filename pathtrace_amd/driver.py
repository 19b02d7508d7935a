"""Headless counterpart of the reference's frame loop (main.go:832-941) and its statistics overlay
(main.go:999-1042), on top of the GPU integrator: render a frame tile by tile, write it as PNG through
ToRGBA (vec3.go:95-106, mirrored as main.go:896 does), orbit the camera one degree and start over.

    python -m pathtrace_amd.driver --scene c1 --frames 3 --out orbit

Camera helpers restate camera.go (SphericalAround :17-41, ApplyRotation :79-103) in float32.
"""
from __future__ import annotations

import argparse
import math
import os
import struct
import time
import zlib

import numpy as np

from .scenes import Camera, F32

DEG = 0.0174533          # main.go:929 (theta += 1 * 0.0174533)


def _n(v):
    l = F32(np.sqrt(F32(F32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])))
    return np.zeros(3, F32) if l == 0 else (v * F32(F32(1.0) / l)).astype(F32)


def _x(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], F32)


def spherical_around(center, radius, phi, theta, frustum=2.0) -> Camera:
    """camera.go:17-41.  (The call site swaps the names: main.go:931 passes (theta, phi) for (phi, theta).)"""
    c = np.asarray(center, F32)
    r, phi, theta = F32(radius), F32(phi), F32(theta)
    pos = np.array([c[0] + r * F32(math.sin(theta)) * F32(math.cos(phi)), c[1] + r * F32(math.cos(theta)),
                    c[2] + r * F32(math.sin(theta)) * F32(math.sin(phi))], F32)
    fwd = _n((c - pos).astype(F32))
    right = _n(_x(np.array([0, 1, 0], F32), fwd))
    if np.sqrt(float(right @ right)) < 1e-6:
        right = np.array([1, 0, 0], F32)
    up = _n(_x(right, fwd))
    return Camera(tuple(pos.tolist()), tuple(fwd.tolist()), tuple(right.tolist()), tuple(up.tolist()), frustum)


def apply_rotation(cam: Camera, rot_y, rot_x) -> Camera:
    """camera.go:79-103: yaw about global Y, pitch about the local right axis, re-orthogonalise."""
    fwd, right, up = (np.asarray(v, F32) for v in (cam.forward, cam.right, cam.up))

    def about_y(v, a):
        c, s = F32(math.cos(a)), F32(math.sin(a))
        return np.array([v[0] * c + v[2] * s, v[1], -v[0] * s + v[2] * c], F32)

    def about_axis(v, axis, a):
        c, s = F32(math.cos(a)), F32(math.sin(a))
        k = _n(axis)
        d = F32(F32(k[0] * v[0] + k[1] * v[1]) + k[2] * v[2])
        cr = _x(k, v)
        return (v * c + cr * s + k * d * F32(F32(1) - c)).astype(F32)

    if rot_y != 0:
        fwd, right = about_y(fwd, rot_y), about_y(right, rot_y)
    if rot_x != 0:
        fwd, up = about_axis(fwd, right, rot_x), about_axis(up, right, rot_x)
    fwd, right = _n(fwd), _n(right)
    up = _n(_x(right, fwd))
    return Camera(cam.position, tuple(fwd.tolist()), tuple(right.tolist()), tuple(up.tolist()), cam.frustum)


def humanize(x) -> str:
    """utils.go:11-22"""
    x = float(x)
    for div, suffix in ((1e9, "G"), (1e6, "M"), (1e3, "K")):
        if x >= div:
            return f"{x / div:.1f}{suffix}"
    return f"{x:.1f}"


def write_png(path, rgba):
    rgba = np.ascontiguousarray(rgba, np.uint8)
    h, w, _ = rgba.shape
    raw = b"".join(b"\x00" + rgba[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


class Stats:
    """The overlay numbers of main.go:999-1042: rays/s now / min / avg / max (a 'ray' is one diffuse shading event,
    tracer.go:666), samples done and the elapsed / remaining clock."""

    def __init__(self, total_samples):
        self.total, self.t0 = total_samples, time.time()
        self.rays = self.samples = 0
        self.min_speed, self.max_speed = float("inf"), 0.0
        self.last_rays, self.last_t = 0, self.t0

    def update(self, rays, samples):
        now = time.time()
        self.rays, self.samples = rays, samples
        dt = max(now - self.last_t, 1e-9)
        cur = (rays - self.last_rays) / dt
        self.min_speed, self.max_speed = min(self.min_speed, cur), max(self.max_speed, cur)
        self.last_rays, self.last_t = rays, now
        el = max(now - self.t0, 1e-9)
        speed = max(1.0, samples / el)
        rem = int((self.total - samples) / speed)
        return (f"{humanize(cur)} rays/s (Min: {humanize(self.min_speed)}, Avg: {humanize(rays / el)}, Max: "
                f"{humanize(self.max_speed)})  {humanize(samples)}/{humanize(self.total)} samples "
                f"({100.0 * samples / self.total:.1f}%)  {int(el) // 60:02d}:{int(el) % 60:02d} / {rem // 60:02d}:{rem % 60:02d}")


def checkpoint_tag(scene, params, cam, spp, splits):
    """What a checkpoint must agree on to be continued: image size, integrator parameters, seed, spp, tiling, camera and
    the scene's geometry and materials (a hash, as hex)."""
    import hashlib
    h = hashlib.sha256()
    h.update(repr((params.width, params.height, params.bounces, params.scatter_rays, params.max_steps, float(params.ambient),
                   float(params.step_size), params.seed, int(spp), tuple(splits))).encode())
    h.update(np.asarray(cam.as_array(), np.float32).tobytes())
    for name in ("verts", "tris", "normals", "materials", "tri_material", "lights", "sky"):
        v = getattr(scene, name, None)
        if v is not None:
            h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()[:32]


def frame_loop(renderer, scene, params, frames=1, spp=32, splits=(4, 4), out_dir=None, orbit=None, log=print,
               checkpoint=None):
    """main.go:454-491 tiles, :832-899 per-tile work, :904-939 write the frame / orbit / restart.
    orbit = (radius, theta0, phi): SphericalAround(origin, radius, theta, phi) with theta += 1 degree per frame.
    checkpoint = a path: the first frame's sums and counts are written there after every tile, and a file found there
    at the start is continued (tiles that already have their samples are skipped); the reference has no such thing."""
    from .render import Framebuffer
    W, H = params.width, params.height
    sx, sy = splits
    tw, th = W // sx, H // sy
    cam = scene.camera
    theta = orbit[1] if orbit else 0.0
    images = []
    for j in range(1, frames + 1):
        fb = renderer.new_framebuffer(W, H)
        tag = checkpoint_tag(scene, params, cam, spp, splits) if checkpoint else None
        if checkpoint and j == 1 and os.path.exists(checkpoint):
            # a truncated file (a kill during an earlier write of a non-atomic writer) or a checkpoint of another
            # frame -- other scene, camera, seed, size, spp or tiling -- is ignored, never mixed into this frame
            try:
                old, old_tag = Framebuffer.load(checkpoint), Framebuffer.load_tag(checkpoint)
            except Exception as e:              # noqa: BLE001
                old, old_tag = None, None
                log(f">> ignoring {checkpoint}: {type(e).__name__}: {e}")
            if old is not None and old.counts.shape == (H, W) and old_tag == tag:
                fb = old
                log(f">> resuming {checkpoint}: {int(fb.counts.sum())} of {W * H * spp} samples done")
            elif old is not None:
                log(f">> ignoring {checkpoint}: it belongs to another frame (tag {old_tag}, this frame {tag})")
        stats = Stats(W * H * spp)
        renderer.bvh.reset_counters()
        for i in range(sx * sy):                               # tile i covers block (i % splitsX, i / splitsX)
            x0, y0 = (i % sx) * tw, (i // sx) * th
            have = int(fb.counts[y0:y0 + th, x0:x0 + tw].min())           # a tile is sampled uniformly (one call)
            if have >= spp:
                continue
            renderer.render_tile(fb, params, cam, (x0, y0, tw, th), spp=spp - have, spp_begin=have)
            c = renderer.bvh.counters()
            log(stats.update(int(c.diffuse_events), int(c.samples)))
            if checkpoint and j == 1:
                fb.save(checkpoint, tag=tag)
        rgba = renderer.to_rgba(fb)
        images.append(rgba)
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)
            write_png(os.path.join(out_dir, f"frame{j}.png"), rgba)
        if orbit:
            theta += 1 * DEG
            cam = spherical_around((0.0, 0.0, 0.0), orbit[0], theta, orbit[2], cam.frustum)
            log(f">> {theta}")
    return images


def orbit_batched(renderer, scene, params, frames=8, spp=32, batch=8, out_dir=None, orbit=None, log=print):
    """The same frames as frame_loop (main.go:903-937: frame j, theta += 1 degree, SphericalAround, next frame), `batch`
    of them per ptmi_render_frames_dev call: a call pays the drain of the path pool once, not once per frame (8 frames
    of C3 at 1080p, 32 spp: 40.6 -> 32.9 ms per frame).  The images are byte-identical to frame_loop's."""
    import torch
    from .render import Framebuffer
    W, H = params.width, params.height
    dev = torch.device("cuda", renderer.ctx.device if hasattr(renderer.ctx, "device") else 0)
    cams, cam, theta = [], scene.camera, orbit[1] if orbit else 0.0
    for _ in range(frames):
        cams.append(cam)
        if orbit:
            theta += 1 * DEG
            cam = spherical_around((0.0, 0.0, 0.0), orbit[0], theta, orbit[2], cam.frustum)
    images = []
    for j0 in range(0, frames, batch):
        part = cams[j0:j0 + batch]
        acc = torch.zeros((len(part), H, W, 3), dtype=torch.float32, device=dev)
        cnt = torch.zeros((len(part), H, W), dtype=torch.int32, device=dev)
        t0 = time.time()
        renderer.render_frames_device(acc, cnt, params, part, spp=spp)
        torch.cuda.synchronize()
        dt = time.time() - t0
        log(f">> frames {j0 + 1}..{j0 + len(part)}: {humanize(len(part) * W * H * spp / max(dt, 1e-9))} samples/s")
        a, c = acc.cpu().numpy(), cnt.cpu().numpy().astype(np.uint32)
        for k in range(len(part)):
            rgba = renderer.to_rgba(Framebuffer(a[k], c[k]))
            images.append(rgba)
            if out_dir:
                os.makedirs(out_dir, exist_ok=True)
                write_png(os.path.join(out_dir, f"frame{j0 + k + 1}.png"), rgba)
    return images


def main():
    from . import scenes
    from .render import Renderer, make_params
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--scene", default="c1", choices=["c1", "c3", "c5", "textured"])
    ap.add_argument("--obj", default=None, help="render an OBJ file instead (lights/sky of the c1 scene)")
    ap.add_argument("--width", type=int, default=512); ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--spp", type=int, default=32); ap.add_argument("--bounces", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1); ap.add_argument("--out", default="orbit")
    ap.add_argument("--orbit-radius", type=float, default=0.0)
    ap.add_argument("--checkpoint", default=None, help="save the first frame's sums/counts here after every tile; continue a file found there")
    ap.add_argument("--batch-frames", type=int, default=0, help="render this many orbit frames per call (ptmi_render_frames_dev); 0 = the reference's tile loop")
    ap.add_argument("--max-spp", type=int, default=0, help="adaptive sampling (tile.go:34-66): --spp samples at a time per pixel until it has this many or its contrast is under 0.01; writes <out>/adaptive.png and the per-pixel counts")
    a = ap.parse_args()
    if a.obj:
        from .obj import LoadObj
        sc = LoadObj(a.obj)
        base = scenes.bunny_scene(small=True)
        sc.lights, sc.sky = base.lights, base.sky
        lo, hi = sc.verts.min(0), sc.verts.max(0)
        sc.camera = scenes.look_at(tuple((hi + (hi - lo) * 0.8).tolist()), tuple(((lo + hi) / 2).tolist()))
        p = make_params(a.width, a.height, bounces=a.bounces)
    elif a.scene == "c5":
        sc = scenes.blackhole_scene(); p = make_params(a.width, a.height, bounces=a.bounces, max_steps=3000, step_size=1.0)
    elif a.scene == "textured":
        sc = scenes.textured_scene(small=False); p = make_params(a.width, a.height, bounces=a.bounces)
    else:
        sc = scenes.bunny_scene(glass=a.scene == "c3", emissive=a.scene == "c3"); p = make_params(a.width, a.height, bounces=a.bounces)
    r = Renderer(sc)
    orbit = None
    if a.orbit_radius > 0 or a.scene == "c5":
        orbit = (a.orbit_radius or 1500.0, 90.0 * DEG, 83.0 * DEG)      # main.go:625-631
    if a.max_spp > a.spp:
        import torch
        from .render import Framebuffer
        dev = torch.device("cuda", 0)
        acc = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device=dev)
        cnt = torch.zeros((p.height, p.width), dtype=torch.int32, device=dev)
        stats = torch.zeros((p.height, p.width, r.PIXEL_STATS), dtype=torch.float32, device=dev)
        t0 = time.time()
        r.render_adaptive_device(acc, cnt, stats, p, sc.camera, samples_per_pixel=a.spp, max_samples=a.max_spp)   # ptmi_render_adaptive_dev
        torch.cuda.synchronize()
        dt = time.time() - t0
        c = cnt.cpu().numpy().astype(np.uint32)
        print(f">> {int(c.sum())} samples, {humanize(c.sum() / max(dt, 1e-9))} samples/s; "
              f"{float(c.mean()):.1f} per pixel (min {int(c.min())}, max {int(c.max())})")
        os.makedirs(a.out, exist_ok=True)
        write_png(os.path.join(a.out, "adaptive.png"), r.to_rgba(Framebuffer(acc.cpu().numpy(), c)))
        np.save(os.path.join(a.out, "adaptive_counts.npy"), c)
    elif a.batch_frames > 0:
        orbit_batched(r, sc, p, frames=a.frames, spp=a.spp, batch=a.batch_frames, out_dir=a.out, orbit=orbit)
    else:
        frame_loop(r, sc, p, frames=a.frames, spp=a.spp, out_dir=a.out, orbit=orbit, checkpoint=a.checkpoint)


if __name__ == "__main__":
    main()
