"""Seeded procedural scenes and ray batches (no mesh assets exist offline).

Everything here is plain data generation: triangle soups, per-corner normals,
material tables and ray batches that are handed, as arrays, to the HIP backend
(and, in tests, to the CPU oracle).  Geometry uses only float32 + - * / sqrt and
integer hashing, so a given (generator, seed) yields the same bits on any IEEE
machine; nothing is read from /root/reference.

Workloads follow SURVEY.md section 8(d):
  C1  bunny-class blob (~70 k tris) + ground quad + tessellated sphere
  C2  1 M-triangle soup + 16 M primary/secondary closest-hit rays
  C3  C1-style scene + glass + emissive quad
  C4  10 M-triangle soup
  C5  black hole + "AccretionDisk" annulus
The camera model is the reference's (main.go:862-874).
"""
from __future__ import annotations

from dataclasses import dataclass, field
import numpy as np

F32 = np.float32
U32 = np.uint32


# --------------------------------------------------------------------------- hashing
def hash_u32(x):
    """lowbias32 integer hash, vectorised over uint32 arrays."""
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x.astype(np.uint32)


def hash3(seed, a, b=0, c=0):
    h = hash_u32(np.asarray(a, dtype=np.uint64) + np.uint64(0x9E3779B9) * np.uint64(seed & 0xFFFFFFFF))
    h = hash_u32(h.astype(np.uint64) ^ (np.asarray(b, dtype=np.uint64) * np.uint64(0x85EBCA6B) & 0xFFFFFFFF))
    h = hash_u32(h.astype(np.uint64) ^ (np.asarray(c, dtype=np.uint64) * np.uint64(0xC2B2AE35) & 0xFFFFFFFF))
    return h


def u01(h):
    """uint32 hash -> float32 in [0,1) with 24 random bits (exact)."""
    return (np.asarray(h, dtype=np.uint32) >> 8).astype(F32) * F32(1.0 / 16777216.0)


# --------------------------------------------------------------------------- f32 vector helpers
def _dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _normalize(v):
    v = np.asarray(v, dtype=F32)
    ln = np.sqrt(_dot(v, v)).astype(F32)
    inv = np.where(ln == 0, F32(0), F32(1.0) / np.where(ln == 0, F32(1), ln)).astype(F32)
    return (v * inv[..., None]).astype(F32)


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1).astype(F32)


# --------------------------------------------------------------------------- primitives
def icosphere(level: int):
    """Unit icosphere, 20*4**level triangles.  Returns (verts f32 [n,3], tris u32 [m,3])."""
    t = F32((1.0 + np.sqrt(np.float64(5.0))) / 2.0)
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0],
                  [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=F32)
    v = _normalize(v)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
                  [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
                  [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        nv = v.shape[0]
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        e.sort(axis=1)
        key = e[:, 0] * np.int64(nv) + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        a = v[(uniq // nv)]
        b = v[(uniq % nv)]
        mid = _normalize(((a + b) * F32(0.5)).astype(F32))
        v = np.concatenate([v, mid], axis=0)
        m = inv.reshape(3, -1) + nv          # m[0]=ab, m[1]=bc, m[2]=ca
        a_, b_, c_ = f[:, 0], f[:, 1], f[:, 2]
        f = np.concatenate([np.stack([a_, m[0], m[2]], 1), np.stack([b_, m[1], m[0]], 1),
                            np.stack([c_, m[2], m[1]], 1), np.stack([m[0], m[1], m[2]], 1)], axis=0)
    return v.astype(F32), f.astype(U32)


def value_noise3(p, seed: int, freq: float):
    """Trilinear value noise in [-1,1], float32, hash lattice; p: [n,3] float32."""
    q = (np.asarray(p, dtype=F32) * F32(freq)).astype(F32)
    fl = np.floor(q).astype(F32)
    fr = (q - fl).astype(F32)
    i = fl.astype(np.int64) + 4096
    out = np.zeros(q.shape[0], dtype=F32)
    w = fr * fr * (F32(3.0) - F32(2.0) * fr)
    for dx in (0, 1):
        wx = w[:, 0] if dx else (F32(1.0) - w[:, 0])
        for dy in (0, 1):
            wy = w[:, 1] if dy else (F32(1.0) - w[:, 1])
            for dz in (0, 1):
                wz = w[:, 2] if dz else (F32(1.0) - w[:, 2])
                h = hash3(seed, i[:, 0] + dx, i[:, 1] + dy, i[:, 2] + dz)
                val = (u01(h) * F32(2.0) - F32(1.0)).astype(F32)
                out = (out + val * ((wx * wy).astype(F32) * wz).astype(F32)).astype(F32)
    return out


def displaced_icosphere(level, seed, radius=1.0, amp=0.25, center=(0, 0, 0), freq=2.0):
    v, f = icosphere(level)
    n = value_noise3(v, seed, freq) * F32(0.6) + value_noise3(v, seed + 17, freq * 2.7) * F32(0.4)
    r = (F32(radius) * (F32(1.0) + F32(amp) * n)).astype(F32)
    v = (v * r[:, None] + np.asarray(center, dtype=F32)[None, :]).astype(F32)
    return v, f


def terrain(nx, nz, seed, extent=10.0, amp=0.8, y0=0.0):
    """Height field with 2*nx*nz triangles on [-extent,extent]^2."""
    xs = (np.arange(nx + 1, dtype=F32) / F32(nx) * F32(2 * extent) - F32(extent)).astype(F32)
    zs = (np.arange(nz + 1, dtype=F32) / F32(nz) * F32(2 * extent) - F32(extent)).astype(F32)
    X, Z = np.meshgrid(xs, zs, indexing="xy")
    P = np.stack([X.ravel(), np.zeros(X.size, dtype=F32), Z.ravel()], axis=1).astype(F32)
    h = value_noise3(P, seed, 0.35) * F32(amp) + value_noise3(P, seed + 3, 1.3) * F32(amp * 0.25)
    P[:, 1] = (F32(y0) + h).astype(F32)
    ix, iz = np.meshgrid(np.arange(nx, dtype=np.int64), np.arange(nz, dtype=np.int64), indexing="xy")
    a = (iz * (nx + 1) + ix).ravel()
    b = a + 1
    c = a + (nx + 1)
    d = c + 1
    f = np.concatenate([np.stack([a, c, b], 1), np.stack([b, c, d], 1)], axis=1).reshape(-1, 3)
    return P, f.astype(U32)


def confetti(n, seed, extent=9.0, ylo=0.5, yhi=6.0, size=0.08):
    """n independent small triangles."""
    k = np.arange(n, dtype=np.int64)
    c = np.stack([(u01(hash3(seed, k, 1)) * F32(2 * extent) - F32(extent)),
                  (u01(hash3(seed, k, 2)) * F32(yhi - ylo) + F32(ylo)),
                  (u01(hash3(seed, k, 3)) * F32(2 * extent) - F32(extent))], axis=1).astype(F32)
    offs = []
    for j in range(3):
        o = np.stack([u01(hash3(seed, k, 10 + 3 * j)), u01(hash3(seed, k, 11 + 3 * j)),
                      u01(hash3(seed, k, 12 + 3 * j))], axis=1).astype(F32)
        offs.append(((o - F32(0.5)) * F32(2 * size)).astype(F32))
    v = np.stack([c + offs[0], c + offs[1], c + offs[2]], axis=1).reshape(-1, 3).astype(F32)
    f = np.arange(3 * n, dtype=U32).reshape(-1, 3)
    return v, f


def quad(p0, p1, p2, p3):
    v = np.array([p0, p1, p2, p3], dtype=F32)
    f = np.array([[0, 1, 2], [0, 2, 3]], dtype=U32)
    return v, f


def annulus(r_in, r_out, segments, y=0.0):
    """Flat ring in the XZ plane (the 'AccretionDisk' mesh of C5), 2*segments tris.
    Uses the rational circle parametrisation (no libm)."""
    k = np.arange(segments, dtype=np.float64)
    ang = k / segments * 2.0 * np.pi
    cs = np.cos(ang).astype(F32)
    sn = np.sin(ang).astype(F32)
    vin = np.stack([cs * F32(r_in), np.full(segments, y, dtype=F32), sn * F32(r_in)], 1)
    vout = np.stack([cs * F32(r_out), np.full(segments, y, dtype=F32), sn * F32(r_out)], 1)
    v = np.concatenate([vin, vout], 0).astype(F32)
    i = np.arange(segments, dtype=np.int64)
    j = (i + 1) % segments
    f = np.concatenate([np.stack([i, i + segments, j + segments], 1), np.stack([i, j + segments, j], 1)], 0)
    return v, f.astype(U32)


def merge(parts):
    """Concatenate (verts, tris) parts into one indexed mesh."""
    vs, fs, base = [], [], 0
    for v, f in parts:
        vs.append(np.asarray(v, dtype=F32))
        fs.append(np.asarray(f, dtype=np.int64) + base)
        base += v.shape[0]
    return np.concatenate(vs, 0).astype(F32), np.concatenate(fs, 0).astype(U32)


def corner_normals_smooth(verts, tris):
    """Per-face-corner normals (3 per triangle, like obj_loader.go's layout): area-weighted
    vertex normals, normalised."""
    a, b, c = verts[tris[:, 0]], verts[tris[:, 1]], verts[tris[:, 2]]
    fn = _cross((b - a).astype(F32), (c - a).astype(F32))
    vn = np.zeros((verts.shape[0], 3), dtype=np.float64)
    for k in range(3):
        np.add.at(vn, tris[:, k], fn.astype(np.float64))
    vn = _normalize(vn.astype(F32))
    return vn[tris.reshape(-1)].reshape(-1, 3).astype(F32)


def corner_normals_flat(verts, tris):
    a, b, c = verts[tris[:, 0]], verts[tris[:, 1]], verts[tris[:, 2]]
    fn = _normalize(_cross((b - a).astype(F32), (c - a).astype(F32)))
    return np.repeat(fn, 3, axis=0).astype(F32)


# --------------------------------------------------------------------------- soups (C2 / C4)
def make_soup(ntris: int, seed: int = 42):
    """Triangle soup with exactly `ntris` triangles: height-field terrain (~70 %), displaced
    icosphere blobs (~28 %), confetti triangles (the remainder).  Fits in [-10,10]^3-ish, well
    inside the reference's hard-coded +-1000 root bounds (main.go:804)."""
    if ntris < 64:
        v, f = confetti(ntris, seed)
        return v, f
    # blobs: pick the icosphere level so that ~14 blobs cover ~28 %
    level = 0
    while 20 * 4 ** (level + 1) * 14 <= 0.30 * ntris:
        level += 1
    per_blob = 20 * 4 ** level
    nblobs = max(1, int(0.28 * ntris) // per_blob)
    n_terrain = int(0.70 * ntris) // 2
    nz = max(1, int(np.sqrt(n_terrain / 1.4)))
    nx = max(1, n_terrain // nz)
    parts = [terrain(nx, nz, seed, extent=10.0, amp=0.9)]
    k = np.arange(nblobs, dtype=np.int64)
    cx = u01(hash3(seed, k, 101)) * F32(16.0) - F32(8.0)
    cy = u01(hash3(seed, k, 102)) * F32(3.0) + F32(1.5)
    cz = u01(hash3(seed, k, 103)) * F32(16.0) - F32(8.0)
    rad = u01(hash3(seed, k, 104)) * F32(0.8) + F32(0.6)
    for i in range(nblobs):
        parts.append(displaced_icosphere(level, seed + 1000 + i, radius=float(rad[i]), amp=0.3,
                                         center=(float(cx[i]), float(cy[i]), float(cz[i]))))
    used = 2 * nx * nz + nblobs * per_blob
    rest = ntris - used
    assert rest >= 0
    if rest:
        parts.append(confetti(rest, seed + 7))
    v, f = merge(parts)
    assert f.shape[0] == ntris
    return v, f


# --------------------------------------------------------------------------- camera (main.go:862-874)
@dataclass
class Camera:
    """camera.go:9-15"""
    position: tuple
    forward: tuple
    right: tuple
    up: tuple
    frustum: float = 2.0

    def as_array(self):
        return np.array([*self.position, *self.forward, *self.right, *self.up, self.frustum], dtype=F32)


def look_at(position, target, frustum=2.0, world_up=(0.0, 1.0, 0.0)):
    """Same basis construction as Camera.SphericalAround (camera.go:25-40): right = up x fwd,
    up = right x fwd."""
    p = np.asarray(position, dtype=F32)
    fwd = _normalize(np.asarray(target, dtype=F32) - p)
    right = _normalize(_cross(np.asarray(world_up, dtype=F32), fwd))
    up = _normalize(_cross(right, fwd))
    return Camera(tuple(p.tolist()), tuple(fwd.tolist()), tuple(right.tolist()), tuple(up.tolist()), frustum)


def camera_rays(cam: Camera, width: int, height: int, jitter=None, px=None, py=None):
    """Primary rays exactly as main.go:862-874, in float32:
        rx=(X+j)/W; ry=(Y+j)/H; px=(rx-.5)*2; py=(ry-.5)*2
        rayOrigin = pos + fwd*F + up*py + right*px; dir = normalize(rayOrigin - pos)
    jitter: None -> 0.5 (pixel centre, 'jitter-free grid'); else float32 array [H*W,2] in [0,1)."""
    if px is None:
        X, Y = np.meshgrid(np.arange(width, dtype=F32), np.arange(height, dtype=F32), indexing="xy")
        X = X.ravel()
        Y = Y.ravel()
    else:
        X, Y = np.asarray(px, dtype=F32), np.asarray(py, dtype=F32)
    if jitter is None:
        jx = jy = F32(0.5)
    else:
        jx, jy = jitter[:, 0].astype(F32), jitter[:, 1].astype(F32)
    rx = ((X + jx) / F32(width)).astype(F32)
    ry = ((Y + jy) / F32(height)).astype(F32)
    pxx = ((rx - F32(0.5)) * F32(2)).astype(F32)
    pyy = ((ry - F32(0.5)) * F32(2)).astype(F32)
    pos = np.asarray(cam.position, dtype=F32)
    fwd = np.asarray(cam.forward, dtype=F32)
    right = np.asarray(cam.right, dtype=F32)
    up = np.asarray(cam.up, dtype=F32)
    ro = (pos + fwd * F32(cam.frustum)).astype(F32)[None, :]
    ro = (ro + up[None, :] * pyy[:, None]).astype(F32)
    ro = (ro + right[None, :] * pxx[:, None]).astype(F32)
    d = _normalize((ro - pos[None, :]).astype(F32))
    o = np.broadcast_to(pos[None, :], d.shape).astype(F32).copy()
    return o, d


def cosine_bounce_rays(verts, tris, o, d, prim, t, seed):
    """Secondary rays of workload C2: cosine-weighted bounce off the primary hits; misses are
    replaced by a re-seeded random origin/direction inside the scene bounds."""
    n = o.shape[0]
    k = np.arange(n, dtype=np.int64)
    u1 = u01(hash3(seed, k, 1)).astype(np.float64)
    u2 = u01(hash3(seed, k, 2)).astype(np.float64)
    hit = prim >= 0
    pid = np.where(hit, prim, 0)
    a, b, c = verts[tris[pid, 0]], verts[tris[pid, 1]], verts[tris[pid, 2]]
    nrm = _normalize(_cross((b - a).astype(F32), (c - a).astype(F32)))
    flip = _dot(nrm, d) > 0
    nrm = np.where(flip[:, None], -nrm, nrm).astype(F32)
    up = np.where((np.abs(nrm[:, 1]) < F32(0.999))[:, None], np.array([0, 1, 0], dtype=F32),
                  np.array([1, 0, 0], dtype=F32)).astype(F32)
    t1 = _normalize(_cross(nrm, up))
    t2 = _cross(nrm, t1)
    r = np.sqrt(u1)
    th = 2.0 * np.pi * u2
    x = (r * np.cos(th)).astype(F32)
    y = (r * np.sin(th)).astype(F32)
    z = np.sqrt(np.maximum(0.0, 1.0 - u1)).astype(F32)
    nd = _normalize((t1 * x[:, None] + t2 * y[:, None] + nrm * z[:, None]).astype(F32))
    p = (o + d * t[:, None]).astype(F32)
    no = (p + nrm * F32(1e-3)).astype(F32)
    # misses: random origin / direction inside the bounds
    lo, hi = verts.min(0), verts.max(0)
    ro = np.stack([u01(hash3(seed, k, 11 + j)) for j in range(3)], 1).astype(F32)
    ro = (lo[None, :] + ro * (hi - lo)[None, :]).astype(F32)
    rd = np.stack([u01(hash3(seed, k, 21 + j)) for j in range(3)], 1).astype(F32)
    rd = _normalize((rd * F32(2) - F32(1)).astype(F32))
    rd = np.where((_dot(rd, rd) == 0)[:, None], np.array([0, -1, 0], dtype=F32), rd).astype(F32)
    no = np.where(hit[:, None], no, ro).astype(F32)
    nd = np.where(hit[:, None], nd, rd).astype(F32)
    return np.ascontiguousarray(no), np.ascontiguousarray(nd)


def soup_camera(verts):
    """Camera looking down at a soup from above one corner."""
    lo, hi = verts.min(0), verts.max(0)
    ctr = (lo + hi) * F32(0.5)
    ext = float((hi - lo).max())
    pos = (float(ctr[0] + 0.55 * ext), float(hi[1] + 0.35 * ext), float(ctr[2] + 0.65 * ext))
    tgt = (float(ctr[0]), float(lo[1] + 0.2 * (hi[1] - lo[1])), float(ctr[2]))
    return look_at(pos, tgt, frustum=2.0)


# --------------------------------------------------------------------------- shaded scenes (C1/C3/C5)
# Material flags (the reference detects these by material *name*: tracer.go:49, :95)
MAT_GLASS = 1
MAT_ACCRETION = 2


@dataclass
class SceneData:
    """Flattened scene in the layout ptmi_upload_scene takes (mirrors VNMU, vnmu.go:8-13,
    and Scene, scene.go:3-9)."""
    verts: np.ndarray                 # [nv,3] f32
    tris: np.ndarray                  # [nt,3] u32
    normals: np.ndarray               # [3*nt,3] f32 per-corner
    tri_material: np.ndarray          # [nt] u32
    materials: np.ndarray             # [nm,16] f32 rows, see MATERIAL_COLUMNS
    lights: np.ndarray                # [nl,12] f32 rows, see LIGHT_COLUMNS
    sky: np.ndarray                   # [16] f32, see SKY_COLUMNS
    blackhole: np.ndarray | None = None   # PtBlackHole: pos xyz, Rs, i32 has_disk, inner, outer, pad
    sky_image: np.ndarray | None = None   # [h,w,3] u16 equirect image for sky kind 3
    camera: Camera | None = None
    names: list = field(default_factory=list)
    uvs: np.ndarray | None = None         # [nt,6] f32: u,v of corners A,B,C (vnmu.UVs, vnmu.go:12)
    textures: list | None = None          # [h,w] u32 images packed R|G<<8|B<<16|A<<24 (CacheImage, renderer.go:94-119)


# Rows are binary images of the C structs of include/ptmi.h (integer fields stored as their bits):
# PtMaterial: Kd rgb, Ks rgb, Ke rgb, Ni, Ns, u32 flags, u32 diffuse_map, u32 bump_map (1-based, 0 = nil), pad x2
#                                                                                      (obj.go:63-79)
MATERIAL_COLUMNS = 16
# PtLight: i32 kind (0 sun, 1 point), color rgb, dir/pos xyz, intensity, pad x4      (lights.go:11-15,32-35)
LIGHT_COLUMNS = 12
# PtSky: i32 kind (0 none,1 solid,2 gradient,3 image), horizon rgb (or solid colour), zenith rgb, ground rgb,
#        intensity, i32 image width, i32 image height, pad x3                        (skybox.go)
SKY_COLUMNS = 16


def material(kd=(0.7, 0.7, 0.7), ks=(0, 0, 0), ke=(0, 0, 0), ni=1.0, ns=0.0, flags=0, diffuse_map=0, bump_map=0):
    row = np.zeros(MATERIAL_COLUMNS, dtype=F32)
    row[12:13].view(np.uint32)[0] = diffuse_map
    row[13:14].view(np.uint32)[0] = bump_map
    row[0:3] = kd
    row[3:6] = ks
    row[6:9] = ke
    row[9] = ni
    row[10] = ns
    row[11:12].view(np.uint32)[0] = flags
    return row


def sun(direction, intensity, color=(1, 1, 1)):
    row = np.zeros(LIGHT_COLUMNS, dtype=F32)
    row[0:1].view(np.int32)[0] = 0
    row[1:4] = color
    row[4:7] = direction
    row[7] = intensity
    return row


def point_light(position, intensity, color=(1, 1, 1)):
    row = np.zeros(LIGHT_COLUMNS, dtype=F32)
    row[0:1].view(np.int32)[0] = 1
    row[1:4] = color
    row[4:7] = position
    row[7] = intensity
    return row


def sky_gradient(horizon, zenith, ground, intensity):
    row = np.zeros(SKY_COLUMNS, dtype=F32)
    row[0:1].view(np.int32)[0] = 2
    row[1:4] = horizon
    row[4:7] = zenith
    row[7:10] = ground
    row[10] = intensity
    return row


def sky_solid(color):
    row = np.zeros(SKY_COLUMNS, dtype=F32)
    row[0:1].view(np.int32)[0] = 1
    row[1:4] = color
    return row


def sky_none():
    return np.zeros(SKY_COLUMNS, dtype=F32)


def sky_image(img_u16, intensity):
    """ImageSkybox (skybox.go:46-107): img_u16 [h,w,3] uint16 equirect image."""
    row = np.zeros(SKY_COLUMNS, dtype=F32)
    iv = row.view(np.int32)
    iv[0] = 3
    row[10] = intensity
    iv[11] = img_u16.shape[1]
    iv[12] = img_u16.shape[0]
    return row


def _sponza_sky():
    # main.go:519-524
    s = F32(1.0 / 255)
    return sky_gradient(np.array([200, 230, 255], dtype=F32) * s, np.array([50, 120, 255], dtype=F32) * s,
                        np.array([76, 76, 76], dtype=F32) * s, 4.0)


def _sponza_sun():
    # main.go:512-516
    d = _normalize(np.array([0.1, 1.0, 0.1], dtype=F32))
    return sun(d, 3.0)


def bunny_scene(blob_level=None, glass=False, emissive=False, seed=1, small=False):
    """C1 (and, with glass/emissive, C3): displaced-icosphere 'bunny-class' blob on a ground quad
    pair with a tessellated sphere; gradient sky + sun as main.go:509-524."""
    level = blob_level if blob_level is not None else (3 if small else 6)
    parts, mats, tri_mat, normals = [], [], [], []

    def add(vf, mat_row, smooth=True):
        v, f = vf
        parts.append((v, f))
        mats.append(mat_row)
        tri_mat.append(np.full(f.shape[0], len(mats) - 1, dtype=U32))
        normals.append(corner_normals_smooth(v, f) if smooth else corner_normals_flat(v, f))

    add(displaced_icosphere(level, seed, radius=1.0, amp=0.35, center=(0.0, 1.05, 0.0)),
        material(kd=(0.75, 0.6, 0.45)))
    add(quad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6)), material(kd=(0.55, 0.55, 0.55)), smooth=False)
    sv, sf = icosphere(1 if small else 3)
    sv = (sv * F32(0.6) + np.array([1.9, 0.6, 0.4], dtype=F32)).astype(F32)
    add((sv, sf), material(kd=(0.2, 0.2, 0.2), ks=(0.95, 0.95, 0.95), ns=400.0))
    if glass:
        gv, gf = icosphere(1 if small else 3)
        gv = (gv * F32(0.55) + np.array([-1.8, 0.55, 0.9], dtype=F32)).astype(F32)
        add((gv, gf), material(kd=(0.9, 0.9, 0.9), ni=1.5, flags=MAT_GLASS))
        gv2, gf2 = icosphere(1 if small else 2)
        gv2 = (gv2 * F32(0.35) + np.array([0.2, 0.35, 1.9], dtype=F32)).astype(F32)
        add((gv2, gf2), material(kd=(0.8, 0.9, 0.8), ks=(0.5, 0.5, 0.5), ns=200.0, ni=1.33, flags=MAT_GLASS))
    if emissive:
        add(quad((-1.0, 3.2, -1.0), (1.0, 3.2, -1.0), (1.0, 3.2, 1.0), (-1.0, 3.2, 1.0)),
            material(kd=(0.0, 0.0, 0.0), ke=(12.0, 11.0, 9.0)), smooth=False)
    v, f = merge(parts)
    cam = look_at((3.2, 2.4, 5.2), (0.0, 0.8, 0.0), frustum=2.0)
    return SceneData(verts=v, tris=f, normals=np.concatenate(normals, 0).astype(F32),
                     tri_material=np.concatenate(tri_mat).astype(U32), materials=np.stack(mats).astype(F32),
                     lights=np.stack([_sponza_sun()]).astype(F32), sky=_sponza_sky(), camera=cam)


def c4_scene(blob_level=9, terrain_n=1600, seed=4):
    """C4-class scene (BASELINE.json configs[3]: 10 M triangles): a level-9 displaced icosphere (5.2 M triangles) over
    a 1600 x 1600 height field (5.1 M), a mirror and a glass sphere, an emissive quad; gradient sky + sun as C1/C3.
    Smaller levels give the same scene for tests."""
    parts, mats, tri_mat, normals = [], [], [], []

    def add(vf, mat_row, smooth=True):
        v, f = vf
        parts.append((v, f))
        mats.append(mat_row)
        tri_mat.append(np.full(f.shape[0], len(mats) - 1, dtype=U32))
        normals.append(corner_normals_smooth(v, f) if smooth else corner_normals_flat(v, f))

    add(displaced_icosphere(blob_level, seed, radius=1.0, amp=0.35, center=(0.0, 1.6, 0.0)), material(kd=(0.75, 0.6, 0.45)))
    add(terrain(terrain_n, terrain_n, seed + 1, extent=8.0, amp=0.5, y0=0.0), material(kd=(0.5, 0.55, 0.45)))
    sub = max(1, min(5, blob_level - 3))
    sv, sf = icosphere(sub)
    add(((sv * F32(0.6) + np.array([2.1, 1.3, 0.4], dtype=F32)).astype(F32), sf),
        material(kd=(0.2, 0.2, 0.2), ks=(0.95, 0.95, 0.95), ns=400.0))
    gv, gf = icosphere(sub)
    add(((gv * F32(0.55) + np.array([-2.0, 1.3, 0.9], dtype=F32)).astype(F32), gf),
        material(kd=(0.9, 0.9, 0.9), ni=1.5, flags=MAT_GLASS))
    add(quad((-1.0, 4.0, -1.0), (1.0, 4.0, -1.0), (1.0, 4.0, 1.0), (-1.0, 4.0, 1.0)),
        material(kd=(0.0, 0.0, 0.0), ke=(12.0, 11.0, 9.0)), smooth=False)
    v, f = merge(parts)
    cam = look_at((3.6, 3.2, 5.6), (0.0, 1.2, 0.0), frustum=2.0)
    return SceneData(verts=v, tris=f, normals=np.concatenate(normals, 0).astype(F32),
                     tri_material=np.concatenate(tri_mat).astype(U32), materials=np.stack(mats).astype(F32),
                     lights=np.stack([_sponza_sun()]).astype(F32), sky=_sponza_sky(), camera=cam)


def pack_rgba(rgb):
    """[h,w,3] (or [h,w,4]) uint8 -> packed texels as CacheImage stores them (renderer.go:103-104)."""
    rgb = np.asarray(rgb, dtype=np.uint32)
    a = rgb[..., 3] if rgb.shape[-1] == 4 else np.full(rgb.shape[:2], 255, dtype=np.uint32)
    return (rgb[..., 0] | (rgb[..., 1] << 8) | (rgb[..., 2] << 16) | (a << 24)).astype(np.uint32)


def procedural_texture(width, height, seed, kind="checker"):
    """Deterministic test image: a coloured checker or smooth value noise (for bump maps)."""
    yy, xx = np.mgrid[0:height, 0:width]
    h = (xx.astype(np.uint32) * np.uint32(0x9E3779B1)) ^ (yy.astype(np.uint32) * np.uint32(0x85EBCA77)) ^ np.uint32(seed * 0x27D4EB2F & 0xFFFFFFFF)
    h ^= h >> np.uint32(15); h *= np.uint32(0x2C1B3C6D); h ^= h >> np.uint32(12)
    if kind == "checker":
        c = ((xx // max(1, width // 8)) + (yy // max(1, height // 8))) & 1
        rgb = np.stack([60 + 150 * c + (h & 31), 200 - 120 * c + ((h >> 5) & 31), 90 + 40 * c + ((h >> 10) & 31)], -1)
    else:
        f = 0.5 + 0.25 * np.sin(xx * (6.2831853 * 3 / width)) + 0.25 * np.cos(yy * (6.2831853 * 5 / height))
        g = np.clip(f * 255.0, 0, 255).astype(np.uint32)
        rgb = np.stack([g, g, g], -1)
    return pack_rgba(np.clip(rgb, 0, 255).astype(np.uint8))


def textured_scene(small=True, seed=1):
    """A Sponza-class stand-in for the texture path (tracer.go:468-514): the blob carries a diffuse map with
    spherical UVs, the ground a bump map with UVs that run outside [0,1) and below 0 (tile(), obj_loader.go:11),
    the small sphere a diffuse + bump pair, and a second quad a bump map with degenerate UVs (the det < 1e-6
    fallback of TransformNormalToWorldSpace, renderer.go:229-250)."""
    level = 3 if small else 5
    parts, mats, tri_mat, normals, uvs = [], [], [], [], []

    def sph_uv(v, f, center):
        d = v[f] - np.asarray(center, dtype=F32)                      # [nt,3,3]
        d = d / np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-9)
        u = (np.arctan2(d[..., 2], d[..., 0]) / (2 * np.pi) + 0.5).astype(F32)
        w = (np.arccos(np.clip(d[..., 1], -1, 1)) / np.pi).astype(F32)
        return np.stack([u, F32(1.0) - w], -1).reshape(-1, 6).astype(F32)     # v = 1 - v (obj_loader.go:92)

    def add(vf, mat_row, uv, smooth=True):
        v, f = vf
        parts.append((v, f)); mats.append(mat_row)
        tri_mat.append(np.full(f.shape[0], len(mats) - 1, dtype=U32))
        normals.append(corner_normals_smooth(v, f) if smooth else corner_normals_flat(v, f))
        uvs.append(uv)

    bv, bf = displaced_icosphere(level, seed, radius=1.0, amp=0.3, center=(0.0, 1.05, 0.0))
    add((bv, bf), material(kd=(0.5, 0.5, 0.5), diffuse_map=1), sph_uv(bv, bf, (0.0, 1.05, 0.0)))
    qv, qf = quad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6))
    quv = (qv[qf][..., [0, 2]] * F32(0.37) - F32(0.8)).reshape(-1, 6).astype(F32)       # spans about [-3, 1.4]
    add((qv, qf), material(kd=(0.55, 0.55, 0.55), bump_map=2), quv, smooth=False)
    sv, sf = icosphere(1 if small else 3)
    sv = (sv * F32(0.6) + np.array([1.9, 0.6, 0.4], dtype=F32)).astype(F32)
    add((sv, sf), material(kd=(0.2, 0.2, 0.2), diffuse_map=3, bump_map=2), sph_uv(sv, sf, (1.9, 0.6, 0.4)))
    wv, wf = quad((-3, 0, -3), (-3, 3, -3), (3, 3, -3), (3, 0, -3))
    add((wv, wf), material(kd=(0.7, 0.6, 0.5), bump_map=2), np.full((wf.shape[0], 6), 0.25, dtype=F32), smooth=False)
    v, f = merge(parts)
    cam = look_at((3.2, 2.4, 5.2), (0.0, 0.8, 0.0), frustum=2.0)
    tex = [procedural_texture(64, 32, 1, "checker"), procedural_texture(48, 48, 2, "noise"),
           procedural_texture(17, 9, 3, "checker")]
    return SceneData(verts=v, tris=f, normals=np.concatenate(normals, 0).astype(F32),
                     tri_material=np.concatenate(tri_mat).astype(U32), materials=np.stack(mats).astype(F32),
                     lights=np.stack([_sponza_sun()]).astype(F32), sky=_sponza_sky(), camera=cam,
                     uvs=np.concatenate(uvs, 0).astype(F32), textures=tex)


def blackhole_scene(segments=256):
    """C5: black hole Rs=100 at the origin, annulus r in [300,450] named AccretionDisk,
    camera on a radius-1500 sphere (main.go:625-660); procedural gradient sky replaces the
    missing nebula PNG."""
    v, f = annulus(300.0, 450.0, segments)
    nrm = np.tile(np.array([0, 1, 0], dtype=F32), (3 * f.shape[0], 1))
    # Ks = 1: the reflective branch keeps the procedural diffuse colour (tracer.go:111, :207-222)
    mats = np.stack([material(kd=(1, 1, 1), ks=(1, 1, 1), ns=1000.0, flags=MAT_ACCRETION)])
    bh = np.array([0, 0, 0, 100.0, 0.0, 300.0, 450.0, 0.0], dtype=F32)
    bh[4:5].view(np.int32)[0] = 1          # PtBlackHole.has_disk
    # SphericalAround(Vec3{}, 1500, theta=90deg, phi=83deg): camera.go:17-41 (argument names are swapped
    # there: radius, phi<-theta(90deg), theta<-phi(83deg))
    ph, th = 90.0 * 0.0174533, 83.0 * 0.0174533
    pos = (1500.0 * np.sin(th) * np.cos(ph), 1500.0 * np.cos(th), 1500.0 * np.sin(th) * np.sin(ph))
    cam = look_at(pos, (0, 0, 0), frustum=2.0)
    sky = sky_gradient((0.02, 0.02, 0.05), (0.0, 0.0, 0.02), (0.01, 0.01, 0.03), 1.0)
    return SceneData(verts=v, tris=f, normals=nrm, tri_material=np.zeros(f.shape[0], dtype=U32),
                     materials=mats, lights=np.zeros((0, LIGHT_COLUMNS), dtype=F32), sky=sky, blackhole=bh,
                     camera=cam)
