"""Shared helpers for the parity tests: seeded ray batches incl. the adversarial cases of SURVEY 8(c)."""
from __future__ import annotations

import numpy as np

from pathtrace_amd import scenes

F32 = np.float32


def random_rays(verts, n, seed):
    """Rays with origins inside/around the mesh bounds and random directions."""
    lo, hi = verts.min(0), verts.max(0)
    ext = (hi - lo)
    k = np.arange(n, dtype=np.int64)
    o = np.stack([scenes.u01(scenes.hash3(seed, k, 1 + j)) for j in range(3)], 1).astype(F32)
    o = (lo - F32(0.25) * ext + o * (F32(1.5) * ext)).astype(F32)
    d = np.stack([scenes.u01(scenes.hash3(seed, k, 11 + j)) for j in range(3)], 1).astype(F32)
    d = scenes._normalize((d * F32(2) - F32(1)).astype(F32))
    d[(d == 0).all(1)] = np.array([0, -1, 0], F32)
    return np.ascontiguousarray(o), np.ascontiguousarray(d)


def adversarial_rays(verts, tris, n, seed):
    """Axis-parallel rays, origins exactly on vertex coordinates / slab planes (the 0*inf = NaN path,
    linear_bvh.go:24-26), rays aimed exactly at vertices and edge midpoints (the equal-t tie path,
    linear_bvh.go:133), zero direction components, tiny tmax, unnormalised directions."""
    k = np.arange(n, dtype=np.int64)
    vi = (scenes.hash3(seed, k, 1).astype(np.int64)) % verts.shape[0]
    ti = (scenes.hash3(seed, k, 2).astype(np.int64)) % tris.shape[0]
    kind = scenes.hash3(seed, k, 3).astype(np.int64) % 8
    o = np.zeros((n, 3), F32)
    d = np.zeros((n, 3), F32)
    tmax = np.full(n, 1000.0, F32)
    p = verts[vi]
    a, b, c = verts[tris[ti, 0]], verts[tris[ti, 1]], verts[tris[ti, 2]]
    axis = scenes.hash3(seed, k, 4).astype(np.int64) % 3
    sign = np.where(scenes.hash3(seed, k, 5) & 1, F32(1), F32(-1)).astype(F32)
    ro, rd = random_rays(verts, n, seed + 99)
    for i in range(n):
        kd = kind[i]
        if kd == 0:      # axis-parallel ray whose origin shares two coordinates with a vertex
            o[i] = p[i]; o[i, axis[i]] -= sign[i] * F32(3.0); d[i, axis[i]] = sign[i]
        elif kd == 1:    # origin exactly at a vertex, axis-parallel
            o[i] = p[i]; d[i, axis[i]] = sign[i]
        elif kd == 2:    # aimed exactly at a vertex (shared by several triangles)
            o[i] = ro[i]; d[i] = p[i] - ro[i]
        elif kd == 3:    # aimed at an edge midpoint, unnormalised direction
            m = ((a[i] + b[i]) * F32(0.5)).astype(F32); o[i] = ro[i]; d[i] = (m - ro[i]) * F32(0.37)
        elif kd == 4:    # one zero direction component, origin on a vertex plane
            o[i] = ro[i]; o[i, axis[i]] = p[i, axis[i]]; d[i] = rd[i]; d[i, axis[i]] = 0
        elif kd == 5:    # tmax right around the hit distance of a centroid shot
            ctr = ((a[i] + b[i] + c[i]) * F32(1.0 / 3.0)).astype(F32); o[i] = ro[i]
            v = ctr - ro[i]; ln = np.sqrt((v * v).sum(dtype=F32), dtype=F32)
            d[i] = v / ln if ln > 0 else np.array([0, 1, 0], F32)
            tmax[i] = ln * F32([0.999999, 1.0, 1.000001][int(k[i]) % 3])
        elif kd == 6:    # ray lying in the plane of a triangle (|det| ~ 0)
            o[i] = a[i] - (b[i] - a[i]); d[i] = (c[i] - a[i])
        else:            # negative zero / tiny components
            o[i] = ro[i]; d[i] = rd[i]; d[i, axis[i]] = F32(-0.0) if k[i] & 1 else F32(1e-42)
    bad = (d == 0).all(1)
    d[bad] = np.array([0, -1, 0], F32)
    return np.ascontiguousarray(o), np.ascontiguousarray(d), tmax


def bits_equal(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def write_obj_scene(dirpath, sc, name="scene", glass_materials=(), emissive=()):
    """Writes a SceneData as an OBJ / MTL pair (+ binary PPM textures) the way the reference's assets look
    (obj_loader.go:19-111 reads them): one object, `usemtl` groups, per-corner `vn` and `vt`, v stored as 1 - v (LoadObj
    flips it back), floats with 9 significant digits so that float32 values survive the text.  Materials listed in
    `glass_materials` get a name with the prefix "Glass" (tracer.go:49 detects glass by name)."""
    import os
    nt = sc.tris.shape[0]
    names = [("Glass%d" % i) if i in glass_materials else ("mat%d" % i) for i in range(sc.materials.shape[0])]
    tex_files = []
    for i, t in enumerate(sc.textures or []):
        fn = f"{name}_tex{i + 1}.ppm"
        rgb = np.stack([t & 0xFF, (t >> 8) & 0xFF, (t >> 16) & 0xFF], -1).astype(np.uint8)
        with open(os.path.join(dirpath, fn), "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (t.shape[1], t.shape[0]) + rgb.tobytes())
        tex_files.append(fn)
    mu = sc.materials.view(np.uint32)
    with open(os.path.join(dirpath, name + ".mtl"), "w") as f:
        for i, m in enumerate(sc.materials):
            f.write(f"newmtl {names[i]}\nKd {m[0]:.9g} {m[1]:.9g} {m[2]:.9g}\nKs {m[3]:.9g} {m[4]:.9g} {m[5]:.9g}\n"
                    f"Ke {m[6]:.9g} {m[7]:.9g} {m[8]:.9g}\nNi {m[9]:.9g}\nNs {m[10]:.9g}\nillum 2\n")
            if mu[i, 12]:
                f.write(f"map_Kd {tex_files[mu[i, 12] - 1]}\n")
            if mu[i, 13]:
                f.write(f"map_Bump {tex_files[mu[i, 13] - 1]}\n")
    uvs = sc.uvs if sc.uvs is not None else np.zeros((nt, 6), np.float32)
    with open(os.path.join(dirpath, name + ".obj"), "w") as f:
        f.write(f"mtllib {name}.mtl\n")
        for v in sc.verts:
            f.write(f"v {v[0]:.9g} {v[1]:.9g} {v[2]:.9g}\n")
        for n in sc.normals:
            f.write(f"vn {n[0]:.9g} {n[1]:.9g} {n[2]:.9g}\n")
        for uv in uvs.reshape(-1, 2):
            f.write(f"vt {uv[0]:.9g} {np.float32(1.0) - uv[1]:.9g}\n")
        f.write("o everything\n")
        cur = -1
        for t in range(nt):
            if sc.tri_material[t] != cur:
                cur = int(sc.tri_material[t])
                f.write(f"usemtl {names[cur]}\n")
            a, b, c = (int(x) + 1 for x in sc.tris[t])
            f.write(f"f {a}/{3 * t + 1}/{3 * t + 1} {b}/{3 * t + 2}/{3 * t + 2} {c}/{3 * t + 3}/{3 * t + 3}\n")
    return os.path.join(dirpath, name + ".obj")
