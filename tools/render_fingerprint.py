#!/usr/bin/env python3
"""Bit-level fingerprints of the integrator's per-sample output on a fixed set of small cases.
A refactor of the shade kernels must not change one of them: run before and after, diff the lines.
usage: tools/render_fingerprint.py [--out file]"""
import argparse, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pathtrace_amd import Context, scenes
from pathtrace_amd.render import Renderer, make_params

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
a = ap.parse_args()


def cases():
    s = scenes.bunny_scene(small=True, glass=True, emissive=True)
    yield "glass+emissive b2", s, make_params(96, 64, bounces=2, seed=11), s.camera, 4
    yield "glass+emissive b3 scatter2", s, make_params(64, 64, bounces=3, scatter_rays=2, seed=11), s.camera, 3
    yield "glass+emissive b0", s, make_params(64, 64, bounces=0, seed=11), s.camera, 3
    yield "glass+emissive b7", s, make_params(32, 32, bounces=7, seed=13), s.camera, 2
    yield "glass+emissive b-1", s, make_params(16, 16, bounces=-1, seed=13), s.camera, 1
    s2 = scenes.bunny_scene(small=True)
    yield "plain b2", s2, make_params(96, 64, bounces=2, seed=5), s2.camera, 4
    s3 = scenes.bunny_scene(small=True, glass=True, emissive=True)
    s3.lights = np.stack([s3.lights[0], scenes.point_light((1.5, 3.0, 2.0), 40.0, (1.0, 0.8, 0.6))]).astype(np.float32)
    s3.sky = scenes.sky_solid((0.3, 0.4, 0.5))
    yield "point light, solid sky, ambient, marching", s3, make_params(48, 48, bounces=2, ambient=0.05, max_steps=40, step_size=0.5, seed=3), s3.camera, 2
    s4 = scenes.bunny_scene(small=True, glass=True, emissive=True)
    s4.sky = scenes.sky_none()
    yield "no sky", s4, make_params(48, 48, bounces=2, seed=4), s4.camera, 3
    s4b = scenes.bunny_scene(small=True)
    s4b.sky = scenes.sky_none()
    s4b.lights = s4b.lights[:0]
    yield "no sky, no lights, no emitters", s4b, make_params(32, 32, bounces=2, seed=4), s4b.camera, 2
    s5 = scenes.textured_scene(small=True)
    yield "textures + bump", s5, make_params(64, 64, bounces=2, seed=7), s5.camera, 3
    s6 = scenes.c4_scene(blob_level=4, terrain_n=96)
    yield "c4 class", s6, make_params(64, 48, bounces=2, seed=17), s6.camera, 3
    s7 = scenes.blackhole_scene(segments=96)
    yield "black hole", s7, make_params(32, 32, bounces=1, max_steps=3000, step_size=1.0, seed=2), s7.camera, 1
    lights = [s.lights[0]]
    for k in range(8):
        t = 0.7 * k
        lights.append(scenes.point_light((2.5 * np.cos(t), 2.0 + 0.2 * k, 2.5 * np.sin(t)), 12.0, (1.0, 0.6 + 0.05 * k, 0.5)))
    s8 = scenes.bunny_scene(small=True, glass=True, emissive=True)
    s8.lights = np.stack(lights).astype(np.float32)
    yield "9 lights", s8, make_params(48, 48, bounces=2, seed=8), s8.camera, 2
    s9 = scenes.bunny_scene(blob_level=5, glass=True, emissive=True)
    yield "c3 class 480x270 (vote kernels)", s9, make_params(480, 270, bounces=2, seed=1), s9.camera, 8
    yield "c3 class 480x270 b4", s9, make_params(480, 270, bounces=4, seed=1), s9.camera, 2


lines = []
ctx = Context(0)
for name, sc, p, cam, spp in cases():
    r = Renderer(sc, ctx=ctx)
    got = r.render_samples(p, cam, spp=spp)
    h = hashlib.sha256(np.ascontiguousarray(got).view(np.uint32).tobytes()).hexdigest()[:16]
    line = f"{name:45s} {h} mean {float(np.nan_to_num(got).mean()):.6f}"
    print(line, flush=True)
    lines.append(line)
ctx.close()
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")
