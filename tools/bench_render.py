#!/usr/bin/env python3
"""Time the wavefront integrator on a named scene (samples/s) and optionally dump a PNG.
usage: tools/bench_render.py [--scene c3|c1|c4|c5] [--width W --height H --spp S] [--png path]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from pathtrace_amd import scenes
from pathtrace_amd.render import Renderer, make_params

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="c3"); ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080); ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--bounces", type=int, default=2); ap.add_argument("--png", default=None)
ap.add_argument("--level", type=int, default=6)
ap.add_argument("--profile", action="store_true", help="per-kernel HIP-event times of the timed frame (ptmi_set_profiling: one lane)")
a = ap.parse_args()
if a.scene == "c4":
    sc = scenes.c4_scene(); p = make_params(a.width, a.height, bounces=a.bounces, seed=1)
elif a.scene == "c5":
    sc = scenes.blackhole_scene(); p = make_params(a.width, a.height, bounces=a.bounces, max_steps=3000, step_size=1.0, seed=1)
else:
    sc = scenes.bunny_scene(blob_level=a.level, glass=a.scene == "c3", emissive=a.scene == "c3")
    p = make_params(a.width, a.height, bounces=a.bounces, seed=1)
t0 = time.time(); r = Renderer(sc); print(f"scene {a.scene}: {sc.tris.shape[0]} tris, set-up {time.time()-t0:.2f}s", flush=True)
dev = torch.device("cuda", 0)
acc = torch.zeros((a.height, a.width, 3), dtype=torch.float32, device=dev); cnt = torch.zeros((a.height, a.width), dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
r.render_device(acc, cnt, p, sc.camera, spp=a.spp, stream=stream)  # warm-up at full size: the pool and the sample buffer are allocated here
acc.zero_(); cnt.zero_(); r.bvh.reset_counters(); torch.cuda.synchronize()
if a.profile: r.set_profiling(True)
t0 = time.time(); r.render_device(acc, cnt, p, sc.camera, spp=a.spp, stream=stream); torch.cuda.synchronize(); dt = time.time() - t0
c = r.bvh.counters(); n = a.width * a.height * a.spp
print(f"{n} samples in {dt*1e3:.1f} ms = {n/dt/1e6:.2f} Msamples/s; closest {c.closest_queries/dt/1e6:.1f} Mq/s, any {c.any_queries/dt/1e6:.1f} Mq/s, "
      f"diffuse events {c.diffuse_events/dt/1e6:.1f} M/s; per sample: {c.closest_queries/n:.2f} closest, {c.any_queries/n:.2f} any")
if a.profile:
    pr = r.profile(); r.set_profiling(False)
    print("profile (ms): " + ", ".join(f"{k[:-3]} {pr[k]:.2f}" for k in pr if k.endswith("_ms")) +
          f"; sum {sum(pr[k] for k in pr if k.endswith('_ms')):.2f} of {dt*1e3:.1f} wall; {pr['iterations']} iterations, {pr['launches']} launches")
if a.png:
    from write_png import write_png
    from pathtrace_amd.render import Framebuffer
    fb = Framebuffer(acc.cpu().numpy(), cnt.cpu().numpy().astype(np.uint32))
    write_png(a.png, r.to_rgba(fb))
