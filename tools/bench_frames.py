#!/usr/bin/env python3
"""An orbit of F frames (the reference's frame loop, main.go:903-937): one ptmi_render_dev call per frame against all
frames in one ptmi_render_frames_dev call.  usage: tools/bench_frames.py [--scene c3|c1] [--frames F] [--spp S] [--width W --height H]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pathtrace_amd import scenes, driver
from pathtrace_amd.render import Renderer, make_params

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="c3"); ap.add_argument("--frames", type=int, default=8); ap.add_argument("--spp", type=int, default=32)
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
sc = scenes.bunny_scene(blob_level=6, glass=a.scene == "c3", emissive=a.scene == "c3")
p = make_params(a.width, a.height, bounces=2, seed=1)
r = Renderer(sc); dev = torch.device("cuda", 0)
radius = float(np.linalg.norm(np.asarray(sc.camera.position)))
cams = [driver.spherical_around((0.0, 0.0, 0.0), radius, (90.0 + k) * driver.DEG, 83.0 * driver.DEG, sc.camera.frustum) for k in range(a.frames)]
acc = torch.zeros((a.frames, a.height, a.width, 3), dtype=torch.float32, device=dev)
cnt = torch.zeros((a.frames, a.height, a.width), dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
n = a.frames * a.width * a.height * a.spp
for name, run in (("one call per frame", lambda: [r.render_device(acc[f], cnt[f], p, cams[f], spp=a.spp, stream=s) for f in range(a.frames)]),
                  ("all frames in one call", lambda: r.render_frames_device(acc, cnt, p, cams, spp=a.spp, stream=s))):
    run(); acc.zero_(); cnt.zero_(); torch.cuda.synchronize()
    t0 = time.time(); run(); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"{name}: {a.frames} frames {a.width}x{a.height} at {a.spp} spp in {dt * 1e3:.1f} ms = {dt * 1e3 / a.frames:.1f} ms per frame, {n / dt / 1e6:.0f} Msamples/s", flush=True)
