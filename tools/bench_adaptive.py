"""Adaptive sampling (pathtrace_amd/adaptive.py; tile.go:34-66) on C3: samples/s of the second run (the first one pays for
the pool and the output buffer), the two-call form against one call per batch, and a fixed-count frame for scale.
usage: tools/bench_adaptive.py [--width 1920 --height 1080 --spp 32 --max-spp 256]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

from pathtrace_amd import adaptive, scenes
from pathtrace_amd.render import Renderer, make_params

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=32); ap.add_argument("--max-spp", type=int, default=256)
a = ap.parse_args()
sc = scenes.bunny_scene(glass=True, emissive=True)
p = make_params(a.width, a.height, bounces=2)
r = Renderer(sc)
for merge in (True, False):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        acc, cnt, lum, calls = adaptive.render_adaptive(r, p, sc.camera, samples_per_pixel=a.spp, max_samples=a.max_spp, merge=merge)
        torch.cuda.synchronize(); dt = time.time() - t0
    n = int(cnt.sum().item())
    print(f"{'two calls' if merge else 'one call per batch'}: {n} samples ({n / cnt.numel():.1f} per pixel, "
          f"{int(cnt.min().item())}..{int(cnt.max().item())}) in {len(calls)} calls, {dt * 1e3:.1f} ms = {n / dt / 1e6:.1f} Msamples/s")
for rep in range(2):
    acc.zero_(); cnt.zero_(); lum.zero_()
    torch.cuda.synchronize(); t0 = time.time()
    r.render_adaptive_device(acc, cnt, lum, p, sc.camera, samples_per_pixel=a.spp, max_samples=a.max_spp)
    torch.cuda.synchronize(); dt = time.time() - t0
n = int(cnt.sum().item())
print(f"ptmi_render_adaptive_dev: {n} samples in {dt * 1e3:.1f} ms = {n / dt / 1e6:.1f} Msamples/s")
dev = acc.device
acc = torch.zeros_like(acc); cnt = torch.zeros_like(cnt)
for rep in range(2):
    acc.zero_(); cnt.zero_()
    torch.cuda.synchronize(); t0 = time.time()
    r.render_device(acc, cnt, p, sc.camera, spp=a.max_spp)
    torch.cuda.synchronize(); dt = time.time() - t0
n = a.width * a.height * a.max_spp
print(f"fixed {a.max_spp} spp: {n} samples in {dt * 1e3:.1f} ms = {n / dt / 1e6:.1f} Msamples/s")
