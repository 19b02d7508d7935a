#!/usr/bin/env python3
"""Cost of call granularity: the C3 frame at 1080p as one call, as the reference's 4x4 split (one call per tile, and
all 16 tiles in one ptmi_render_tiles_dev call) and as 64x64 tiles.  usage: tools/bench_tiles.py [--spp 64]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathtrace_amd import scenes, sharded
from pathtrace_amd.render import Renderer, make_params

ap = argparse.ArgumentParser(); ap.add_argument("--spp", type=int, default=64); a = ap.parse_args()
sc = scenes.bunny_scene(blob_level=6, glass=True, emissive=True)
W, H, spp = 1920, 1080, a.spp
p = make_params(W, H, bounces=2, seed=1)
r = Renderer(sc)
dev = torch.device("cuda", 0)
acc = torch.zeros((H, W, 3), dtype=torch.float32, device=dev); cnt = torch.zeros((H, W), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
rr = sharded.make_device_render_rect(r, p, sc.camera, st)
rr(acc, cnt, (0, 0, W, H), 0, spp); torch.cuda.synchronize()   # warm-up at full size: pool and sample buffer allocation
rr(acc, cnt, (8, 8, 16, 16), 0, 1); torch.cuda.synchronize()    # ... and torch's strided add of a sub-rectangle (its first use loads a code object: ~55 ms)
split = [(x, y, 480, 270) for y in range(0, H, 270) for x in range(0, W, 480)]
small = sharded.tile_rects(W, H, 64)
for name, calls in (("whole frame, one call", [[(0, 0, W, H)]]), ("4x4 split, one call per tile", [[t] for t in split]),
                    ("4x4 split, one call for the 16 tiles", [split]), ("64x64 tiles, one call per tile", [[t] for t in small]),
                    ("64x64 tiles, one call for the 510 tiles", [small])):
    acc.zero_(); cnt.zero_(); torch.cuda.synchronize(); t0 = time.time()
    per_call = []
    for tiles in calls:
        t1 = time.time()
        if len(tiles) == 1: rr(acc, cnt, tiles[0], 0, spp)
        else: r.render_tiles_device(acc, cnt, p, sc.camera, tiles, spp=spp, stream=st)
        per_call.append(time.time() - t1)           # the render calls return when the region is done
    torch.cuda.synchronize(); dt = time.time() - t0
    assert bool((cnt == spp).all())
    if 1 < len(per_call) <= 16: print("   per call, in order (ms):", " ".join(f"{t * 1e3:.1f}" for t in per_call))
    per_call.sort()
    print(f"{name}: {len(calls)} calls, {W * H * spp / dt / 1e6:.1f} Msamples/s; per call min / median / max "
          f"{per_call[0] * 1e3:.2f} / {per_call[len(per_call) // 2] * 1e3:.2f} / {per_call[-1] * 1e3:.2f} ms")
