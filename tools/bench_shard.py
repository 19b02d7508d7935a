#!/usr/bin/env python3
"""What one GPU of N does for the C3 frame (1920x1080, 256 spp): the time of its row-interleaved share, shard 0 of N,
in one ptmi_render_rows_dev call -- the strong-scaling ceiling of the render leg before the framebuffer reduce.
usage: tools/bench_shard.py [--spp 256]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathtrace_amd import scenes
from pathtrace_amd.render import Renderer, make_params

ap = argparse.ArgumentParser(); ap.add_argument("--spp", type=int, default=256); a = ap.parse_args()
sc = scenes.bunny_scene(blob_level=6, glass=True, emissive=True)
W, H = 1920, 1080
p = make_params(W, H, bounces=2, seed=1)
r = Renderer(sc)
dev = torch.device("cuda", 0)
acc = torch.zeros((H, W, 3), dtype=torch.float32, device=dev); cnt = torch.zeros((H, W), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
t1 = None
for n in (1, 2, 4, 8):
    r.render_rows_device(acc, cnt, p, sc.camera, row_block=8, shard=0, nshards=n, spp=a.spp, stream=st)   # warm-up at size
    acc.zero_(); cnt.zero_(); torch.cuda.synchronize(); t0 = time.time()
    r.render_rows_device(acc, cnt, p, sc.camera, row_block=8, shard=0, nshards=n, spp=a.spp, stream=st)
    torch.cuda.synchronize(); dt = time.time() - t0
    t1 = t1 or dt
    print(f"shard 0 of {n}: {dt * 1e3:.1f} ms -> frame rate of {n} GPUs {W * H * a.spp / dt / 1e6:.0f} Msamples/s, efficiency {t1 / (n * dt):.3f}")
