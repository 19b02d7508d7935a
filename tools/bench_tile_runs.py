"""ptmi_render_tiles_stats_dev on tile lists of very different granularity (what an adaptive pass hands down): the whole
frame as rows, a random half of the pixels as 1-row runs, a random half of the 8x8 blocks.  usage: tools/bench_tile_runs.py"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from pathtrace_amd import adaptive, scenes
from pathtrace_amd.render import Renderer, make_params
sc = scenes.bunny_scene(glass=True, emissive=True)
W, H, spp = 1920, 1080, 64
p = make_params(W, H, bounces=2)
r = Renderer(sc)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
for name, mask in (("all", np.ones((H, W), bool)), ("random 50%", rng.random((H, W)) < 0.5), ("blocks 8x8 50%", np.kron(rng.random((H // 8, W // 8)) < 0.5, np.ones((8, 8), bool)))):
    t0 = time.time(); tiles = adaptive.mask_runs(mask); t_runs = time.time() - t0
    for rep in range(2):
        acc = torch.zeros((H, W, 3), dtype=torch.float32, device=dev); cnt = torch.zeros((H, W), dtype=torch.int32, device=dev)
        st = torch.zeros((H, W, 9), dtype=torch.float32, device=dev)
        torch.cuda.synchronize(); t0 = time.time()
        r.render_tiles_device(acc, cnt, p, sc.camera, tiles, spp=spp, stats=st)
        torch.cuda.synchronize(); dt = time.time() - t0
    n = int(mask.sum()) * spp
    print(f"{name}: {len(tiles)} runs (mask_runs {t_runs*1e3:.1f} ms), {n} samples in {dt*1e3:.1f} ms = {n/dt/1e6:.1f} Msamples/s")
