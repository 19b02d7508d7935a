#!/usr/bin/env python3
"""The one leg on which "fraction of the HBM roofline" can be read off: closest-hit on a tree that does not fit the
Infinity Cache (10 M triangles, 0.67 GB packed) with INCOHERENT rays -- origins uniform in the bounds, directions uniform on
the sphere, batch order random -- so that neighbouring lanes and waves touch unrelated subtrees and only the top of the
tree stays cached.  Prints Mrays/s, the counted work (N_inner, N_tri) and the algorithmic GB/s; under `rocprofv3 --pmc`
(tools/pmc_incoherent.sh) the same launches give the fabric bytes and the L2 hit rate.
usage: tools/bench_incoherent.py [--tris N] [--rays N] [--steps K]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from pathtrace_amd import BuildBVH, ConstructLinearBVH, Context, scenes
from helpers import random_rays

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=10_000_000); ap.add_argument("--rays", type=int, default=1 << 24)
ap.add_argument("--steps", type=int, default=5); ap.add_argument("--tmax", type=float, default=1000.0)
a = ap.parse_args()
t0 = time.time()
verts, tris = scenes.make_soup(a.tris, seed=42)
box = BuildBVH(verts, tris); st = box.GetStats()
ctx = Context(0); bvh = ConstructLinearBVH(box, ctx=ctx)
o, d = random_rays(verts, a.rays, 77)
print(f"mesh {tris.shape[0]} tris, BVH depth {st.max_depth}, packed {(st.packed_inner_nodes * 64 + tris.shape[0] * 48) / 1e6:.0f} MB, "
      f"{a.rays} incoherent rays (set-up {time.time() - t0:.1f} s)", flush=True)
dev = torch.device("cuda", 0)
O, D = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
T = torch.full((a.rays,), a.tmax, dtype=torch.float32, device=dev)
P = torch.empty(a.rays, dtype=torch.int32, device=dev); t = torch.empty(a.rays, dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
bvh.set_counting(True); bvh.reset_counters(); bvh.CheckIntersectionDevice(O, D, T, P, t, s); c = bvh.counters(); bvh.set_counting(False)
alg = 64 * c.inner_visits + 48 * c.tri_tests + 40 * a.rays
bvh.CheckIntersectionDevice(O, D, T, P, t, s); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps):
    bvh.CheckIntersectionDevice(O, D, T, P, t, s)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.steps
print(f"extend_vote_kernel: {ms:.3f} ms per launch = {a.rays / ms / 1e3:.1f} Mrays/s; hit rate {(P >= 0).float().mean().item():.3f}; "
      f"per ray {c.inner_visits / a.rays:.2f} inner nodes, {c.tri_tests / a.rays:.2f} triangles = {alg / a.rays:.0f} B; "
      f"algorithmic {alg / 1e9:.2f} GB per launch = {alg / ms / 1e6:.0f} GB/s = {alg / ms / 1e6 / 8000:.3f} of the 8 TB/s HBM peak")
ctx.close()
