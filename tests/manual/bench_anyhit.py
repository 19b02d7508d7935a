#!/usr/bin/env python3
"""Any-hit (QuickCheckIntersection) rate on the bench's ray batch, with an oracle check on a strided sample.
usage: tests/manual/bench_anyhit.py [--tris N] [--rays N]    (PTMI_EXTEND_VARIANT=0 selects the one-shot kernel)"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=1_000_000); ap.add_argument("--rays", type=int, default=1 << 24)
ap.add_argument("--tmax", type=float, default=1000.0); ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
wl = bench.build_workload(a, 0, torch, dev)
n = wl["n"]
occ = torch.empty(n, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
wl["bvh"].QuickCheckIntersectionDevice(wl["O"], wl["D"], wl["T"], occ, stream); torch.cuda.synchronize()
t0 = time.time()
for _ in range(a.steps):
    wl["bvh"].QuickCheckIntersectionDevice(wl["O"], wl["D"], wl["T"], occ, stream)
torch.cuda.synchronize(); dt = (time.time() - t0) / a.steps
import oracle_lib
orc = oracle_lib.OracleBVH(wl["verts"], wl["tris"], linear=wl["box"].linear())
idx = np.arange(0, n, max(1, n // (1 << 20)))
want = orc.any_hit(np.ascontiguousarray(wl["o_host"][idx]), np.ascontiguousarray(wl["d_host"][idx]), a.tmax, nthreads=16, counters=False)[0]
got = occ.cpu().numpy()[idx]
print(f"any-hit: {n / dt / 1e6:.1f} Mrays/s ({dt * 1e3:.2f} ms per {n} rays), occluded {got.mean():.3f}, mismatches vs oracle on {len(idx)} rays: {int((got != want.astype(np.uint8)).sum())}")
