#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point (ptmi_closest_hit: rays in host memory, results back to host
memory) on the C2 workload, next to the device-resident rate bench.py reports as `value`.
usage: tools/bench_host_buffers.py [--tris 1000000] [--rays 16777216]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--tris", type=int, default=1_000_000); ap.add_argument("--rays", type=int, default=1 << 24)
a = ap.parse_args()
dev = torch.device("cuda", 0)
wl = bench.build_workload(a.tris, a.rays, 1000.0, 0, torch, dev, "C2")
o, d, n = wl["o_host"], wl["d_host"], wl["n"]
bvh = wl["bvh"]
hit, t, prim = bvh.CheckIntersection((o, d), 1000.0)          # warm-up: staging buffers are allocated here
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); hit, t, prim = bvh.CheckIntersection((o, d), 1000.0); best = min(best, time.perf_counter() - t0)
bench.extend_step(wl, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
same = bool(np.array_equal(prim, wl["P"].cpu().numpy()) and np.array_equal(t.view(np.uint32), wl["t"].cpu().numpy().view(np.uint32)))
print(f"pageable host buffers: {n} rays in {best * 1e3:.1f} ms = {n / best / 1e6:.1f} Mrays/s PCIe-inclusive (40 B per ray each way over the link: "
      f"{n * 40 / best / 1e9:.1f} GB/s); results equal the device-resident call's: {same}")
# the same call with the rays in page-locked host memory (what a host that cares about the link would hand over)
op, dp = torch.from_numpy(o).pin_memory().numpy(), torch.from_numpy(d).pin_memory().numpy()
bvh.CheckIntersection((op, dp), 1000.0)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); hit2, t2, prim2 = bvh.CheckIntersection((op, dp), 1000.0); best = min(best, time.perf_counter() - t0)
print(f"page-locked ray buffers (results still into pageable arrays): {best * 1e3:.1f} ms = {n / best / 1e6:.1f} Mrays/s; equal: "
      f"{bool(np.array_equal(prim2, prim))}")
