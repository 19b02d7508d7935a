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
# the same batch straight through the C ABI with all five buffers page-locked (what a host that cares about the link hands over)
import ctypes as C
from pathtrace_amd import _lib
lib = _lib.load()
pin = lambda x: torch.from_numpy(np.ascontiguousarray(x)).pin_memory()
O, D, T = pin(o), pin(d), pin(np.full(n, 1000.0, np.float32))
P, Tt = torch.empty(n, dtype=torch.int32).pin_memory(), torch.empty(n, dtype=torch.float32).pin_memory()
fp = lambda x: C.cast(x.data_ptr(), C.POINTER(C.c_float))
call = lambda: _lib.check(lib.ptmi_closest_hit(bvh.ctx.handle, fp(O), fp(D), fp(T), n, C.cast(P.data_ptr(), C.POINTER(C.c_int32)), fp(Tt)),
                          bvh.ctx.handle)
call()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); call(); best = min(best, time.perf_counter() - t0)
print(f"C ABI, all buffers page-locked: {best * 1e3:.1f} ms = {n / best / 1e6:.1f} Mrays/s ({n * 40 / best / 1e9:.1f} GB/s over the link); equal: "
      f"{bool(np.array_equal(P.numpy(), prim) and np.array_equal(Tt.numpy().view(np.uint32), t.view(np.uint32)))}")
