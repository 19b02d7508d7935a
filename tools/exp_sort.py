#!/usr/bin/env python3
"""Experiment: how much does ray ordering matter for the extend kernel?  Times the kernel on the bench
batch in its original order and under several host-side orderings (the sort itself is NOT timed here;
this only bounds what a device-side binning pass could win).  usage: python tools/exp_sort.py"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def part1by2(x):
    x = x.astype(np.uint64) & 0x3FF
    x = (x | (x << 16)) & 0x30000FF
    x = (x | (x << 8)) & 0x300F00F
    x = (x | (x << 4)) & 0x30C30C3
    x = (x | (x << 2)) & 0x9249249
    return x


def morton3(q):
    return part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)


def keys(o, d, obits, dbits, lo, hi):
    qo = np.clip(((o - lo) / (hi - lo) * (1 << obits)).astype(np.int64), 0, (1 << obits) - 1)
    qd = np.clip(((d * 0.5 + 0.5) * (1 << dbits)).astype(np.int64), 0, (1 << dbits) - 1)
    return (morton3(qd) << np.uint64(3 * obits)) | morton3(qo), (morton3(qo) << np.uint64(3 * dbits)) | morton3(qd)


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--tmax", type=float, default=1000.0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    wl = bench.build_workload(args, 0, torch, dev)
    bvh, n = wl["bvh"], wl["n"]
    o, d = wl["o_host"], wl["d_host"]
    stream = torch.cuda.current_stream().cuda_stream
    lo, hi = wl["verts"].min(0), wl["verts"].max(0)

    def time_order(name, perm):
        oo = np.ascontiguousarray(o[perm]) if perm is not None else o
        dd = np.ascontiguousarray(d[perm]) if perm is not None else d
        O, D = torch.from_numpy(oo).to(dev), torch.from_numpy(dd).to(dev)
        m = O.shape[0]
        T = torch.full((m,), args.tmax, dtype=torch.float32, device=dev)
        P = torch.empty(m, dtype=torch.int32, device=dev)
        t = torch.empty(m, dtype=torch.float32, device=dev)
        for _ in range(2):
            bvh.CheckIntersectionDevice(O, D, T, P, t, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            bvh.CheckIntersectionDevice(O, D, T, P, t, stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{name:40s} {m:9d} rays  {ms:8.3f} ms  {m / ms / 1e3:8.1f} Mrays/s", flush=True)

    half = n // 2
    time_order("original (all)", None)
    time_order("primary only", np.arange(half))
    time_order("secondary only", np.arange(half, n))
    w, h = wl["grid"]
    yy, xx = np.divmod(np.arange(half), w)
    tile = ((yy // 8) * (w // 8) + (xx // 8)) * 64 + (yy % 8) * 8 + (xx % 8)
    time_order("primary 8x8 tiles", np.argsort(tile, kind="stable"))
    rng = np.random.default_rng(0)
    time_order("random shuffle (all)", rng.permutation(n))
    for ob, db in ((5, 1), (6, 2), (8, 2), (10, 3)):
        kd, ko = keys(o, d, ob, db, lo, hi)
        time_order(f"sort dir-major o{ob} d{db} (all)", np.argsort(kd, kind="stable"))
        time_order(f"sort origin-major o{ob} d{db} (all)", np.argsort(ko, kind="stable"))
    kd, ko = keys(o[half:], d[half:], 6, 2, lo, hi)
    time_order("secondary sort dir-major o6 d2", half + np.argsort(kd, kind="stable"))
    time_order("secondary sort origin-major o6 d2", half + np.argsort(ko, kind="stable"))


if __name__ == "__main__":
    main()
