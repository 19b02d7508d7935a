#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json): closest-hit Mrays/s.

Workload at N=1 (BASELINE.json configs[1], SURVEY.md 8(d) "C2"): a 16 M-ray primary+secondary
closest-hit batch on a 1 M-triangle mesh, rays resident in HBM when the timed region starts.
A "step" is one pass of the `extend` kernel (LinearBVH.CheckIntersection, linear_bvh.go:116)
over the whole batch.  With --gpus N each rank holds a replica of the BVH and traces its own
16 M-ray batch (weak scaling, no data-path collective); `value` = all rays of all ranks divided
by the slowest rank's time.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Besides the contract's keys it carries

  roofline              the extend kernel on the headline batch: ALGORITHMIC bytes (64*N_inner + 48*N_tri + 40*R per
                        launch, N_* counted by the instrumented kernel variant) / the kernel's duration (HIP events on
                        the launch stream, inside this run), against the ceiling that binds it -- the CU's L1 gather
                        rate -- with the HBM figures beside it: `hbm_algorithmic_frac` (the same bytes / 8 TB/s; above
                        1 only says the tree is served from cache) and `hbm_frac_counter` (fabric bytes measured in
                        this run with rocprofv3 PMC passes over a child process / time / 8 TB/s); `traffic` = those
                        fabric bytes per launch.
  roofline_hbm_resident the same for a second leg whose tree (10 M triangles, 0.67 GB) does not fit the 256 MiB
                        Infinity Cache (N=1 only).
  cpu_baseline          the C restatement of the Go reference (the Go code cannot be built here) on the host cores:
                        CheckIntersection single-threaded and on all cores of this job's CPU share, on a bounded
                        sample of the same batch; and the goroutine tile renderer (one worker thread per tile,
                        main.go:454-491, :832-899) on C1 with threadCount = 16 and = the CPU share.
  render, render_bounces4, render_c4, render_c5 (and render_orbit: 8 frames of C3 at 32 spp in one ptmi_render_frames_dev call)
                        samples/s of the wavefront integrator on BASELINE.json configs[2] (C3; also with bounces = 4),
                        configs[3] (C4: 10 M triangles, 4K, 64x64 tiles) and configs[4] (C5: black hole, 4K), at every N: with
                        N > 1 the frame is sharded over the ranks and reduced with RCCL through the C ABI ("comm": "rccl";
                        anything else, a failed rank or a hang makes the run exit non-zero after the line is printed).  At
                        N = 1 each carries `roofline`: algorithmic bytes per kernel class (instrumented traversal kernels) /
                        that class's time (the library's own per-kernel events).
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# The ceiling the extend kernel runs against: divergent 4 x dwordx4 gathers of random 64-byte records from an
# L1-resident table arrive at 22.6 B/clk/CU = 13.9 TB/s chip-wide, the same whether the lanes of a quad share a line or
# not (tools/ubench_gather.hip; profiles/r01b/ubench_gather.txt, re-measured in profiles/r02a/).
L1_GATHER_PEAK_GBS = 13900.0
NODE_BYTES, TRI_BYTES, RAY_BYTES = 64, 48, 40


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_share():
    """Host threads this job may use: the box exports its CPU share as OMP_NUM_THREADS (os.cpu_count() there shows the
    whole machine)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    v = os.environ.get("OMP_NUM_THREADS", "")
    if v.isdigit() and int(v) > 0:
        n = min(n, int(v))
    return max(1, n)


class _HostStagedDist:
    """torch.distributed over gloo for the rehearsal mode: gloo has no CUDA `reduce`, so tensors go through the host."""

    def __init__(self, dist):
        self._d = dist
        self.ReduceOp = dist.ReduceOp

    def barrier(self):
        self._d.barrier()

    def destroy_process_group(self):
        self._d.destroy_process_group()

    def all_reduce(self, t, op=None):
        h = t.cpu()
        self._d.all_reduce(h, op=op)
        t.copy_(h)

    def broadcast(self, t, src=0):
        h = t.cpu()
        self._d.broadcast(h, src=src)
        t.copy_(h)

    def reduce(self, t, dst=0, op=None, group=None):
        h = t.cpu()
        self._d.reduce(h, dst=dst, op=op, group=group)
        t.copy_(h)


# ------------------------------------------------------------------------------------------------ workload
def build_workload(ntris, nrays, tmax, rank, torch, dev, label, ctx=None):
    """Mesh + BVH + ray batch, all untimed set-up.  Returns dict with device tensors."""
    from pathtrace_amd import BuildBVH, ConstructLinearBVH, Context, scenes

    t0 = time.time()
    verts, tris = scenes.make_soup(ntris, seed=42)
    box = BuildBVH(verts, tris)
    st = box.GetStats()
    ctx = ctx or Context(dev.index)
    bvh = ConstructLinearBVH(box, ctx=ctx)
    t1 = time.time()
    log(f"[rank {rank}] {label}: mesh {tris.shape[0]} tris, BVH {st.linear_nodes} nodes depth {st.max_depth} "
        f"built in {st.build_ms:.0f} ms (set-up {t1 - t0:.1f} s)")

    n_primary = nrays // 2
    w = 1
    while w * w // 2 < n_primary:
        w *= 2
    h = n_primary // w
    n_primary = w * h
    cam = scenes.soup_camera(verts)
    if rank:   # each rank traces its own batch (weak scaling): move the camera around the scene
        p = np.asarray(cam.position, np.float32)
        c, s = np.cos(0.7 * rank), np.sin(0.7 * rank)
        cam = scenes.look_at((float(c * p[0] - s * p[2]), float(p[1]), float(s * p[0] + c * p[2])), (0.0, 0.5, 0.0))
    o1, d1 = scenes.camera_rays(cam, w, h)
    stream = torch.cuda.current_stream().cuda_stream
    O1 = torch.from_numpy(o1).to(dev)
    D1 = torch.from_numpy(d1).to(dev)
    T1 = torch.full((n_primary,), tmax, dtype=torch.float32, device=dev)
    P1 = torch.empty(n_primary, dtype=torch.int32, device=dev)
    t1_ = torch.empty(n_primary, dtype=torch.float32, device=dev)
    bvh.CheckIntersectionDevice(O1, D1, T1, P1, t1_, stream)
    torch.cuda.synchronize()
    prim = P1.cpu().numpy()
    tt = t1_.cpu().numpy()
    o2, d2 = scenes.cosine_bounce_rays(verts, tris, o1, d1, prim, tt, seed=42 + rank)
    o = np.concatenate([o1, o2], 0)
    d = np.concatenate([d1, d2], 0)
    n = o.shape[0]
    del O1, D1, T1, P1, t1_
    packed_mb = (st.packed_inner_nodes * NODE_BYTES + tris.shape[0] * TRI_BYTES) / 1e6
    wl = dict(ctx=ctx, bvh=bvh, box=box, verts=verts, tris=tris, n=n, o_host=o, d_host=d, tmax=tmax,
              O=torch.from_numpy(o).to(dev), D=torch.from_numpy(d).to(dev),
              T=torch.full((n,), tmax, dtype=torch.float32, device=dev),
              P=torch.empty(n, dtype=torch.int32, device=dev), t=torch.empty(n, dtype=torch.float32, device=dev),
              primary_hit_rate=float((prim >= 0).mean()), stats=st, grid=(w, h), packed_mb=packed_mb)
    log(f"[rank {rank}] {label}: rays {n} ({w}x{h} primary, hit rate {wl['primary_hit_rate']:.3f}), packed tree "
        f"{packed_mb:.0f} MB (set-up {time.time() - t0:.1f} s)")
    return wl


def extend_step(wl, stream):
    wl["bvh"].CheckIntersectionDevice(wl["O"], wl["D"], wl["T"], wl["P"], wl["t"], stream)


def count_work(wl, stream):
    """Algorithmic work of one launch, from the instrumented kernel variant (untimed)."""
    bvh = wl["bvh"]
    bvh.set_counting(True)
    bvh.reset_counters()
    extend_step(wl, stream)
    c = bvh.counters()
    bvh.set_counting(False)
    alg = NODE_BYTES * c.inner_visits + TRI_BYTES * c.tri_tests + RAY_BYTES * wl["n"]
    return c, int(alg)


def time_extend(wl, steps, warmup, torch, dist, stream):
    """`warmup` untimed launches, then exactly `steps` launches between barrier + synchronize on both sides.
    Returns (wall seconds, mean kernel ms from HIP events recorded on the launch stream)."""
    for _ in range(warmup):
        extend_step(wl, stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                                   # torch's current stream == `stream`, the one the kernel is launched on
    for _ in range(steps):
        extend_step(wl, stream)
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return wall, ev0.elapsed_time(ev1) / steps


def roofline_block(wl, c, alg_bytes, kernel_ms, traffic, note):
    n = wl["n"]
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    out = {"bound": "l1_gather", "achieved": round(achieved, 2), "peak": L1_GATHER_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / L1_GATHER_PEAK_GBS, 4),
           "traffic": None if not traffic else int(traffic["hbm_bytes_per_launch"]),
           "peak_source": "measured: divergent 4 x dwordx4 gathers of random 64-B records from an L1-resident table, "
                          "22.6 B/clk/CU x 256 CUs x 2.4 GHz (tools/ubench_gather.hip, profiles/r02a/ubench_gather.txt)",
           "hbm_peak": HBM_PEAK_GBS, "hbm_algorithmic_frac": round(achieved / HBM_PEAK_GBS, 4),
           "hbm_frac_counter": None,
           "kernel": f"extend_vote_kernel<{wl['stack']},false>", "kernel_ms": round(kernel_ms, 4),
           "algorithmic_bytes_per_launch": int(alg_bytes),
           "inner_visits_per_ray": round(c.inner_visits / n, 3), "tri_tests_per_ray": round(c.tri_tests / n, 3),
           "packed_tree_MB": round(wl["packed_mb"], 1), "note": note}
    if traffic:
        gbs = traffic["hbm_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9
        out["hbm_frac_counter"] = round(gbs / HBM_PEAK_GBS, 4)
        out["hbm_frac_counter_lower"] = round(traffic["hbm_bytes_per_launch_lower"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out["traffic_GBs"] = round(gbs, 1)
        out["traffic_over_algorithmic"] = round(traffic["hbm_bytes_per_launch"] / alg_bytes, 4)
        out["traffic_detail"] = traffic
    return out


def stack_class(depth):
    """Stack size of the kernel instantiation a tree of this depth runs (ptmi_ctx::stack_entries: depth - 1 entries)."""
    for s in (16, 20, 24, 28, 32, 48):
        if depth - 1 <= s:
            return s
    return 64


# ------------------------------------------------------------------------------------------------ live PMC traffic
def pmc_child(args):
    """Child of measure_traffic(), run under `rocprofv3 --pmc`: the bench's set-up and `steps` full-batch launches per
    leg, nothing else.  The extend launches happen in a fixed order the parent relies on:
    per leg one primary-rays launch (set-up), then `steps` full-batch launches."""
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream().cuda_stream
    legs = [("c2", args.tris)] + ([("hbm", args.hbm_tris)] if not args.no_hbm_leg else [])
    for label, ntris in legs:
        wl = build_workload(ntris, args.rays, args.tmax, 0, torch, dev, f"pmc-child {label}")
        for _ in range(args.steps):
            extend_step(wl, stream)
        torch.cuda.synchronize()
        wl["ctx"].close()
        del wl


def _parse_pmc(outdir, nlegs, steps):
    """Per counter: the mean value of the `steps` full-batch extend launches of every leg (a dispatch's rows are
    summed: rocprofv3 writes one row per dispatch, counter and instance)."""
    paths = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    if not paths:
        raise RuntimeError("no counter_collection.csv written")
    acc = {}
    for path in paths:
        for row in csv.DictReader(open(path)):
            if "extend_vote_kernel" not in row["Kernel_Name"]:
                continue
            d = acc.setdefault(row["Counter_Name"], {})
            k = int(row["Dispatch_Id"])
            d[k] = d.get(k, 0.0) + float(row["Counter_Value"])
    out = [dict() for _ in range(nlegs)]
    for name, d in acc.items():
        ids = sorted(d)
        if len(ids) != nlegs * (steps + 1):
            raise RuntimeError(f"{name}: {len(ids)} extend dispatches, expected {nlegs * (steps + 1)}")
        for leg in range(nlegs):
            full = ids[leg * (steps + 1) + 1: (leg + 1) * (steps + 1)]
            out[leg][name] = sum(d[i] for i in full) / len(full)
    return out


def measure_traffic(args):
    """HBM-side bytes of one extend launch per leg, measured in THIS run: two rocprofv3 --pmc passes (FETCH_SIZE needs 3
    of the 4 TCC slots, WRITE_SIZE 2: they cannot share a pass) over a child process that repeats the bench's set-up
    and launches.  Units and correction as MI355X_MICROARCH.md "HBM" prescribes: both counters are in KiB; on gfx950
    FETCH_SIZE tallies 128-byte fabric reads at 64 bytes, so it is doubled -- an UPPER bound here: calibrated on this
    kernel's access patterns (tools/ubench_fetchcal.hip, profiles/r02a/fetch_size_calibration.txt) the counter is exact
    for random 64-byte record gathers (one 64-byte request each) and half only where a whole 128-byte line is
    requested at once (coalesced ray reads, 128-byte gathers), so the true figure lies between FETCH_SIZE + WRITE_SIZE
    (reported as hbm_bytes_per_launch_lower) and 2 x FETCH_SIZE + WRITE_SIZE, close to the former.
    Returns [leg dicts] or None."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        log("[traffic] rocprofv3 not found: roofline.traffic = null")
        return None
    steps = 3
    nlegs = 1 if args.no_hbm_leg else 2
    tmp = tempfile.mkdtemp(prefix="ptmi_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", str(steps), "--tris", str(args.tris),
             "--rays", str(args.rays), "--hbm-tris", str(args.hbm_tris), "--tmax", str(args.tmax)]
    if args.no_hbm_leg:
        child.append("--no-hbm-leg")
    passes = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"]}
    vals = [dict() for _ in range(nlegs)]
    try:
        for name, counters in passes.items():
            outdir = os.path.join(tmp, name)
            # the program itself follows `--` (no shell, env or launcher hop in between)
            cmd = ["timeout", "-s", "KILL", "420", exe, "--pmc", *counters, "--output-format", "csv", "-d", outdir, "--", *child]
            t0 = time.time()
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            log(f"[traffic] pass {name} ({' '.join(counters)}): rc {r.returncode}, {time.time() - t0:.0f} s")
            if r.returncode != 0:
                log(r.stderr[-1500:])
                return None
            for leg, d in enumerate(_parse_pmc(outdir, nlegs, steps)):
                vals[leg].update(d)
        out = []
        for d in vals:
            fetch_kib, write_kib = d["FETCH_SIZE"], d["WRITE_SIZE"]
            out.append({"hbm_bytes_per_launch": int((2.0 * fetch_kib + write_kib) * 1024.0),
                        "hbm_bytes_per_launch_lower": int((fetch_kib + write_kib) * 1024.0),
                        "FETCH_SIZE_KiB": round(fetch_kib, 1), "WRITE_SIZE_KiB": round(write_kib, 1),
                        "fetch_correction": "x2 as MI355X_MICROARCH.md (HBM) prescribes for gfx950 = upper bound; calibrated on random "
                                            "64-B gathers the counter is exact (x1, profiles/r02a/fetch_size_calibration.txt) = lower bound",
                        "l2_hit_rate": round(d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 4),
                        "how": f"rocprofv3 --pmc, mean of {steps} full-batch launches in a child process of this run"})
        return out
    except Exception as e:       # a profiler problem must not cost the benchmark line
        log(f"[traffic] failed: {e!r}: roofline.traffic = null")
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ------------------------------------------------------------------------------------------------ CPU baselines
def cpu_closest_hit_baseline(wl, args):
    """The C restatement of CheckIntersection (linear_bvh.go:116-170) on a bounded, strided sample of the same batch:
    one thread, and every core of this job's CPU share (rank 0, N=1 only).  The all-core pass doubles as a parity
    check of the GPU results."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    orc = oracle_lib.OracleBVH(wl["verts"], wl["tris"], linear=wl["box"].linear())
    n = wl["n"]
    gp_all, gt_all = wl["P"].cpu().numpy(), wl["t"].cpu().numpy()

    def run(threads, seconds):
        pilot = np.arange(0, n, max(1, n // 32768))[:32768]
        t0 = time.time()
        orc.closest_hit(wl["o_host"][pilot], wl["d_host"][pilot], wl["tmax"], nthreads=threads, counters=False)
        rate = len(pilot) / max(time.time() - t0, 1e-4)
        ns = int(min(n, max(len(pilot), rate * seconds)))
        stride = max(1, n // ns)
        idx = np.arange(0, n, stride)[:ns]
        oo, dd = np.ascontiguousarray(wl["o_host"][idx]), np.ascontiguousarray(wl["d_host"][idx])
        t0 = time.time()
        prim, t, _ = orc.closest_hit(oo, dd, wl["tmax"], nthreads=threads, counters=False)
        dt = time.time() - t0
        mism = int((gp_all[idx] != prim).sum()) + int((gt_all[idx].view(np.uint32) != t.view(np.uint32)).sum())
        return {"value": round(len(idx) / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
                "sample": f"{len(idx)} of the {n} rays (every {stride}th), {dt:.1f} s", "gpu_mismatches_on_sample": mism}

    share = cpu_share()
    single = run(1, args.cpu_seconds * 0.5)
    allc = run(share, args.cpu_seconds * 0.5)
    allc["sample"] += (f"; C restatement of the Go reference's CheckIntersection (the Go code cannot be built here), "
                       f"one OpenMP thread per core of this job's CPU share ({share})")
    allc["single_thread"] = single
    return allc


def cpu_tile_renderer_baseline(args, dev):
    """The reference's goroutine tile renderer restated in C (oracle/orc_trace.c: orc_render_tile_workers), on C1
    (BASELINE.json configs[0]: bunny-class mesh + ground + sphere, 512x512, 16 spp, bounces 2, 4x4 tiles), one worker
    thread per tile: threadCount = 16 (the reference's default, main.go:435) and threadCount = this job's CPU share,
    beside the GPU on the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import torch
    from pathtrace_amd import scenes
    from pathtrace_amd.render import Renderer, make_params
    sc = scenes.bunny_scene()
    W = H = 512
    spp = 16
    osc = oracle_lib.OracleScene(sc)
    op = osc.params(W, H, bounces=2, scatter_rays=1, max_steps=1, ambient=0.0, step_size=1000.0, seed=1)
    share = cpu_share()
    out = {"workload": f"C1: {sc.tris.shape[0]}-triangle bunny-class scene, {W}x{H}, {spp} spp, bounces 2 (BASELINE.json configs[0])",
           "kind": "port", "what": "C restatement of the Go reference's goroutine tile renderer, one worker thread per tile "
                                   "(main.go:454-491, :832-899); the Go code itself cannot be built here",
           "cpu_share": share}
    splits = {16: (4, 4)}
    p2 = 1
    while p2 * 2 <= share:
        p2 *= 2
    sx = 1
    while sx * sx < p2:
        sx *= 2
    splits[p2] = (sx, p2 // sx)
    ref = None
    out["threadCount_nproc"] = f"threadCount_{p2}" + (" (this job's CPU share is 16 host threads: the two cases coincide)" if p2 == 16 else "")
    for threads, (a, b) in sorted(splits.items()):
        acc, cnt, oc, secs = osc.render_tile_workers(sc.camera.as_array(), op, splits=(a, b), spp=spp)
        ref = acc if ref is None else ref
        out[f"threadCount_{threads}"] = {"value": round(W * H * spp / secs / 1e6, 4), "unit": "Msamples/s", "seconds": round(secs, 3),
                                         "threadCount": threads, "tiles": f"{a}x{b}", "cores": min(threads, share),
                                         "diffuse_events_M_per_s": round(oc["diffuse_events"] / secs / 1e6, 3),
                                         "frame_complete": bool((cnt == spp).all())}
    # the GPU on the same frame: one ptmi_render_tiles_dev call with the reference's 4x4 tiles
    r = Renderer(sc, device=dev.index)
    p = make_params(W, H, bounces=2, scatter_rays=1, max_steps=1, step_size=1000.0, seed=1)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    cnt = torch.zeros((H, W), dtype=torch.int32, device=dev)
    tiles = [((i % 4) * (W // 4), (i // 4) * (H // 4), W // 4, H // 4) for i in range(16)]
    st = torch.cuda.current_stream().cuda_stream
    r.render_tiles_device(acc, cnt, p, sc.camera, tiles, spp=spp, stream=st)
    torch.cuda.synchronize()
    acc.zero_(); cnt.zero_()
    t0 = time.perf_counter()
    r.render_tiles_device(acc, cnt, p, sc.camera, tiles, spp=spp, stream=st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g = acc.cpu().numpy()
    rmse = float(np.sqrt(((g - ref) ** 2).mean()) / spp / max(1e-9, (ref / spp).mean()))
    out["gpu"] = {"value": round(W * H * spp / dt / 1e6, 2), "unit": "Msamples/s", "seconds": round(dt, 5),
                  "image_rmse_vs_cpu_rel": round(rmse, 6), "frame_complete": bool((cnt == spp).all().item())}
    r.ctx.close()
    return out


# ------------------------------------------------------------------------------------------------ render legs
# The configurations BASELINE.json names for the integrator, each run at every N so that the N = 1 and N > 1 lines compare:
RENDER_LEGS = {
    "render": dict(scene="c3", width=1920, height=1080, spp=256, bounces=2, mode="rows",
                   workload="C3: blob + glass spheres + emissive quad + mirror sphere, gradient sky + sun (BASELINE.json configs[2])"),
    "render_bounces4": dict(scene="c3", width=1920, height=1080, spp=256, bounces=4, mode="rows",
                            workload="C3 with bounces = 4 (SURVEY 8(d): 'bounces = 2, also report 4')"),
    "render_c4": dict(scene="c4", width=3840, height=2160, spp=128, bounces=2, mode="tile",
                      workload="C4: 10.4 M-triangle scene (blob over a height field, glass, mirror, emissive quad), 4K, 64x64 tiles dealt "
                               "tile_id % N, frame reduced onto rank 0 (BASELINE.json configs[3]; 128 of its 1024 spp per timed frame: "
                               "about the pass the library itself cuts a 1024-spp call at 4K into)"),
    "render_c5": dict(scene="c5", width=3840, height=2160, spp=2, bounces=2, mode="rows", max_steps=3000, step_size=1.0,
                      workload="C5: black hole Rs = 100 + accretion disk, RK4 geodesic march maxSteps = 3000, stepSize = 1, 4K "
                               "(BASELINE.json configs[4])"),
    "render_orbit": dict(scene="c3", width=1920, height=1080, spp=32, bounces=2, mode="frames", frames=8,
                         workload="an orbit of 8 frames of C3 at the reference's 32 samples per pixel (main.go:441-446, :903-937: "
                                  "theta += 1 degree per frame), all of a rank's frames in one ptmi_render_frames_dev call"),
}


def _leg_scene(kind):
    from pathtrace_amd import scenes
    if kind == "c4":
        return scenes.c4_scene()
    if kind == "c5":
        return scenes.blackhole_scene()
    return scenes.bunny_scene(blob_level=6, glass=True, emissive=True)


def render_roofline(r, run_one_gpu, leg, npix, nshadow):
    """The render leg's roofline (N = 1): per kernel class, ALGORITHMIC bytes / that class's time.  Times: the library's
    own per-kernel HIP events (ptmi_set_profiling: one wavefront loop, so the classes add up to the wall time).  Work:
    the instrumented traversal kernels (ptmi_set_counting) on the same frame -- inner nodes popped (64 B each) and
    triangles tested (48 B) by closest-hit and any-hit walks, query counts, queue entries shaded per kernel, frames
    read back.  Per-unit constants: extend 64 N_inner + 48 N_tri + 41 per query (32 B ray, 8 B
    answer, 1 B class); connect 64 N_inner + 48 N_tri + 33 per query; shade: 9 B per queue entry for the binning,
    136 B per queue record read (a fresh sample has no record where rays are straight single segments: 12 B, its answer
    and its free-list entry; else its one-sector record, 72 B), 80 B triangle record per hit shaded, 64 B
    per frame read or parked by the lean kernels (128 by the general kernel), 128 B per child record, 32 B per shadow
    ray + 16 B per term, 20 B per finished sample (radiance + free list)."""
    r.set_profiling(True)
    run_one_gpu()
    prof = r.profile()
    r.set_profiling(False)
    r.bvh.set_counting(True)
    r.bvh.reset_counters()
    run_one_gpu()
    c = r.bvh.counters()
    r.bvh.set_counting(False)
    ci, ct = c.inner_visits - c.any_inner_visits, c.tri_tests - c.any_tri_tests
    visits = c.shade_ret_visits + c.shade_diff_visits + c.shade_general_visits
    hits_shaded = c.shade_diff_visits + c.shade_general_visits
    entries = c.extend_entries                     # pending calls traced (== closest_queries unless a call marches in segments)
    children = max(0, entries - c.samples)
    implicit = leg.get("max_steps", 1) == 1        # fresh samples without queue records (generate writes 32-byte rays for extend)
    shade_bytes = (9 * entries + 136 * children + (12 if implicit else 72) * c.samples + 80 * hits_shaded + 64 * c.frames_read +
                   64 * c.shade_diff_visits + 128 * c.shade_general_visits + 128 * children + (32 + 16) * c.any_queries + 20 * c.samples)
    rows = {"extend": (NODE_BYTES * ci + TRI_BYTES * ct + 41 * entries, prof["extend_ms"]),
            "connect": (NODE_BYTES * c.any_inner_visits + TRI_BYTES * c.any_tri_tests + 33 * c.any_queries, prof["connect_ms"]),
            "shade": (shade_bytes, prof["shade_ms"]),
            "generate": ((4 + (32 if implicit else 64)) * c.samples, prof["generate_ms"]),
            "accumulate": (12 * c.samples + 16 * npix, prof["accumulate_ms"])}
    out = {"how": "per kernel class: algorithmic bytes (instrumented kernels of this run) / time (ptmi_set_profiling events, one wavefront loop); "
                  "l1_gather_peak was measured for divergent gathers: coherent rays (primary rays are half of extend's) share cache lines, so a "
                  "fraction near or above 1 says the kernel is at that unit's limit",
           "hbm_peak": HBM_PEAK_GBS, "l1_gather_peak": L1_GATHER_PEAK_GBS}
    total_b = total_ms = 0.0
    for k, (bts, ms) in rows.items():
        gbs = bts / max(ms, 1e-9) / 1e6
        out[k] = {"algorithmic_bytes": int(bts), "ms": round(ms, 3), "GBs": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                  "l1_gather_frac": round(gbs / L1_GATHER_PEAK_GBS, 4)}
        total_b += bts
        total_ms += ms
    out["extend"].update(inner_per_query=round(ci / max(1, c.closest_queries), 3), tri_per_query=round(ct / max(1, c.closest_queries), 3),
                         queries=int(c.closest_queries), entries=int(entries))
    if leg.get("max_steps", 1) > 1:
        out["extend"]["note"] = ("a marching call makes up to max_steps segment queries, most of them answered without a walk; the kernel is "
                                 "bound by the RK4 step's arithmetic (four accelerations, each with a correctly rounded square root and reciprocal), not by bytes")
    out["connect"].update(inner_per_query=round(c.any_inner_visits / max(1, c.any_queries), 2),
                          tri_per_query=round(c.any_tri_tests / max(1, c.any_queries), 2))
    out["shade"].update(visits=int(visits), ret_visits=int(c.shade_ret_visits), diff_visits=int(c.shade_diff_visits),
                        general_visits=int(c.shade_general_visits), frames_read=int(c.frames_read))
    total_ms += prof["prepare_ms"]
    out["frame"] = {"algorithmic_bytes": int(total_b), "ms": round(total_ms, 3), "GBs": round(total_b / total_ms / 1e6, 1),
                    "hbm_frac": round(total_b / total_ms / 1e6 / HBM_PEAK_GBS, 4), "iterations": int(prof["iterations"]),
                    "launches": int(prof["launches"])}
    return out


def render_leg(name, args, rank, world, dist, dev, torch):
    """Secondary figure (not `value`): samples/s of the full wavefront integrator on one of RENDER_LEGS.  With N > 1 the
    frame is sharded (blocks of 8 rows, or 64x64 tiles, round-robin over the ranks), every rank renders its share in one
    call and the frame is summed onto rank 0 through the C ABI's RCCL communicator (strong scaling)."""
    from pathtrace_amd import sharded
    from pathtrace_amd.render import Renderer, make_params
    leg = dict(RENDER_LEGS[name])
    if name == "render":
        leg.update(width=args.render_width, height=args.render_height, spp=args.render_spp)
    sc = _leg_scene(leg["scene"])
    r = Renderer(sc, device=dev.index)
    W, H, spp = leg["width"], leg["height"], leg["spp"]
    p = make_params(W, H, bounces=leg["bounces"], scatter_rays=1, max_steps=leg.get("max_steps", 1), step_size=leg.get("step_size", 1000.0), seed=1)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    cnt = torch.zeros((H, W), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    rehearse = os.environ.get("PTMI_BENCH_REHEARSE") == "1"
    comm = sharded.make_comm(r, rank, world, dist, dev, rehearse=rehearse) if world > 1 else None
    render_tiles = sharded.make_device_render_tiles(r, p, sc.camera, stream)

    def run(spp_):
        if leg["mode"] == "tile":       # the reference's tiles (main.go:454-491), a rank's share in one ptmi_render_tiles_dev call
            sharded.render_sharded(None, acc, cnt, W, H, spp_, rank=rank, world=world, dst=0, mode="tile", tile=64,
                                   render_tiles=render_tiles, comm=comm, stream=stream)
        else:                           # every rank renders its row-interleaved share in one call, then one reduce onto rank 0
            sharded.render_frame_sharded(r, comm, acc, cnt, p, sc.camera, spp_, row_block=8, root=0, stream=stream)
    run(spp)                                 # warm-up at full size: path pool and sample buffer allocation, RCCL channel set-up
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    r.bvh.reset_counters()
    t0 = time.perf_counter()
    run(spp)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    c = r.bvh.counters()
    q = torch.tensor([dt, float(c.closest_queries), float(c.any_queries), float(c.diffuse_events)], dtype=torch.float64,
                     device=dev)
    if dist is not None:
        mx = q.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(q, op=dist.ReduceOp.SUM)
        q[0] = mx[0]
    q = q.cpu().tolist()
    complete = bool((cnt == spp).all().item()) if rank == 0 else True
    share = "blocks of 8 rows" if leg["mode"] == "rows" else "64x64 tiles"
    if world == 1:
        how, kind = f"single GPU ({share} in one call)", "single"
    else:
        how = f"{share} round-robin over {world} ranks, one render call per rank + "
        if comm.kind == "rccl":
            how += "RCCL ncclReduce behind the C ABI (ptmi_render_sharded / ptmi_reduce_frame)"
            kind = "rccl"
        elif rehearse:
            how += "host-staged gloo reduce (rehearsal)"
            kind = "rehearsal"
        else:
            how += f"torch.distributed reduce (the C ABI's communicator was not available: {comm.note})"
            kind = "fallback"
    out = {"metric": "samples/s, full multi-bounce integrator", "value": round(W * H * spp / q[0] / 1e6, 2),
           "unit": "Msamples/s", "seconds": round(q[0], 4), "comm": kind,
           "config": {"workload": leg["workload"], "width": W, "height": H, "spp": spp, "bounces": leg["bounces"],
                      "max_steps": leg.get("max_steps", 1), "triangles": int(sc.tris.shape[0]), "sharding": how},
           "closest_hit_Mqueries_per_s": round(q[1] / q[0] / 1e6, 1), "any_hit_Mqueries_per_s": round(q[2] / q[0] / 1e6, 1),
           "diffuse_events_M_per_s": round(q[3] / q[0] / 1e6, 1), "frame_complete": complete}
    if world == 1 and not args.no_render_roofline:
        out["roofline"] = render_roofline(r, lambda: (run(spp), torch.cuda.synchronize()), leg, W * H, int(sc.lights.shape[0]) + 2)
    if comm is not None:
        comm.close()
    r.ctx.close()
    del r, acc, cnt
    torch.cuda.empty_cache()
    return out


def orbit_leg(name, args, rank, world, dist, dev, torch):
    """Secondary figure: an orbit of frames at the reference's samples per pixel.  Frames are independent, so with N > 1
    rank r renders frames r, r + N, ... and nothing is exchanged; `value` = all frames' samples / the slowest rank.  Beside
    it the same frames as one call each (what the reference's loop would issue)."""
    from pathtrace_amd import driver
    from pathtrace_amd.render import Renderer, make_params
    leg = dict(RENDER_LEGS[name])
    sc = _leg_scene(leg["scene"])
    r = Renderer(sc, device=dev.index)
    W, H, spp, F = leg["width"], leg["height"], leg["spp"], leg["frames"]
    p = make_params(W, H, bounces=leg["bounces"], seed=1)
    radius = float(np.linalg.norm(np.asarray(sc.camera.position)))
    cams = [driver.spherical_around((0.0, 0.0, 0.0), radius, (90.0 + f) * driver.DEG, 83.0 * driver.DEG, sc.camera.frustum) for f in range(F)]
    mine = cams[rank::world]
    acc = torch.zeros((max(len(mine), 1), H, W, 3), dtype=torch.float32, device=dev)
    cnt = torch.zeros((max(len(mine), 1), H, W), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def timed(fn):
        fn()                                       # warm-up at full size
        acc.zero_(); cnt.zero_(); torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    t_batch = timed(lambda: mine and r.render_frames_device(acc[:len(mine)], cnt[:len(mine)], p, mine, spp=spp, stream=stream))
    complete = bool((cnt[:len(mine)] == spp).all().item())
    t_each = timed(lambda: [r.render_device(acc[f], cnt[f], p, mine[f], spp=spp, stream=stream) for f in range(len(mine))])
    n = F * W * H * spp
    out = {"metric": "samples/s, full multi-bounce integrator, frames of an orbit", "value": round(n / t_batch / 1e6, 2), "unit": "Msamples/s",
           "seconds": round(t_batch, 4), "ms_per_frame": round(t_batch * 1e3 / F, 2), "comm": "single" if world == 1 else "none",
           "one_call_per_frame": {"value": round(n / t_each / 1e6, 2), "unit": "Msamples/s", "ms_per_frame": round(t_each * 1e3 / F, 2)},
           "config": {"workload": leg["workload"], "width": W, "height": H, "spp": spp, "bounces": leg["bounces"], "frames": F,
                      "triangles": int(sc.tris.shape[0]),
                      "sharding": "single GPU" if world == 1 else f"frame f on rank f % {world}, no exchange (frames are independent)"},
           "frame_complete": complete}
    r.ctx.close()
    del r, acc, cnt
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-render", action="store_true", help="skip the secondary samples/s figure")
    ap.add_argument("--render-width", type=int, default=1920)
    ap.add_argument("--render-height", type=int, default=1080)
    ap.add_argument("--render-spp", type=int, default=256)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--tmax", type=float, default=1000.0)
    ap.add_argument("--hbm-tris", type=int, default=10_000_000, help="triangles of the HBM-resident second leg")
    ap.add_argument("--no-hbm-leg", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC child passes (roofline.traffic = null)")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="CPU time budget of the closest-hit baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--render-timeout", type=float, default=420.0, help="N > 1: seconds after which stuck render legs are given up")
    ap.add_argument("--render-legs", default="render,render_bounces4,render_c4,render_c5,render_orbit", help="comma-separated subset of the render legs")
    ap.add_argument("--no-render-roofline", action="store_true", help="skip the profiled + counted frames of the render legs (N = 1)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.pmc_child:
        pmc_child(args)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        args.no_hbm_leg = True           # the second leg, the traffic passes and the CPU baselines are N=1 work
    # The PMC passes run first, before this process touches the GPU (their child repeats the set-up on its own).
    traffic = None
    if rank == 0 and world == 1 and not args.no_traffic:
        traffic = measure_traffic(args)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # PTMI_BENCH_REHEARSE=1: every rank on cuda:0 over gloo -- exercises the world > 1 code on a one-GPU box (RCCL
    # refuses two ranks on one device); the numbers of such a run mean nothing
    rehearse = os.environ.get("PTMI_BENCH_REHEARSE") == "1"
    dev = torch.device("cuda", 0 if rehearse else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist = _HostStagedDist(dist)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    stream = torch.cuda.current_stream().cuda_stream
    wl = build_workload(args.tris, args.rays, args.tmax, rank, torch, dev, "C2")
    wl["stack"] = stack_class(int(wl["stats"].max_depth))
    n = wl["n"]
    c, alg_bytes = count_work(wl, stream)
    wall, kernel_ms = time_extend(wl, args.steps, args.warmup, torch, dist, stream)
    if dist is not None:
        tt = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())
        nn = torch.tensor([float(n)], dtype=torch.float64, device=dev)
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        total_rays = float(nn.item())
    else:
        total_rays = float(n)

    # secondary: the any-hit (QuickCheckIntersection) form of the same batch, rank-local, outside the timed region
    occ = torch.empty(n, dtype=torch.uint8, device=dev)
    wl["bvh"].QuickCheckIntersectionDevice(wl["O"], wl["D"], wl["T"], occ, stream)
    torch.cuda.synchronize()
    ea0, ea1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea0.record()
    for _ in range(3):
        wl["bvh"].QuickCheckIntersectionDevice(wl["O"], wl["D"], wl["T"], occ, stream)
    ea1.record()
    torch.cuda.synchronize()
    any_ms = ea0.elapsed_time(ea1) / 3
    any_hit = {"metric": "any-hit Mrays/s on the same batch (one GPU)", "value": round(n / any_ms / 1e3, 2), "unit": "Mrays/s",
               "kernel": "connect_vote_kernel", "kernel_ms": round(any_ms, 4), "occluded_fraction": round(float(occ.float().mean().item()), 4)}
    del occ

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_closest_hit_baseline(wl, args)

    out = None
    if rank == 0:
        out = {
            "metric": "closest-hit Mrays/s (16M-ray primary+secondary batch, 1M-triangle mesh)",
            "value": round(total_rays * args.steps / wall / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall * 1e3 / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: closest-hit ray batch on a procedurally generated triangle soup "
                                   "(BASELINE.json configs[1])",
                       "triangles": int(wl["tris"].shape[0]), "rays_per_gpu": int(n),
                       "primary_grid": list(wl["grid"]), "tmax": args.tmax,
                       "bvh_nodes": int(wl["stats"].linear_nodes), "bvh_depth": int(wl["stats"].max_depth),
                       "sharding": f"replicated BVH, {world} independent ray batches"},
            "roofline": roofline_block(wl, c, alg_bytes, kernel_ms, traffic[0] if traffic else None,
                                       "cache-resident leg: the 67 MB tree is served from L1/L2/Infinity Cache, so the kernel runs "
                                       "against the CU's L1 gather rate, not HBM; hbm_algorithmic_frac above 1 says only that"),
        }
        if rehearse:
            out["rehearsal"] = "all ranks share cuda:0 over gloo: exercises the world > 1 code, not a measurement"
        out["any_hit"] = any_hit
        if cpu is not None:
            out["cpu_baseline"] = cpu
    # free the first leg before the second: both hold a 16.7 M-ray batch
    wl["ctx"].close()
    del wl
    torch.cuda.empty_cache()

    if not args.no_hbm_leg:          # N=1 only
        wl2 = build_workload(args.hbm_tris, args.rays, args.tmax, rank, torch, dev, "HBM-resident leg")
        wl2["stack"] = stack_class(int(wl2["stats"].max_depth))
        c2, alg2 = count_work(wl2, stream)
        wall2, kms2 = time_extend(wl2, max(3, args.steps // 2), 2, torch, None, stream)
        steps2 = max(3, args.steps // 2)
        rb = roofline_block(wl2, c2, alg2, kms2, traffic[1] if traffic and len(traffic) > 1 else None,
                            "HBM-resident leg: the packed tree exceeds the 256 MiB Infinity Cache; the hot top of the tree is "
                            "still served from L1/L2, the leaves and lower nodes come over the fabric (hbm_frac_counter)")
        rb["Mrays_per_s"] = round(wl2["n"] * steps2 / wall2 / 1e6, 2)
        rb["config"] = {"triangles": int(wl2["tris"].shape[0]), "rays": int(wl2["n"]), "bvh_nodes": int(wl2["stats"].linear_nodes),
                        "bvh_depth": int(wl2["stats"].max_depth), "steps": steps2}
        if not args.no_cpu_baseline:
            # parity spot check of this leg too: the oracle on every 256th ray
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            orc = oracle_lib.OracleBVH(wl2["verts"], wl2["tris"], linear=wl2["box"].linear())
            idx = np.arange(0, wl2["n"], 256)
            op, ot, _ = orc.closest_hit(np.ascontiguousarray(wl2["o_host"][idx]), np.ascontiguousarray(wl2["d_host"][idx]),
                                        args.tmax, nthreads=cpu_share(), counters=False)
            gp, gt = wl2["P"].cpu().numpy()[idx], wl2["t"].cpu().numpy()[idx]
            rb["gpu_mismatches_on_sample"] = int((gp != op).sum()) + int((gt.view(np.uint32) != ot.view(np.uint32)).sum())
            rb["parity_sample"] = f"{len(idx)} rays (every 256th) against the CPU oracle"
            del orc
        if out is not None:
            out["roofline_hbm_resident"] = rb
        wl2["ctx"].close()
        del wl2
        torch.cuda.empty_cache()

    # The primary figure is complete here.  With N > 1 the render legs enter collectives (communicator set-up, the frame
    # reduce): if a rank fails or hangs there, the others would wait for ever and the whole line would be lost, so a
    # watchdog per rank ends the process after --render-timeout seconds; rank 0 prints the line with what it has.  Such
    # a run -- and one in which a rank raised, or whose reduce did not go through the C ABI's RCCL communicator although
    # it should have -- exits NON-ZERO after the line is out: it must not pass for a measurement of the RCCL path.
    state = {"printed": False, "failed": ""}
    watchdog = None
    if world > 1:
        import threading

        def give_up():
            if rank == 0 and not state["printed"]:
                out["render_error"] = (f"the render legs did not finish within {args.render_timeout:.0f} s (a rank failed or a "
                                       "collective hung); the primary figure above is unaffected")
                print(json.dumps(out), flush=True)
            os._exit(3)
        watchdog = threading.Timer(args.render_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
    legs = [] if args.no_render else [l for l in args.render_legs.split(",") if l]
    for name in legs:
        if name not in RENDER_LEGS:
            raise SystemExit(f"unknown render leg {name!r}; known: {', '.join(RENDER_LEGS)}")
        try:
            leg = (orbit_leg if RENDER_LEGS[name]["mode"] == "frames" else render_leg)(name, args, rank, world, dist, dev, torch)
        except Exception as e:               # noqa: BLE001 -- reported in the line, the primary figure stands
            if world == 1:
                raise
            leg = {"error": f"rank {rank}: {type(e).__name__}: {e}"}
            state["failed"] = leg["error"]
        if leg.get("comm") == "fallback":
            state["failed"] = state["failed"] or f"{name}: the frame reduce fell back to torch.distributed"
        if rank == 0:
            out[name] = leg
        log(f"[rank {rank}] {name}: {leg.get('value', leg.get('error'))} {leg.get('unit', '')}")
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            tb = cpu_tile_renderer_baseline(args, dev)
            out["cpu_baseline"]["tile_renderer"] = tb
        print(json.dumps(out), flush=True)
        state["printed"] = True
    failed = state["failed"]
    if dist is not None:
        # every rank learns whether any rank failed, so that all exit with the same code
        flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=dev)
        try:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            failed = failed or ("another rank failed in a render leg" if int(flag.item()) else "")
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:               # noqa: BLE001
            failed = failed or f"{type(e).__name__}: {e}"
    if watchdog is not None:
        watchdog.cancel()
    if failed:
        log(f"[rank {rank}] FAILED: {failed}")
        sys.exit(4)


if __name__ == "__main__":
    main()
