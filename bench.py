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

Prints ONE JSON line on rank 0.  `roofline` prices the extend kernel against HBM bandwidth with
ALGORITHMIC bytes = 64*N_inner + 48*N_tri + 40*R per launch; `cpu_baseline` is the
CPU oracle (a C restatement of the Go reference -- the Go code itself cannot be built here)
timed on a bounded sample of the same batch.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L1_GATHER_PEAK_GBS = 13900.0   # measured: divergent 4 x dwordx4 gathers of L1-resident 64-B records (profiles/r01b/ubench_gather.txt)
NODE_BYTES, TRI_BYTES, RAY_BYTES = 64, 48, 40


def log(*a):
    print(*a, file=sys.stderr, flush=True)


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

    def reduce(self, t, dst=0, op=None, group=None):
        h = t.cpu()
        self._d.reduce(h, dst=dst, op=op, group=group)
        t.copy_(h)


def build_workload(args, rank, torch, dev):
    """Mesh + BVH + ray batch, all untimed set-up.  Returns dict with device tensors."""
    from pathtrace_amd import BuildBVH, ConstructLinearBVH, Context, scenes

    t0 = time.time()
    verts, tris = scenes.make_soup(args.tris, seed=42)
    box = BuildBVH(verts, tris)
    st = box.GetStats()
    ctx = Context(dev.index)
    bvh = ConstructLinearBVH(box, ctx=ctx)
    t1 = time.time()
    log(f"[rank {rank}] mesh {tris.shape[0]} tris, BVH {st.linear_nodes} nodes depth {st.max_depth} "
        f"built in {st.build_ms:.0f} ms (set-up {t1 - t0:.1f} s)")

    n_primary = args.rays // 2
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
    T1 = torch.full((n_primary,), args.tmax, dtype=torch.float32, device=dev)
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
    wl = dict(ctx=ctx, bvh=bvh, box=box, verts=verts, tris=tris, n=n, o_host=o, d_host=d,
              O=torch.from_numpy(o).to(dev), D=torch.from_numpy(d).to(dev),
              T=torch.full((n,), args.tmax, dtype=torch.float32, device=dev),
              P=torch.empty(n, dtype=torch.int32, device=dev), t=torch.empty(n, dtype=torch.float32, device=dev),
              primary_hit_rate=float((prim >= 0).mean()), stats=st, grid=(w, h))
    log(f"[rank {rank}] rays {n} ({w}x{h} primary, hit rate {wl['primary_hit_rate']:.3f}) "
        f"(set-up {time.time() - t0:.1f} s)")
    return wl


def cpu_baseline(wl, args):
    """Time the CPU oracle on a bounded, strided sample of the same batch (rank 0, N=1 only)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    orc = oracle_lib.OracleBVH(wl["verts"], wl["tris"], linear=wl["box"].linear())
    threads = max(1, min(16, os.cpu_count() or 1))
    n = wl["n"]
    pilot = np.arange(0, n, max(1, n // 65536))[:65536]
    t0 = time.time()
    orc.closest_hit(wl["o_host"][pilot], wl["d_host"][pilot], args.tmax, nthreads=threads, counters=False)
    dt = max(time.time() - t0, 1e-4)
    rate = len(pilot) / dt
    ns = int(min(n, max(len(pilot), rate * args.cpu_seconds)))
    idx = np.arange(0, n, max(1, n // ns))[:ns]
    oo, dd = np.ascontiguousarray(wl["o_host"][idx]), np.ascontiguousarray(wl["d_host"][idx])
    t0 = time.time()
    prim, t, _ = orc.closest_hit(oo, dd, args.tmax, nthreads=threads, counters=False)
    dt = time.time() - t0
    # the sample doubles as a full-size parity spot check of the GPU results
    gp = wl["P"].cpu().numpy()[idx]
    gt = wl["t"].cpu().numpy()[idx]
    mism = int((gp != prim).sum()) + int((gt.view(np.uint32) != t.view(np.uint32)).sum())
    return {"value": round(len(idx) / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{len(idx)} of the {n} rays (every {max(1, n // ns)}th), {dt:.1f} s, C restatement of the Go "
                      f"reference's CheckIntersection with {threads} OpenMP threads",
            "gpu_mismatches_on_sample": mism}


def render_leg(args, rank, world, dist, dev, torch):
    """Secondary figure (not `value`): samples/s of the full wavefront integrator on BASELINE.json configs[2]
    (C3: glass + emissive + gradient sky, 1920x1080, 256 spp, bounces 2).  With N > 1 the frame is sharded by
    blocks of 8 rows over the ranks and reduced onto rank 0 with RCCL (strong scaling)."""
    from pathtrace_amd import scenes, sharded
    from pathtrace_amd.render import Renderer, make_params
    sc = scenes.bunny_scene(blob_level=6, glass=True, emissive=True)
    r = Renderer(sc, device=dev.index)
    W, H, spp = args.render_width, args.render_height, args.render_spp
    p = make_params(W, H, bounces=2, scatter_rays=1, max_steps=1, step_size=1000.0, seed=1)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    cnt = torch.zeros((H, W), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    rows = sharded.make_device_render_rows(r, p, sc.camera, stream)

    def run(spp_):
        # every rank renders its row-interleaved share of the frame in one call, then one reduce onto rank 0
        sharded.render_sharded(None, acc, cnt, W, H, spp_, rank=rank, world=world, dist=dist, dst=0, mode="rows",
                               render_rows=rows, row_block=8)
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
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the CPU oracle's integrator on a bounded part of the same frame: bands of 16 rows every 64, 8 samples per pixel
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        osc = oracle_lib.OracleScene(sc)
        threads = max(1, min(16, os.cpu_count() or 1))
        op = osc.params(W, H, bounces=2, scatter_rays=1, max_steps=1, ambient=p.ambient, step_size=1000.0, seed=1)
        rows, spp_cpu, nsamp, t0, sb = 16, 8, 0, time.time(), 0
        while time.time() - t0 < args.cpu_seconds * 0.6:          # passes of 8 more samples per pixel over the bands
            for y0 in range(0, H, 64):
                osc.render(sc.camera.as_array(), op, rect=(0, y0, W, min(rows, H - y0)), spp_begin=sb, spp_count=spp_cpu,
                           nthreads=threads)
                nsamp += W * min(rows, H - y0) * spp_cpu
            sb += spp_cpu
        dtc = time.time() - t0
        cpu = {"value": round(nsamp / dtc / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
               "sample": f"{nsamp} samples of the same frame (bands of {rows} rows every 64, {sb} spp), {dtc:.1f} s, C "
                         f"restatement of TraceRay with {threads} OpenMP threads"}
    return {"metric": "samples/s, full multi-bounce integrator", "value": round(W * H * spp / q[0] / 1e6, 2),
            "unit": "Msamples/s", "seconds": round(q[0], 4),
            "config": {"workload": "C3: blob + glass spheres + emissive quad + mirror sphere, gradient sky + sun "
                                   "(BASELINE.json configs[2])", "width": W, "height": H, "spp": spp, "bounces": 2,
                       "triangles": int(sc.tris.shape[0]),
                       "sharding": "single GPU" if world == 1 else f"blocks of 8 rows round-robin over {world} ranks, one render call per rank + RCCL reduce"},
            "closest_hit_Mqueries_per_s": round(q[1] / q[0] / 1e6, 1), "any_hit_Mqueries_per_s": round(q[2] / q[0] / 1e6, 1),
            "diffuse_events_M_per_s": round(q[3] / q[0] / 1e6, 1), "frame_complete": complete, "cpu_baseline": cpu}


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
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    wl = build_workload(args, rank, torch, dev)
    bvh, n = wl["bvh"], wl["n"]
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        bvh.CheckIntersectionDevice(wl["O"], wl["D"], wl["T"], wl["P"], wl["t"], stream)

    # algorithmic work of one launch, from the instrumented kernel variant (untimed)
    bvh.set_counting(True)
    bvh.reset_counters()
    step()
    c = bvh.counters()
    bvh.set_counting(False)
    alg_bytes = NODE_BYTES * c.inner_visits + TRI_BYTES * c.tri_tests + RAY_BYTES * n

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps        # HIP events on the launch stream
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
    bvh.QuickCheckIntersectionDevice(wl["O"], wl["D"], wl["T"], occ, stream)
    torch.cuda.synchronize()
    ea0, ea1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea0.record()
    for _ in range(3):
        bvh.QuickCheckIntersectionDevice(wl["O"], wl["D"], wl["T"], occ, stream)
    ea1.record()
    torch.cuda.synchronize()
    any_ms = ea0.elapsed_time(ea1) / 3
    any_hit = {"metric": "any-hit Mrays/s on the same batch (one GPU)", "value": round(n / any_ms / 1e3, 2), "unit": "Mrays/s",
               "kernel": "connect_vote_kernel", "kernel_ms": round(any_ms, 4), "occluded_fraction": round(float(occ.float().mean().item()), 4)}
    del occ

    render = None
    if not args.no_render:
        render = render_leg(args, rank, world, dist, dev, torch)

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("tris") == args.tris and tj.get("rays") == n:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "closest-hit Mrays/s (16M-ray primary+secondary batch, 1M-triangle mesh)",
            "value": round(total_rays * args.steps / wall / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: closest-hit ray batch on a procedurally generated triangle soup "
                                   "(BASELINE.json configs[1])",
                       "triangles": int(wl["tris"].shape[0]), "rays_per_gpu": int(n),
                       "primary_grid": list(wl["grid"]), "tmax": args.tmax,
                       "bvh_nodes": int(wl["stats"].linear_nodes), "bvh_depth": int(wl["stats"].max_depth),
                       "sharding": f"replicated BVH, {world} independent ray batches"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "extend_vote_kernel", "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "inner_visits_per_ray": round(c.inner_visits / n, 3),
                         "tri_tests_per_ray": round(c.tri_tests / n, 3),
                         "note": "the 67 MB tree is cache resident (traffic = fabric bytes from profiles/), so frac "
                                 "can exceed 1; the ceiling this kernel runs against is the CU's L1 gather rate, "
                                 "13.9 TB/s chip-wide measured by tools/ubench_gather.hip (profiles/r01b/ubench_gather.txt)",
                         "l1_gather_peak_GBs": L1_GATHER_PEAK_GBS,
                         "frac_of_l1_gather_peak": round(achieved / L1_GATHER_PEAK_GBS, 4)},
        }
        if rehearse:
            out["rehearsal"] = "all ranks share cuda:0 over gloo: exercises the world > 1 code, not a measurement"
        out["any_hit"] = any_hit
        if render is not None:
            out["render"] = render
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, args)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
