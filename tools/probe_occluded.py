#!/usr/bin/env python3
"""Experiment build -DPTMI_PROBE_OCCLUDED=1 (tools/build_variant.sh occl ...; PTMI_LIB=tools/bin/libptmi_occl.so): which
share of the any-hit walk's node and triangle fetches belongs to rays that turn out occluded?  (only the vote-scheduled
connect kernel is instrumented: use frames large enough for it)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtrace_amd import scenes
from pathtrace_amd.render import Renderer, make_params
dev = torch.device("cuda", 0)
for name, sc, W, H, spp in (("C3", scenes.bunny_scene(blob_level=6, glass=True, emissive=True), 1920, 1080, 64), ("C4-class", scenes.c4_scene(), 3840, 2160, 16)):
    p = make_params(W, H, bounces=2, seed=1)
    r = Renderer(sc)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device=dev); cnt = torch.zeros((H, W), dtype=torch.int32, device=dev)
    os.environ["PTMI_RENDER_LANES"] = "1"
    r.bvh.set_counting(True); r.bvh.reset_counters()
    r.render_device(acc, cnt, p, sc.camera, spp=spp, stream=torch.cuda.current_stream().cuda_stream)
    c = r.bvh.counters(); r.bvh.set_counting(False)
    print(f"{name}: {c.any_queries} shadow rays, {c.shade_ret_visits} occluded ({c.shade_ret_visits / c.any_queries:.3f}); inner nodes {c.any_inner_visits}, "
          f"of occluded rays {c.shade_diff_visits} ({c.shade_diff_visits / c.any_inner_visits:.3f}); triangles {c.any_tri_tests}, of occluded rays "
          f"{c.shade_general_visits} ({c.shade_general_visits / c.any_tri_tests:.3f}); per occluded ray {c.shade_diff_visits / max(1, c.shade_ret_visits):.1f} nodes, "
          f"per free ray {(c.any_inner_visits - c.shade_diff_visits) / max(1, c.any_queries - c.shade_ret_visits):.1f}")
    del r
