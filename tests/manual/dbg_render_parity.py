#!/usr/bin/env python3
"""Debug helper: per-configuration mismatch fraction of the GPU integrator against the CPU oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pathtrace_amd import scenes, Context
from pathtrace_amd.render import Renderer, make_params
import oracle_lib
ctx = Context(0)
for glass, emissive, bounces, scatter in [(False, False, 1, 2), (False, False, 2, 2), (True, True, 3, 2), (True, True, 2, 1)]:
    sc = scenes.bunny_scene(small=True, glass=glass, emissive=emissive)
    p = make_params(64, 64, bounces=bounces, scatter_rays=scatter, seed=11)
    r = Renderer(sc, ctx=ctx)
    got = r.render_samples(p, sc.camera, spp=4)
    osc = oracle_lib.OracleScene(sc)
    want = osc.render_samples(sc.camera.as_array(), oracle_lib.OrcParams(p.width, p.height, p.bounces, p.scatter_rays, p.max_steps, p.ambient, p.step_size, p.seed), spp_count=4)
    err = (np.abs(got - want) / (1.0 + np.abs(want))).max(-1)
    bad = err > 2e-4
    idx = np.argwhere(bad)[:6]
    for a in idx: print("   sample", tuple(a), "got", got[tuple(a)], "want", want[tuple(a)])
    print(f"glass={glass} emissive={emissive} bounces={bounces} scatter={scatter}: {bad.mean():.4f} differ, max {np.nanmax(err):.3e}", flush=True)
