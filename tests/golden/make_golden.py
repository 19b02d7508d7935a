#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz.

The reference ships no tests or vectors and cannot be run here (Go, no toolchain), so these
vectors come from the CPU oracle (oracle/, a C restatement of the Go code): inputs and expected
outputs only.  They pin (a) the oracle against regressions / platform drift and (b) the HIP path
against fixed data that does not depend on regenerating the scenes.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from pathtrace_amd import scenes  # noqa: E402
import oracle_lib  # noqa: E402
from helpers import random_rays, adversarial_rays  # noqa: E402


def case(name, verts, tris, o, d, tmax):
    orc = oracle_lib.OracleBVH(verts, tris)          # literal builder
    lin = orc.linear()
    prim, t, cc = orc.closest_hit(o, d, tmax)
    occ, ca = orc.any_hit(o, d, tmax)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), verts=verts, tris=tris, o=o, d=d,
                        tmax=np.broadcast_to(np.asarray(tmax, np.float32), (o.shape[0],)).copy(),
                        prim=prim, t=t, occluded=occ.astype(np.uint8),
                        node_bounds=lin.node_bounds, is_leaf=lin.is_leaf, tri_offset=lin.tri_offset,
                        tri_count=lin.tri_count, second_child=lin.second_child, tri_ids=lin.tri_ids,
                        closest_counters=np.array([cc["inner_visits"], cc["tri_tests"], cc["leaf_visits"]], np.uint64),
                        any_counters=np.array([ca["inner_visits"], ca["tri_tests"], ca["leaf_visits"]], np.uint64))
    print(name, "tris", tris.shape[0], "rays", o.shape[0], "hits", int((prim >= 0).sum()), "occluded", int(occ.sum()))


def main():
    v, f = scenes.make_soup(37, 101)
    o, d, tm = adversarial_rays(v, f, 1024, 1)
    case("soup37_aimed", v, f, o, d, tm)

    v, f = scenes.make_soup(2500, 102)
    o, d = random_rays(v, 2048, 2)
    o2, d2, tm2 = adversarial_rays(v, f, 2048, 3)
    case("soup2500_mixed", v, f, np.concatenate([o, o2]), np.concatenate([d, d2]),
         np.concatenate([np.full(2048, 1000.0, np.float32), tm2]))

    sc = scenes.bunny_scene(small=True, glass=True, emissive=True)
    o, d = scenes.camera_rays(sc.camera, 64, 64)
    case("bunny_small_primary", sc.verts, sc.tris, o, d, 1000.0)
    case("bunny_small_short_step", sc.verts, sc.tris, o, d, 6.0)


if __name__ == "__main__":
    main()
