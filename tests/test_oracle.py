"""CPU oracle (oracle/, C restatement of the Go reference) against hand-derived known answers and
the committed golden vectors.  PARITY UNPINNED: the reference holds no tests or vectors, so the
known answers below are derived by hand from the cited reference lines."""
import glob
import os

import numpy as np
import pytest

import oracle_lib
from oracle_lib import probe_aabb, probe_segment_triangle, probe_shadow_triangle
from helpers import bits_equal

INF = np.float32(3.4028234663852886e38)      # math.MaxFloat32, linear_bvh.go:9
TRI = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def test_segment_triangle_known_answers():
    # utils.go:63-125 worked by hand for A=(0,0,0) B=(1,0,0) C=(0,1,0), o=(.25,.25,1), d=(0,0,-1):
    # pvec=(1,0,0) det=1 u=.25 qvec=(0,1,-.25) v=.25 t=1
    assert probe_segment_triangle([.25, .25, 1], [0, 0, -1], 10.0, TRI) == (True, np.float32(1.0))
    # t == stepSize passes (`t > stepSize` is the rejection, utils.go:121)
    assert probe_segment_triangle([.25, .25, 1], [0, 0, -1], 1.0, TRI) == (True, np.float32(1.0))
    assert probe_segment_triangle([.25, .25, 1], [0, 0, -1], np.float32(0.99999994), TRI)[0] is False
    # t <= EPSILON rejected
    assert probe_segment_triangle([.25, .25, 1e-7], [0, 0, -1], 10.0, TRI)[0] is False
    assert probe_segment_triangle([.25, .25, 2e-6], [0, 0, -1], 10.0, TRI)[0] is True
    # +-1e-6 slack on u, v and u+v (utils.go:96, :111)
    assert probe_segment_triangle([-5e-7, .25, 1], [0, 0, -1], 10.0, TRI)[0] is True
    assert probe_segment_triangle([-2e-6, .25, 1], [0, 0, -1], 10.0, TRI)[0] is False
    assert probe_segment_triangle([.25, -5e-7, 1], [0, 0, -1], 10.0, TRI)[0] is True
    assert probe_segment_triangle([.25, -2e-6, 1], [0, 0, -1], 10.0, TRI)[0] is False
    assert probe_segment_triangle([.5, .5, 1], [0, 0, -1], 10.0, TRI)[0] is True          # u+v == 1
    assert probe_segment_triangle([.5, .500002, 1], [0, 0, -1], 10.0, TRI)[0] is False
    # no back-face culling: det = -1
    assert probe_segment_triangle([.25, .25, -1], [0, 0, 1], 10.0, TRI) == (True, np.float32(1.0))
    # |det| < 1e-6: parallel ray
    assert probe_segment_triangle([.25, .25, 1], [1, 0, 0], 10.0, TRI)[0] is False
    # unnormalised direction: t is in units of |d|
    assert probe_segment_triangle([.25, .25, 1], [0, 0, -4], 10.0, TRI) == (True, np.float32(0.25))


def test_shadow_triangle_known_answers():
    # linear_bvh.go:173-200: no slack on u,v; 1e-5 < t < tmax strictly; divides by a
    assert probe_shadow_triangle([.25, .25, 1], [0, 0, -1], 10.0, TRI) is True
    assert probe_shadow_triangle([.25, .25, 1], [0, 0, -1], 1.0, TRI) is False        # t < tmax is strict
    assert probe_shadow_triangle([.25, .25, 1], [0, 0, -1], np.float32(1.0000001), TRI) is True
    assert probe_shadow_triangle([-5e-7, .25, 1], [0, 0, -1], 10.0, TRI) is False     # u < 0, no tolerance
    assert probe_shadow_triangle([0, 0, 1], [0, 0, -1], 10.0, TRI) is True            # u = v = 0 allowed
    assert probe_shadow_triangle([.5, .5, 1], [0, 0, -1], 10.0, TRI) is True          # u+v = 1 allowed
    assert probe_shadow_triangle([.25, .25, 5e-6], [0, 0, -1], 10.0, TRI) is False    # t <= 1e-5
    assert probe_shadow_triangle([.25, .25, -1], [0, 0, 1], 10.0, TRI) is True        # no culling


def test_intersect_aabb_known_answers():
    lo, hi = [0, 0, 0], [1, 1, 1]
    # linear_bvh.go:23-70 by hand: lb=(1,-inf,-inf) rb=(2,+inf,+inf) -> tMin=1 tMax=2
    assert probe_aabb(lo, hi, [-1, .5, .5], [1, 0, 0], 10.0) == np.float32(1.0)
    assert probe_aabb(lo, hi, [-1, .5, .5], [1, 0, 0], 0.5) == INF            # tMin > min(tMax, step)
    assert probe_aabb(lo, hi, [-1, .5, .5], [1, 0, 0], 1.0) == np.float32(1.0)
    assert probe_aabb(lo, hi, [.5, .5, .5], [1, 0, 0], 10.0) == np.float32(0.0)   # inside: max(0, tMin)
    assert probe_aabb(lo, hi, [2, .5, .5], [1, 0, 0], 10.0) == INF            # behind: tMax < 0
    assert probe_aabb(lo, hi, [-1, 2, .5], [1, 0, 0], 10.0) == INF            # parallel, outside the slab
    # origin exactly on a slab plane of an axis-parallel ray: 0 * inf = NaN flows through Go's max
    assert np.isnan(probe_aabb(lo, hi, [-1, 0, .5], [1, 0, 0], 10.0))
    # ... unless tMax < 0 decides first (linear_bvh.go:60)
    assert probe_aabb(lo, hi, [2, 0, .5], [1, 0, 0], 10.0) == INF
    # NaN only on the tMax side: tMin survives (gomin gives NaN, comparisons are false)
    assert probe_aabb(lo, hi, [-1, 1, .5], [1, 0, 0], 10.0) == np.float32(1.0)
    # the miss value is MaxFloat32, not +Inf
    assert np.isfinite(probe_aabb(lo, hi, [2, .5, .5], [1, 0, 0], 10.0))


def test_first_found_wins_ties():
    # two coincident triangles in one (root) leaf: strict t < best_t keeps the first (linear_bvh.go:133)
    v = np.concatenate([TRI, TRI]); f = np.array([[3, 4, 5], [0, 1, 2]], np.uint32)
    orc = oracle_lib.OracleBVH(v, f)
    prim, t, _ = orc.closest_hit([[.25, .25, 1]], [[0, 0, -1]], 10.0)
    assert prim[0] == 0 and t[0] == 1.0
    occ, _ = orc.any_hit([[.25, .25, 1]], [[0, 0, -1]], 10.0)
    assert occ[0]
    # a miss returns (false, 0, nil) -> (-1, 0)
    prim, t, _ = orc.closest_hit([[5, 5, 1]], [[0, 0, -1]], 10.0)
    assert prim[0] == -1 and t[0] == 0.0


def test_root_box_is_never_tested():
    # linear_bvh.go:118-119 pushes node 0 unconditionally: a hit outside the caller's root bounds counts
    v = (TRI + np.float32(5000.0)).astype(np.float32); f = np.array([[0, 1, 2]], np.uint32)
    orc = oracle_lib.OracleBVH(v, f)
    prim, t, _ = orc.closest_hit([[5000.25, 5000.25, 5001]], [[0, 0, -1]], 10.0)
    assert prim[0] == 0


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_vectors(path):
    g = np.load(path)
    orc = oracle_lib.OracleBVH(g["verts"], g["tris"])
    lin = orc.linear()
    for k in ("node_bounds", "is_leaf", "tri_offset", "tri_count", "second_child", "tri_ids"):
        assert bits_equal(getattr(lin, k), g[k]), k
    prim, t, cc = orc.closest_hit(g["o"], g["d"], g["tmax"])
    assert np.array_equal(prim, g["prim"]) and bits_equal(t, g["t"])
    assert [cc["inner_visits"], cc["tri_tests"], cc["leaf_visits"]] == g["closest_counters"].tolist()
    occ, ca = orc.any_hit(g["o"], g["d"], g["tmax"])
    assert np.array_equal(occ, g["occluded"].astype(bool))
    assert [ca["inner_visits"], ca["tri_tests"], ca["leaf_visits"]] == g["any_counters"].tolist()


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_product_builder_reproduces_golden_tree(path):
    from pathtrace_amd import BuildBVH
    g = np.load(path)
    lin = BuildBVH(g["verts"], g["tris"]).linear()
    for k in ("node_bounds", "is_leaf", "tri_offset", "tri_count", "second_child", "tri_ids"):
        assert bits_equal(getattr(lin, k), g[k]), k


def test_generated_scenes_are_reproducible():
    """The procedural generators are pure functions of their seed (the fixtures embed their output)."""
    from pathtrace_amd import scenes
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "soup2500_mixed.npz"))
    v, f = scenes.make_soup(2500, 102)
    assert bits_equal(v, g["verts"]) and np.array_equal(f, g["tris"])


# ---- textures (obj_loader.go:11-17, renderer.go:122-169, :199-291): hand-derived known answers ----

def _tlib():
    return oracle_lib._load_trace()


def test_tile_wraps_like_modf():
    lib = _tlib()
    f = np.float32
    assert lib.orc_probe_tile(f(0.25)) == f(0.25)
    assert lib.orc_probe_tile(f(3.75)) == f(0.75)
    assert lib.orc_probe_tile(f(-0.25)) == f(1) + f(-0.25)          # negative fractions wrap up (obj_loader.go:13-15)
    assert lib.orc_probe_tile(f(-2.5)) == f(0.5)
    assert lib.orc_probe_tile(f(2.0)) == f(0.0)


def test_sample_diffuse_map_truncates_and_packs():
    lib = _tlib()
    img = np.arange(4 * 3, dtype=np.uint32).reshape(3, 4) * np.uint32(0x01020304)       # [h=3, w=4]
    p = img.ctypes.data_as(oracle_lib._U32)
    # px = int(x*(W-1)), py = int(y*(H-1))  (renderer.go:123-124): 0.5*3 = 1.5 -> 1; 0.99*2 = 1.98 -> 1
    assert lib.orc_probe_texel(p, 4, 3, np.float32(0.5), np.float32(0.99)) == img[1, 1]
    assert lib.orc_probe_texel(p, 4, 3, np.float32(1.0), np.float32(1.0)) == img[2, 3]
    assert lib.orc_probe_texel(p, 4, 3, np.float32(0.0), np.float32(0.0)) == img[0, 0]
    # restatement-only clamp (the reference would index out of the slice)
    assert lib.orc_probe_texel(p, 4, 3, np.float32(np.nan), np.float32(7.0)) == img[2, 0]


def test_sample_bump_map_gradient():
    lib = _tlib()
    # grey ramp along x: texel (x, y) = 10*x in R,G,B -> height = 10*x/255 * (0.299+0.587+0.114)
    w, h = 8, 4
    g = (np.arange(w, dtype=np.uint32) * 10)[None, :].repeat(h, 0)
    img = np.ascontiguousarray(g | (g << 8) | (g << 16) | np.uint32(0xFF000000))
    out = np.zeros(3, np.float32)
    x, y = np.float32(0.4), np.float32(0.4)
    lib.orc_probe_bump(img.ctypes.data_as(oracle_lib._U32), w, h, x, y, np.float32(1.0), out.ctypes.data_as(oracle_lib._F))
    f = np.float32
    def height(px):
        v = f(10 * px)
        return f(f(f(f(0.299) * v + f(0.587) * v) + f(0.114) * v) / f(255))
    pc = int(x * f(w - 1)); pr = int(np.fmod(x + f(1) / f(w), f(1)) * f(w - 1))
    dx = f(height(pr) - height(pc))                         # up neighbour has the same height: dy = 0
    n = np.array([-dx, f(-0.0), f(1)], np.float32)
    ln = f(np.sqrt(f(f(n[0] * n[0] + n[1] * n[1]) + n[2] * n[2])))
    want = (n * f(f(1) / ln)).astype(np.float32)
    assert np.array_equal(out, want), (out, want)
    assert out[0] < 0 and out[2] > 0.9


def test_transform_normal_identity_fallback_and_frame():
    lib = _tlib()
    F = oracle_lib._F
    def tbn(tn, wn, abc, uv):
        a = [np.ascontiguousarray(x, np.float32) for x in (tn, wn, abc, uv)]
        out = np.zeros(3, np.float32)
        lib.orc_probe_tbn(*[x.ctypes.data_as(F) for x in a], out.ctypes.data_as(F))
        return out
    tri = [0, 0, 0, 1, 0, 0, 0, 0, 1]                       # A, B = +x, C = +z; geometric normal -y / +y
    wn = [0, 1, 0]
    # (0,0,1) in tangent space is returned untouched (renderer.go:201-203)
    assert np.array_equal(tbn([0, 0, 1], wn, tri, [0, 0, 1, 0, 0, 1]), np.float32(wn))
    # UVs aligned with the edges: tangent = +x, calculated bitangent = n x t = (0,1,0)x(1,0,0) = -z, UV bitangent = +z
    # -> flipped to +z (renderer.go:273-279); tangent-space (1,0,0) -> +x, (0,1,0) -> +z
    assert np.allclose(tbn([1, 0, 0], wn, tri, [0, 0, 1, 0, 0, 1]), [1, 0, 0], atol=1e-6)
    assert np.allclose(tbn([0, 1, 0], wn, tri, [0, 0, 1, 0, 0, 1]), [0, 0, 1], atol=1e-6)
    # degenerate UVs -> arbitrary frame from up = X because |n.y| >= 0.9 (renderer.go:229-250):
    # tangent = n x (1,0,0) = (0,0,-1), bitangent = n x tangent = (-1,0,0)
    assert np.allclose(tbn([1, 0, 0], wn, tri, [0.25] * 6), [0, 0, -1], atol=1e-6)
    assert np.allclose(tbn([0, 1, 0], wn, tri, [0.25] * 6), [-1, 0, 0], atol=1e-6)
    got = tbn([0.6, 0, 0.8], wn, tri, [0, 0, 1, 0, 0, 1])
    assert np.allclose(got, [0.6, 0.8, 0], atol=1e-6) and abs(np.linalg.norm(got) - 1) < 1e-6


def test_tile_worker_renderer_equals_the_pixel_loop():
    """The CPU baseline's goroutine-style tile renderer (one thread per tile, main.go:454-491, :832-899) adds the same
    samples in the same order as the OpenMP pixel loop the parity tests use: identical sums, full coverage when the
    splits divide the image, uncovered remainder otherwise (as in the reference)."""
    from pathtrace_amd import scenes
    sc = scenes.bunny_scene(small=True, glass=True, emissive=True)
    osc = oracle_lib.OracleScene(sc)
    p = osc.params(48, 40, bounces=2, seed=3)
    acc, cnt, oc = osc.render(sc.camera.as_array(), p, spp_count=3)
    acc2, cnt2, oc2, secs = osc.render_tile_workers(sc.camera.as_array(), p, splits=(4, 4), spp=3)
    assert secs > 0 and np.array_equal(acc, acc2) and np.array_equal(cnt, cnt2) and oc == oc2
    acc3, cnt3, _, _ = osc.render_tile_workers(sc.camera.as_array(), p, splits=(5, 3), spp=2)      # 48/5 = 9, 40/3 = 13
    assert (cnt3[:39, :45] == 2).all() and (cnt3[39:] == 0).all() and (cnt3[:, 45:] == 0).all()
