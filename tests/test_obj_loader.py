"""LoadObj / DecomposeObjects restated in csrc/obj_load.cpp (obj_loader.go:19-111, obj.go:405-939,
renderer.go:293-324): hand-written OBJ/MTL files with hand-derived expected arrays.  CPU only."""
import struct
import zlib

import numpy as np
import pytest

from pathtrace_amd import PtmiError
from pathtrace_amd.obj import LoadObj, read_image
from pathtrace_amd.scenes import MAT_GLASS, MAT_ACCRETION

OBJ = """# two objects; only the first one reaches the renderer (obj_loader.go:82)
mtllib scene.mtl
v 0 0 0
v 1 0 0
v 0 1 0
v 0 0 2
vn 0 0 3
vn 0 2 0
vt 0.25 0.75
vt 1.5 -0.5
vt 0 0
o first
usemtl GlassBlue
s off
f 1/1/1 2/2/1 3/3/2
usemtl matte
f -1/3/2 1/1/1 2/-2/-2
g second
usemtl AccretionDisk
f 1/1/1 3/3/1 4/2/1
"""
MTL = """newmtl GlassBlue
Kd 0.1 0.2 0.3
Ks 0.4 0.5 0.6
Ke 0 0 0
Ni 1.5
Ns 96.5
illum 4
d 0.5
Tf 1 1 1
newmtl matte
Kd 0.7 0.6 0.5
Ke 2 3 4
map_Kd -s 1 1 1 wall.ppm
map_Bump -bm 0.2 bump.png
"""


def _write_ppm(path, rgb):
    h, w, _ = rgb.shape
    open(path, "wb").write(b"P6\n# comment\n%d %d\n255\n" % (w, h) + rgb.astype(np.uint8).tobytes())


def _write_png(path, px, filters):
    """8-bit PNG writer that applies the given filter type per row (to exercise all five decoders)."""
    h, w, nch = px.shape
    raw = bytearray()
    prev = np.zeros(w * nch, np.int32)
    for y in range(h):
        line = px[y].reshape(-1).astype(np.int32)
        ft = filters[y % len(filters)]
        a = np.concatenate([np.zeros(nch, np.int32), line[:-nch]])
        c = np.concatenate([np.zeros(nch, np.int32), prev[:-nch]])
        if ft == 0: pred = 0
        elif ft == 1: pred = a
        elif ft == 2: pred = prev
        elif ft == 3: pred = (a + prev) >> 1
        else:
            p0 = a + prev - c
            pa, pb, pc = abs(p0 - a), abs(p0 - prev), abs(p0 - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        raw += bytes([ft]) + ((line - pred) & 255).astype(np.uint8).tobytes()
        prev = line
    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[nch]
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))


@pytest.fixture
def asset(tmp_path):
    (tmp_path / "scene.obj").write_text(OBJ)
    (tmp_path / "scene.mtl").write_text(MTL)
    rng = np.random.default_rng(5)
    wall = rng.integers(0, 256, (3, 5, 3))
    bump = rng.integers(0, 256, (6, 4, 4))
    _write_ppm(tmp_path / "wall.ppm", wall)
    _write_png(tmp_path / "bump.png", bump, [0, 1, 2, 3, 4])
    return tmp_path, wall, bump


def test_load_obj_arrays(asset):
    d, wall, bump = asset
    sc = LoadObj(str(d / "scene.obj"), scale_factor=2.0, position=(10.0, 0.0, -1.0), load_textures=False)
    f = np.float32
    # vertices: float32(v) * scale (obj_loader.go:76-78) + Position (renderer.go:301-303); all four are kept
    want_v = (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 2]], f) * f(2.0) + np.array([10, 0, -1], f)).astype(f)
    assert np.array_equal(sc.verts, want_v)
    # only the two faces of the first object; -1 = last vertex so far (4 -> index 3), "-2" of 3 vt = index 1
    assert sc.tris.tolist() == [[0, 1, 2], [3, 0, 1]]
    # normals are normalised per vn (obj_loader.go:68-74) and gathered per corner
    assert np.array_equal(sc.normals, np.array([[0, 0, 1], [0, 0, 1], [0, 1, 0], [0, 1, 0], [0, 0, 1], [0, 0, 1]], f))   # "-2" of two vn = the first
    # uv: u as is, v -> 1 - v (obj_loader.go:91-92)
    vt = np.array([[0.25, 0.75], [1.5, -0.5], [0, 0]], f)
    flip = lambda i: [vt[i][0], f(1.0) - vt[i][1]]
    assert np.array_equal(sc.uvs, np.array([flip(0) + flip(1) + flip(2), flip(2) + flip(0) + flip(1)], f))
    # materials in first-seen order: usemtl order in the OBJ, then new names of the MTL
    assert sc.names == ["GlassBlue", "matte", "AccretionDisk"]
    assert sc.tri_material.tolist() == [0, 1]
    m = sc.materials
    assert np.array_equal(m[0, :11], np.array([0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0, 0, 0, 1.5, 96.5], f))
    assert np.array_equal(m[1, :9], np.array([0.7, 0.6, 0.5, 0, 0, 0, 2, 3, 4], f))
    flags = m.view(np.uint32)[:, 11]
    assert flags.tolist() == [MAT_GLASS, 0, MAT_ACCRETION]        # by NAME: prefix "Glass", == "AccretionDisk"
    assert (m[2, :11] == 0).all()                                 # named by usemtl, absent from the MTL: zero material
    # map_Kd keeps the FIRST field after the keyword, map_Bump the LAST (obj.go:917 vs :929): the option
    # "-s" is therefore what the reference would try to open for map_Kd -- kept, and reported
    assert sc.map_kd[1] == "-s" and sc.map_bump[1] == "bump.png"
    assert any("Tf" in w for w in sc.warnings)


def test_textures_are_decoded_and_bound(asset, tmp_path):
    d, wall, bump = asset
    (d / "scene.mtl").write_text(MTL.replace("map_Kd -s 1 1 1 wall.ppm", "map_Kd wall.ppm"))
    sc = LoadObj(str(d / "scene.obj"))
    maps = sc.materials.view(np.uint32)[:, 12:14]
    assert maps.tolist() == [[0, 0], [2, 1], [0, 0]]              # bump decoded first (obj_loader.go:29), 1-based
    b, w = sc.textures
    assert b.shape == (6, 4) and w.shape == (3, 5)
    assert np.array_equal(w, (wall[..., 0] | (wall[..., 1] << 8) | (wall[..., 2] << 16) | (255 << 24)).astype(np.uint32))
    assert np.array_equal(b, (bump[..., 0] | (bump[..., 1] << 8) | (bump[..., 2] << 16) | (bump[..., 3] << 24)).astype(np.uint32))


def test_image_readers_grey_and_rgb(tmp_path):
    rng = np.random.default_rng(1)
    g = rng.integers(0, 256, (4, 7, 1))
    _write_png(tmp_path / "g.png", g, [4, 3, 1])
    got = read_image(tmp_path / "g.png")
    assert np.array_equal(got & 0xFF, g[..., 0]) and np.array_equal((got >> 8) & 0xFF, g[..., 0]) and (got >> 24 == 255).all()
    rgb = rng.integers(0, 256, (5, 3, 3))
    _write_png(tmp_path / "c.png", rgb, [2, 4])
    assert np.array_equal(read_image(tmp_path / "c.png") & 0xFFFFFF, rgb[..., 0] | (rgb[..., 1] << 8) | (rgb[..., 2] << 16))
    open(tmp_path / "g.pgm", "wb").write(b"P5 7 4 255\n" + g.astype(np.uint8).tobytes())
    assert np.array_equal(read_image(tmp_path / "g.pgm") & 0xFF, g[..., 0])


def test_default_material_and_mtl_fallback(tmp_path):
    body = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nusemtl GlassX\nf 1//1 2//1 3//1\n"
    (tmp_path / "a.obj").write_text(body)                         # no mtllib, no a.mtl -> defaultMat (obj.go:82-87, :209-219)
    sc = LoadObj(str(tmp_path / "a.obj"))
    assert np.array_equal(sc.materials[0, :11], np.array([.7, .7, .7, .5, .5, .5, 0, 0, 0, 0, 30], np.float32))
    assert sc.materials.view(np.uint32)[0, 11] == 0               # defaultMat has no name: no glass detection
    assert sc.uvs is None and any("default material" in w for w in sc.warnings)
    (tmp_path / "a.mtl").write_text("newmtl GlassX\nKd 1 0 0\n")  # <name>.mtl is tried when mtllib is absent (obj.go:186-206)
    sc = LoadObj(str(tmp_path / "a.obj"))
    assert sc.materials[0, :3].tolist() == [1, 0, 0] and sc.materials.view(np.uint32)[0, 11] == MAT_GLASS
    assert sc.tris.tolist() == [[0, 1, 2]]                        # the face opened the object "unnamed<line>"


@pytest.mark.parametrize("body,msg", [
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvn 0 0 1\nusemtl m\nf 1//1 2//1 3//1 4//1\n", "not a triangle"),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl m\nf 1 2 3\n", "without a normal"),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n", "without usemtl"),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nusemtl m\nf 1//1 2//1 3//1\n", "without vt"),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nusemtl m\nf 1//1 2//1 9//1\n", "out of range"),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nusemtl m\nf 0//1 2//1 3//1\n", "equal to 0"),
    ("v 0 0\n", "Less than 3"),
])
def test_what_panics_in_the_reference_is_an_error_here(tmp_path, body, msg):
    (tmp_path / "bad.obj").write_text(body)
    with pytest.raises(PtmiError) as e:
        LoadObj(str(tmp_path / "bad.obj"))
    assert msg in str(e.value)


def test_missing_file(tmp_path):
    with pytest.raises(PtmiError):
        LoadObj(str(tmp_path / "nope.obj"))
