"""Host-side pieces of the headless frame loop (pathtrace_amd/driver.py): camera.go helpers, Humanize, PNG."""
import numpy as np

from pathtrace_amd import driver
from pathtrace_amd.obj import read_image


def test_humanize_matches_utils_go():
    assert driver.humanize(999) == "999.0" and driver.humanize(1500) == "1.5K"
    assert driver.humanize(2.5e6) == "2.5M" and driver.humanize(3e9) == "3.0G"


def test_spherical_around_basis():
    # camera.go:17-41 with the black-hole call of main.go:625-631: radius 1500, (phi, theta) = (90 deg, 83 deg)
    cam = driver.spherical_around((0, 0, 0), 1500.0, 90.0 * driver.DEG, 83.0 * driver.DEG)
    pos, f, r, u = (np.array(v) for v in (cam.position, cam.forward, cam.right, cam.up))
    assert abs(np.linalg.norm(pos) - 1500) < 1e-2 and abs(pos[1] - 1500 * np.cos(83 * driver.DEG)) < 1e-2
    assert np.allclose(f, -pos / np.linalg.norm(pos), atol=1e-6)
    for a, b in ((f, r), (f, u), (r, u)):
        assert abs(a @ b) < 1e-6
    assert np.allclose(r, np.cross([0, 1, 0], f) / np.linalg.norm(np.cross([0, 1, 0], f)), atol=1e-6)   # worldUp x forward
    assert np.allclose(u, np.cross(r, f), atol=1e-6)                                                     # right x forward


def test_apply_rotation_keeps_an_orthonormal_frame():
    cam = driver.spherical_around((0, 0, 0), 10.0, 0.3, 1.1)
    rot = driver.apply_rotation(cam, 0.4, -0.2)
    f, r, u = (np.array(v) for v in (rot.forward, rot.right, rot.up))
    assert all(abs(np.linalg.norm(v) - 1) < 1e-6 for v in (f, r, u))
    assert abs(f @ r) < 1e-6 and np.allclose(u, np.cross(r, f), atol=1e-6)
    # yaw only: rotateAroundY (camera.go:44-54) on forward
    y = driver.apply_rotation(cam, 0.5, 0.0)
    c, s = np.cos(0.5), np.sin(0.5)
    f0 = np.array(cam.forward)
    want = np.array([f0[0] * c + f0[2] * s, f0[1], -f0[0] * s + f0[2] * c])
    assert np.allclose(np.array(y.forward), want / np.linalg.norm(want), atol=1e-6)


def test_png_round_trip(tmp_path):
    rgba = np.random.default_rng(0).integers(0, 256, (9, 7, 4)).astype(np.uint8)
    driver.write_png(tmp_path / "f.png", rgba)
    got = read_image(tmp_path / "f.png")
    want = rgba[..., 0].astype(np.uint32) | (rgba[..., 1].astype(np.uint32) << 8) | (rgba[..., 2].astype(np.uint32) << 16) | (rgba[..., 3].astype(np.uint32) << 24)
    assert np.array_equal(got, want)


def test_framebuffer_checkpoint_round_trip(tmp_path):
    """The accumulation state of a frame in progress survives save / load bit for bit; a file of another shape is refused."""
    import pytest
    from pathtrace_amd.render import Framebuffer
    rng = np.random.default_rng(1)
    fb = Framebuffer(rng.random((5, 7, 3), dtype=np.float32) * 9, np.full((5, 7), 12, np.uint32))
    fb.counts[2, 3] = 11
    fb.save(tmp_path / "fb.npz")
    got = Framebuffer.load(tmp_path / "fb.npz")
    assert got.accum.dtype == np.float32 and got.counts.dtype == np.uint32
    assert np.array_equal(got.accum.view(np.uint32), fb.accum.view(np.uint32)) and np.array_equal(got.counts, fb.counts)
    assert got.samples_done == 11
    np.savez(tmp_path / "bad.npz", accum=np.zeros((5, 7), np.float32), counts=np.zeros((5, 7), np.uint32))
    with pytest.raises(ValueError):
        Framebuffer.load(tmp_path / "bad.npz")
