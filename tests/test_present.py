"""Presentation pass of the reference (raytrace.cpp:545-551: the render texture drawn over the window with
GL_NEAREST) and the PPM writer.  Host code only."""
import numpy as np
import pytest

import raytrace_amd as rt
from raytrace_amd import _lib


def test_window_height_is_the_references():
    L = _lib.host_lib()
    assert L.rt_host_height_for_width(1200) == 675   # SCREEN_WIDTH, raytrace.cpp:84,124
    assert L.rt_host_height_for_width(320) == 180    # RENDER_WIDTH, raytrace.cpp:87,127
    assert L.rt_host_height_for_width(1) == 1        # the clamp at :125


def test_blit_identity_and_integer_zoom():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(9, 16, 3), dtype=np.uint8)
    L = _lib.host_lib()
    same = np.empty_like(img)
    assert L.rt_host_blit_nearest(img.ctypes.data, 16, 9, same.ctypes.data, 16, 9) == 0
    assert (same == img).all()
    big = np.empty((27, 48, 3), dtype=np.uint8)
    assert L.rt_host_blit_nearest(img.ctypes.data, 16, 9, big.ctypes.data, 48, 27) == 0
    assert (big == np.repeat(np.repeat(img, 3, axis=0), 3, axis=1)).all()


def test_blit_matches_texel_under_pixel_centre():
    """320x180 -> 1200x675, the reference's sizes: every window pixel shows the texel under its centre."""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(180, 320, 3), dtype=np.uint8)
    out = rt.present(img, 1200)
    assert out.shape == (675, 1200, 3)
    from fractions import Fraction
    for (i, jt) in [(0, 0), (7, 3), (1199, 674), (600, 337), (8, 674), (1192, 1)]:
        j = 674 - jt
        c = int(Fraction(2 * i + 1, 2 * 1200) * 320)
        r = int(Fraction(2 * j + 1, 2 * 675) * 180)
        assert (out[jt, i] == img[179 - r, c]).all()
    # rows keep their orientation: the top row of the window comes from the top row of the image
    assert (out[0, 0] == img[0, 0]).all() and (out[-1, -1] == img[-1, -1]).all()


def test_blit_rejects_bad_arguments():
    L = _lib.host_lib()
    a = np.zeros((2, 2, 3), dtype=np.uint8)
    assert L.rt_host_blit_nearest(None, 2, 2, a.ctypes.data, 2, 2) != 0
    assert L.rt_host_blit_nearest(a.ctypes.data, 0, 2, a.ctypes.data, 2, 2) != 0


def test_ppm_round_trip(tmp_path):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, size=(5, 7, 3), dtype=np.uint8)
    path = tmp_path / "frame.ppm"
    rt.write_ppm(str(path), img)
    raw = path.read_bytes()
    head = b"P6\n7 5\n255\n"
    assert raw.startswith(head) and raw[len(head):] == img.tobytes()
    with pytest.raises(rt.RtError):
        rt.write_ppm(str(tmp_path / "missing_dir" / "x.ppm"), img)
