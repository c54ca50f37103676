"""The C-ABI libraries load and export every symbol the public headers declare (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest

from raytrace_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.findall(r"RT_API\s+[\w\s\*]+?\b(rt_\w+)\s*\(", text)


def test_device_library_exports_every_declared_symbol():
    names = declared("raytrace_amd.h")
    assert len(names) >= 24 and set(_lib.DEVICE_SYMBOLS) == set(names)
    test_names = declared("raytrace_amd_test.h")
    assert set(_lib.DEVICE_TEST_SYMBOLS) == set(test_names) and not set(test_names) & set(names)
    lib = ctypes.CDLL(_lib.DEVICE_LIB)
    for n in names + test_names:
        assert hasattr(lib, n), n


def test_host_library_exports_every_declared_symbol():
    names = declared("raytrace_amd_host.h")
    assert set(names) == {n for n in _lib.HOST_SYMBOLS if n.startswith("rt_host_")}
    assert {"rt_host_camera", "rt_host_build_scene", "rt_host_scene_camera", "rt_host_gen_rays",
            "rt_host_blit_nearest", "rt_host_write_ppm"} <= set(names)
    lib = ctypes.CDLL(_lib.HOST_LIB)
    for n in names + ["rt_gl_bind_buffer", "rt_gl_buffer_sub_data", "rt_gl_buffer_data"]:
        assert hasattr(lib, n), n


def test_struct_layouts_match_the_header():
    assert ctypes.sizeof(_lib.RtParams) == 136
    assert _lib.RtParams.delta_u.offset == 40 and _lib.RtParams.seed_tex.offset == 128
    assert ctypes.sizeof(_lib.RtCounters) == 20 * 8
    assert ctypes.sizeof(_lib.RtCamera) == 52


def test_no_gpu_means_loud_failure():
    """Without a device rt_init must fail with RT_ERR_NO_GPU; nothing falls back to a CPU path."""
    lib = _lib.device_lib()
    if lib.rt_device_count() > 0:
        pytest.skip("a GPU is visible")
    ctx = ctypes.c_void_p()
    assert lib.rt_init(0, ctypes.byref(ctx)) == -4
    assert b"no HIP device" in lib.rt_last_error()
    assert not ctx.value


def test_product_never_imports_the_oracle():
    """The shipped package and the native sources must not reference oracle/."""
    pkg = os.path.join(ROOT, "raytrace_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f == "build.py":  # compiles the checker (building it is not using it)
                continue
            if f.endswith((".py", ".hip", ".cuh", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "rt_oracle" not in text and "orc_" not in text, f
