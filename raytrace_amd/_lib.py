"""ctypes bindings of the two native libraries.

  libraytrace_amd.so  include/raytrace_amd.h       (HIP; the product)
  librt_host.so       include/raytrace_amd_host.h  (plain C++ host harness)

The libraries are built in-tree by raytrace_amd/build.py and are required:
nothing here falls back to another implementation when one is missing.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# RAYTRACE_AMD_LIB names another build of the same library (kernel tuning experiments).
DEVICE_LIB = os.environ.get("RAYTRACE_AMD_LIB") or os.path.join(HERE, "libraytrace_amd.so")
# RAYTRACE_AMD_HOST_LIB names another build of the host harness (the sanitizer build of `make -C oracle ubsan`).
HOST_LIB = os.environ.get("RAYTRACE_AMD_HOST_LIB") or os.path.join(HERE, "librt_host.so")

c_float3 = ctypes.c_float * 3


class RtParams(ctypes.Structure):
    """struct rt_params of include/raytrace_amd.h."""
    _fields_ = [
        ("width", ctypes.c_int32), ("height", ctypes.c_int32),
        ("tile_x0", ctypes.c_int32), ("tile_y0", ctypes.c_int32),
        ("tile_x1", ctypes.c_int32), ("tile_y1", ctypes.c_int32),
        ("sample_begin", ctypes.c_int32), ("sample_end", ctypes.c_int32),
        ("bounce_limit", ctypes.c_uint32), ("time_u32t", ctypes.c_uint32),
        ("delta_u", c_float3), ("delta_v", c_float3), ("camera_origin", c_float3),
        ("viewport_top_left", c_float3), ("defocus_disk_u", c_float3), ("defocus_disk_v", c_float3),
        ("defocus_angle", ctypes.c_float),
        ("rng_mode", ctypes.c_int32), ("seed", ctypes.c_uint32),
        ("strip_count", ctypes.c_uint16), ("strip_index", ctypes.c_uint16),
        ("seed_tex", ctypes.c_void_p),
    ]

    def copy(self):
        other = RtParams()
        ctypes.memmove(ctypes.byref(other), ctypes.byref(self), ctypes.sizeof(RtParams))
        return other


COUNTER_FIELDS = ["paths", "segments", "sphere_tests", "roots", "node_tests", "hits", "sky", "lambertian",
                  "metallic", "dialectric", "perlin", "unit_tries", "disk_tries", "limit_hits",
                  "trav_wave_steps", "trav_lane_steps", "path_wave_phases", "path_lane_phases",
                  "leaf_wave_steps", "leaf_lane_steps"]


class RtCounters(ctypes.Structure):
    """struct rt_counters of include/raytrace_amd.h."""
    _fields_ = [(name, ctypes.c_uint64) for name in COUNTER_FIELDS]

    def as_dict(self):
        return {name: int(getattr(self, name)) for name in COUNTER_FIELDS}


class RtSceneInfo(ctypes.Structure):
    """struct rt_scene_info of include/raytrace_amd.h."""
    _fields_ = [(n, ctypes.c_int32) for n in ("n_spheres", "n_materials", "accel", "builder", "n_big", "n_leaf", "n_nodes",
                                               "depth", "tree_in_lds", "pad_per_ray", "node_bytes")] + [("reserved", ctypes.c_int32 * 5)]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_[:-1]}
        d["builder"] = {0: "none", 1: "host_sah", 2: "device_lbvh"}[d["builder"]]
        return d


class RtCamera(ctypes.Structure):
    """struct rt_camera of include/raytrace_amd_host.h."""
    _fields_ = [
        ("lookfrom", c_float3), ("lookat", c_float3), ("vup", c_float3),
        ("vfov", ctypes.c_float), ("defocus_angle", ctypes.c_float), ("focus_dist", ctypes.c_float),
        ("origin_is_lookat", ctypes.c_int32),
    ]


# Entry points every build of the device library must export (checked by tests).
DEVICE_SYMBOLS = [
    "rt_last_error", "rt_init", "rt_destroy", "rt_gl_bind_buffer", "rt_gl_buffer_sub_data", "rt_gl_buffer_data",
    "rt_gl_buffer_bytes", "rt_gl_reset", "rt_gl_dropped_writes", "rt_scene_upload", "rt_scene_upload_objects", "rt_comm_unique_id", "rt_comm_init", "rt_comm_reduce", "rt_comm_destroy", "rt_render", "rt_render_counted",
    "rt_clear", "rt_set_accumulator", "rt_set_accumulated_samples", "rt_readback", "rt_trace_batch",
    "rt_trace_batch_device", "rt_synchronize", "rt_get_counters", "rt_last_kernel_ms", "rt_last_build_ms",
    "rt_lbvh_depth", "rt_device_count", "rt_seed_texture", "rt_accumulator_read", "rt_accumulator_write",
    "rt_render_chunk", "rt_trace_batch_noalloc", "rt_eval_perlin", "rt_eval_perlin_gradient", "rt_set_scratch_budget", "rt_get_scene_info", "rt_get_stream",
]
# ... and the test-only ones of include/raytrace_amd_test.h
DEVICE_TEST_SYMBOLS = ["rt_eval_sqrt", "rt_check_sqrt"]
HOST_SYMBOLS = ["rt_host_camera", "rt_host_build_scene", "rt_host_scene_object_bytes", "rt_host_scene_camera", "rt_host_gen_rays",
                "rt_host_blit_nearest", "rt_host_height_for_width", "rt_host_write_ppm",
                "rt_gl_bind_buffer", "rt_gl_buffer_sub_data", "rt_gl_buffer_data", "rt_gl_buffer_bytes", "rt_gl_reset"]

_device = None
_host = None


def _declare_capture(lib):
    lib.rt_gl_bind_buffer.restype = None
    lib.rt_gl_bind_buffer.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    lib.rt_gl_buffer_sub_data.restype = None
    lib.rt_gl_buffer_sub_data.argtypes = [ctypes.c_uint32, ctypes.c_ssize_t, ctypes.c_ssize_t, ctypes.c_void_p]
    lib.rt_gl_buffer_data.restype = None
    lib.rt_gl_buffer_data.argtypes = [ctypes.c_uint32, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_uint32]
    lib.rt_gl_buffer_bytes.restype = ctypes.c_void_p
    lib.rt_gl_buffer_bytes.argtypes = [ctypes.c_uint32, ctypes.POINTER(ctypes.c_size_t)]
    lib.rt_gl_reset.restype = None
    lib.rt_gl_reset.argtypes = []


def host_lib():
    """The host harness (no GPU needed)."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError(f"{HOST_LIB} is missing: run `python raytrace_amd/build.py`")
        lib = ctypes.CDLL(HOST_LIB)
        _declare_capture(lib)
        lib.rt_host_camera.restype = ctypes.c_int
        lib.rt_host_camera.argtypes = [ctypes.POINTER(RtCamera), ctypes.c_int, ctypes.c_int, ctypes.POINTER(RtParams)]
        lib.rt_host_scene_object_bytes.restype = ctypes.c_int
        lib.rt_host_scene_object_bytes.argtypes = []
        lib.rt_host_build_scene.restype = ctypes.c_int
        lib.rt_host_build_scene.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int)]
        lib.rt_host_scene_camera.restype = ctypes.c_int
        lib.rt_host_scene_camera.argtypes = [ctypes.c_int, ctypes.POINTER(RtCamera)]
        lib.rt_host_blit_nearest.restype = ctypes.c_int
        lib.rt_host_blit_nearest.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        lib.rt_host_height_for_width.restype = ctypes.c_int
        lib.rt_host_height_for_width.argtypes = [ctypes.c_int]
        lib.rt_host_write_ppm.restype = ctypes.c_int
        lib.rt_host_write_ppm.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        lib.rt_host_gen_rays.restype = ctypes.c_int
        lib.rt_host_gen_rays.argtypes = [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        _host = lib
    return _host


def device_lib():
    """The HIP library.  Raises if it has not been built; never substitutes anything for it."""
    global _device
    if _device is None:
        if not os.path.exists(DEVICE_LIB):
            raise RuntimeError(f"{DEVICE_LIB} is missing: run `python raytrace_amd/build.py` (no CPU fallback exists)")
        # PyTorch's wheel carries its own copy of the HIP runtime.  If torch is
        # going to be used in this process it must be loaded first, so that this
        # library binds to the runtime torch already loaded instead of bringing
        # in a second one.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(DEVICE_LIB)
        _declare_capture(lib)
        P = ctypes.POINTER
        lib.rt_last_error.restype = ctypes.c_char_p
        lib.rt_last_error.argtypes = []
        lib.rt_device_count.restype = ctypes.c_int
        lib.rt_init.restype = ctypes.c_int
        lib.rt_init.argtypes = [ctypes.c_int, P(ctypes.c_void_p)]
        lib.rt_destroy.restype = None
        lib.rt_destroy.argtypes = [ctypes.c_void_p]
        lib.rt_comm_unique_id.restype = ctypes.c_int
        lib.rt_comm_unique_id.argtypes = [ctypes.c_void_p]
        lib.rt_comm_init.restype = ctypes.c_int
        lib.rt_comm_init.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        lib.rt_comm_reduce.restype = ctypes.c_int
        lib.rt_comm_reduce.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.rt_comm_destroy.restype = ctypes.c_int
        lib.rt_comm_destroy.argtypes = [ctypes.c_void_p]
        lib.rt_scene_upload_objects.restype = ctypes.c_int
        lib.rt_scene_upload_objects.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                                ctypes.c_int, ctypes.c_int]
        lib.rt_scene_upload.restype = ctypes.c_int
        lib.rt_scene_upload.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        for name in ("rt_render", "rt_render_counted"):
            fn = getattr(lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, P(RtParams)]
        lib.rt_clear.restype = ctypes.c_int
        lib.rt_clear.argtypes = [ctypes.c_void_p]
        lib.rt_set_accumulator.restype = ctypes.c_int
        lib.rt_set_accumulator.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        lib.rt_set_accumulated_samples.restype = ctypes.c_int
        lib.rt_set_accumulated_samples.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.rt_readback.restype = ctypes.c_int
        lib.rt_readback.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.rt_trace_batch.restype = ctypes.c_int
        lib.rt_trace_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.rt_trace_batch_device.restype = ctypes.c_int
        lib.rt_trace_batch_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        lib.rt_trace_batch_noalloc.restype = ctypes.c_int
        lib.rt_trace_batch_noalloc.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.rt_set_scratch_budget.restype = ctypes.c_int
        lib.rt_set_scratch_budget.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]
        lib.rt_get_scene_info.restype = ctypes.c_int
        lib.rt_get_scene_info.argtypes = [ctypes.c_void_p, P(RtSceneInfo)]
        lib.rt_get_stream.restype = ctypes.c_int
        lib.rt_get_stream.argtypes = [ctypes.c_void_p, P(ctypes.c_void_p)]
        lib.rt_eval_perlin.restype = ctypes.c_int
        lib.rt_eval_perlin.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        lib.rt_eval_sqrt.restype = ctypes.c_int
        lib.rt_eval_sqrt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        lib.rt_check_sqrt.restype = ctypes.c_int
        lib.rt_check_sqrt.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64),
                                      ctypes.POINTER(ctypes.c_uint32)]
        lib.rt_eval_perlin_gradient.restype = ctypes.c_int
        lib.rt_eval_perlin_gradient.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        lib.rt_synchronize.restype = ctypes.c_int
        lib.rt_synchronize.argtypes = [ctypes.c_void_p]
        lib.rt_get_counters.restype = ctypes.c_int
        lib.rt_get_counters.argtypes = [ctypes.c_void_p, P(RtCounters)]
        lib.rt_last_kernel_ms.restype = ctypes.c_int
        lib.rt_last_kernel_ms.argtypes = [ctypes.c_void_p, P(ctypes.c_float), P(ctypes.c_char_p)]
        lib.rt_last_build_ms.restype = ctypes.c_int
        lib.rt_last_build_ms.argtypes = [ctypes.c_void_p, P(ctypes.c_float)]
        lib.rt_lbvh_depth.restype = ctypes.c_int
        lib.rt_lbvh_depth.argtypes = [ctypes.c_void_p]
        lib.rt_accumulator_read.restype = ctypes.c_int
        lib.rt_accumulator_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int),
                                            ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        lib.rt_accumulator_write.restype = ctypes.c_int
        lib.rt_accumulator_write.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.rt_render_chunk.restype = ctypes.c_int
        lib.rt_render_chunk.argtypes = [ctypes.c_void_p, ctypes.POINTER(RtParams), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int]
        lib.rt_seed_texture.restype = ctypes.c_int
        lib.rt_seed_texture.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        _device = lib
    return _device
