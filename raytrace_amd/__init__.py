"""raytrace_amd — MI355X-native sphere path tracer behind the not-bottle/raytrace scene API.

Python here is a thin driver over the C ABI (include/raytrace_amd.h) for tests
and benchmarks; the product is the HIP library.  Scene construction and camera
set-up run in the C++ host harness (raytrace_amd/host), rendering and hit tests
in HIP kernels (raytrace_amd/csrc).
"""
import ctypes
import os
import struct

import numpy as np

from . import _lib
from ._lib import RtCamera, RtCounters, RtParams  # noqa: F401

ACCEL_LINEAR, ACCEL_LBVH = 0, 1
RNG_XORSHIFT_REF, RNG_COUNTER = 0, 1
SCENE_DEFAULT, SCENE_RANDOM_SPHERES, SCENE_PERLIN, SCENE_CLOUD, SCENE_QUAD_ROOM, SCENE_QUAD_CLOUD = 0, 1, 2, 3, 4, 5
MATERIAL_BUFFER, SPHERE_BUFFER = 1, 2
MAT_NOTHING, MAT_LAMBERTIAN, MAT_METALLIC, MAT_DIALECTRIC, MAT_PERLIN_LAMBERTIAN = 0, 1, 2, 3, 4
RECORD_BYTES = 32


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rt error {code}: {message}")
        self.code = code


class Scene:
    """std140 bytes of a scene as the scene headers' add() calls produced them."""

    def __init__(self, spheres: bytes, materials: bytes):
        assert len(spheres) % RECORD_BYTES == 0 and len(materials) % RECORD_BYTES == 0
        self.spheres = spheres
        self.materials = materials

    @property
    def n_spheres(self):
        return len(self.spheres) // RECORD_BYTES

    @property
    def n_materials(self):
        return len(self.materials) // RECORD_BYTES


OBJECT_SPHERE, OBJECT_QUAD = 0, 1
QUAD_RECORD_BYTES = 64


def sphere_record(mat, radius, origin) -> bytes:
    """One 32-byte sphere record (mat @0, radius @4, kind 0 @8, origin @16)."""
    return struct.pack("<ifi4x3f4x", int(mat), float(radius), OBJECT_SPHERE, *[float(x) for x in origin])


def quad_record(mat, corner, edge_u, edge_v) -> bytes:
    """One 64-byte quad record (mat @0, kind 1 @8, Q @16, u @32, v @48)."""
    f = [float(x) for x in (*corner, *edge_u, *edge_v)]
    return struct.pack("<i4xi4x3f4x3f4x3f4x", int(mat), OBJECT_QUAD, *f)


def material_record(ident, kind, albedo=(0.0, 0.0, 0.0), param1=0.0) -> bytes:
    """One 32-byte material record (id @0, type @4, param1 @8, albedo @16)."""
    return struct.pack("<iif4x3f4x", int(ident), int(kind), float(param1), *[float(x) for x in albedo])


class ObjectScene(Scene):
    """A list of spheres and quads: records of rt_scene_upload_objects, back to back, kinds at byte 8."""

    def __init__(self, objects: bytes, n_objects: int, materials: bytes):
        assert len(materials) % RECORD_BYTES == 0
        self.objects = objects
        self.n_objects = int(n_objects)
        self.materials = materials

    @property
    def n_spheres(self):  # objects of either kind: what ids and counts refer to
        return self.n_objects


def build_scene(scene_id, n=0, seed=0) -> Scene:
    """Run a scene builder of the host harness (through the scene description API) and collect the captured bytes."""
    lib = _lib.host_lib()
    nm = ctypes.c_int(0)
    ns = lib.rt_host_build_scene(int(scene_id), int(n), int(seed), ctypes.byref(nm))
    if ns < 0:
        raise RtError(ns, "rt_host_build_scene failed")

    def grab(buffer_id, count):
        size = ctypes.c_size_t(0)
        ptr = lib.rt_gl_buffer_bytes(buffer_id, ctypes.byref(size))
        need = count * RECORD_BYTES
        if ptr is None or size.value < need:
            raise RtError(-3, f"captured buffer {buffer_id} holds {size.value} bytes, need {need}")
        return ctypes.string_at(ptr, need)

    if scene_id in (SCENE_QUAD_ROOM, SCENE_QUAD_CLOUD):
        nbytes = lib.rt_host_scene_object_bytes()
        size = ctypes.c_size_t(0)
        ptr = lib.rt_gl_buffer_bytes(SPHERE_BUFFER, ctypes.byref(size))
        if ptr is None or size.value < nbytes:
            raise RtError(-3, f"captured object buffer holds {size.value} bytes, need {nbytes}")
        scene = ObjectScene(ctypes.string_at(ptr, nbytes), ns, grab(MATERIAL_BUFFER, nm.value))
    else:
        scene = Scene(grab(SPHERE_BUFFER, ns), grab(MATERIAL_BUFFER, nm.value))
    lib.rt_gl_reset()
    return scene


def scene_camera(scene_id) -> RtCamera:
    cam = RtCamera()
    rc = _lib.host_lib().rt_host_scene_camera(int(scene_id), ctypes.byref(cam))
    if rc:
        raise RtError(rc, "rt_host_scene_camera failed")
    return cam


def make_params(cam: RtCamera, width, height, samples, bounce_limit=50, rng_mode=RNG_COUNTER, seed=0x5EED0001,
                sample_begin=0) -> RtParams:
    """Camera block + uniform wiring (raytrace.cpp:341-366, 522-533) for a full-image render."""
    p = RtParams()
    rc = _lib.host_lib().rt_host_camera(ctypes.byref(cam), int(width), int(height), ctypes.byref(p))
    if rc:
        raise RtError(rc, "rt_host_camera failed")
    p.sample_begin = sample_begin
    p.sample_end = sample_begin + samples
    p.bounce_limit = bounce_limit
    p.rng_mode = rng_mode
    p.seed = seed
    p.seed_tex = None
    return p


def gen_rays(seed, first, n):
    """Synthetic hit-test batch of the host harness: (origins, directions), float32 [n,3] each."""
    o = np.empty((n, 3), dtype=np.float32)
    d = np.empty((n, 3), dtype=np.float32)
    rc = _lib.host_lib().rt_host_gen_rays(int(seed), int(first), int(n), o.ctypes.data, d.ctypes.data)
    if rc:
        raise RtError(rc, "rt_host_gen_rays failed")
    return o, d


def present(rgb8, window_width=1200):
    """The reference's presentation pass: the rendered RGB8 image drawn over its 16:9 window with nearest
    sampling (raytrace.cpp:84-88,545-551).  rgb8: uint8 [H,W,3], rows from the top."""
    src = np.ascontiguousarray(rgb8, dtype=np.uint8)
    sh, sw, _ = src.shape
    L = _lib.host_lib()
    dh = L.rt_host_height_for_width(int(window_width))
    dst = np.empty((dh, int(window_width), 3), dtype=np.uint8)
    rc = L.rt_host_blit_nearest(src.ctypes.data, sw, sh, dst.ctypes.data, int(window_width), dh)
    if rc:
        raise RtError(rc, "rt_host_blit_nearest failed")
    return dst


def write_ppm(path, rgb8):
    """Binary PPM of a uint8 [H,W,3] image, rows from the top."""
    img = np.ascontiguousarray(rgb8, dtype=np.uint8)
    rc = _lib.host_lib().rt_host_write_ppm(os.fsencode(path), img.ctypes.data, img.shape[1], img.shape[0])
    if rc:
        raise RtError(rc, f"cannot write {path}")


class Renderer:
    """One rt_ctx.  Every method is a direct call of the C ABI."""

    def __init__(self, device=0):
        self.lib = _lib.device_lib()
        self.ctx = ctypes.c_void_p()
        self._check(self.lib.rt_init(int(device), ctypes.byref(self.ctx)))
        self.width = self.height = 0

    def _check(self, rc):
        if rc != 0:
            raise RtError(rc, self.lib.rt_last_error().decode())

    def close(self):
        if self.ctx:
            self.lib.rt_destroy(self.ctx)
            self.ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, scene: Scene, accel=ACCEL_LINEAR):
        if isinstance(scene, ObjectScene):
            self._check(self.lib.rt_scene_upload_objects(self.ctx, scene.objects, len(scene.objects), scene.n_objects,
                                                         scene.materials, scene.n_materials, int(accel)))
            return
        self._check(self.lib.rt_scene_upload(self.ctx, scene.spheres, scene.n_spheres, scene.materials,
                                             scene.n_materials, int(accel)))

    def render(self, params: RtParams, counted=False, seed_tex=None):
        p = params.copy()
        if seed_tex is not None:
            seed_tex = np.ascontiguousarray(seed_tex, dtype=np.uint32)
            assert seed_tex.size == p.width * p.height
            p.seed_tex = seed_tex.ctypes.data
        fn = self.lib.rt_render_counted if counted else self.lib.rt_render
        self._check(fn(self.ctx, ctypes.byref(p)))
        self.width, self.height = p.width, p.height

    def clear(self):
        self._check(self.lib.rt_clear(self.ctx))

    def set_accumulator(self, device_ptr, width, height):
        self._check(self.lib.rt_set_accumulator(self.ctx, ctypes.c_void_p(device_ptr), int(width), int(height)))
        self.width, self.height = width, height

    def set_scratch_budget(self, split_bytes=0, probe_bytes=0):
        """Device memory the render schedule may take (0 = default); see rt_set_scratch_budget."""
        self._check(self.lib.rt_set_scratch_budget(self.ctx, int(split_bytes), int(probe_bytes)))

    def set_accumulated_samples(self, samples):
        self._check(self.lib.rt_set_accumulated_samples(self.ctx, int(samples)))

    def _accumulator_size(self):
        w, h = ctypes.c_int(), ctypes.c_int()
        self._check(self.lib.rt_accumulator_read(self.ctx, None, ctypes.byref(w), ctypes.byref(h), None))
        self.width, self.height = w.value, h.value

    def readback(self, linear=True, rgb8=True):
        self._accumulator_size()
        n = self.width * self.height * 3
        lin = np.empty(n, dtype=np.float32) if linear else None
        q = np.empty(n, dtype=np.uint8) if rgb8 else None
        self._check(self.lib.rt_readback(self.ctx, lin.ctypes.data if linear else None, q.ctypes.data if rgb8 else None))
        shape = (self.height, self.width, 3)
        return (lin.reshape(shape) if linear else None, q.reshape(shape) if rgb8 else None)

    def trace(self, origins, directions):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        assert d.shape[0] == n
        ids = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32)
        nrm = np.empty((n, 3), dtype=np.float32)
        self._check(self.lib.rt_trace_batch(self.ctx, o.ctypes.data, d.ctypes.data, n, ids.ctypes.data, t.ctypes.data,
                                            nrm.ctypes.data))
        return ids, t, nrm

    def trace_device(self, o_ptr, d_ptr, n, id_ptr, t_ptr, nrm_ptr, counted=False):
        self._check(self.lib.rt_trace_batch_device(self.ctx, ctypes.c_void_p(o_ptr), ctypes.c_void_p(d_ptr), int(n),
                                                   ctypes.c_void_p(id_ptr), ctypes.c_void_p(t_ptr),
                                                   ctypes.c_void_p(nrm_ptr), int(bool(counted))))

    def trace_noalloc(self, o_ptr, d_ptr, n, id_ptr, t_ptr, nrm_ptr):
        """rt_trace_batch_noalloc: the same batch without any allocation by the library (fixed per-GPU scratch)."""
        self._check(self.lib.rt_trace_batch_noalloc(self.ctx, ctypes.c_void_p(o_ptr), ctypes.c_void_p(d_ptr), int(n),
                                                    ctypes.c_void_p(id_ptr), ctypes.c_void_p(t_ptr),
                                                    ctypes.c_void_p(nrm_ptr)))

    def eval_perlin(self, xy, use_table=True):
        """perlin(x, y) of perlin.h:59-85 on the device for float32 [n,2] points."""
        xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        out = np.empty(xy.shape[0], dtype=np.float32)
        self._check(self.lib.rt_eval_perlin(self.ctx, xy.ctypes.data, xy.shape[0], int(bool(use_table)), out.ctypes.data))
        return out

    def eval_perlin_gradient(self, ixy):
        """randomGradient(ix, iy) of perlin.h:31-43 on the device for int32 [n,2] lattice points -> float32 [n,2]."""
        ixy = np.ascontiguousarray(ixy, dtype=np.int32).reshape(-1, 2)
        out = np.empty((ixy.shape[0], 2), dtype=np.float32)
        self._check(self.lib.rt_eval_perlin_gradient(self.ctx, ixy.ctypes.data, ixy.shape[0], out.ctypes.data))
        return out

    def eval_sqrt(self, x):
        """The kernels' square root (correctly rounded binary32) of float32 values, evaluated on the device."""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
        out = np.empty_like(x)
        self._check(self.lib.rt_eval_sqrt(self.ctx, x.ctypes.data, x.shape[0], out.ctypes.data))
        return out

    def check_sqrt(self, first_bits=0, count=1 << 32):
        """(mismatches, example bits) of the kernels' square root against the compiler's correctly rounded one over
        `count` bit patterns from `first_bits` on."""
        bad, ex = ctypes.c_uint64(0), ctypes.c_uint32(0)
        self._check(self.lib.rt_check_sqrt(self.ctx, first_bits, count, ctypes.byref(bad), ctypes.byref(ex)))
        return int(bad.value), int(ex.value)

    def synchronize(self):
        self._check(self.lib.rt_synchronize(self.ctx))

    def counters(self):
        c = RtCounters()
        self._check(self.lib.rt_get_counters(self.ctx, ctypes.byref(c)))
        return c.as_dict()

    def last_kernel_ms(self):
        ms = ctypes.c_float(0)
        name = ctypes.c_char_p()
        self._check(self.lib.rt_last_kernel_ms(self.ctx, ctypes.byref(ms), ctypes.byref(name)))
        return float(ms.value), (name.value or b"").decode()

    def last_build_ms(self):
        ms = ctypes.c_float(0)
        self._check(self.lib.rt_last_build_ms(self.ctx, ctypes.byref(ms)))
        return float(ms.value)

    def scene_info(self):
        info = _lib.RtSceneInfo()
        self._check(self.lib.rt_get_scene_info(self.ctx, ctypes.byref(info)))
        return info.as_dict()

    @staticmethod
    def comm_unique_id() -> bytes:
        """rt_comm_unique_id: the 128 bytes one rank makes and every rank of the group passes to comm_init."""
        buf = ctypes.create_string_buffer(128)
        rc = _lib.device_lib().rt_comm_unique_id(buf)
        if rc != 0:
            raise RtError(rc, _lib.device_lib().rt_last_error().decode())
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == 128
        self._check(self.lib.rt_comm_init(self.ctx, unique_id, int(rank), int(world)))

    def comm_reduce(self, root=0):
        """rt_comm_reduce: the ranks' accumulators summed onto `root`'s, queued on the context's stream."""
        self._check(self.lib.rt_comm_reduce(self.ctx, int(root)))

    def comm_destroy(self):
        self._check(self.lib.rt_comm_destroy(self.ctx))

    def stream(self):
        """The context's hipStream_t as an integer (wrap it with torch.cuda.ExternalStream to order torch work after it)."""
        st = ctypes.c_void_p()
        self._check(self.lib.rt_get_stream(self.ctx, ctypes.byref(st)))
        return st.value or 0

    def lbvh_depth(self):
        return int(self.lib.rt_lbvh_depth(self.ctx))

    def render_chunk(self, params: RtParams, c_min, c_max):
        """shader_chunk_pass of the reference (raytrace.cpp:598-630): pixels [c_min, c_max) of the frame."""
        self._check(self.lib.rt_render_chunk(self.ctx, ctypes.byref(params), int(c_min[0]), int(c_min[1]), int(c_max[0]),
                                             int(c_max[1])))

    def render_progressive(self, params: RtParams, chunk=(256, 256), samples_per_pass=None):
        """Generator: renders the frame chunk by chunk and pass by pass, yielding (samples_done, chunk_rect) after
        each chunk so that a caller can display or checkpoint in between (counter RNG)."""
        spp0, spp1 = params.sample_begin, params.sample_end
        step = samples_per_pass or (spp1 - spp0)
        for s in range(spp0, spp1, step):
            p = params.copy()
            p.sample_begin, p.sample_end = s, min(spp1, s + step)
            for y0 in range(0, params.height, chunk[1]):
                for x0 in range(0, params.width, chunk[0]):
                    rect = (x0, y0, min(params.width, x0 + chunk[0]), min(params.height, y0 + chunk[1]))
                    self.render_chunk(p, rect[:2], rect[2:])
                    yield p.sample_end, rect

    def checkpoint(self):
        """(sums float32 [H,W,3], samples): the state a later restore() continues from."""
        w, h, n = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self._check(self.lib.rt_accumulator_read(self.ctx, None, ctypes.byref(w), ctypes.byref(h), ctypes.byref(n)))
        sums = np.empty((h.value, w.value, 3), dtype=np.float32)
        self._check(self.lib.rt_accumulator_read(self.ctx, sums.ctypes.data, None, None, None))
        return sums, n.value

    def restore(self, sums, samples):
        sums = np.ascontiguousarray(sums, dtype=np.float32)
        self._check(self.lib.rt_accumulator_write(self.ctx, sums.ctypes.data, sums.shape[1], sums.shape[0], int(samples)))

    def seed_texture(self, width, height, grid=400.0, contrast=1.5, layers=12, lacunarity=1.5):
        """The reference's seed image (perlin.fs pass, raytrace.cpp:309-328) as rt_params.seed_tex words."""
        out = np.zeros((height, width), dtype=np.uint32)
        self._check(self.lib.rt_seed_texture(self.ctx, width, height, grid, contrast, layers, lacunarity, out.ctypes.data))
        return out


# Algorithmic FP32 work per event (add/sub/mul/div/sqrt/min count 1 each; no
# compares, no integer work): the flop model every roofline figure here uses.
FLOPS = {"sphere_test": 23, "root": 3, "hit_record": 17, "primary": 36, "sky": 26, "lambertian": 23,
         "metallic": 56, "dialectric": 60, "extra_unit_try": 12, "node_test": 22}


def algorithmic_flops(c: dict) -> float:
    """FP32 flops of a workload from its work counters."""
    scatters = c["lambertian"] + c["metallic"] + c["perlin"]
    extra_tries = max(0, c["unit_tries"] - scatters)
    return float(
        c["paths"] * FLOPS["primary"] + c["sphere_tests"] * FLOPS["sphere_test"] + c["roots"] * FLOPS["root"]
        + c["hits"] * FLOPS["hit_record"] + c["sky"] * FLOPS["sky"]
        + (c["lambertian"] + c["perlin"]) * FLOPS["lambertian"] + c["metallic"] * FLOPS["metallic"]
        + c["dialectric"] * FLOPS["dialectric"] + extra_tries * FLOPS["extra_unit_try"]
        + c["node_tests"] * FLOPS["node_test"])
