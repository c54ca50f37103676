"""Builds the native libraries in-tree (run on the CPU; hipcc cross-compiles gfx950).

  raytrace_amd/libraytrace_amd.so   HIP kernels + the C ABI of include/raytrace_amd.h
  raytrace_amd/librt_host.so        host harness of include/raytrace_amd_host.h (plain g++)
  oracle/librt_oracle.so            the CPU checker (test infrastructure; built here, used only by tests/bench)
  oracle/_ref/ref_harness           the reference's own headers, only where /root/reference exists
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off and the correctly rounded divide / sqrt are part of the
# numerical contract: every f32 operation of the shader is one IEEE operation.
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
    "-Wl,-rpath,/opt/rocm/lib",
]
CXX_FLAGS = ["-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-Wall", "-Wextra", "-fvisibility=hidden"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _glob(d, exts):
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts))


DEVICE_SOURCES = ["rt_render_stored.hip", "rt_render_pixels.hip", "rt_api.hip", "rt_build.hip", "rt_render.hip", "rt_comm.hip", "rt_eval.hip", "rt_gl_capture.cpp"]


def build_device(force=False, extra_flags=(), out=None, objdir=None, jobs=None):
    """Every translation unit to an object of its own (side by side: rt_render.hip alone is most of the time), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    out = out or os.path.join(HERE, "libraytrace_amd.so")
    objdir = objdir or os.path.join(HERE, "build")
    headers = _glob(CSRC, (".cuh", ".hpp")) + _glob(os.path.join(ROOT, "include"), (".h",))
    os.makedirs(objdir, exist_ok=True)
    compile_flags = [f for f in HIP_FLAGS if f != "-shared" and not f.startswith("-Wl,")] + list(extra_flags)
    todo, objs = [], []
    for src in DEVICE_SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _newer(obj, [os.path.join(CSRC, src)] + headers):
            todo.append([HIPCC] + compile_flags + ["-c", "-o", obj, os.path.join(CSRC, src)])
    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(len(todo), os.cpu_count() or 1)) as pool:
            list(pool.map(_run, todo))
    if todo or force or _newer(out, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined", "-o", out] + objs)
    return out


def kernel_sources_sha():
    """SHA-256 over the device library's sources (csrc/, include/*.h), in name order: what a committed profile of the
    kernels was taken at.  bench.py replays counter figures from profiles/ only when this still matches."""
    import hashlib
    h = hashlib.sha256()
    for path in _glob(CSRC, (".hip", ".cuh", ".hpp", ".cpp")) + _glob(os.path.join(ROOT, "include"), (".h",)):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_host(force=False):
    out = os.path.join(HERE, "librt_host.so")
    srcs = [os.path.join(HOST, "rt_host.cpp"), os.path.join(CSRC, "rt_gl_capture.cpp")]
    deps = srcs + _glob(HOST, (".hpp",)) + _glob(os.path.join(ROOT, "include"), (".h",))
    if force or _newer(out, deps):
        _run(["g++"] + CXX_FLAGS + ["-o", out] + srcs)
    return out


def build_oracle():
    _run(["make", "-C", os.path.join(ROOT, "oracle")])
    if os.path.isdir("/root/reference"):
        _run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])


def build_all(force=False):
    build_host(force)
    build_device(force)
    build_oracle()


def build_sanitized():
    """The host-side C++ under AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile: asan runs its driver, ubsan
    builds the two libraries the Python CPU tests can load)."""
    _run(["make", "-C", os.path.join(ROOT, "oracle"), "asan", "ubsan"])


if __name__ == "__main__":
    if "--asan" in sys.argv:
        build_sanitized()
    else:
        build_all(force="--force" in sys.argv)
