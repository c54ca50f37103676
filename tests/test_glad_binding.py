"""The binding a maintainer of the reference would add — glad's function pointers pointed at the capture functions,
without casts — compiled against the reference's own glad.h and linked with its src/glad.c
(container only: skipped where /root/reference is absent, as on the GPU box).  Nothing of the reference is copied:
its files are read where they lie and the outputs go to a temporary directory."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not present")
def test_capture_functions_fit_the_glad_pointers(tmp_path):
    from raytrace_amd import _lib
    _lib.host_lib()  # built
    glad_o = str(tmp_path / "glad.o")
    exe = str(tmp_path / "glad_binding_check")
    subprocess.check_call(["gcc", "-c", "-O1", f"-I{REF}/include", f"{REF}/src/glad.c", "-o", glad_o])
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", f"-I{REF}/include", f"-I{REF}", f"-I{ROOT}/include",
                           os.path.join(ROOT, "tests", "cxx", "glad_binding_check.cpp"), glad_o, "-o", exe,
                           f"-L{ROOT}/raytrace_amd", "-l:librt_host.so", f"-Wl,-rpath,{ROOT}/raytrace_amd", "-ldl"])
    out = subprocess.check_output([exe]).decode()
    assert out.startswith("OK"), out
