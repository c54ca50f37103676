#!/usr/bin/env python3
"""Build a tuning variant of the device library: tools/build_variant.py <name> [-DFLAG=...]...
The result is variants/<name>.so; select it at run time with RAYTRACE_AMD_LIB=variants/<name>.so."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytrace_amd import build
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(build.ROOT, "variants", name + ".so")
os.makedirs(os.path.dirname(out), exist_ok=True)
build.build_device(force=True, extra_flags=flags, out=out, objdir=os.path.join(build.ROOT, "variants", "obj_" + name))
print(out)
