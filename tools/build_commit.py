#!/usr/bin/env python3
"""Build the device library as it was at a commit, for same-box comparisons with the current one:
  tools/build_commit.py <commit> <name>  ->  variants/<name>.so   (sources unpacked under variants/tree_<name>/)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytrace_amd import build
commit, name = sys.argv[1], sys.argv[2]
tree = os.path.join(build.ROOT, "variants", "tree_" + name)
os.makedirs(tree, exist_ok=True)
archive = subprocess.Popen(["git", "-C", build.ROOT, "archive", commit, "raytrace_amd/csrc", "include"], stdout=subprocess.PIPE)
subprocess.check_call(["tar", "-x", "-C", tree], stdin=archive.stdout)
assert archive.wait() == 0
build.CSRC = os.path.join(tree, "raytrace_amd", "csrc")
build.ROOT = tree
out = os.path.join(os.path.dirname(tree), name + ".so")
build.build_device(force=True, out=out, objdir=os.path.join(os.path.dirname(tree), "obj_" + name))
print(out)
