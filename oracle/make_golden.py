#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference (this container only).

Test infrastructure.  Builds oracle/_ref/ref_harness from the reference's own
vec3.h / perlin.h / hittable_list.h (see oracle/ref_harness.cpp), runs it, and
stores what it printed as data fixtures:

  tests/golden/ref_values.json          perlin() table, gradient table, camera
                                        uniform vectors, hittable_list bookkeeping
  tests/golden/random_scene_spheres.bin 487 x 32-byte std140 sphere slots
  tests/golden/random_scene_materials.bin  487 x 32-byte std140 material slots

The fixtures hold inputs and outputs only; no reference source text.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("the reference tree is not present; fixtures are already committed")
    subprocess.check_call(["make", "-C", HERE, "ref"])
    out = subprocess.check_output([os.path.join(HERE, "_ref", "ref_harness")])
    data = json.loads(out)
    scene = data.pop("random_scene")
    os.makedirs(GOLD, exist_ok=True)
    with open(os.path.join(GOLD, "random_scene_spheres.bin"), "wb") as f:
        f.write(bytes.fromhex(scene.pop("spheres")))
    with open(os.path.join(GOLD, "random_scene_materials.bin"), "wb") as f:
        f.write(bytes.fromhex(scene.pop("materials")))
    data["random_scene"] = scene  # counts only
    with open(os.path.join(GOLD, "ref_values.json"), "w") as f:
        json.dump(data, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", GOLD, {k: (len(v) if isinstance(v, list) else v) for k, v in data.items() if k in ("perlin", "gradients", "random_scene")})


if __name__ == "__main__":
    main()
