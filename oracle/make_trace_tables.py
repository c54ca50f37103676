#!/usr/bin/env python3
"""Write tests/golden/trace_<scene>.npz: closest-hit tables (id, t, normal) of fixed 4096-ray batches and a
small rendered frame, computed by the CPU checker.

Test infrastructure.  The reference holds no such vectors (its arithmetic is GLSL and needs a GL device);
these tables freeze what the checker answers today, so that a later change to the checker or to the HIP path
shows up as a difference against committed data, not only as a difference between the two.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import raytrace_amd as rt  # noqa: E402  (host harness only: scenes, cameras, rays)

GOLD = os.path.join(ROOT, "tests", "golden")


def rays_toward_scene(scene_id, n, seed):
    """Origins around the scene, directions through it (un-normalised, as the shader's are)."""
    o, d = rt.gen_rays(seed, 0, n)
    scale = {rt.SCENE_DEFAULT: 4.0, rt.SCENE_RANDOM_SPHERES: 14.0, rt.SCENE_PERLIN: 8.0}[scene_id]
    o = (o / np.float32(600.0) * np.float32(scale)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1]) + np.float32(0.3)
    target = (d / np.float32(2.0) * np.float32(scale * 0.4)).astype(np.float32)
    target[:, 1] = np.float32(0.2)
    return o, (target - o).astype(np.float32)


def main():
    os.makedirs(GOLD, exist_ok=True)
    for name, sid in (("default", rt.SCENE_DEFAULT), ("random_spheres", rt.SCENE_RANDOM_SPHERES), ("perlin", rt.SCENE_PERLIN)):
        scene = rt.build_scene(sid)
        o, d = rays_toward_scene(sid, 4096, 1234 + sid)
        ids, t, nrm, _ = oracle.trace(scene, o, d, threads=8)
        cam = rt.scene_camera(sid)
        cam.origin_is_lookat = 0
        p = rt.make_params(cam, 48, 27, 4, rng_mode=rt.RNG_COUNTER)
        acc, _ = oracle.render(scene, p, threads=8)
        np.savez_compressed(os.path.join(GOLD, f"trace_{name}.npz"), o=o, d=d, id=ids, t=t, n=nrm, frame_sums=acc)
        print(name, "hits", int((ids >= 0).sum()), "of", len(ids), "frame mean", float(acc.mean()) / 4)


if __name__ == "__main__":
    main()
