#!/usr/bin/env python3
"""Rate of the C2 scene on tiles of decreasing height (full width, 256 spp): what a small share of a frame costs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402

r = rt.Renderer(0)
scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
r.upload(scene, rt.ACCEL_LBVH)
w, h, spp = 1920, 1080, 256
for rows in ([int(a) for a in sys.argv[1:]] or [1080, 540, 272, 136, 64, 32, 16]):
    p = rt.make_params(cam, w, h, spp)
    p.tile_y0, p.tile_y1 = 540 - rows // 2, 540 - rows // 2 + rows
    best = 1e9
    for i in range(4):
        r.clear()
        t0 = time.perf_counter()
        r.render(p)
        r.synchronize()
        if i:
            best = min(best, time.perf_counter() - t0)
    paths = w * rows * spp
    print(f"{rows:5d} rows: {best * 1e3:7.2f} ms {paths / best / 1e6:8.1f} Mpaths/s  {r.last_kernel_ms()[1]}", flush=True)
r.close()
