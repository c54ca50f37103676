#!/usr/bin/env python3
"""Tree against scan on small scenes: the ground sphere, the three large spheres and n - 4 of the small ones of the
487-sphere scene, 960x540x64 spp, both intersectors.  Where the scan is faster, the upload should choose it."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402

full = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
rec = np.frombuffer(full.spheres, dtype=np.uint8).reshape(-1, 32)
radius = rec[:, 4:8].copy().view(np.float32).ravel()
large = np.flatnonzero(radius >= 0.9)
small = np.flatnonzero(radius < 0.9)
rng = np.random.default_rng(5)
r = rt.Renderer(0)
p = rt.make_params(cam, 960, 540, 64)
for n in [int(a) for a in sys.argv[1:]] or [4, 6, 8, 12, 16, 24, 32, 48, 64, 96]:
    keep = np.sort(np.concatenate([large, rng.choice(small, max(n - len(large), 0), replace=False)]))[:max(n, 1)]
    scene = rt.Scene(rec[keep].tobytes(), full.materials)
    out = []
    for accel in (rt.ACCEL_LBVH, rt.ACCEL_LINEAR):
        r.upload(scene, accel)
        best = 1e9
        for i in range(4):
            r.clear()
            r.synchronize()
            t0 = time.perf_counter()
            r.render(p)
            r.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            name = r.last_kernel_ms()[1]
            if i:
                best = min(best, ms)
        out.append((best, name))
    print(f"{len(keep):4d} spheres: tree {out[0][0]:7.3f} ms ({out[0][1]})   scan {out[1][0]:7.3f} ms ({out[1][1]})   scan/tree {out[1][0] / out[0][0]:5.2f}", flush=True)
r.close()
