#!/usr/bin/env python3
"""Find non-finite pixels of the full C2 frame and compare them, sample by sample, with the CPU checker."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytrace_amd as rt
import oracle
scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
p = rt.make_params(cam, 1920, 1080, 256, rng_mode=rt.RNG_COUNTER)
r = rt.Renderer(0)
r.upload(scene, rt.ACCEL_LBVH)
r.clear(); r.render(p)
a, _ = r.readback(rgb8=False)
bad = np.argwhere(~np.isfinite(a).all(axis=2))
print("non-finite pixels:", len(bad), bad[:10].tolist())
for (y, x) in bad[:4]:
    y, x = int(y), int(x)
    q = p.copy(); q.tile_x0, q.tile_x1, q.tile_y0, q.tile_y1 = x, x + 1, y, y + 1
    acc, _ = oracle.render(scene, q, threads=1)
    print("pixel", x, y, "gpu sum/256", a[y, x], "oracle sum", acc[y, x] / 256)
    for s in range(256):
        q.sample_begin, q.sample_end = s, s + 1
        acc, oc = oracle.render(scene, q, threads=1)
        r.clear(); r.render(q, counted=True)
        g, _ = r.readback(rgb8=False)
        gc = r.counters()
        same = (g[y, x].view(np.uint32) == acc[y, x].view(np.uint32)).all()
        if not same or not np.isfinite(g[y, x]).all() or not np.isfinite(acc[y, x]).all():
            print("  sample", s, "gpu", g[y, x], "oracle", acc[y, x], "segments gpu/oracle", gc["segments"], oc["segments"],
                  "kinds gpu", gc["lambertian"], gc["metallic"], gc["dialectric"], "oracle", oc["lambertian"], oc["metallic"], oc["dialectric"])
