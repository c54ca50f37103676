#!/usr/bin/env python3
"""Rate of both RNG modes at several frame sizes.  usage: tools/rng_mode_rate.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt
scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES); cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
r = rt.Renderer(0); r.upload(scene, rt.ACCEL_LBVH)
for (w, h, spp) in ((1920, 1080, 256), (1920, 1080, 64), (1920, 1080, 32), (1920, 1080, 16), (320, 180, 8)):
    for mode, name in ((rt.RNG_XORSHIFT_REF, "xorshift_ref"), (rt.RNG_COUNTER, "counter")):
        p = rt.make_params(cam, w, h, spp, rng_mode=mode)
        best = 1e9
        for _ in range(3):
            r.clear(); r.render(p); best = min(best, r.last_kernel_ms()[0])
        print(f"{w}x{h}x{spp} {name:12s} {best:8.2f} ms {w*h*spp/best/1e3:8.1f} Mpaths/s", flush=True)
