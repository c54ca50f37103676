#!/usr/bin/env python3
"""Wall time of rt_scene_upload (flatten + LBVH build) for repeated uploads.  usage: tools/upload_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt
r = rt.Renderer(0)
for name, scene in (("487 spheres", rt.build_scene(rt.SCENE_RANDOM_SPHERES)), ("60000 spheres", rt.build_scene(rt.SCENE_CLOUD, n=60000, seed=1)),
                    ("2^20 spheres", rt.build_scene(rt.SCENE_CLOUD, n=1 << 20, seed=1))):
    for accel, an in ((rt.ACCEL_LBVH, "lbvh"), (rt.ACCEL_LINEAR, "linear")):
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            r.upload(scene, accel)
            ts.append((time.perf_counter() - t0) * 1e3)
        b = r.last_build_ms() if accel == rt.ACCEL_LBVH else 0.0
        print(f"{name:14s} {an:6s} upload wall ms: " + " ".join(f"{t:.2f}" for t in ts) + f"   device build {b:.2f} ms", flush=True)
