#!/usr/bin/env python3
"""Time one rank's share of the strong-scaling C2 job on one GPU: the strips of rank 0 of N, for N = 1, 2, 4, 8.
The ideal time of a share is the whole frame's time / N; what is lost is the schedule's (fewer pixels per lane)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402
from raytrace_amd import distributed as rtd  # noqa: E402

r = rt.Renderer(0)
scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
r.upload(scene, rt.ACCEL_LBVH)
w, h, spp = 1920, 1080, 256
base = None
for n in ([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]):
    for rank in sorted({0, n - 1}):
        for bands in ((1,) if len(sys.argv) > 1 else (1, 2, 4)):   # rows rendered in bands, as render_and_combine does to overlap the sums
            p = rtd.apply_share(rt.make_params(cam, w, h, spp), rtd.split_interleaved(h, spp, n)[rank])
            best = 1e9
            for i in range(4):
                r.clear()
                t0 = time.perf_counter()
                for y0, y1 in rtd.band_rows(h, bands, p.strip_count):
                    q = p.copy()
                    q.tile_y0, q.tile_y1 = y0, y1
                    r.render(q)
                r.synchronize()
                if i:
                    best = min(best, time.perf_counter() - t0)
            if base is None:
                base = best * n
            print(f"N={n} rank {rank} bands {bands}: {best * 1e3:7.2f} ms  ideal {base * 1e3 / n:7.2f} ms  efficiency {base / n / best:5.3f}", flush=True)
r.close()
