#!/usr/bin/env python3
"""Kernel-tuning loop: tree-vs-scan bit check on a small frame, then the C2 frame (1920x1080x256 spp, 487 spheres)
timed a few times.  Prints one line per configuration.  RAYTRACE_AMD_LIB selects another build of the library.

  python tools/quick_c2.py [--frames 3] [--spp 256] [--counted] [--c4 RAYS]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--counted", action="store_true")
    ap.add_argument("--c4", type=int, default=0, help="also time a closest-hit batch of this many rays on 2^20 spheres")
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()
    r = rt.Renderer(0)
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    if not args.no_check:
        frames = {}
        for spp in (8, 64):   # one-pass / sweeps and the two-pass schedule
            p = rt.make_params(cam, 192, 108, spp, rng_mode=rt.RNG_COUNTER)
            for accel in (rt.ACCEL_LINEAR, rt.ACCEL_LBVH):
                r.upload(scene, accel)
                r.clear()
                r.render(p)
                frames[accel] = r.readback(rgb8=False)[0]
            same = np.array_equal(frames[rt.ACCEL_LINEAR].view(np.uint32), frames[rt.ACCEL_LBVH].view(np.uint32))
            print(f"check {spp:3d} spp tree == scan: {same}", flush=True)
            if not same:
                sys.exit(1)
    r.upload(scene, rt.ACCEL_LBVH)
    p = rt.make_params(cam, args.width, args.height, args.spp, rng_mode=rt.RNG_COUNTER)
    paths = args.width * args.height * args.spp
    for i in range(args.frames + 1):
        r.clear()
        t0 = time.perf_counter()
        r.render(p)
        r.synchronize()
        dt = time.perf_counter() - t0
        ms, name = r.last_kernel_ms()
        if i:
            print(f"c2 {args.width}x{args.height}x{args.spp}: {ms:8.2f} ms kernel {dt * 1e3:8.2f} ms wall {paths / dt / 1e6:9.1f} Mpaths/s  {name}", flush=True)
    if args.counted:
        r.clear()
        r.render(p, counted=True)
        c = r.counters()
        seg = max(c["segments"], 1)
        print("counters per segment:", {k: round(v / seg, 3) for k, v in c.items() if v}, flush=True)
        print("lanes per trav step %.1f, per path phase %.1f" % (c["trav_lane_steps"] / max(c["trav_wave_steps"], 1),
                                                                 c["path_lane_phases"] / max(c["path_wave_phases"], 1)), flush=True)
    if args.c4:
        import torch
        cloud = rt.build_scene(rt.SCENE_CLOUD, n=1 << 20, seed=0xC4C4C4C4)
        r.upload(cloud, rt.ACCEL_LBVH)
        o, d = rt.gen_rays(0xC4C4C4C4, 0, args.c4)
        o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
        ids = torch.empty(args.c4, dtype=torch.int32, device="cuda")
        t = torch.empty(args.c4, dtype=torch.float32, device="cuda")
        nrm = torch.empty((args.c4, 3), dtype=torch.float32, device="cuda")
        for i in range(3):
            r.trace_device(o.data_ptr(), d.data_ptr(), args.c4, ids.data_ptr(), t.data_ptr(), nrm.data_ptr())
            r.synchronize()
            ms, name = r.last_kernel_ms()
            if i:
                print(f"c4 {args.c4} rays: {ms:8.2f} ms {args.c4 / ms / 1e3:9.1f} Mrays/s  {name}  hits {int((ids >= 0).sum())} idsum {int(ids.long().sum())}", flush=True)
    r.close()


if __name__ == "__main__":
    main()
