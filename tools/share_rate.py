#!/usr/bin/env python3
"""Time one rank's share of a strong-scaling job on one GPU, with the shares bench.py hands out.

  python tools/share_rate.py [--config c2|c5] [--ranks 1 2 4 8] [--sample-groups G ...] [--bands B ...]

c2: the 487-sphere frame at 1920x1080x256 spp, N interleaved sets of 8-row strips, all samples each (rows gathered).
c5: the same scene at 3840x2160x4096 spp, N / G strip sets x G sample ranges (frames summed).
The ideal time of a share is the whole job's time on this GPU / N; what a share loses against it is the schedule's
(fewer pixels per lane, the probe pass's tail) — the data exchange of N real ranks is not in this number.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402
from raytrace_amd import distributed as rtd  # noqa: E402

SHAPES = {"c2": (1920, 1080, 256), "c5": (3840, 2160, 4096)}


def time_share(r, p, bands, h, repeats, pause=0.0):
    best = 1e9
    for i in range(repeats + 1):
        r.clear()
        r.synchronize()
        if pause and i:
            time.sleep(pause)
        t0 = time.perf_counter()
        for y0, y1 in rtd.band_rows(h, bands, p.strip_count):
            q = p.copy()
            q.tile_y0, q.tile_y1 = y0, y1
            r.render(q)
        r.synchronize()
        if i:
            best = min(best, time.perf_counter() - t0)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=sorted(SHAPES))
    ap.add_argument("--ranks", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--sample-groups", type=int, nargs="*", default=None, help="sample ranges per strip set (c5: default 1 and 2)")
    ap.add_argument("--bands", type=int, nargs="*", default=[1], help="launches per share (render_and_combine overlaps the sums with > 1)")
    ap.add_argument("--pause", type=float, default=0.0, help="seconds of rest before each measurement (minutes of full load lower the clocks)")
    ap.add_argument("--bounce-limit", type=int, default=50, help="path depth (50 is the configurations'; smaller values answer what long paths cost)")
    ap.add_argument("--whole-ms", type=float, default=0.0, help="the whole job's time if N = 1 is not in --ranks")
    args = ap.parse_args()
    w, h, spp = SHAPES[args.config]
    repeats = 3 if args.config == "c2" else 1
    r = rt.Renderer(0)
    r.upload(rt.build_scene(rt.SCENE_RANDOM_SPHERES), rt.ACCEL_LBVH)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    whole = args.whole_ms * 1e-3 or None
    print(f"# {args.config}: {w}x{h}x{spp} spp, one share timed on one GPU; ideal = whole job / N", flush=True)
    for n in args.ranks:
        groups_list = args.sample_groups or ([1, 2] if args.config == "c5" and n >= 4 else [1])
        for groups in groups_list:
            if n % groups:
                continue
            shares = rtd.split_interleaved(h, spp, n, sample_groups=groups)
            for rank in sorted({0, n - 1}):
                for bands in args.bands:
                    p = rtd.apply_share(rt.make_params(cam, w, h, spp, bounce_limit=args.bounce_limit), shares[rank])
                    t = time_share(r, p, bands, h, repeats, args.pause)
                    if whole is None:
                        whole = t * n
                    print(f"N={n} strip sets {n // groups} x sample ranges {groups}, rank {rank}, bands {bands}: {t * 1e3:8.2f} ms  "
                          f"ideal {whole * 1e3 / n:8.2f} ms  efficiency {whole / n / t:5.3f}", flush=True)
    r.close()


if __name__ == "__main__":
    main()
