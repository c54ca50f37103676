#!/usr/bin/env python3
"""Closest-hit batch (2^20 spheres) timed on several builds of the library in one process, through the few entry
points every build has.   python tools/ab_c4.py lib_a.so lib_b.so [--rays N] [--rounds K]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    cloud = rt.build_scene(rt.SCENE_CLOUD, n=1 << 20, seed=0xC4C4C4C4)
    o, d = rt.gen_rays(0xC4C4C4C4, 0, args.rays)
    o, d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    ids = torch.empty(args.rays, dtype=torch.int32, device="cuda")
    t = torch.empty(args.rays, dtype=torch.float32, device="cuda")
    nrm = torch.empty((args.rays, 3), dtype=torch.float32, device="cuda")
    ctxs = []
    for path in args.libs:
        L = ctypes.CDLL(os.path.abspath(path))
        ctx = ctypes.c_void_p()
        assert L.rt_init(0, ctypes.byref(ctx)) == 0
        assert L.rt_scene_upload(ctx, cloud.spheres, cloud.n_spheres, cloud.materials, cloud.n_materials, rt.ACCEL_LBVH) == 0
        ctxs.append((path, L, ctx))
    vp = ctypes.c_void_p
    for rnd in range(args.rounds + 1):
        for path, L, ctx in ctxs:
            rc = L.rt_trace_batch_device(ctx, vp(o.data_ptr()), vp(d.data_ptr()), ctypes.c_size_t(args.rays), vp(ids.data_ptr()),
                                         vp(t.data_ptr()), vp(nrm.data_ptr()), 0)
            assert rc == 0
            L.rt_synchronize(ctx)
            ms, name = ctypes.c_float(0), ctypes.c_char_p()
            L.rt_last_kernel_ms(ctx, ctypes.byref(ms), ctypes.byref(name))
            if rnd:
                print(f"{os.path.basename(path):28s} {ms.value:8.2f} ms {args.rays / ms.value / 1e3:9.1f} Mrays/s  idsum {int(ids.long().sum())}",
                      flush=True)


if __name__ == "__main__":
    main()
