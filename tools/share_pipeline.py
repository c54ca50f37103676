#!/usr/bin/env python3
"""Does a rank's share of the strong-scaling C2 frame run faster as two half shares on two contexts (two streams), the
second's probe pass filling the compute units the first's tail leaves idle?  One GPU; N = 8 by default.
  python tools/share_pipeline.py [N] [pieces]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pieces = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w, h, spp = 1920, 1080, 256
dev = torch.device("cuda", 0)
scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
accum = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
rs = []
for i in range(pieces):
    r = rt.Renderer(0)
    r.upload(scene, rt.ACCEL_LBVH)
    r.set_accumulator(accum.data_ptr(), w, h)
    rs.append(r)


def params(count, index):
    p = rt.make_params(cam, w, h, spp)
    p.strip_count, p.strip_index = count, index
    return p


def run(parts):
    best = 1e9
    for i in range(5):
        rs[0].clear()
        rs[0].synchronize()
        t0 = time.perf_counter()
        for r, p in parts:
            r.render(p)
        for r, _ in parts:
            r.synchronize()
        if i:
            best = min(best, time.perf_counter() - t0)
    return best, accum.cpu().numpy().copy()


one, img1 = run([(rs[0], params(n, 0))])
print(f"N={n} rank 0, one call: {one * 1e3:.2f} ms")
two, img2 = run([(rs[k], params(n * pieces, k * n)) for k in range(pieces)])
same = np.array_equal(img1.view(np.uint32), img2.view(np.uint32))
print(f"N={n} rank 0, {pieces} pieces on {pieces} streams: {two * 1e3:.2f} ms, same image: {same}")
seq, _ = run([(rs[0], params(n * pieces, k * n)) for k in range(pieces)])
print(f"N={n} rank 0, {pieces} pieces on one stream: {seq * 1e3:.2f} ms")
