#!/usr/bin/env python3
"""How much does ray order matter for the closest-hit kernel on the 2^20-sphere cloud?
Traces the same rays in generated order, sorted origin-major and sorted direction-major (torch sort
on the GPU, not timed), and prints the kernel time of each.   usage: tools/coherence_probe.py [n_rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytrace_amd as rt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
dev = torch.device("cuda", 0)
scene = rt.build_scene(rt.SCENE_CLOUD, n=1 << 20, seed=0xC4C4C4C4)
r = rt.Renderer(0)
r.upload(scene, rt.ACCEL_LBVH)
o_h, d_h = rt.gen_rays(0xC4C4C4C4, 0, n)
o = torch.from_numpy(o_h).to(dev)
d = torch.from_numpy(d_h).to(dev)
ids = torch.empty(n, dtype=torch.int32, device=dev)
t = torch.empty(n, dtype=torch.float32, device=dev)
nrm = torch.empty((n, 3), dtype=torch.float32, device=dev)


def spread(v):  # 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def origin_code(bits):
    q = ((o + 600.0) / 1200.0 * (1 << bits)).long().clamp(0, (1 << bits) - 1)
    return (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])


def direction_code(res):
    a = d.abs()
    axis = a.argmax(dim=1)
    major = d.gather(1, axis[:, None])[:, 0]
    face = axis * 2 + (major < 0).long()
    u_ax = (axis + 1) % 3
    v_ax = (axis + 2) % 3
    u = d.gather(1, u_ax[:, None])[:, 0] / major.abs()
    v = d.gather(1, v_ax[:, None])[:, 0] / major.abs()
    qu = ((u + 1) * 0.5 * res).long().clamp(0, res - 1)
    qv = ((v + 1) * 0.5 * res).long().clamp(0, res - 1)
    return (face * res + qu) * res + qv


def run(name, order):
    oo, dd = (o, d) if order is None else (o[order].contiguous(), d[order].contiguous())
    torch.cuda.synchronize()  # the gather above runs on torch's stream, the trace on the library's
    best = 1e9
    for _ in range(3):
        r.trace_device(oo.data_ptr(), dd.data_ptr(), n, ids.data_ptr(), t.data_ptr(), nrm.data_ptr(), False)
        best = min(best, r.last_kernel_ms()[0])
    hits = int((ids >= 0).sum())
    print(f"{name:40s} {best:8.2f} ms  {n / best / 1e3:8.1f} Mrays/s  hits {hits}", flush=True)


run("generated order", None)
for ob, res in ((5, 4), (6, 4), (6, 8)):
    oc = origin_code(ob)
    dc = direction_code(res)
    nd = 6 * res * res
    run(f"origin {ob}b major, dir {res}x{res} minor", torch.argsort(oc * nd + dc))
    run(f"dir {res}x{res} major, origin {ob}b minor", torch.argsort(dc * (1 << (3 * ob)) + oc))
