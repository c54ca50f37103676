#!/usr/bin/env python3
"""Rate of the host-buffer closest-hit entry point (rt_trace_batch: rays and results in host memory, PCIe inclusive)
beside the device-resident one.   usage: tools/host_batch_rate.py [n_rays]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytrace_amd as rt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
scene = rt.build_scene(rt.SCENE_CLOUD, n=1 << 20, seed=0xC4C4C4C4)
r = rt.Renderer(0)
r.upload(scene, rt.ACCEL_LBVH)
o, d = rt.gen_rays(0xC4C4C4C4, 0, n)
r.trace(o[:1000], d[:1000])
for rep in range(3):
    t0 = time.perf_counter()
    ids, t, nrm = r.trace(o, d)
    dt = time.perf_counter() - t0
    print(f"rt_trace_batch: {n} rays in {dt*1e3:.1f} ms = {n/dt/1e6:.1f} Mrays/s ({44.0*n/dt/1e9:.1f} GB/s over PCIe), kernel {r.last_kernel_ms()[0]:.1f} ms, hits {(ids>=0).sum()}", flush=True)
