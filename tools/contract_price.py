#!/usr/bin/env python3
"""What the one-IEEE-operation-per-written-operation contract costs and buys: run with RAYTRACE_AMD_LIB naming a build
of the library compiled with -ffp-contract=fast (tools/build_variant.py contract -ffp-contract=fast), on the GPU box.

Against the CPU checker's frozen answers (tests/golden/trace_*.npz, 4 096 rays and a 48x27x4 frame per scene) and a
160x90x4 frame of the 487-sphere scene: how many hit ids change, the largest relative error of t and of the normal,
the frame's RMSE; then the C2 frame's time.  Prints one JSON object.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (the checker: this is a measurement tool, not product code)
import raytrace_amd as rt  # noqa: E402
from raytrace_amd import _lib  # noqa: E402


def main():
    out = {"library": os.path.relpath(_lib.DEVICE_LIB, ROOT), "tables": {}}
    os.environ["RT_TUNING"] = "1"
    os.environ["RT_SCAN_UP_TO"] = "0"   # tree kernels on the small scenes too
    r = rt.Renderer(0)
    for name, sid in (("default", rt.SCENE_DEFAULT), ("random_spheres", rt.SCENE_RANDOM_SPHERES), ("perlin", rt.SCENE_PERLIN)):
        g = np.load(os.path.join(ROOT, "tests", "golden", f"trace_{name}.npz"), allow_pickle=False)
        scene = rt.build_scene(sid)
        for accel, aname in ((rt.ACCEL_LINEAR, "scan"), (rt.ACCEL_LBVH, "tree")):
            r.upload(scene, accel)
            ids, t, nrm = r.trace(g["o"], g["d"])
            same = ids == g["id"]
            hit = same & (ids >= 0)
            rel_t = np.abs(t[hit].astype(np.float64) - g["t"][hit]) / np.abs(g["t"][hit])
            err_n = np.linalg.norm(nrm[hit].astype(np.float64) - g["n"][hit], axis=1)
            cam = rt.scene_camera(sid)
            cam.origin_is_lookat = 0
            p = rt.make_params(cam, 48, 27, 4, rng_mode=rt.RNG_COUNTER)
            r.clear()
            r.render(p)
            sums, _ = r.checkpoint()
            ref = g["frame_sums"]
            ok = np.isfinite(ref) & np.isfinite(sums)
            out["tables"][f"{name}/{aname}"] = {
                "rays": int(ids.size), "ids_changed": int((~same).sum()),
                "t_bits_changed": int((t[hit].view(np.uint32) != g["t"][hit].view(np.uint32)).sum()),
                "max_rel_err_t": float(rel_t.max()) if rel_t.size else 0.0,
                "max_err_normal": float(err_n.max()) if err_n.size else 0.0,
                "frame_values_changed": int((sums.view(np.uint32) != ref.view(np.uint32)).sum()), "frame_values": int(ref.size),
                "frame_rmse_of_mean": float(np.sqrt(np.mean(((sums[ok] - ref[ok]) / 4.0) ** 2))),
                "frame_mean": float(np.mean(ref[ok]) / 4.0)}
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    p = rt.make_params(cam, 160, 90, 4, rng_mode=rt.RNG_COUNTER)
    acc, _ = oracle.render(scene, p, threads=8)
    want = oracle.resolve(acc, 4)[0]
    r.upload(scene, rt.ACCEL_LBVH)
    r.clear()
    r.render(p)
    got = r.readback(rgb8=False)[0]
    out["c2_small_frame"] = {"values_changed": int((got.view(np.uint32) != want.view(np.uint32)).sum()), "values": int(want.size),
                             "rmse": float(np.sqrt(np.mean((got - want) ** 2))), "mean": float(want.mean()),
                             "note": "paths diverge after the first differing scatter: the RMSE is that of two independent 4-spp renders"}
    p = rt.make_params(cam, 1920, 1080, 256, rng_mode=rt.RNG_COUNTER)
    ms = []
    for i in range(4):
        r.clear()
        t0 = time.perf_counter()
        r.render(p)
        r.synchronize()
        if i:
            ms.append((time.perf_counter() - t0) * 1e3)
    out["c2_frame_ms"] = ms
    out["c2_mpaths_per_s"] = 1920 * 1080 * 256 / (min(ms) * 1e-3) / 1e6
    r.clear()
    r.render(p, counted=True)
    c = r.counters()
    out["c2_counters"] = {k: c[k] for k in ("segments", "sphere_tests", "node_tests", "hits")}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
