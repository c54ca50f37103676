"""Committed closest-hit tables and small frames (tests/golden/trace_*.npz, written by oracle/make_trace_tables.py).
The reference has no vectors of its own for this path; these freeze the checker's answers, so that the CPU
checker and the HIP path are each held against data, not only against one another."""
import os

import numpy as np
import pytest

import oracle
import raytrace_amd as rt

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENES = [("default", rt.SCENE_DEFAULT), ("random_spheres", rt.SCENE_RANDOM_SPHERES), ("perlin", rt.SCENE_PERLIN)]


def same_f32(a, b):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


def load(name):
    return np.load(os.path.join(GOLD, f"trace_{name}.npz"), allow_pickle=False)


def frame_params(sid):
    cam = rt.scene_camera(sid)
    cam.origin_is_lookat = 0
    return rt.make_params(cam, 48, 27, 4, rng_mode=rt.RNG_COUNTER)


@pytest.mark.parametrize("name,sid", SCENES)
def test_checker_reproduces_its_tables(name, sid):
    g = load(name)
    scene = rt.build_scene(sid)
    ids, t, nrm, _ = oracle.trace(scene, g["o"], g["d"], threads=4)
    assert np.array_equal(ids, g["id"]) and same_f32(t, g["t"]) and same_f32(nrm, g["n"])
    acc, _ = oracle.render(scene, frame_params(sid), threads=4)
    assert same_f32(acc, g["frame_sums"])
    assert len(np.unique(g["id"])) >= 2 and (g["id"] >= 0).sum() > 3000


@pytest.mark.gpu
@pytest.mark.parametrize("accel", [rt.ACCEL_LINEAR, rt.ACCEL_LBVH])
@pytest.mark.parametrize("name,sid", SCENES)
def test_gpu_reproduces_the_tables(renderer, name, sid, accel):
    g = load(name)
    scene = rt.build_scene(sid)
    renderer.upload(scene, accel)
    ids, t, nrm = renderer.trace(g["o"], g["d"])
    assert np.array_equal(ids, g["id"]), f"{(ids != g['id']).sum()} ids differ"
    assert same_f32(t, g["t"]) and same_f32(nrm, g["n"])
    renderer.clear()
    renderer.render(frame_params(sid))
    sums, n = renderer.checkpoint()
    assert n == 4 and same_f32(sums, g["frame_sums"])
