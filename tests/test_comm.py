"""rt_comm_*: the sum of the ranks' accumulators as a library call (RCCL underneath, on the context's stream).

One GPU here, so the group has one rank: what is tested is the plumbing — RCCL found and loaded next to PyTorch's
copy, the communicator made on the context's device, the reduce queued behind the render on the context's stream and
ahead of the next clear, the error paths.  Groups of several GPUs are the driver's runs.
"""
import numpy as np
import pytest

import raytrace_amd as rt


@pytest.mark.gpu
def test_reduce_in_a_group_of_one(renderer):
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    p = rt.make_params(cam, 320, 200, 64)
    r = rt.Renderer(0)
    try:
        r.upload(scene, rt.ACCEL_LBVH)
        with pytest.raises(rt.RtError, match="before rt_comm_init"):
            r.comm_reduce(0)
        uid = rt.Renderer.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        with pytest.raises(rt.RtError, match="bad group"):
            r.comm_init(uid, 1, 1)
        r.comm_init(uid, 0, 1)
        with pytest.raises(rt.RtError, match="already"):
            r.comm_init(uid, 0, 1)
        with pytest.raises(rt.RtError, match="nothing rendered"):
            r.comm_reduce(0)
        frames = []
        for _ in range(2):   # no host wait between render, sum, and the next frame's clear
            r.clear()
            r.render(p)
            r.comm_reduce(0)
        frames.append(r.readback(rgb8=False)[0])
        with pytest.raises(rt.RtError, match="root"):
            r.comm_reduce(3)
        r.comm_destroy()
        r.comm_destroy()   # leaving twice is harmless
        renderer.upload(scene, rt.ACCEL_LBVH)
        renderer.clear()
        renderer.render(p)
        want = renderer.readback(rgb8=False)[0]
        assert np.array_equal(frames[0].view(np.uint32), want.view(np.uint32))
    finally:
        r.close()
