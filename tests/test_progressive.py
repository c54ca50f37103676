"""Chunked / progressive rendering (the reference's shader_chunk_pass idea, raytrace.cpp:598-630) and
checkpoint / resume of the accumulator."""
import numpy as np
import pytest

import oracle
import raytrace_amd as rt

pytestmark = pytest.mark.gpu


def _setup(renderer, w=160, h=96, spp=12):
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    p = rt.make_params(cam, w, h, spp, rng_mode=rt.RNG_COUNTER)
    renderer.upload(scene, rt.ACCEL_LBVH)
    return scene, p


def _whole(renderer, p):
    renderer.clear()
    renderer.render(p)
    return renderer.readback()[0]


def test_chunks_compose_to_the_frame_and_match_the_oracle(renderer):
    scene, p = _setup(renderer)
    want = _whole(renderer, p)
    renderer.clear()
    steps = list(renderer.render_progressive(p, chunk=(64, 40), samples_per_pass=5))
    assert len(steps) == 3 * 3 * 3 and steps[-1][0] == 12
    got = renderer.readback()[0]
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    acc, _ = oracle.render(scene, p, threads=8)
    olin, _ = oracle.resolve(acc, 12)
    assert (got.view(np.uint32) == olin.view(np.uint32)).all()


def test_chunk_ignores_the_params_tile_and_checks_bounds(renderer):
    scene, p = _setup(renderer, 64, 48, 4)
    q = p.copy()
    q.tile_x0, q.tile_y0, q.tile_x1, q.tile_y1 = 5, 5, 6, 6
    renderer.clear()
    renderer.render_chunk(q, (0, 0), (64, 48))
    a = renderer.readback()[0]
    b = _whole(renderer, p)
    assert (a.view(np.uint32) == b.view(np.uint32)).all()
    with pytest.raises(rt.RtError):
        renderer.render_chunk(p, (0, 0), (65, 48))
    with pytest.raises(rt.RtError):
        renderer.render_chunk(p, (10, 0), (5, 48))


def test_checkpoint_and_resume_equal_one_uninterrupted_render(renderer):
    scene, p = _setup(renderer)
    want = _whole(renderer, p)
    first = p.copy()
    first.sample_end = 7
    renderer.clear()
    renderer.render(first)
    sums, n = renderer.checkpoint()
    assert n == 7 and sums.shape == (96, 160, 3) and sums.sum() > 0
    # something else happens to the context in between
    other = rt.make_params(rt.scene_camera(rt.SCENE_RANDOM_SPHERES), 32, 32, 2, rng_mode=rt.RNG_COUNTER)
    renderer.clear()
    renderer.render(other)
    renderer.restore(sums, n)
    rest = p.copy()
    rest.sample_begin = n
    renderer.render(rest)
    got = renderer.readback()[0]
    assert renderer.checkpoint()[1] == 12
    assert (got.view(np.uint32) == want.view(np.uint32)).all()


def test_checkpoint_needs_a_render_and_write_checks_arguments(renderer):
    fresh = rt.Renderer(0)
    try:
        with pytest.raises(rt.RtError):
            fresh.checkpoint()
        with pytest.raises(rt.RtError):
            fresh.restore(np.zeros((4, 4, 3), np.float32), -1)
        fresh.restore(np.ones((4, 5, 3), np.float32), 2)
        sums, n = fresh.checkpoint()
        assert n == 2 and sums.shape == (4, 5, 3) and (sums == 1).all()
        lin, _ = fresh.readback()
        assert np.allclose(lin, 0.5)
    finally:
        fresh.close()
