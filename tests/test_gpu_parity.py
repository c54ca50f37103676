"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle.

Bars: hit ids bit-exact; t, normals and linear radiance bit-exact as well
(the kernels perform the oracle's IEEE operations in the oracle's order), which
is stricter than the 1e-5 relative tolerance the task states for t / normal;
RGB8 within one level (device powf vs libm powf).
"""
import numpy as np
import pytest

import oracle
import raytrace_amd as rt

# (small_trees: tree kernels also on the scenes of a few spheres that the library would scan, see conftest.py)
pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("small_trees")]


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_same_f32(a, b, what):
    a, b = _bits(a), _bits(b)
    bad = np.flatnonzero(a.ravel() != b.ravel())
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} values differ, first at {bad[:5]}"


def rays_for(scene_id, n, seed):
    """Rays that actually meet the scene: origins on a shell, aimed at jittered targets near the spheres."""
    rng = np.random.default_rng(seed)
    if scene_id == rt.SCENE_CLOUD:
        return rt.gen_rays(seed, 0, n)
    span = 12.0 if scene_id == rt.SCENE_RANDOM_SPHERES else 2.0
    o = rng.normal(size=(n, 3)).astype(np.float32)
    o /= np.linalg.norm(o, axis=1, keepdims=True)
    o *= rng.uniform(0.0, 3.0 * span, size=(n, 1)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1]) * 0.3 + 0.05
    target = rng.uniform(-span, span, size=(n, 3)).astype(np.float32)
    target[:, 1] = rng.uniform(-0.2, 1.0, size=n)
    d = (target - o) * rng.uniform(0.5, 2.0, size=(n, 1)).astype(np.float32)
    return o.astype(np.float32), d.astype(np.float32)


@pytest.mark.parametrize("accel", [rt.ACCEL_LINEAR, rt.ACCEL_LBVH])
@pytest.mark.parametrize("scene_id,n_cloud", [(rt.SCENE_DEFAULT, 0), (rt.SCENE_RANDOM_SPHERES, 0), (rt.SCENE_PERLIN, 0),
                                              (rt.SCENE_CLOUD, 60000)])
def test_trace_batch_matches_oracle(renderer, accel, scene_id, n_cloud):
    scene = rt.build_scene(scene_id, n=n_cloud, seed=0xC4C4C4C4)
    n = 4096
    o, d = rays_for(scene_id, n, 1234 + scene_id)
    renderer.upload(scene, accel)
    ids, t, nrm = renderer.trace(o, d)
    oid, ot, onrm, _ = oracle.trace(scene, o, d, threads=8)
    assert (oid >= 0).sum() > n // 50, "the batch should hit something"
    assert np.array_equal(ids, oid), f"{(ids != oid).sum()} hit ids differ"
    assert_same_f32(t, ot, "t")
    assert_same_f32(nrm, onrm, "normal")


def test_lbvh_equals_linear_on_large_cloud(renderer):
    """Size-independent property at a size the oracle cannot reach: both intersectors agree on every ray."""
    scene = rt.build_scene(rt.SCENE_CLOUD, n=50000, seed=7)
    o, d = rt.gen_rays(99, 0, 200000)
    renderer.upload(scene, rt.ACCEL_LINEAR)
    a = renderer.trace(o, d)
    renderer.upload(scene, rt.ACCEL_LBVH)
    b = renderer.trace(o, d)
    assert (a[0] >= 0).sum() > 1000
    assert np.array_equal(a[0], b[0]), f"{(a[0] != b[0]).sum()} ids differ"
    assert_same_f32(a[1], b[1], "t")
    assert_same_f32(a[2], b[2], "normal")


def _render_both(renderer, scene, params, accel, seed_tex=None):
    renderer.upload(scene, accel)
    renderer.clear()
    renderer.render(params, counted=True, seed_tex=seed_tex)
    lin, q = renderer.readback()
    counters = renderer.counters()
    acc, oc = oracle.render(scene, params, threads=8, seed_tex=seed_tex)
    olin, oq = oracle.resolve(acc, params.sample_end - params.sample_begin)
    return lin, q, counters, olin, oq, oc


@pytest.mark.parametrize("accel", [rt.ACCEL_LINEAR, rt.ACCEL_LBVH])
def test_default_scene_reference_rng(renderer, accel):
    """The reference's own configuration: default scene, 320x180, 8 spp, depth 50, xorshift per pixel."""
    scene = rt.build_scene(rt.SCENE_DEFAULT)
    cam = rt.scene_camera(rt.SCENE_DEFAULT)
    p = rt.make_params(cam, 320, 180, 8, rng_mode=rt.RNG_XORSHIFT_REF)
    lin, q, c, olin, oq, oc = _render_both(renderer, scene, p, accel)
    assert_same_f32(lin, olin, "linear radiance")
    assert np.abs(q.astype(int) - oq.astype(int)).max() <= 1
    for k in ("paths", "segments", "hits", "sky", "lambertian", "metallic", "dialectric", "unit_tries", "disk_tries",
              "limit_hits"):
        assert c[k] == oc[k], k
    if accel == rt.ACCEL_LINEAR:
        # same spheres tested; the scan skips the exact roots of spheres that provably lie beyond the closest hit
        assert c["sphere_tests"] == oc["sphere_tests"] and 0 < c["roots"] <= oc["roots"]


@pytest.mark.parametrize("accel", [rt.ACCEL_LINEAR, rt.ACCEL_LBVH])
@pytest.mark.parametrize("rng_mode", [rt.RNG_XORSHIFT_REF, rt.RNG_COUNTER])
def test_default_scene_from_lookfrom(renderer, accel, rng_mode):
    """Same scene seen from lookfrom (the reference's camera_origin = lookat starts every path inside the ground)."""
    scene = rt.build_scene(rt.SCENE_DEFAULT)
    cam = rt.scene_camera(rt.SCENE_DEFAULT)
    cam.origin_is_lookat = 0
    for i, v in enumerate((0.0, 1.0, 2.5)):
        cam.lookfrom[i] = v
    for i, v in enumerate((0.0, 0.0, -1.0)):
        cam.lookat[i] = v
    for i, v in enumerate((0.0, 1.0, 0.0)):
        cam.vup[i] = v
    cam.vfov = 50.0
    cam.focus_dist = 3.4
    p = rt.make_params(cam, 200, 120, 6, rng_mode=rng_mode)
    seed_tex = (np.arange(200 * 120, dtype=np.uint32) * np.uint32(2654435761)) if rng_mode == rt.RNG_XORSHIFT_REF else None
    lin, q, c, olin, oq, oc = _render_both(renderer, scene, p, accel, seed_tex=seed_tex)
    assert olin.mean() > 0.05, "the image must not be empty"
    assert c["dialectric"] > 0 and c["metallic"] > 0 and c["lambertian"] > 0
    assert_same_f32(lin, olin, "linear radiance")
    assert np.abs(q.astype(int) - oq.astype(int)).max() <= 1
    for k in ("paths", "segments", "hits", "sky", "lambertian", "metallic", "dialectric", "unit_tries", "disk_tries"):
        assert c[k] == oc[k], k


@pytest.mark.parametrize("accel", [rt.ACCEL_LINEAR, rt.ACCEL_LBVH])
def test_random_sphere_scene_small(renderer, accel):
    """The 487-sphere scene at a size the oracle renders in seconds."""
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    p = rt.make_params(cam, 160, 90, 4, rng_mode=rt.RNG_COUNTER)
    lin, q, c, olin, oq, oc = _render_both(renderer, scene, p, accel)
    assert_same_f32(lin, olin, "linear radiance")
    assert np.abs(q.astype(int) - oq.astype(int)).max() <= 1
    for k in ("paths", "segments", "hits", "sky", "lambertian", "metallic", "dialectric", "unit_tries", "disk_tries"):
        assert c[k] == oc[k], k


def test_perlin_scene_small(renderer):
    scene = rt.build_scene(rt.SCENE_PERLIN)
    cam = rt.scene_camera(rt.SCENE_PERLIN)
    p = rt.make_params(cam, 128, 72, 4, rng_mode=rt.RNG_COUNTER)
    lin, q, c, olin, oq, oc = _render_both(renderer, scene, p, rt.ACCEL_LBVH)
    assert c["perlin"] == oc["perlin"] and c["perlin"] > 0
    # cos / sin of the gradient angle are evaluated in double on both sides and narrowed: the device function gives
    # the bits of the reference's own perlin.h on every stored vector (test_gpu_edges.py), and so does the frame
    assert_same_f32(lin, olin, "linear radiance")


def test_tiles_and_sample_ranges_compose(renderer):
    """Counter RNG: four tiles x two sample ranges add up to the one-call image (bitwise per pixel for the tiles)."""
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    W, H, S = 150, 85, 6   # ragged against the 16x16 workgroup tile
    p = rt.make_params(cam, W, H, S, rng_mode=rt.RNG_COUNTER)
    renderer.upload(scene, rt.ACCEL_LBVH)
    renderer.clear()
    renderer.render(p)
    whole, _ = renderer.readback(rgb8=False)
    # tiles only: disjoint pixels, same sample range -> bit-identical
    renderer.clear()
    for (x0, x1) in ((0, 70), (70, W)):
        for (y0, y1) in ((0, 33), (33, H)):
            q = p.copy()
            q.tile_x0, q.tile_x1, q.tile_y0, q.tile_y1 = x0, x1, y0, y1
            renderer.render(q)
    tiled, _ = renderer.readback(rgb8=False)
    assert_same_f32(tiled, whole, "tiled image")
    # sample ranges in order: every pixel's sum goes on from the accumulator -> bit-identical
    renderer.clear()
    for (s0, s1) in ((0, 2), (2, S)):
        q = p.copy()
        q.sample_begin, q.sample_end = s0, s1
        renderer.render(q)
    split, _ = renderer.readback(rgb8=False)
    assert_same_f32(split, whole, "image from two sample ranges")
    # out of order: the same samples, summed in another order
    renderer.clear()
    for (s0, s1) in ((2, S), (0, 2)):
        q = p.copy()
        q.sample_begin, q.sample_end = s0, s1
        renderer.render(q)
    swapped, _ = renderer.readback(rgb8=False)
    np.testing.assert_allclose(swapped, whole, rtol=1e-5, atol=1e-6)


def test_edge_cases(renderer):
    cam = rt.scene_camera(rt.SCENE_DEFAULT)
    cam.origin_is_lookat = 0
    p = rt.make_params(cam, 33, 17, 2, rng_mode=rt.RNG_COUNTER)
    # empty scene: every path is sky
    empty = rt.Scene(b"", b"")
    for accel in (rt.ACCEL_LINEAR, rt.ACCEL_LBVH):
        renderer.upload(empty, accel)
        renderer.clear()
        renderer.render(p, counted=True)
        lin, _ = renderer.readback(rgb8=False)
        acc, oc = oracle.render(empty, p)
        assert_same_f32(lin, oracle.resolve(acc, 2)[0], "empty scene")
        assert renderer.counters()["sky"] == 33 * 17 * 2
        ids, t, nrm = renderer.trace(np.zeros((5, 3), np.float32), np.ones((5, 3), np.float32))
        assert (ids == -1).all() and (t == -1).all() and (nrm == 0).all()
    # one sphere: the LBVH is a single leaf
    scene = rt.build_scene(rt.SCENE_DEFAULT)
    one = rt.Scene(scene.spheres[32:64], scene.materials)
    o, d = rays_for(rt.SCENE_DEFAULT, 512, 5)
    renderer.upload(one, rt.ACCEL_LBVH)
    a = renderer.trace(o, d)
    b = oracle.trace(one, o, d)
    assert np.array_equal(a[0], b[0]) and (a[0] >= 0).any()
    assert_same_f32(a[1], b[1], "t")
    # zero rays, zero-size tile, zero samples: no-ops
    renderer.trace(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    q = p.copy()
    q.tile_x1 = q.tile_x0
    renderer.render(q)
    # degenerate rays (zero direction) follow the scan's NaN rule: no hit
    renderer.upload(scene, rt.ACCEL_LBVH)
    z = renderer.trace(np.array([[0, 0, 0]], np.float32), np.array([[0, 0, 0]], np.float32))
    assert z[0][0] == oracle.trace(scene, [[0, 0, 0]], [[0, 0, 0]])[0][0]
    # a sphere naming a material that was not uploaded is refused
    bad = bytearray(scene.spheres)
    bad[0:4] = (99).to_bytes(4, "little")
    with pytest.raises(rt.RtError):
        renderer.upload(rt.Scene(bytes(bad), scene.materials), rt.ACCEL_LINEAR)
    renderer.upload(scene, rt.ACCEL_LINEAR)


@pytest.mark.parametrize("rng_mode,spp", [(rt.RNG_COUNTER, 64), (rt.RNG_XORSHIFT_REF, 64), (rt.RNG_XORSHIFT_REF, 16)])
@pytest.mark.parametrize("accel", [rt.ACCEL_LBVH, rt.ACCEL_LINEAR])
def test_two_pass_render_matches_oracle(renderer, accel, rng_mode, spp):
    """Enough pixels and samples for the probe / sort / main schedule: the per-pixel sums and (for the
    reference RNG) the per-pixel RNG state are carried from the probe pass into the main pass.  The reference's
    generator takes that schedule from 16 samples on, the counter generator from 64."""
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    p = rt.make_params(cam, 160, 104, spp, rng_mode=rng_mode)   # 16 640 pixels
    renderer.upload(scene, accel)
    renderer.clear()
    renderer.render(p, counted=True)
    lin, _ = renderer.readback(rgb8=False)
    c = renderer.counters()
    acc, oc = oracle.render(scene, p, threads=8)
    assert_same_f32(lin, oracle.resolve(acc, spp)[0], "linear radiance")
    for k in ("paths", "segments", "hits", "sky", "lambertian", "metallic", "dialectric", "unit_tries", "disk_tries"):
        assert c[k] == oc[k], k


def _scaled_cloud(n, seed, scale):
    """The sphere cloud with its centres pulled towards the origin (radii kept): dense enough that paths bounce."""
    scene = rt.build_scene(rt.SCENE_CLOUD, n=n, seed=seed)
    rec = np.frombuffer(scene.spheres, dtype=np.uint8).reshape(-1, 32).copy()
    centres = rec[:, 16:28].copy().view(np.float32) * np.float32(scale)
    rec[:, 16:28] = centres.view(np.uint8)
    return rt.Scene(rec.tobytes(), scene.materials)


@pytest.mark.parametrize("scale,lookfrom_z", [(0.05, 60.0), (1.0, 300.0)])
def test_render_with_a_tree_too_big_for_lds(renderer, scale, lookfrom_z):
    """20 000 spheres: the hierarchy is read from HBM (staged node fetch, overflowing traversal stack) by the
    render kernel.  Packed tightly the per-ray box pad applies, spread out the per-box one; both must
    reproduce the scan bit for bit, on the GPU and in the CPU restatement."""
    scene = _scaled_cloud(20000, 0xC10D, scale)
    cam = rt.scene_camera(rt.SCENE_CLOUD)
    cam.lookfrom[2] = lookfrom_z
    p = rt.make_params(cam, 96, 54, 4, rng_mode=rt.RNG_COUNTER)
    frames = {}
    for accel in (rt.ACCEL_LBVH, rt.ACCEL_LINEAR):
        renderer.upload(scene, accel)
        renderer.clear()
        renderer.render(p, counted=True)
        frames[accel] = renderer.readback(rgb8=False)[0]
        if accel == rt.ACCEL_LBVH:
            c = renderer.counters()
            assert c["node_tests"] > 0 and renderer.lbvh_depth() > 8
            # the device build opens the largest child until a node has four: nearly full nodes, a third to a half as many as leaves
            info = renderer.scene_info()
            assert 20000 // 3 <= info["n_nodes"] <= 11000, info
            assert renderer.lbvh_depth() <= 2 * 20, renderer.lbvh_depth()
            want = "rt_render_kernel<lbvh4,hbm,pad_ray>" if scale < 1.0 else "rt_render_kernel<lbvh4,hbm,pad_node>"
            assert renderer.last_kernel_ms()[1] == want
            assert c["hits"] > (1000 if scale < 1.0 else 20)
    assert_same_f32(frames[rt.ACCEL_LBVH], frames[rt.ACCEL_LINEAR], "tree vs scan")
    acc, oc = oracle.render(scene, p, threads=16)
    assert_same_f32(frames[rt.ACCEL_LBVH], oracle.resolve(acc, 4)[0], "tree vs CPU restatement")
    assert c["hits"] == oc["hits"] and c["segments"] == oc["segments"]


def test_batch_result_does_not_depend_on_ray_order(renderer):
    """Lanes of a group fetch parts of each other's nodes and rays are handed out in chunks: neither may couple a
    ray's answer to its neighbours.  The same rays sorted by origin cell give the same answers, permuted."""
    scene = rt.build_scene(rt.SCENE_CLOUD, n=60000, seed=11)
    o, d = rt.gen_rays(123, 0, 300000)
    renderer.upload(scene, rt.ACCEL_LBVH)
    a = renderer.trace(o, d)
    cell = np.clip(((o + 600.0) / 1200.0 * 32).astype(np.int64), 0, 31)
    order = np.argsort((cell[:, 0] * 32 + cell[:, 1]) * 32 + cell[:, 2], kind="stable")
    b = renderer.trace(o[order], d[order])
    assert (a[0] >= 0).sum() > 1000
    assert np.array_equal(b[0], a[0][order])
    assert_same_f32(b[1], a[1][order], "t")
    assert_same_f32(b[2], a[2][order], "normal")
    c = renderer.trace(o[order], d[order])
    assert np.array_equal(b[0], c[0]) and np.array_equal(_bits(b[1]), _bits(c[1]))


@pytest.mark.parametrize("accel", [rt.ACCEL_LBVH, rt.ACCEL_LINEAR])
def test_few_samples_in_sweeps_match_oracle(renderer, accel):
    """Enough pixels but too few samples for a probe: with the counter RNG the whole frame is rendered as short
    jobs whose per-sample radiance is summed in order afterwards.  Rendered in two sample ranges onto the same
    image, so that the second sum starts from what the first left."""
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    renderer.upload(scene, accel)
    renderer.clear()
    for s0, n in ((0, 5), (5, 3)):
        renderer.render(rt.make_params(cam, 160, 104, n, rng_mode=rt.RNG_COUNTER, sample_begin=s0))
    lin, _ = renderer.readback(rgb8=False)
    # the CPU restatement adds the eight samples in order in one go: the same sum
    acc, _ = oracle.render(scene, rt.make_params(cam, 160, 104, 8, rng_mode=rt.RNG_COUNTER), threads=8)
    assert_same_f32(lin, oracle.resolve(acc, 8)[0], "linear radiance")


def test_host_and_device_hierarchy_builds_agree(renderer, monkeypatch):
    """A tree small enough for LDS is built on the host (surface area heuristic) unless RT_BUILDER=lbvh keeps the
    device's Morton-order build: both trees, and a cloud with coincident and nested spheres, must give the scan's
    answers bit for bit — frames and closest-hit batches."""
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    p = rt.make_params(cam, 160, 90, 4, rng_mode=rt.RNG_COUNTER)
    frames, steps = {}, {}
    for builder in ("sah", "lbvh"):
        monkeypatch.setenv("RT_BUILDER", builder)
        renderer.upload(scene, rt.ACCEL_LBVH)
        renderer.clear()
        renderer.render(p, counted=True)
        frames[builder] = renderer.readback(rgb8=False)[0]
        steps[builder] = renderer.counters()["node_tests"]
        assert renderer.last_kernel_ms()[1] == "rt_render_kernel<lbvh4,lds,pad_ray>"
    assert_same_f32(frames["sah"], frames["lbvh"], "host-built tree vs device-built tree")
    assert steps["sah"] < steps["lbvh"]  # what the host build is for
    acc, _ = oracle.render(scene, p, threads=8)
    assert_same_f32(frames["sah"], oracle.resolve(acc, 4)[0], "host-built tree vs CPU restatement")

    # duplicates, concentric spheres and a line of touching ones: split positions with equal cost, zero-extent boxes
    rng = np.random.default_rng(7)
    centres = np.concatenate([np.zeros((40, 3)), np.repeat(rng.uniform(-5, 5, (30, 3)), 3, axis=0),
                              np.stack([np.arange(60) * 0.5, np.zeros(60), np.zeros(60)], axis=1),
                              np.full((300, 3), 3.0)]).astype(np.float32)
    radii = np.concatenate([np.linspace(0.1, 2.0, 40), np.tile([0.2, 0.4, 0.8], 30), np.full(60, 0.25),
                            np.full(300, 0.3)]).astype(np.float32)  # the last 300 are one sphere 300 times: ties go to the first
    rec = np.zeros((len(radii), 32), dtype=np.uint8)
    rec[:, 4:8] = radii.view(np.uint8).reshape(-1, 4)
    rec[:, 16:28] = centres.view(np.uint8).reshape(-1, 12)
    odd = rt.Scene(rec.tobytes(), scene.materials)
    o = rng.uniform(-8, 8, (20000, 3)).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    o[:200] = centres[rng.integers(0, len(centres), 200)]  # rays that start at centres
    monkeypatch.delenv("RT_BUILDER")
    renderer.upload(odd, rt.ACCEL_LINEAR)
    want = renderer.trace(o, d)
    for builder in ("sah", "lbvh"):
        monkeypatch.setenv("RT_BUILDER", builder)
        renderer.upload(odd, rt.ACCEL_LBVH)
        got = renderer.trace(o, d)
        assert np.array_equal(got[0], want[0]), builder
        assert_same_f32(got[1], want[1], f"t ({builder})")
        assert_same_f32(got[2], want[2], f"normal ({builder})")
    assert (want[0] >= 0).sum() > 2000


def _scene_of(centres, radii, materials):
    rec = np.zeros((len(radii), 32), dtype=np.uint8)
    rec[:, 4:8] = np.ascontiguousarray(radii, dtype=np.float32).view(np.uint8).reshape(-1, 4)
    rec[:, 16:28] = np.ascontiguousarray(centres, dtype=np.float32).view(np.uint8).reshape(-1, 12)
    return rt.Scene(rec.tobytes(), materials)


def test_device_build_on_degenerate_clouds(renderer, monkeypatch):
    """The device build's top-down collapse on trees that are anything but balanced: a few spheres only (nodes with
    fewer than four children), and 24 000 spheres of which thousands are one sphere, thousands share a Morton cell
    and thousands lie on a line with growing gaps (chains of nodes with one internal child each).  Every ray must get
    the scan's answer bit for bit."""
    materials = rt.build_scene(rt.SCENE_DEFAULT).materials
    rng = np.random.default_rng(99)
    monkeypatch.setenv("RT_BUILDER", "lbvh")
    for n in (2, 3, 4, 5, 9, 17):
        centres = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
        radii = rng.uniform(0.2, 0.8, n).astype(np.float32)
        scene = _scene_of(centres, radii, materials)
        o = rng.uniform(-4, 4, (2048, 3)).astype(np.float32)
        d = (centres[rng.integers(0, n, 2048)] + rng.normal(scale=0.4, size=(2048, 3)) - o).astype(np.float32)
        renderer.upload(scene, rt.ACCEL_LBVH)
        assert renderer.scene_info()["builder"] == "device_lbvh"
        got = renderer.trace(o, d)
        want = oracle.trace(scene, o, d, threads=4)
        assert np.array_equal(got[0], want[0]), n
        assert_same_f32(got[1], want[1], f"t ({n} spheres)")
        assert (want[0] >= 0).sum() > 500
    monkeypatch.delenv("RT_BUILDER")

    same = np.tile(np.array([[1.0, 1.0, 1.0]], np.float32), (6000, 1))
    cell = (np.array([[-3.0, 2.0, 0.5]]) + rng.uniform(0, 1e-3, (6000, 3))).astype(np.float32)
    line = np.stack([1.0015 ** np.arange(6000) - 1.0, np.full(6000, -2.0), np.zeros(6000)], axis=1).astype(np.float32)
    spread = rng.uniform(-6, 6, (6000, 3)).astype(np.float32)
    centres = np.concatenate([same, cell, line, spread])
    radii = np.concatenate([np.full(6000, 0.3), rng.uniform(1e-4, 0.2, 6000), np.full(6000, 0.05),
                            rng.uniform(0.05, 0.3, 6000)]).astype(np.float32)
    scene = _scene_of(centres, radii, materials)
    n_rays = 16384
    o = rng.uniform(-8, 8, (n_rays, 3)).astype(np.float32)
    aim = centres[rng.integers(0, 18000, n_rays)] + rng.normal(scale=0.05, size=(n_rays, 3))
    d = ((aim - o) * rng.uniform(0.5, 2.0, (n_rays, 1))).astype(np.float32)
    renderer.upload(scene, rt.ACCEL_LINEAR)
    want = renderer.trace(o, d)
    renderer.upload(scene, rt.ACCEL_LBVH)
    info = renderer.scene_info()
    assert info["builder"] == "device_lbvh" and not info["tree_in_lds"], info
    got = renderer.trace(o, d)
    assert np.array_equal(got[0], want[0]), f"{(got[0] != want[0]).sum()} hit ids differ"
    assert_same_f32(got[1], want[1], "t")
    assert_same_f32(got[2], want[2], "normal")
    assert (want[0] >= 0).sum() > n_rays // 4
    ref = oracle.trace(scene, o[:1024], d[:1024], threads=8)
    assert np.array_equal(got[0][:1024], ref[0])
    assert_same_f32(got[1][:1024], ref[1], "t vs CPU restatement")


def test_hbm_tree_with_spheres_kept_out_of_it(renderer):
    """A tree read from HBM takes a leaf's sphere, index and 1/r from one record, and the hit record from that same
    record — unless the closest hit is one of the spheres kept out of the tree (much larger than the rest), whose
    record lies in the upload-order arrays.  20 000 small spheres around two giants: rays that end on a giant, on a
    small sphere in front of one, and on nothing; ids, t and normals are the scan's, bit for bit."""
    materials = rt.build_scene(rt.SCENE_DEFAULT).materials
    rng = np.random.default_rng(123)
    small = rng.uniform(-40, 40, (20000, 3)).astype(np.float32)
    centres = np.concatenate([np.array([[0.0, -60.0, 0.0], [55.0, 10.0, -5.0]], np.float32), small])
    radii = np.concatenate([np.array([50.0, 30.0], np.float32), rng.uniform(0.2, 0.6, 20000).astype(np.float32)])
    order = rng.permutation(len(radii))          # the giants somewhere in the middle of the upload order
    scene = _scene_of(centres[order], radii[order], materials)
    n_rays = 32768
    o = rng.uniform(-45, 45, (n_rays, 3)).astype(np.float32)
    d = (rng.normal(size=(n_rays, 3)) * rng.uniform(0.5, 2.0, (n_rays, 1))).astype(np.float32)
    renderer.upload(scene, rt.ACCEL_LINEAR)
    want = renderer.trace(o, d)
    renderer.upload(scene, rt.ACCEL_LBVH)
    info = renderer.scene_info()
    assert not info["tree_in_lds"] and info["n_big"] == 2, info
    got = renderer.trace(o, d)
    giants = np.isin(want[0], np.flatnonzero(order < 2))
    assert giants.sum() > n_rays // 20 and ((want[0] >= 0) & ~giants).sum() > n_rays // 20, (giants.sum(), (want[0] >= 0).sum())
    assert np.array_equal(got[0], want[0]), f"{(got[0] != want[0]).sum()} hit ids differ"
    assert_same_f32(got[1], want[1], "t")
    assert_same_f32(got[2], want[2], "normal")
    ref = oracle.trace(scene, o[:512], d[:512], threads=8)
    assert np.array_equal(got[0][:512], ref[0])
    assert_same_f32(got[2][:512], ref[2], "normal vs CPU restatement")


def test_small_scenes_are_scanned(renderer, monkeypatch):
    """Up to 16 spheres the scan is faster than any tree, so an upload that asks for a tree gets the scan (no tree is
    built); the frame and a closest-hit batch are those of the tree, bit for bit."""
    scene = rt.build_scene(rt.SCENE_DEFAULT)
    cam = rt.scene_camera(rt.SCENE_DEFAULT)
    p = rt.make_params(cam, 160, 90, 8, rng_mode=rt.RNG_XORSHIFT_REF)
    o, d = rays_for(rt.SCENE_DEFAULT, 2048, 77)
    out = {}
    for limit in (None, "0"):
        if limit is None:
            monkeypatch.delenv("RT_SCAN_UP_TO")
        else:
            monkeypatch.setenv("RT_SCAN_UP_TO", limit)
        renderer.upload(scene, rt.ACCEL_LBVH)
        info = renderer.scene_info()
        renderer.clear()
        renderer.render(p)
        kernel = renderer.last_kernel_ms()[1]
        out[limit] = (renderer.readback(rgb8=False)[0], renderer.trace(o, d))
        if limit is None:
            assert kernel == "rt_render_kernel<linear,lds>" and info["builder"] == "none" and info["n_nodes"] == 0, (kernel, info)
        else:
            assert kernel == "rt_render_kernel<lbvh4,lds,pad_ray>" and info["n_nodes"] > 0, (kernel, info)
    assert_same_f32(out[None][0], out["0"][0], "scan chosen for size vs tree")
    assert np.array_equal(out[None][1][0], out["0"][1][0])
    assert_same_f32(out[None][1][1], out["0"][1][1], "t")
    acc, _ = oracle.render(scene, p)
    assert_same_f32(out[None][0], oracle.resolve(acc, 8)[0], "scan chosen for size vs CPU restatement")
    # 17 spheres: the tree stays
    big = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    monkeypatch.delenv("RT_SCAN_UP_TO")
    renderer.upload(rt.Scene(big.spheres[:17 * 32], big.materials), rt.ACCEL_LBVH)
    assert renderer.scene_info()["n_nodes"] > 0
    renderer.upload(rt.Scene(big.spheres[:16 * 32], big.materials), rt.ACCEL_LBVH)
    assert renderer.scene_info()["n_nodes"] == 0
