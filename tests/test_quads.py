"""Quads: the second kind of object behind the hittable interface (hittable.h:6-15 is a polymorphic base whose
records carry their own size; the reference ships one subclass, sphere.h:11).

The reference holds no quad, so no reference-held vector can pin this primitive: the CPU tests pin the checker's
hit_quad by hand-computed cases (parity otherwise unpinned), and the GPU tests hold the HIP path — scan and
hierarchy — to the checker bit for bit.
"""
import struct

import numpy as np
import pytest

import oracle
import raytrace_amd as rt


def rays_at(scene_id, n, seed, scale=1.0):
    o, d = rt.gen_rays(seed, 0, n)
    return (o * np.float32(scale)).astype(np.float32), d


# ---- checker, CPU ------------------------------------------------------------------------------------------------

UNIT = rt.quad_record(0, (0, 0, 0), (1, 0, 0), (0, 1, 0))   # the unit square in z = 0


@pytest.mark.parametrize("o,d,want", [
    ((0.25, 0.5, 2.0), (0, 0, -1), 2.0),        # straight down onto it
    ((0.25, 0.5, -4.0), (0, 0, 2), 2.0),        # from behind, direction of length 2
    ((0.25, 0.5, 2.0), (0, 0, 1), -1.0),        # pointing away
    ((1.5, 0.5, 2.0), (0, 0, -1), -1.0),        # beside it
    ((0.25, -0.01, 2.0), (0, 0, -1), -1.0),     # just below an edge
    ((0.0, 0.0, 1.0), (0, 0, -1), 1.0),         # a corner counts
    ((1.0, 1.0, 1.0), (0, 0, -1), 1.0),         # the far corner too
    ((0.25, 0.5, 1.0), (1, 0, 0), -1.0),        # along the plane
    ((0.25, 0.5, 0.0005), (0, 0, -1), -1.0),    # nearer than the shader's 0.001
    ((-1.0, 0.5, 1.0), (1.5, 0, -1), 1.0),      # slanted, lands at x = 0.5
])
def test_hit_quad_known_answers(o, d, want):
    assert oracle.hit_quad(UNIT, d, o) == want


def test_hit_quad_sheared_parallelogram():
    rec = rt.quad_record(0, (1, 1, 1), (2, 0, 0), (1, 2, 0))   # corners (1,1) (3,1) (2,3) (4,3) in z = 1
    assert oracle.hit_quad(rec, (0, 0, -1), (2.0, 2.0, 3.0)) == 2.0
    assert oracle.hit_quad(rec, (0, 0, -1), (1.2, 2.0, 3.0)) == -1.0   # left of the slanted edge
    assert oracle.hit_quad(rec, (0, 0, -1), (3.8, 2.9, 3.0)) == 2.0


def test_quad_record_layout_and_host_class():
    """The quad class of the scene API writes material @0, kind 1 @8, Q @16, u @32, v @48; a sphere written into
    the zeroed buffer has kind 0 @8 (sphere.h:22-35 leaves bytes 8..15 alone)."""
    sc = rt.build_scene(rt.SCENE_QUAD_ROOM)
    assert (sc.n_objects, sc.n_materials, len(sc.objects)) == (9, 7, 6 * 64 + 3 * 32)
    assert sc.objects[:64] == rt.quad_record(0, (-3, -2, 1), (0, 0, -6), (0, 4, 0))
    assert sc.objects[64:96] == rt.sphere_record(6, 0.8, (1.2, -1.2, -2.5))
    assert struct.unpack_from("<i", sc.objects, 96 + 8)[0] == rt.OBJECT_QUAD


def test_checker_orders_objects_by_distance_then_upload_index():
    mats = rt.material_record(0, rt.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))
    near = rt.quad_record(0, (-1, -1, -2), (2, 0, 0), (0, 2, 0))
    far = rt.quad_record(0, (-1, -1, -5), (2, 0, 0), (0, 2, 0))
    ball = rt.sphere_record(0, 1.0, (0, 0, -3.5))
    o = np.zeros((1, 3), np.float32)
    d = np.array([[0, 0, -1]], np.float32)
    for order, want_id, want_t in (((far, ball, near), 2, 2.0), ((far, ball), 1, 2.5), ((far,), 0, 5.0),
                                   ((near, near), 0, 2.0)):      # coincident quads: the first uploaded wins
        sc = rt.ObjectScene(b"".join(order), len(order), mats)
        ids, t, nrm, _ = oracle.trace(sc, o, d)
        assert (int(ids[0]), float(t[0])) == (want_id, want_t)
    # normals face the ray from either side
    sc = rt.ObjectScene(near, 1, mats)
    assert oracle.trace(sc, o, d)[2][0].tolist() == [0.0, 0.0, 1.0]
    assert oracle.trace(sc, np.array([[0, 0, -4]], np.float32), -d)[2][0].tolist() == [0.0, 0.0, -1.0]


def test_checker_rejects_malformed_lists():
    mats = rt.material_record(0, rt.MAT_LAMBERTIAN)
    o = np.zeros((1, 3), np.float32)
    with pytest.raises(RuntimeError):
        oracle.trace(rt.ObjectScene(UNIT[:48], 1, mats), o, o)            # quad cut short
    bad = bytearray(UNIT)
    bad[8] = 7
    with pytest.raises(RuntimeError):
        oracle.trace(rt.ObjectScene(bytes(bad), 1, mats), o, o)          # unknown kind


# ---- HIP path --------------------------------------------------------------------------------------------------

def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def frame_pair(renderer, scene, params, seed_tex=None):
    renderer.clear()
    renderer.render(params, seed_tex=seed_tex)
    got = renderer.readback(rgb8=False)[0]
    acc, _ = oracle.render(scene, params, threads=8, seed_tex=seed_tex)
    return got, oracle.resolve(acc, params.sample_end - params.sample_begin)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("accel", [rt.ACCEL_LINEAR, rt.ACCEL_LBVH])
@pytest.mark.parametrize("scene_id,n,ray_scale", [(rt.SCENE_QUAD_ROOM, 0, 0.01), (rt.SCENE_QUAD_CLOUD, 3000, 0.1),
                                                  (rt.SCENE_QUAD_CLOUD, 40000, 0.1)])
def test_closest_hits_match_the_checker(renderer, scene_id, n, ray_scale, accel):
    scene = rt.build_scene(scene_id, n=n, seed=21)
    n_rays = 200_000 if n <= 3000 else 30_000
    o, d = rays_at(scene_id, n_rays, 0xABCD0000 + n, ray_scale)
    want_id, want_t, want_n, _ = oracle.trace(scene, o, d, threads=8)
    renderer.upload(scene, accel)
    ids, t, nrm = renderer.trace(o, d)
    hits = int((want_id >= 0).sum())
    is_quad = np.array([struct.unpack_from("<i", scene.objects, off + 8)[0] for off in object_offsets(scene)])
    quad_hits = int(is_quad[want_id[want_id >= 0]].sum())
    assert hits > n_rays // 50 and quad_hits > hits // 20, (hits, quad_hits)
    assert np.array_equal(ids, want_id)
    assert same_bits(t, want_t) and same_bits(nrm, want_n)
    name = renderer.last_kernel_ms()[1]
    assert "quads" in name and ("lbvh4" in name) == (accel == rt.ACCEL_LBVH), name


def object_offsets(scene):
    off, out = 0, []
    for _ in range(scene.n_objects):
        out.append(off)
        off += 64 if struct.unpack_from("<i", scene.objects, off + 8)[0] == rt.OBJECT_QUAD else 32
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("accel", [rt.ACCEL_LINEAR, rt.ACCEL_LBVH])
@pytest.mark.parametrize("rng_mode", [rt.RNG_COUNTER, rt.RNG_XORSHIFT_REF])
def test_room_frames_match_the_checker(renderer, accel, rng_mode):
    scene = rt.build_scene(rt.SCENE_QUAD_ROOM)
    cam = rt.scene_camera(rt.SCENE_QUAD_ROOM)
    p = rt.make_params(cam, 96, 54, 6, rng_mode=rng_mode)
    seeds = oracle.seed_texture(96, 54) if rng_mode == rt.RNG_XORSHIFT_REF else None
    renderer.upload(scene, accel)
    got, want = frame_pair(renderer, scene, p, seeds)
    assert same_bits(got, want)
    assert float(want.mean()) > 0.05


@pytest.mark.gpu
def test_cloud_frames_stored_pass_and_tree_match_the_checker(renderer):
    """Enough samples for the two-pass schedule (probe, cost-sorted jobs, stored samples), on the hierarchy."""
    scene = rt.build_scene(rt.SCENE_QUAD_CLOUD, n=2000, seed=5)
    cam = rt.scene_camera(rt.SCENE_QUAD_CLOUD)
    p = rt.make_params(cam, 64, 36, 40, bounce_limit=8)
    for accel in (rt.ACCEL_LINEAR, rt.ACCEL_LBVH):
        renderer.upload(scene, accel)
        got, want = frame_pair(renderer, scene, p)
        assert same_bits(got, want), accel
        assert "quads" in renderer.last_kernel_ms()[1]


@pytest.mark.gpu
def test_sphere_only_list_through_the_object_entry_point(renderer):
    """A list without quads takes the sphere kernels and gives what rt_scene_upload gives."""
    plain = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    tagged = rt.ObjectScene(plain.spheres, plain.n_spheres, plain.materials)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    p = rt.make_params(cam, 96, 54, 4)
    frames = []
    for sc in (plain, tagged):
        renderer.upload(sc, rt.ACCEL_LBVH)
        renderer.clear()
        renderer.render(p)
        frames.append(renderer.readback(rgb8=False)[0])
        assert "quads" not in renderer.last_kernel_ms()[1]
    assert same_bits(*frames)


@pytest.mark.gpu
def test_object_upload_rejects_malformed_lists(renderer):
    mats = rt.material_record(0, rt.MAT_LAMBERTIAN)
    with pytest.raises(rt.RtError, match="beyond"):
        renderer.upload(rt.ObjectScene(UNIT[:48], 1, mats))
    with pytest.raises(rt.RtError, match="beyond"):
        renderer.upload(rt.ObjectScene(UNIT, 2, mats))
    bad = bytearray(UNIT)
    bad[8] = 7
    with pytest.raises(rt.RtError, match="unknown kind"):
        renderer.upload(rt.ObjectScene(bytes(bad), 1, mats))
    with pytest.raises(rt.RtError, match="material"):
        renderer.upload(rt.ObjectScene(rt.quad_record(3, (0, 0, 0), (1, 0, 0), (0, 1, 0)), 1, mats))
    # a failed upload leaves no scene behind
    o = np.zeros((1, 3), np.float32)
    with pytest.raises(rt.RtError):
        renderer.trace(o, o)


@pytest.mark.gpu
def test_all_quads_and_degenerate_quads(renderer):
    """No sphere at all; a quad with parallel edges (normal 0/0 = NaN) is never hit, as in the checker."""
    mats = rt.material_record(0, rt.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))
    recs = [rt.quad_record(0, (-1, -1, -2 - i), (2, 0, 0), (0, 2, 0)) for i in range(5)]
    recs.insert(2, rt.quad_record(0, (-1, -1, -1), (2, 0, 0), (4, 0, 0)))
    scene = rt.ObjectScene(b"".join(recs), len(recs), mats)
    o, d = rays_at(0, 50_000, 77, 0.004)
    want = oracle.trace(scene, o, d, threads=8)
    assert 2 not in want[0] and (want[0] >= 0).sum() > 1000
    for accel in (rt.ACCEL_LINEAR, rt.ACCEL_LBVH):
        renderer.upload(scene, accel)
        ids, t, nrm = renderer.trace(o, d)
        assert np.array_equal(ids, want[0]) and same_bits(t, want[1]) and same_bits(nrm, want[2])
