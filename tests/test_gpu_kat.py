"""Known answers of the reference shader (shaders/testFragment.fs) asked of the HIP path directly, through the C ABI,
and asserted against literals — not against the CPU checker.  The reference holds no test vectors, so the answers
that can be derived by hand from its source are the only ones that do not pass through the checker:
  hit_sphere  (fs:143-163)  c = (0,0,-1), r = 0.5, d = (0,0,-1): t = 0.5 from o = 0, and 0.5 from o = c (far root)
  hit_any     (fs:165-199)  a miss reports nothing; of two spheres met at the same t the lower index wins (fs:180)
  shade_sky   (fs:318-324)  straight up (0.5, 0.7, 1.0), straight down (1, 1, 1)
  xorshift32  (fs:326-345)  xorshift32(1) = 270369, hence the first draw 270369 / 2^23; cantor (fs:394-398)
Every case runs with the dispatch as shipped and, for scenes of a few spheres that the library would scan, once more
with the tree kernels forced (RT_SCAN_UP_TO=0).
"""
import numpy as np
import pytest

import raytrace_amd as rt

pytestmark = pytest.mark.gpu

F = np.float32


@pytest.fixture(params=["as_shipped", "tree_kernels"])
def dispatch(request, monkeypatch):
    if request.param == "tree_kernels":
        monkeypatch.setenv("RT_SCAN_UP_TO", "0")
    return request.param


def _materials():
    return rt.material_record(0, rt.MAT_LAMBERTIAN, (0.5, 0.5, 0.5)) + rt.material_record(1, rt.MAT_METALLIC, (0.9, 0.9, 0.9), 0.0)


def _scene(spheres):
    return rt.Scene(b"".join(rt.sphere_record(m, r, c) for (m, r, c) in spheres), _materials())


def _trace(renderer, scene, accel, o, d):
    renderer.upload(scene, accel)
    ids, t, nrm = renderer.trace(np.array(o, F).reshape(-1, 3), np.array(d, F).reshape(-1, 3))
    return ids.tolist(), t.tolist(), nrm.tolist()


@pytest.mark.parametrize("accel", [rt.ACCEL_LINEAR, rt.ACCEL_LBVH])
def test_hit_sphere_roots_and_miss(renderer, dispatch, accel):
    """fs:143-163: t = (h - sqrt(disc)) / a, the far root when the near one is below 0.001, -1 otherwise."""
    scene = _scene([(0, 0.5, (0.0, 0.0, -1.0))])
    ids, t, nrm = _trace(renderer, scene, accel,
                         [(0, 0, 0), (0, 0, -1), (0, 0, 0), (0, 0, -2), (0, 0, 0), (0, 2, 0)],
                         [(0, 0, -1), (0, 0, -1), (0, 0, 1), (0, 0, -1), (0, 0, -2), (0, 0, -1)])
    # from outside: near root 0.5, outward normal (0, 0, 1) faces the ray
    assert (ids[0], t[0], nrm[0]) == (0, 0.5, [0.0, 0.0, 1.0])
    # from the centre: near root -0.5 fails the gate, far root 0.5; the outward normal (0, 0, -1) is flipped (fs:190-193)
    assert (ids[1], t[1], nrm[1]) == (0, 0.5, [0.0, 0.0, 1.0])
    # looking away, and from behind the sphere: both roots negative -> no hit: id -1, t -1, normal 0
    for k in (2, 3):
        assert (ids[k], t[k], nrm[k]) == (-1, -1.0, [0.0, 0.0, 0.0])
    # an un-normalised direction: a = 4, h = 2, disc = 4 - 4 * 0.75 = 1 -> t = (2 - 1) / 4
    assert (ids[4], t[4], nrm[4]) == (0, 0.25, [0.0, 0.0, 1.0])
    # a ray that passes the sphere: negative discriminant
    assert (ids[5], t[5]) == (-1, -1.0)


@pytest.mark.parametrize("accel", [rt.ACCEL_LINEAR, rt.ACCEL_LBVH])
def test_equal_t_goes_to_the_lower_index(renderer, dispatch, accel):
    """fs:180 accepts a sphere only if new_t < t: of two spheres met at exactly the same parameter the one uploaded
    first stays.  A: c = (0,0,-1), r = 0.5 and B: c = (0,0,-2), r = 1.5 both meet the ray 0 + t (0,0,-1) at t = 0.5
    exactly (A: h = 1, disc = 0.25; B: h = 2, disc = 2.25)."""
    a, b = (0, 0.5, (0.0, 0.0, -1.0)), (1, 1.5, (0.0, 0.0, -2.0))
    far = (0, 0.25, (5.0, 5.0, 5.0))
    for order, first_mat in (([a, b], 0), ([b, a], 1), ([far, b, far, a], 1), ([far, a, b], 0)):
        ids, t, _ = _trace(renderer, _scene(order), accel, [(0, 0, 0)], [(0, 0, -1)])
        want = min(i for i, s in enumerate(order) if s is not far)
        assert (ids[0], t[0]) == (want, 0.5), order
        assert order[ids[0]][0] == first_mat
    # the same sphere twenty times (more than the library scans by default): index 0
    ids, t, _ = _trace(renderer, _scene([a] * 20), accel, [(0, 0, 0)], [(0, 0, -1)])
    assert (ids[0], t[0]) == (0, 0.5)


def _fixed_ray_params(direction, rng_mode, width=1, height=1, vtl=None, delta_u=(0, 0, 0), bounce_limit=1):
    """A camera whose every sample is the same ray: no pixel pitch, no defocus disk (fs:117-131)."""
    p = rt.RtParams()
    p.width, p.height = width, height
    p.tile_x0, p.tile_y0, p.tile_x1, p.tile_y1 = 0, 0, width, height
    p.sample_begin, p.sample_end = 0, 1
    p.bounce_limit = bounce_limit
    for i in range(3):
        p.delta_u[i] = float(delta_u[i])
        p.delta_v[i] = 0.0
        p.camera_origin[i] = 0.0
        p.viewport_top_left[i] = float((vtl or direction)[i])
        p.defocus_disk_u[i] = p.defocus_disk_v[i] = 0.0
    p.defocus_angle = 0.0
    p.rng_mode = rng_mode
    p.seed = 12345
    p.strip_count = p.strip_index = 0
    p.seed_tex = None
    return p


@pytest.mark.parametrize("accel", [rt.ACCEL_LINEAR, rt.ACCEL_LBVH])
@pytest.mark.parametrize("rng_mode", [rt.RNG_XORSHIFT_REF, rt.RNG_COUNTER])
@pytest.mark.parametrize("bounce_limit", [1, 50])
def test_sky_straight_up_and_straight_down(renderer, dispatch, accel, rng_mode, bounce_limit):
    """fs:318-324 with throughput 1: a = 0.5 (1 + normalize(d).y); (1 - a) (1,1,1) + a (0.5,0.7,1.0).  One sample of a
    one-pixel image, in an empty scene and past a sphere that is out of the way."""
    for scene in (rt.Scene(b"", b""), _scene([(0, 0.5, (40.0, 0.0, 0.0))] * 3)):
        renderer.upload(scene, accel)
        for direction, rgb, rgb8 in (((0, 1, 0), (F(0.5), F(0.7), F(1.0)), (186, 217, 255)),
                                     ((0, -1, 0), (F(1.0), F(1.0), F(1.0)), (255, 255, 255)),
                                     ((0, 3, 0), (F(0.5), F(0.7), F(1.0)), (186, 217, 255))):
            renderer.clear()
            renderer.render(_fixed_ray_params(direction, rng_mode, bounce_limit=bounce_limit))
            lin, q = renderer.readback()
            assert tuple(lin[0, 0]) == rgb, (direction, lin[0, 0])
            assert np.abs(q[0, 0].astype(int) - np.array(rgb8)).max() <= 1   # 0.5^(1/2.2) = 0.7297, 0.7^(1/2.2) = 0.8503


def test_bounce_limit_zero_is_black(renderer, dispatch):
    """fs:203-205: the first bounce() of a path already finds count >= bounce_limit."""
    renderer.upload(rt.Scene(b"", b""), rt.ACCEL_LINEAR)
    renderer.clear()
    renderer.render(_fixed_ray_params((0, 1, 0), rt.RNG_XORSHIFT_REF, bounce_limit=0))
    assert tuple(renderer.readback(rgb8=False)[0][0, 0]) == (0.0, 0.0, 0.0)


def _sky_of(dy_over_len):
    a = F(0.5) * (F(1.0) + dy_over_len)
    one_minus = F(1.0) - a
    return tuple(F(one_minus * F(1.0)) + F(a * c) for c in (F(0.5), F(0.7), F(1.0)))


@pytest.mark.parametrize("accel", [rt.ACCEL_LINEAR, rt.ACCEL_LBVH])
def test_first_draw_of_seed_one_is_xorshift32_of_one(renderer, dispatch, accel):
    """fs:326-345: xorshift32(1) = 1 ^ (1 << 13) = 8193, >> 17 changes nothing, ^ (8193 << 5) = 270369; rand_float
    keeps the low 23 bits as a mantissa: u = 270369 / 2^23.  With seed = texel ^ cantor(x, y) = 1 the first draw of the
    pixel is the x jitter u - 0.5 (fs:119); a pixel pitch of (0, 1, 0) and a viewport corner of (1, 0.5 - x, 0) turn it
    into the direction (1, u, 0) exactly, whose sky colour is a literal computed from u here.  cantor (fs:394-398):
    (0,0) -> 0, (1,0) -> 1, (0,1) -> 2, (1,1) -> 4, so texels 1, 0, 3, 5 all seed their pixel with 1."""
    x = 1
    x ^= (x << 13) & 0xFFFFFFFF
    x ^= x >> 17
    x ^= (x << 5) & 0xFFFFFFFF
    assert x == 270369
    u = F(270369) / F(8388608.0)
    length = np.sqrt(F(F(F(1.0) * F(1.0)) + F(u * u)) + F(0.0))   # dot = (x x + y y) + z z, correctly rounded sqrt
    want = _sky_of(F(F(F(1.0) / length) * u))                         # v / s = (1 / s) v
    assert want != _sky_of(F(F(F(1.0) / np.sqrt(F(2.0))) * F(1.0)))
    renderer.upload(rt.Scene(b"", b"") if accel == rt.ACCEL_LINEAR else _scene([(0, 0.5, (40.0, 0.0, 0.0))] * 3), accel)
    seed_tex = np.array([[1, 0], [3, 5]], dtype=np.uint32)
    for (px, py) in ((0, 0), (1, 0), (0, 1), (1, 1)):
        p = _fixed_ray_params(None, rt.RNG_XORSHIFT_REF, width=2, height=2, vtl=(1.0, 0.5 - px, 0.0), delta_u=(0, 1, 0))
        p.tile_x0, p.tile_x1, p.tile_y0, p.tile_y1 = px, px + 1, py, py + 1
        renderer.clear()
        renderer.render(p, seed_tex=seed_tex)
        lin = renderer.readback(rgb8=False)[0]
        assert tuple(lin[py, px]) == want, ((px, py), lin[py, px], want)
        assert not lin[1 - py, 1 - px].any()
    # another seed, another colour: the pixel does show the draw
    renderer.clear()
    p = _fixed_ray_params(None, rt.RNG_XORSHIFT_REF, width=2, height=2, vtl=(1.0, 0.5, 0.0), delta_u=(0, 1, 0))
    p.tile_x1, p.tile_y1 = 1, 1
    renderer.render(p, seed_tex=np.array([[2, 0], [0, 0]], dtype=np.uint32))
    assert tuple(renderer.readback(rgb8=False)[0][0, 0]) != want


def test_reference_configuration_as_shipped_matches_the_checker(renderer):
    """The reference's own render (default scene, 320x180, 8 spp, depth 50, its generator) with nothing overridden:
    the library scans this six-sphere scene whatever is asked for, and the frame is the CPU restatement's bit for bit."""
    import oracle
    scene = rt.build_scene(rt.SCENE_DEFAULT)
    p = rt.make_params(rt.scene_camera(rt.SCENE_DEFAULT), 320, 180, 8, rng_mode=rt.RNG_XORSHIFT_REF)
    acc, _ = oracle.render(scene, p, threads=8)
    want = oracle.resolve(acc, 8)[0]
    for accel in (rt.ACCEL_LBVH, rt.ACCEL_LINEAR):
        renderer.upload(scene, accel)
        assert renderer.scene_info()["builder"] == "none"
        renderer.clear()
        renderer.render(p)
        got = renderer.readback(rgb8=False)[0]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
