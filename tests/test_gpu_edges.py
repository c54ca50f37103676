"""GPU tests of what rendered frames of the stock scenes do not reach: the noise function against the
reference's own vectors, the shader's rare branches against the CPU oracle, and the memory-budget fallbacks.

Parity status of the branch tests: unpinned by the reference (it holds no vectors for its shader); they hold the
HIP path to the line-by-line CPU restatement.  The noise tests ARE pinned: tests/golden/ref_values.json was
printed by the reference's unmodified perlin.h."""
import struct

import numpy as np
import pytest

import oracle
import raytrace_amd as rt

# (small_trees: tree kernels also on the scenes of a few spheres that the library would scan, see conftest.py)
pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("small_trees")]


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_same_f32(a, b, what):
    """Bit-identical; any NaN equals any NaN (x86 and gfx950 produce different default NaN payloads)."""
    fa, fb = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    bad = np.flatnonzero(((_bits(fa) != _bits(fb)) & ~(np.isnan(fa) & np.isnan(fb))).ravel())
    assert bad.size == 0, f"{what}: {bad.size} of {fa.size} values differ, first at {bad[:5]}"


def sphere(mat, r, c):
    return struct.pack("<if2i3fi", mat, r, 0, 0, c[0], c[1], c[2], 0)


def material(ident, kind, param1, albedo):
    return struct.pack("<iifi3fi", ident, kind, param1, 0, albedo[0], albedo[1], albedo[2], 0)


# ---- perlin.h:31-85 on the device against the reference's own values ---------------------------------------

@pytest.mark.parametrize("use_table", [True, False])
@pytest.mark.parametrize("block", ["perlin", "perlin_far"])
def test_device_perlin_matches_reference_header(renderer, golden, use_table, block):
    """Every vector through the device function the noise material calls: with the gradient table (lattice points
    inside +-2048, the others computed on the spot) and without it.  Bit-exact."""
    v = np.array(golden[block], dtype=np.uint32)
    xy = v[:, :2].copy().view(np.float32)
    got = renderer.eval_perlin(xy, use_table=use_table)
    bad = np.flatnonzero(_bits(got) != v[:, 2])
    assert bad.size == 0, f"{bad.size} of {len(v)} perlin values differ from the reference's, first {xy[bad[:3]]}"
    if block == "perlin_far":
        cell = np.floor(xy)
        assert ((np.abs(cell) > 2048).any(axis=1)).sum() > 100, "the far block must leave the table"
        assert ((np.abs(cell) < 2047).all(axis=1)).sum() > 10, "and cross its border from inside"


def test_device_gradient_matches_reference_header(renderer, golden):
    v = np.array(golden["gradients"] + golden["gradients_far"], dtype=np.int64)
    got = renderer.eval_perlin_gradient(v[:, :2].astype(np.int32))
    want = v[:, 2:].astype(np.uint32)
    assert np.array_equal(_bits(got), want), f"{(_bits(got) != want).any(axis=1).sum()} of {len(v)} gradients differ"


def test_perlin_table_equals_direct_on_random_points(renderer):
    """Size-independent property: table and direct evaluation agree bit for bit on a million points
    that straddle the table's border."""
    rng = np.random.default_rng(5)
    xy = rng.uniform(-2300, 2300, size=(1 << 20, 2)).astype(np.float32)
    a = renderer.eval_perlin(xy, use_table=True)
    b = renderer.eval_perlin(xy, use_table=False)
    assert np.array_equal(_bits(a), _bits(b))
    assert 0.0 <= a.min() and a.max() <= 1.0 and 0.4 < a.mean() < 0.6


def test_perlin_material_outside_the_table(renderer):
    """A noise-textured ground whose texture coordinates k * p run far beyond the table (k = 4000)."""
    mats = material(0, rt.MAT_PERLIN_LAMBERTIAN, 4000.0, (0.9, 0.8, 0.7))
    sc = rt.Scene(sphere(0, 1000.0, (0, -1000, 0)) + sphere(0, 2.0, (0, 2, 0)), mats)
    cam = rt.scene_camera(rt.SCENE_PERLIN)
    p = rt.make_params(cam, 96, 54, 3, rng_mode=rt.RNG_COUNTER)
    for accel in (rt.ACCEL_LINEAR, rt.ACCEL_LBVH):
        renderer.upload(sc, accel)
        renderer.clear()
        renderer.render(p, counted=True)
        lin, _ = renderer.readback(rgb8=False)
        c = renderer.counters()
        acc, oc = oracle.render(sc, p, threads=8)
        assert c["perlin"] == oc["perlin"] > 1000
        assert_same_f32(lin, oracle.resolve(acc, 3)[0], f"noise far outside the table, accel {accel}")


# ---- rare branches of the shader ---------------------------------------------------------------------------

def _camera():
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
    return cam


def _frame_vs_oracle(renderer, scene, p, what, seed_tex=None):
    out = {}
    for accel in (rt.ACCEL_LINEAR, rt.ACCEL_LBVH):
        renderer.upload(scene, accel)
        renderer.clear()
        renderer.render(p, counted=True, seed_tex=seed_tex)
        lin, _ = renderer.readback(rgb8=False)
        c = renderer.counters()
        acc, oc = oracle.render(scene, p, threads=8, seed_tex=seed_tex)
        assert_same_f32(lin, oracle.resolve(acc, p.sample_end - p.sample_begin)[0], f"{what}, accel {accel}")
        for k in ("paths", "segments", "hits", "sky", "lambertian", "metallic", "dialectric", "unit_tries", "disk_tries",
                  "limit_hits"):
            assert c[k] == oc[k], (what, accel, k)
        out[accel] = (lin, c)
    return out[rt.ACCEL_LBVH]


@pytest.mark.parametrize("rng_mode", [rt.RNG_COUNTER, rt.RNG_XORSHIFT_REF])
def test_unknown_material_paints_red_and_stops(renderer, rng_mode):
    """testFragment.fs:235-237: a type outside 1..3 sets the throughput to (1,0,0) and ends the path."""
    mats = (material(0, 1, 0.0, (0.8, 0.8, 0.0)) + material(1, 0, 0.0, (0.3, 0.3, 0.3)) +     # NOTHING
            material(2, 7, 0.5, (0.1, 0.9, 0.1)) + material(3, -5, 0.0, (0.0, 0.0, 0.0)))     # never defined
    sc = rt.Scene(sphere(0, 100.0, (0, -100.5, -1)) + sphere(1, 0.5, (0, 0, -1.2)) + sphere(2, 0.5, (-1, 0, -1)) +
                  sphere(3, 0.5, (1, 0, -1)), mats)
    p = rt.make_params(_camera(), 120, 68, 4, rng_mode=rng_mode)
    lin, c = _frame_vs_oracle(renderer, sc, p, "unknown material")
    assert c["hits"] > c["lambertian"] + c["metallic"] + c["dialectric"] > 0, "some hits must take the red branch"
    centre = lin[34, 60]
    assert centre[0] > 0.5 and centre[1] < 0.3 * centre[0] and centre[2] < 0.3 * centre[0], f"centre pixel {centre}"


@pytest.mark.parametrize("limit", [0, 1, 2])
@pytest.mark.parametrize("rng_mode", [rt.RNG_COUNTER, rt.RNG_XORSHIFT_REF])
def test_small_bounce_limits(renderer, limit, rng_mode):
    """testFragment.fs:203-205: count >= bounce_limit zeroes the path before any hit test (limit 0: a black frame
    with no segments at all; limit 1: only what the camera ray sees of the sky)."""
    scene = rt.build_scene(rt.SCENE_DEFAULT)
    p = rt.make_params(_camera(), 120, 68, 3, bounce_limit=limit, rng_mode=rng_mode)
    lin, c = _frame_vs_oracle(renderer, scene, p, f"bounce_limit {limit}")
    if limit == 0:
        assert c["segments"] == 0 and c["limit_hits"] == c["paths"] == 120 * 68 * 3 and not lin.any()
    else:
        assert c["limit_hits"] > 0 and c["sky"] > 0 and lin.any()


def test_fuzzy_metal_absorbs(renderer):
    """testFragment.fs:260-263: a fuzzed reflection that ends below the surface is absorbed."""
    mats = material(0, 1, 0.0, (0.5, 0.5, 0.5)) + material(1, 2, 0.9, (0.9, 0.9, 0.9)) + material(2, 2, 3.0, (0.7, 0.6, 0.5))
    sc = rt.Scene(sphere(0, 100.0, (0, -100.5, -1)) + sphere(1, 0.5, (0, 0, -1.2)) + sphere(2, 0.5, (-1, 0, -1)) +
                  sphere(1, 0.5, (1, 0, -1)), mats)
    p = rt.make_params(_camera(), 120, 68, 4, rng_mode=rt.RNG_COUNTER)
    lin, c = _frame_vs_oracle(renderer, sc, p, "fuzzy metal")
    # a metallic scatter either continues (another segment) or is absorbed: with fuzz 0.9 and 3.0 many are
    ended_by_metal = c["paths"] - c["sky"] - c["limit_hits"]
    assert c["metallic"] > 1000 and ended_by_metal > c["metallic"] // 20


@pytest.mark.parametrize("ground_r,inner_r,side_r", [(100.0, -0.4, 0.5), (100.0, 0.0, 0.0), (-100.0, 0.4, -0.5)])
def test_negative_and_zero_radius(renderer, ground_r, inner_r, side_r):
    """testFragment.fs:148,189: r enters as r*r in the test and as 1/r in the normal (a negative radius flips the
    normal: the hollow-glass trick; zero gives an infinite factor)."""
    mats = (material(0, 1, 0.0, (0.8, 0.8, 0.0)) + material(1, 3, 1.5, (0.0, 0.0, 0.0)) + material(2, 1, 0.0, (0.1, 0.2, 0.5)))
    sc = rt.Scene(sphere(0, ground_r, (0, -100.5, -1)) + sphere(1, 0.5, (0, 0, -1.2)) + sphere(1, inner_r, (0, 0, -1.2)) +
                  sphere(2, side_r, (1, 0, -1)), mats)
    p = rt.make_params(_camera(), 120, 68, 3, rng_mode=rt.RNG_COUNTER)
    _frame_vs_oracle(renderer, sc, p, f"radii {ground_r} {inner_r} {side_r}")
    rng = np.random.default_rng(3)
    o = rng.uniform(-2, 2, (4096, 3)).astype(np.float32)
    d = rng.normal(size=(4096, 3)).astype(np.float32)
    o[:64] = (0, 0, -1.2)  # from the centre of the zero-radius sphere: its discriminant is exactly h*h
    want = oracle.trace(sc, o, d, threads=4)
    for accel in (rt.ACCEL_LINEAR, rt.ACCEL_LBVH):
        renderer.upload(sc, accel)
        got = renderer.trace(o, d)
        assert np.array_equal(got[0], want[0])
        assert_same_f32(got[1], want[1], "t")
        assert_same_f32(got[2], want[2], "normal")


def test_more_big_spheres_than_the_pretest_list_holds(renderer):
    """Spheres far larger than the rest are tested exactly ahead of the tree, at most eight of them; with twelve
    the four smallest go into the tree and blow its boxes up.  Results must not notice."""
    rng = np.random.default_rng(11)
    mats = b"".join(material(i, 1 + i % 3, 0.3 if i % 3 == 1 else 1.5, rng.uniform(0.2, 0.9, 3)) for i in range(16))
    recs = [sphere(int(rng.integers(0, 16)), 0.2, rng.uniform(-8, 8, 3) * (1, 0.2, 1)) for _ in range(300)]
    recs += [sphere(i % 16, 20.0 + 3.0 * i, (60.0 * np.cos(i), -25.0 - 3.0 * i, 60.0 * np.sin(i))) for i in range(12)]
    order = rng.permutation(len(recs))
    sc = rt.Scene(b"".join(recs[i] for i in order), mats)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    p = rt.make_params(cam, 128, 72, 3, rng_mode=rt.RNG_COUNTER)
    _frame_vs_oracle(renderer, sc, p, "twelve big spheres")
    o = rng.uniform(-10, 10, (8192, 3)).astype(np.float32)
    d = rng.normal(size=(8192, 3)).astype(np.float32)
    want = oracle.trace(sc, o, d, threads=8)
    renderer.upload(sc, rt.ACCEL_LBVH)
    got = renderer.trace(o, d)
    assert np.array_equal(got[0], want[0]) and (want[0] >= 0).sum() > 2000
    assert_same_f32(got[1], want[1], "t")
    assert_same_f32(got[2], want[2], "normal")


# ---- argument rules and memory budgets ---------------------------------------------------------------------

def test_reference_rng_refuses_a_later_sample_range(renderer):
    scene = rt.build_scene(rt.SCENE_DEFAULT)
    renderer.upload(scene, rt.ACCEL_LBVH)
    p = rt.make_params(_camera(), 64, 36, 4, rng_mode=rt.RNG_XORSHIFT_REF, sample_begin=4)
    with pytest.raises(rt.RtError) as err:
        renderer.render(p)
    assert err.value.code == -1 and "sample_begin" in str(err.value)


@pytest.mark.parametrize("split_mb,probe_mb", [(1, 1), (64, 1), (1, 4096), (0, 0)])
def test_tiny_scratch_budgets_give_the_same_frame(renderer, split_mb, probe_mb):
    """The per-sample stores of the schedule are optional: with budgets too small for them (and with the
    defaults) the frame is bit for bit the oracle's."""
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    p = rt.make_params(cam, 192, 108, 64, rng_mode=rt.RNG_COUNTER)   # two-pass territory: 20 736 pixels, 64 spp
    acc, _ = oracle.render(scene, p, threads=8)
    want = oracle.resolve(acc, 64)[0]
    try:
        renderer.set_scratch_budget(split_mb << 20, probe_mb << 20)
        renderer.upload(scene, rt.ACCEL_LBVH)
        renderer.clear()
        renderer.render(p)
        assert_same_f32(renderer.readback(rgb8=False)[0], want, f"budgets {split_mb} / {probe_mb} MiB")
        if split_mb == 1:
            # only some pixels have rows in the store: the main pass ran in the form of the stored-samples kernel whose
            # lanes sum whole pixels in registers (and not in the kernel without camera rays in reserve)
            assert renderer.last_kernel_ms()[1] == "rt_render_stored_kernel<lbvh4,lds,pad_ray>"
        few = rt.make_params(cam, 192, 108, 6, rng_mode=rt.RNG_COUNTER)    # whole frame in sweeps, or not
        renderer.clear()
        renderer.render(few)
        acc6, _ = oracle.render(scene, few, threads=8)
        assert_same_f32(renderer.readback(rgb8=False)[0], oracle.resolve(acc6, 6)[0], "6 spp")
    finally:
        renderer.set_scratch_budget(0, 0)


@pytest.mark.parametrize("strips", [0, 2])
def test_sample_chunks_give_the_frame_of_one_pass(renderer, strips):
    """A sample range too long for the store's budget to give every pixel a row is rendered in chunks of samples that
    continue each other's sums: bit for bit the oracle's frame, whole and as one rank's strips."""
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    w, h, spp = 256, 144, 200   # (the strips' half of it still has the 16 384 pixels two-pass rendering asks for)
    p = rt.make_params(cam, w, h, spp, rng_mode=rt.RNG_COUNTER)
    if strips:
        p.strip_count, p.strip_index = strips, 1
    acc, _ = oracle.render(scene, p, threads=8)
    want = oracle.resolve(acc, spp)[0]
    try:
        renderer.upload(scene, rt.ACCEL_LBVH)
        # rows of 72 samples for every pixel of the whole frame: chunks of 68 (whole), of 100 (the strips' half of it)
        renderer.set_scratch_budget(w * h * 16 * 72, 0)
        renderer.clear()
        renderer.render(p, counted=True)
        chunked = renderer.counters()
        assert_same_f32(renderer.readback(rgb8=False)[0], want, "in chunks")
        assert renderer.last_kernel_ms()[1] == "rt_render_stored_kernel<lbvh4,lds,pad_ray>"
        renderer.set_scratch_budget(0, 0)
        renderer.clear()
        renderer.render(p, counted=True)
        whole = renderer.counters()
        assert_same_f32(renderer.readback(rgb8=False)[0], want, "one pass")
        for k in ("paths", "segments", "hits"):
            assert chunked[k] == whole[k], k
    finally:
        renderer.set_scratch_budget(0, 0)


@pytest.mark.parametrize("give_up_after", [1, 3, 12, 0])
def test_probe_paths_given_up_and_rendered_again(renderer, monkeypatch, give_up_after):
    """The probe pass of a two-pass frame gives a path up after so many segments and leaves the sample to a job of the
    main pass, which renders it again from its counter key; the sums are formed once it is there.  Bit for bit the
    oracle's frame whether nearly every sample goes that way (1), many (3), few (12, the default), none (0); whole and
    as one rank's strips; with the tree and with the scan over the 487 spheres."""
    monkeypatch.setenv("RT_PROBE_ABORT", str(give_up_after))
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    w, h, spp = 256, 144, 64
    for strips in (0, 2):
        p = rt.make_params(cam, w, h, spp, rng_mode=rt.RNG_COUNTER)
        both = rt.make_params(cam, w, h, 2 * spp, rng_mode=rt.RNG_COUNTER)
        if strips:
            p.strip_count, p.strip_index = strips, 1
            both.strip_count, both.strip_index = strips, 1
        want = oracle.resolve(oracle.render(scene, p, threads=8)[0], spp)[0]
        want2 = oracle.resolve(oracle.render(scene, both, threads=8)[0], 2 * spp)[0]
        for accel, kernel in ((rt.ACCEL_LBVH, "rt_render_stored_kernel<lbvh4,lds,pad_ray>"), (rt.ACCEL_LINEAR, "rt_render_stored_kernel<linear,lds>")):
            renderer.upload(scene, accel)
            renderer.clear()
            renderer.render(p)
            assert_same_f32(renderer.readback(rgb8=False)[0], want, f"given up after {give_up_after}, strips {strips}, accel {accel}")
            assert renderer.last_kernel_ms()[1] == kernel
            # a second render onto the first: the sums go on from the image
            renderer.render(_next_range(p, spp))
            renderer.set_accumulated_samples(2 * spp)
            assert_same_f32(renderer.readback(rgb8=False)[0], want2, "two ranges in a row")


def _next_range(p, spp):
    q = p.copy()
    q.sample_begin, q.sample_end = p.sample_end, p.sample_end + spp
    return q


# ---- the kernels' square root ---------------------------------------------------------------------------------

def test_short_square_root_is_the_correctly_rounded_one(renderer):
    """The kernels take square roots with a shortened sequence wherever a whole wavefront holds ordinary arguments
    (sqrt_rn in csrc/rt_device.cuh).  Held to the compiler's correctly rounded sqrtf over ALL 2^32 bit patterns on the
    device, and to numpy's (IEEE, correctly rounded) on values around every boundary of the short path, in wavefronts
    that are all ordinary and in mixed ones."""
    bad, example = renderer.check_sqrt(0, 1 << 32)
    assert bad == 0, f"{bad} arguments differ, e.g. bits {example:#010x}"
    edge_bits = np.array([0x00000000, 0x00000001, 0x007FFFFF, 0x00800000, 0x0F7FFFFF, 0x0F800000, 0x0F800001, 0x3F7FFFFF, 0x3F800000,
                          0x3F800001, 0x40490FDB, 0x7F7FFFFF, 0x7F800000, 0x7FC00000, 0x80000000, 0x80000001, 0xBF800000, 0xFF800000],
                         dtype=np.uint32)
    rng = np.random.default_rng(11)
    ordinary = rng.integers(0x0F800000, 0x7F800000, size=64 * 512, dtype=np.uint32)      # whole wavefronts on the short path
    mixed = np.concatenate([edge_bits, rng.integers(0, 1 << 32, size=64 * 512 - edge_bits.size, dtype=np.uint64).astype(np.uint32)])
    for bits in (ordinary, mixed):
        x = bits.view(np.float32)
        with np.errstate(invalid="ignore"):
            want = np.sqrt(x)
        assert_same_f32(renderer.eval_sqrt(x), want, "sqrt")
