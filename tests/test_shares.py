"""Shares of a frame as the multi-GPU path cuts it — row bands, interleaved strips of eight rows, sample ranges —
rendered one after the other on ONE device into one accumulator: together they must give the one-call frame.
Disjoint pixels: bit for bit.  Sample ranges summed in another order: within float32 rounding of the sum."""
import numpy as np
import pytest

import raytrace_amd as rt
from raytrace_amd import distributed as rtd


def test_interleaved_shares_cover_every_row_once():
    for height in (1, 7, 8, 9, 85, 1080, 2160):
        for world, groups in ((1, 1), (2, 1), (3, 1), (8, 1), (8, 2), (4, 4)):
            cover = np.zeros((height, groups), dtype=np.int32)
            for sh in rtd.split_interleaved(height, 64 * groups, world, groups):
                rows = np.arange(height)
                mine = rows[(rows // 8) % max(sh.strips, 1) == sh.strip] if sh.strips else rows
                cover[mine, sh.s0 // 64] += 1
                assert (sh.s1 - sh.s0) == 64
            assert (cover == 1).all(), (height, world, groups)


def test_oracle_strips_compose():
    """The CPU restatement honours strip_count / strip_index the same way (it is what the GPU shares are held to)."""
    import oracle
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    p = rt.make_params(rt.scene_camera(rt.SCENE_RANDOM_SPHERES), 40, 29, 2, rng_mode=rt.RNG_COUNTER)
    whole, _ = oracle.render(scene, p, threads=4)
    parts = np.zeros_like(whole)
    for sh in rtd.split_interleaved(29, 2, 3):
        a, _ = oracle.render(scene, rtd.apply_share(p, sh), threads=4)
        assert not (a != 0)[parts != 0].any(), "shares overlap"
        parts += a
    assert np.array_equal(parts, whole)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _one_call(renderer, p):
    renderer.clear()
    renderer.render(p)
    return renderer.readback(rgb8=False)[0]


def _shares(renderer, p, shares):
    renderer.clear()
    for sh in shares:
        renderer.render(rtd.apply_share(p, sh))
    renderer.set_accumulated_samples(p.sample_end - p.sample_begin)
    return renderer.readback(rgb8=False)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("accel", [rt.ACCEL_LBVH, rt.ACCEL_LINEAR])
@pytest.mark.parametrize("size,spp,world", [((150, 85), 6, 3), ((333, 203), 64, 8), ((160, 9), 5, 4)])
def test_interleaved_strips_compose_bit_for_bit(renderer, accel, size, spp, world):
    """Ragged heights (a short last strip, fewer strips than ranks), the one-pass, swept and two-pass schedules."""
    import oracle
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    p = rt.make_params(rt.scene_camera(rt.SCENE_RANDOM_SPHERES), size[0], size[1], spp, rng_mode=rt.RNG_COUNTER)
    renderer.upload(scene, accel)
    whole = _one_call(renderer, p)
    parts = _shares(renderer, p, rtd.split_interleaved(size[1], spp, world))
    assert np.array_equal(_bits(parts), _bits(whole))
    if size[0] * size[1] * spp < 200000:
        acc, _ = oracle.render(scene, p, threads=8)
        assert np.array_equal(_bits(whole), _bits(oracle.resolve(acc, spp)[0]))
    # one share alone equals the oracle's rendering of that share (the other rows stay zero)
    sh = rtd.split_interleaved(size[1], spp, world)[world - 1]
    if size[0] * size[1] * spp < 200000:
        renderer.clear()
        renderer.render(rtd.apply_share(p, sh))
        renderer.set_accumulated_samples(spp)
        got = renderer.readback(rgb8=False)[0]
        acc, _ = oracle.render(scene, rtd.apply_share(p, sh), threads=8)
        assert np.array_equal(_bits(got), _bits(oracle.resolve(acc, spp)[0]))


@pytest.mark.gpu
def test_c5_frame_from_the_shares_of_eight_ranks(renderer):
    """3840x2160 (the 8-GPU configuration's frame) at 64 samples per pixel: the shares of split_frame(2160, S, 8, 2)
    — two row bands x four sample ranges — of eight interleaved strip sets (what bench.py --config c5 gives eight
    ranks), and of four strip sets x two sample ranges (the kind of share it gives ranks whose strips alone would leave
    a lane about eight pixels: four ranks run as 2 x 2), each rendered into one accumulator, against the one-call
    frame; and the tree against the scan on the same frame."""
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    W, H, S = 3840, 2160, 64
    p = rt.make_params(rt.scene_camera(rt.SCENE_RANDOM_SPHERES), W, H, S, rng_mode=rt.RNG_COUNTER)
    renderer.upload(scene, rt.ACCEL_LBVH)
    whole = _one_call(renderer, p)
    assert 0.2 < np.nanmean(whole) < 1.5
    strips = _shares(renderer, p, rtd.split_interleaved(H, S, 8))
    assert np.array_equal(_bits(strips), _bits(whole)), "interleaved strips: disjoint pixels, same samples"
    # bands x sample ranges in rank order: within a band the ranges arrive in increasing order, so each pixel's sum
    # goes on from the accumulator exactly as in one call
    banded = _shares(renderer, p, rtd.split_frame(H, S, 8, 2))
    assert np.array_equal(_bits(banded), _bits(whole)), "row bands x sample ranges in order"
    # strip sets x sample ranges: four interleaved strip sets x two sample ranges (in rank order the first range of
    # every strip set comes before the second, so the sums go on in sample order)
    mixed = _shares(renderer, p, rtd.split_interleaved(H, S, 8, sample_groups=2))
    assert np.array_equal(_bits(mixed), _bits(whole)), "interleaved strips x sample ranges in order"
    # what an RCCL sum of separate per-rank accumulators gives: the same samples added in another order
    for shares in (rtd.split_interleaved(H, S, 8, sample_groups=2),):
        total = np.zeros_like(whole, dtype=np.float64)
        for sh in shares:
            renderer.clear()
            renderer.render(rtd.apply_share(p, sh))
            renderer.set_accumulated_samples(S)
            total += renderer.readback(rgb8=False)[0].astype(np.float32)
        ok = np.isfinite(whole)
        np.testing.assert_allclose(total.astype(np.float32)[ok], whole[ok], rtol=2e-5, atol=1e-6)
    total = np.zeros_like(whole, dtype=np.float64)
    for sh in rtd.split_frame(H, S, 8, 2):
        renderer.clear()
        renderer.render(rtd.apply_share(p, sh))
        renderer.set_accumulated_samples(S)
        total += renderer.readback(rgb8=False)[0].astype(np.float32)
    ok = np.isfinite(whole)
    np.testing.assert_allclose(total.astype(np.float32)[ok], whole[ok], rtol=2e-5, atol=1e-6)
    renderer.upload(scene, rt.ACCEL_LINEAR)
    q = rt.make_params(rt.scene_camera(rt.SCENE_RANDOM_SPHERES), W, H, 8, rng_mode=rt.RNG_COUNTER)
    scan = _one_call(renderer, q)
    renderer.upload(scene, rt.ACCEL_LBVH)
    tree = _one_call(renderer, q)
    assert np.array_equal(_bits(tree), _bits(scan)), "tree vs scan on the 4K frame"


@pytest.mark.gpu
def test_c5_frame_at_its_stated_count_strips_equal_one_call(renderer):
    """3840x2160 x 4096 spp (34 G paths, the 8-GPU configuration's whole job): at this count the per-sample store
    cannot hold every pixel's samples, so only the most expensive pixels are cut into jobs and the main pass runs in the
    form of the kernel whose lanes sum whole pixels in registers — a schedule no smaller frame reaches.  The eight
    interleaved strip sets (each a frame of 270 rows, scheduled on its own: four pixels per lane, so each is rendered
    in two chunks of 2048 samples with every sample stored, the second continuing the first's sums) against the frame
    in one call."""
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    W, H, S = 3840, 2160, 4096
    p = rt.make_params(rt.scene_camera(rt.SCENE_RANDOM_SPHERES), W, H, S, rng_mode=rt.RNG_COUNTER)
    renderer.upload(scene, rt.ACCEL_LBVH)
    whole = _one_call(renderer, p)
    assert renderer.last_kernel_ms()[1] == "rt_render_stored_kernel<lbvh4,lds,pad_ray>"
    assert 0.2 < np.nanmean(whole) < 1.5
    strips = _shares(renderer, p, rtd.split_interleaved(H, S, 8))
    same = (_bits(strips) == _bits(whole)) | (np.isnan(strips) & np.isnan(whole))   # (NaN payloads are not part of the contract)
    assert same.all(), f"{(~same).sum()} values differ"
    # 34 G paths reach the rare chains of refractions that end in a NaN (about one path in 2 * 10^8, as in the shader): the
    # CPU restatement gives a NaN at the same pixels
    bad = np.argwhere(np.isnan(whole).any(axis=2))
    assert len(bad) < 1000, len(bad)
    import oracle
    for (y, x) in bad[:2]:
        q = p.copy()
        q.tile_x0, q.tile_x1, q.tile_y0, q.tile_y1 = int(x), int(x) + 1, int(y), int(y) + 1
        acc, _ = oracle.render(scene, q, threads=1)
        assert np.isnan(oracle.resolve(acc, S)[0][y, x]).any(), (x, y)
