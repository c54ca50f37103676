"""The seed image of the reference (shaders/perlin.fs drawn into an RGB8 texture, raytrace.cpp:309-328, read back
at testFragment.fs:105-106).  No GL device exists here and GLSL leaves sin() to the driver, so the reference
cannot pin these values: parity is between the HIP kernel and the CPU restatement, which both take sin as the
double sine rounded to binary32."""
import numpy as np
import pytest

import oracle
import raytrace_amd as rt


def _word(level):
    c = np.float32(level) / np.float32(255.0)
    return np.array([(c + c) + c], dtype=np.float32).view(np.uint32)[0]


LEVEL_WORDS = np.array([_word(k) for k in range(256)], dtype=np.uint32)


def test_oracle_words_are_rgb8_levels():
    tex = oracle.seed_texture(96, 54)
    assert tex.shape == (54, 96) and tex.dtype == np.uint32
    assert np.isin(tex, LEVEL_WORDS).all()
    assert len(np.unique(tex)) > 8, "twelve octaves must give more than a few grey levels"


def test_oracle_no_layers_is_mid_grey():
    # value 0 -> 0.5 -> round(127.5) = 128 -> three times 128/255
    tex = oracle.seed_texture(8, 5, layers=0)
    assert (tex == _word(128)).all()


def test_oracle_rows_are_flipped():
    """Ray-shader row y reads texture row H-1-y: a taller image keeps its bottom rows."""
    a = oracle.seed_texture(40, 10)
    b = oracle.seed_texture(40, 16)
    assert (a == b[6:]).all()


def test_oracle_clamps():
    hi = oracle.seed_texture(32, 32, contrast=1e6)
    assert set(np.unique(hi)) <= {_word(0), _word(255), _word(128)}


@pytest.mark.gpu
@pytest.mark.parametrize("shape,kw", [((320, 180), {}), ((37, 21), dict(grid=50.0, contrast=2.0, layers=5, lacunarity=2.0)),
                                      ((1, 1), {}), ((130, 7), dict(layers=0))])
def test_gpu_seed_texture_matches_oracle(renderer, shape, kw):
    w, h = shape
    got = renderer.seed_texture(w, h, **kw)
    want = oracle.seed_texture(w, h, **kw)
    assert got.shape == want.shape
    assert (got == want).all(), f"{int((got != want).sum())} of {w * h} seed words differ"


@pytest.mark.gpu
def test_gpu_seed_texture_rejects_bad_arguments(renderer):
    with pytest.raises(Exception):
        renderer.seed_texture(0, 4)
    with pytest.raises(Exception):
        renderer.seed_texture(4, 4, grid=0.0)
    with pytest.raises(Exception):
        renderer.seed_texture(4, 4, layers=-1)


@pytest.mark.gpu
def test_reference_configuration_with_its_seed_image(renderer):
    """The reference's run end to end: perlin.fs seed image, then the 320x180x8 render of the default scene
    seen from lookfrom (from lookat, as the reference sets camera_origin, every path dies inside the ground)."""
    scene = rt.build_scene(rt.SCENE_DEFAULT)
    cam = rt.scene_camera(rt.SCENE_DEFAULT)
    cam.origin_is_lookat = 0
    p = rt.make_params(cam, 320, 180, 8, rng_mode=rt.RNG_XORSHIFT_REF)
    seed = renderer.seed_texture(320, 180)
    renderer.upload(scene, rt.ACCEL_LBVH)
    renderer.clear()
    renderer.render(p, seed_tex=seed)
    lin, _ = renderer.readback()
    acc, _ = oracle.render(scene, p, threads=8, seed_tex=oracle.seed_texture(320, 180))
    olin, _ = oracle.resolve(acc, 8)
    assert olin.mean() > 0.01
    assert (lin.view(np.uint32) == olin.view(np.uint32)).all()
