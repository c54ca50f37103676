"""Image agreement with a converged render, stated as RMSE / PSNR bounds.

The GPU renders the 487-sphere scene at S samples per pixel; the CPU oracle renders the same
frame at 16 S with a different seed (the "converged" reference: no sample is shared, so this is
a statistical comparison, unlike the bit-exact tests elsewhere).  The bound comes from the
measured variance: two independent GPU renders a, b at S samples differ by 2 sigma^2 / S per
value in mean square, so the S-sample image should differ from the 16 S reference by
sigma^2 / S * (1 + 1/16).  The test allows 1.5 times that in RMSE (3.5 dB in PSNR) and, for the
image mean — where a biased sampler would show first — four standard errors.
"""
import math

import numpy as np
import pytest

import oracle
import raytrace_amd as rt

pytestmark = pytest.mark.gpu

W, H, S = 160, 90, 64
SEEDS = (0x5EED0001, 0x0BADCAFE, 0x13579BDF)  # render a, render b, reference


def _gpu_frame(renderer, scene, cam, seed):
    p = rt.make_params(cam, W, H, S, rng_mode=rt.RNG_COUNTER, seed=seed)
    renderer.clear()
    renderer.render(p)
    return renderer.readback()


@pytest.mark.parametrize("accel", [rt.ACCEL_LBVH, rt.ACCEL_LINEAR])
def test_frame_converges_to_the_high_spp_reference(renderer, accel):
    scene = rt.build_scene(rt.SCENE_RANDOM_SPHERES)
    cam = rt.scene_camera(rt.SCENE_RANDOM_SPHERES)
    renderer.upload(scene, accel)
    a_lin, a_q = _gpu_frame(renderer, scene, cam, SEEDS[0])
    b_lin, _ = _gpu_frame(renderer, scene, cam, SEEDS[1])
    p_ref = rt.make_params(cam, W, H, 16 * S, rng_mode=rt.RNG_COUNTER, seed=SEEDS[2])
    acc, _ = oracle.render(scene, p_ref, threads=16)
    r_lin, r_q = oracle.resolve(acc, 16 * S)

    ok = np.isfinite(a_lin) & np.isfinite(b_lin) & np.isfinite(r_lin)  # the shader can produce NaN pixels
    assert ok.mean() > 0.999
    n = int(ok.sum())
    var_over_s = float(np.mean((a_lin[ok].astype(np.float64) - b_lin[ok]) ** 2)) / 2.0  # sigma^2 / S
    assert var_over_s > 0.0
    expected_mse = var_over_s * (1.0 + 1.0 / 16.0)
    diff = a_lin[ok].astype(np.float64) - r_lin[ok]
    rmse = math.sqrt(float(np.mean(diff ** 2)))
    assert rmse <= 1.5 * math.sqrt(expected_mse), f"linear RMSE {rmse:.5f} vs expected {math.sqrt(expected_mse):.5f}"
    assert rmse >= 0.5 * math.sqrt(expected_mse), "suspiciously close: are the seeds really independent?"
    # the image mean: a biased sampler shows here long before it shows in the RMSE
    stderr = math.sqrt(expected_mse / n)
    assert abs(float(np.mean(diff))) <= 4.0 * stderr + 1e-6, f"mean difference {np.mean(diff):.2e}, standard error {stderr:.2e}"

    # RGB8, as displayed: PSNR against the reference's RGB8 image, bounded the same way through
    # the PSNR of the two S-sample renders against each other
    okq = ok
    mse_q = float(np.mean((a_q[okq].astype(np.float64) - r_q[okq]) ** 2))
    psnr = 10.0 * math.log10(255.0 ** 2 / max(mse_q, 1e-12))
    b_q = np.clip(np.floor(np.power(np.clip(b_lin, 0.0, 1.0), 1.0 / 2.2) * 255.0 + 0.5), 0, 255)
    mse_ab = float(np.mean((a_q[okq].astype(np.float64) - b_q[okq]) ** 2)) / 2.0 * (1.0 + 1.0 / 16.0)
    psnr_expected = 10.0 * math.log10(255.0 ** 2 / max(mse_ab, 1e-12))
    assert psnr >= psnr_expected - 3.5, f"PSNR {psnr:.2f} dB vs expected {psnr_expected:.2f} dB"
    assert psnr >= 20.0
