"""GPU parity (through the C ABI) of ov2_compute_keypoints against the oracle."""
import numpy as np
import pytest

import ov2slam_amd

pytestmark = pytest.mark.gpu

K = (458.654, 457.296, 367.215, 248.375)
D4 = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)


def _pts(n, seed):
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(-20, 772, n), rng.uniform(-20, 500, n)], 1).astype(np.float32)


@pytest.mark.parametrize("coeffs", [D4, D4 + (0.01,), D4 + (0.01, 0.02, -0.01, 0.005), D4 + (0.01, 0.02, -0.01, 0.005, 1e-3, -2e-3, 1e-3, 5e-4), None])
def test_pinhole_bit_exact(gpu_ctx, oracle, coeffs):
    cal = ov2slam_amd.CameraCalibration(gpu_ctx, "pinhole", *K, D=coeffs)
    for n in (1, 308, 5000):
        px = _pts(n, n)
        u, b = cal.computeKeypoints(px)
        ru, rb = oracle.compute_keypoints(oracle.CAM_PINHOLE, K, coeffs, cal.iK, px)
        assert np.array_equal(u, ru)                 # +,-,*,/ in fp64 without contraction: bit-exact
        assert np.array_equal(b, rb)
    assert np.array_equal(cal.undistortImagePoint((100.5, 200.25)), oracle.compute_keypoints(oracle.CAM_PINHOLE, K, coeffs, cal.iK, [[100.5, 200.25]])[0][0])


def test_pinhole_strong_distortion_negative_icdist(gpu_ctx, oracle):
    """icdist < 0 branch (far outside the valid radius): the point falls back to the normalised input."""
    coeffs = (-2.5, 0.0, 0.0, 0.0)
    cal = ov2slam_amd.CameraCalibration(gpu_ctx, "pinhole", *K, D=coeffs)
    px = np.array([[900.0, 700.0], [-300.0, -200.0], [367.0, 248.0]], np.float32)
    u, b = cal.computeKeypoints(px)
    ru, rb = oracle.compute_keypoints(oracle.CAM_PINHOLE, K, coeffs, cal.iK, px)
    assert np.array_equal(u, ru) and np.array_equal(b, rb)


def test_fisheye(gpu_ctx, oracle):
    kf = (-0.02, 0.004, -0.001, 0.0002)
    cal = ov2slam_amd.CameraCalibration(gpu_ctx, "fisheye", *K, D=kf)
    px = np.concatenate([_pts(4000, 3), np.array([[K[2], K[3]]], np.float32)])
    u, b = cal.computeKeypoints(px)
    ru, rb = oracle.compute_keypoints(oracle.CAM_FISHEYE, K, kf, cal.iK, px)
    # tan() may differ in its last bit between the device and the host libm: allow 1 float ulp on the pixel
    assert np.abs(u - ru).max() <= 6.2e-5 * 2
    assert (u != ru).mean() < 0.01
    same = np.all(u == ru, axis=1)
    assert np.array_equal(b[same], rb[same])


def test_empty_and_errors(gpu_ctx):
    cal = ov2slam_amd.CameraCalibration(gpu_ctx, "pinhole", *K, D=D4)
    u, b = cal.computeKeypoints(np.zeros((0, 2), np.float32))
    assert u.shape == (0, 2) and b.shape == (0, 3)
    bad = ov2slam_amd.CameraCalibration(gpu_ctx, "pinhole", *K, D=(0.1, 0.2, 0.3))
    with pytest.raises(ov2slam_amd.Ov2Error):
        bad.computeKeypoints(np.zeros((2, 2), np.float32))
