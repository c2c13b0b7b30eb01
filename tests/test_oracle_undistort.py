"""Oracle checks for per-keypoint undistortion + bearing (oracle/undistort.c): no OpenCV here ("parity
unpinned"), so the restated cv::undistortPoints / cv::fisheye::undistortPoints are pinned by inverting the
published forward distortion models with an independent numpy implementation."""
import numpy as np
import pytest

# EuRoC cam0 (parameters_files/accurate/euroc/euroc_stereo.yaml)
K = (458.654, 457.296, 367.215, 248.375)
D = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)


def distort_pinhole(xn, yn, k):
    k = list(k) + [0.0] * (14 - len(k))
    r2 = xn * xn + yn * yn
    cd = (1 + k[0] * r2 + k[1] * r2 ** 2 + k[4] * r2 ** 3) / (1 + k[5] * r2 + k[6] * r2 ** 2 + k[7] * r2 ** 3)
    xd = xn * cd + 2 * k[2] * xn * yn + k[3] * (r2 + 2 * xn * xn) + k[8] * r2 + k[9] * r2 * r2
    yd = yn * cd + k[2] * (r2 + 2 * yn * yn) + 2 * k[3] * xn * yn + k[10] * r2 + k[11] * r2 * r2
    return xd, yd


def distort_fisheye(xn, yn, k):
    r = np.sqrt(xn * xn + yn * yn)
    th = np.arctan(r)
    thd = th * (1 + k[0] * th ** 2 + k[1] * th ** 4 + k[2] * th ** 6 + k[3] * th ** 8)
    s = np.where(r > 1e-12, thd / np.maximum(r, 1e-12), 1.0)
    return xn * s, yn * s


def iK_of(K):
    return np.linalg.inv(np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1.0]]))


@pytest.mark.parametrize("coeffs", [D, D + (0.01,), D + (0.01, 0.02, -0.01, 0.005), D + (0.01, 0.02, -0.01, 0.005, 1e-3, -2e-3, 1e-3, 5e-4)])
def test_pinhole_inverts_forward_model(oracle, coeffs):
    rng = np.random.default_rng(0)
    und = np.stack([rng.uniform(40, 710, 400), rng.uniform(30, 450, 400)], 1)
    xn, yn = (und[:, 0] - K[2]) / K[0], (und[:, 1] - K[3]) / K[1]
    xd, yd = distort_pinhole(xn, yn, coeffs)
    px = np.stack([xd * K[0] + K[2], yd * K[1] + K[3]], 1).astype(np.float32)
    unpx, bv = oracle.compute_keypoints(oracle.CAM_PINHOLE, K, coeffs, iK_of(K), px)
    # 5 fixed-point iterations (OpenCV's fixed count): ~0.15 px left in the image corners of the EuRoC lens,
    # < 2e-3 px inside the central half of the field of view
    err = np.abs(unpx - und).max(axis=1)
    assert err.max() < 0.25
    assert err[(np.abs(xn) < 0.4) & (np.abs(yn) < 0.25)].max() < 2e-3
    b = np.stack([xn, yn, np.ones_like(xn)], 1); b /= np.linalg.norm(b, axis=1, keepdims=True)
    assert np.abs(bv - b).max() < 6e-4
    assert np.allclose(np.linalg.norm(bv, axis=1), 1.0, atol=1e-15)


def test_pinhole_five_iterations_exactly(oracle):
    """The fixed-point map is x <- (x0 - delta(x)) / cdist(x), applied 5 times from x0 (MAX_ITER 5, no EPS test)."""
    px = np.array([[100.25, 80.5], [700.0, 20.0]], np.float32)
    unpx, _ = oracle.compute_keypoints(oracle.CAM_PINHOLE, K, D, iK_of(K), px)
    for p, u in zip(px.astype(np.float64), unpx):
        x0, y0 = (p[0] - K[2]) * (1. / K[0]), (p[1] - K[3]) * (1. / K[1])
        x, y = x0, y0
        for _ in range(5):
            r2 = x * x + y * y
            ic = 1.0 / (1 + (D[1] * r2 + D[0]) * r2)
            dx = 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x)
            dy = D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y
            x, y = (x0 - dx) * ic, (y0 - dy) * ic
        assert abs(np.float32(K[0] * x + K[2]) - u[0]) <= 1e-4 and abs(np.float32(K[1] * y + K[3]) - u[1]) <= 1e-4


def test_fisheye_inverts_forward_model(oracle):
    kf = (-0.02, 0.004, -0.001, 0.0002)
    rng = np.random.default_rng(1)
    und = np.stack([rng.uniform(5, 745, 400), rng.uniform(5, 475, 400)], 1)
    xn, yn = (und[:, 0] - K[2]) / K[0], (und[:, 1] - K[3]) / K[1]
    xd, yd = distort_fisheye(xn, yn, kf)
    px = np.stack([xd * K[0] + K[2], yd * K[1] + K[3]], 1).astype(np.float32)
    unpx, _ = oracle.compute_keypoints(oracle.CAM_FISHEYE, K, kf, iK_of(K), px)
    assert np.abs(unpx - und).max() < 2e-3          # Newton to 1e-8 on theta; float32 pixel storage dominates
    # principal point: theta_d < EPS branch
    c, _ = oracle.compute_keypoints(oracle.CAM_FISHEYE, K, kf, iK_of(K), np.array([[K[2], K[3]]], np.float32))
    assert np.allclose(c[0], [K[2], K[3]], atol=1e-4)


def test_no_distortion_returns_input(oracle):
    px = np.array([[10.5, 20.25], [300.0, 200.0]], np.float32)
    unpx, bv = oracle.compute_keypoints(oracle.CAM_PINHOLE, K, None, iK_of(K), px)
    assert np.array_equal(unpx, px)                   # Dcv_.empty(): `return pt`
    assert np.allclose(bv[:, :2] / bv[:, 2:], (px - [K[2], K[3]]) / [K[0], K[1]], atol=1e-12)
