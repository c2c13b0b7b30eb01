"""The oracle's (and the device's) reprojection factors against THE REFERENCE'S OWN CODE: /root/reference/src/ceres_parametrization.cpp is
compiled from where it lies, unchanged, against stand-in Eigen / Sophus / Ceres headers (oracle/ref/standin: the real libraries are not
in this image) into oracle/_ref/libref_factors.so (recipe: oracle/ref/Makefile).  Every Evaluate() of namespace DirectLeftSE3
(src/ceres_parametrization.cpp:104-712) and SE3LeftParameterization (se3left_parametrization.hpp:39-73) is then called on random inputs
and compared with oracle/ba.c / oracle/xyz_ba.c / oracle/struct_ba.c -- residuals, chi2err_, isdepthpositive_, every Jacobian block.

Tolerance 1e-11 relative: the stand-in evaluates the reference's expressions with plain loops, not with Eigen's evaluation order, so
agreement is to rounding, not bit for bit.  What this pins is the reference's FORMULAS (signs, frames, which block gets which
derivative) without anybody reading them.  The library is built here (where /root/reference exists) and travels to the GPU box."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_factors.so")
RTOL = 1e-11

SIZES = {0: (4, 7, 7, 1), 1: (4, 4, 7, 7, 7, 1), 2: (4, 4, 7, 1), 3: (7,), 4: (4, 7, 3), 5: (4, 7, 7, 3)}


@pytest.fixture(scope="module")
def ref():
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref")])
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libref_factors.so is absent and /root/reference is not here to build it from")
    lib = C.CDLL(REF_SO)
    lib.ref_invdepth_block.restype = C.c_double
    lib.ref_invdepth_block.argtypes = [C.c_double]
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def ref_eval(lib, ftype, params, uv, anch_uv=(0., 0.), sigma=1.0, K=(1., 1., 0., 0.), xyz=(0., 0., 1.), want_jac=True):
    """-> (r (2,), [J_block (2, size)], chi2, depthpos) from the reference's Evaluate"""
    params = [np.ascontiguousarray(p, np.float64) for p in params]
    assert tuple(len(p) for p in params) == SIZES[ftype]
    pp = (C.c_void_p * len(params))(*[p.ctypes.data for p in params])
    J = [np.full((2, len(p)), np.nan) for p in params]
    jp = (C.c_void_p * len(params))(*[j.ctypes.data for j in J])
    r = np.zeros(2); chi2 = C.c_double(0); dp = C.c_int(0)
    uv = np.ascontiguousarray(uv, np.float64); auv = np.ascontiguousarray(anch_uv, np.float64)
    Kk = np.ascontiguousarray(K, np.float64); X = np.ascontiguousarray(xyz, np.float64)
    rc = lib.ref_factor_eval(int(ftype), pp, _p(uv), _p(auv), C.c_double(sigma), _p(Kk), _p(X), _p(r), jp if want_jac else None, C.byref(chi2), C.byref(dp))
    assert rc == 0
    return r, J, chi2.value, bool(dp.value)


def _rand_pose(rng, t_scale=1.0, rot=0.4):
    w = rng.normal(0, rot, 3)
    th = np.linalg.norm(w)
    q = np.concatenate([np.sin(th / 2) * w / th, [np.cos(th / 2)]])
    return np.concatenate([rng.normal(0, t_scale, 3), q])


def _close(a, b, scale=None):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    s = max(1.0, float(np.abs(b).max()) if b.size else 1.0) if scale is None else scale
    return float(np.abs(a - b).max()) <= RTOL * s


CAL_L = np.array([458.654, 457.296, 367.215, 248.375])
CAL_R = np.array([457.587, 456.134, 379.999, 255.238])


def _case(rng):
    anchor, obs = _rand_pose(rng), _rand_pose(rng)
    T_rl = _rand_pose(rng, 0.1, 0.05); T_rl[0] -= 0.11
    lam = 1.0 / rng.uniform(0.6, 15.0)
    if rng.uniform() < 0.1:
        lam = -lam                                                   # behind the camera: isdepthpositive_ false somewhere
    auv = np.array([rng.uniform(0, 752), rng.uniform(0, 480)])
    uv = np.array([rng.uniform(0, 752), rng.uniform(0, 480)])
    return anchor, obs, T_rl, lam, auv, uv, float(rng.choice([1.0, 2.0, 4.0]))


def test_inverse_depth_factors_match_the_reference_source(ref, oracle):
    rng = np.random.default_rng(11)
    n_neg = 0
    for _ in range(300):
        anchor, obs, T_rl, lam, auv, uv, sigma = _case(rng)
        lm = np.array([lam])
        # ---- LEFT: ReprojectionErrorKSE3AnchInvDepth {calib, anchor, obs, lambda}   (:361-473)
        r, J, chi2, dp = ref_eval(ref, 0, [CAL_L, anchor, obs, lm], uv, auv, sigma)
        o_r, o_Ja, o_Jo, o_Jl, o_chi2, o_dp = oracle.ba_residual(0, CAL_L, CAL_R, T_rl, anchor, obs, lam, auv, uv, sigma)
        assert _close(o_r, r) and _close(o_chi2, chi2, max(1.0, chi2)) and o_dp == dp
        assert _close(o_Ja, J[1][:, :6]) and _close(o_Jo, J[2][:, :6]) and _close(o_Jl, J[3][:, 0])
        assert np.all(J[1][:, 6] == 0) and np.all(J[2][:, 6] == 0) and np.all(J[0] == 0)          # 7th column / calib block: zero (TODO in the reference)
        n_neg += not dp
        # ---- RIGHT: ReprojectionErrorRightCamKSE3AnchInvDepth {calib_l, calib_r, anchor, obs, T_rl, lambda}   (:579-712)
        r, J, chi2, dp = ref_eval(ref, 1, [CAL_L, CAL_R, anchor, obs, T_rl, lm], uv, auv, sigma)
        o_r, o_Ja, o_Jo, o_Jl, o_chi2, o_dp = oracle.ba_residual(1, CAL_L, CAL_R, T_rl, anchor, obs, lam, auv, uv, sigma)
        assert _close(o_r, r) and _close(o_chi2, chi2, max(1.0, chi2)) and o_dp == dp
        assert _close(o_Ja, J[2][:, :6]) and _close(o_Jo, J[3][:, :6]) and _close(o_Jl, J[5][:, 0])
        assert np.all(J[4] == 0) and np.all(J[0] == 0)                                               # extrinsic / left calib: zero
        # ---- RIGHT_ANCH: ReprojectionErrorRightAnchCamKSE3AnchInvDepth {calib_l, calib_r, T_rl, lambda}   (:476-577)
        r, J, chi2, dp = ref_eval(ref, 2, [CAL_L, CAL_R, T_rl, lm], uv, auv, sigma)
        o_r, o_Ja, o_Jo, o_Jl, o_chi2, o_dp = oracle.ba_residual(2, CAL_L, CAL_R, T_rl, anchor, obs, lam, auv, uv, sigma)
        assert _close(o_r, r) and _close(o_chi2, chi2, max(1.0, chi2)) and o_dp == dp and _close(o_Jl, J[3][:, 0])
        assert np.all(o_Ja == 0) and np.all(J[2] == 0)                                               # no pose enters this factor
    assert n_neg > 5


def test_xyz_and_pnp_factors_match_the_reference_source(ref, oracle):
    rng = np.random.default_rng(12)
    for _ in range(300):
        _, pose, T_rl, _, _, uv, sigma = _case(rng)
        X = rng.normal(0, 3.0, 3)
        # ---- ReprojectionErrorKSE3XYZ {calib, pose, X}   (:107-195)
        r, J, chi2, dp = ref_eval(ref, 4, [CAL_L, pose, X], uv, sigma=sigma)
        o_r, o_Jp, o_Jx, o_chi2, o_dp = oracle.xyzba_residual(0, CAL_L, CAL_R, T_rl, pose, X, uv, sigma)
        assert _close(o_r, r) and _close(o_chi2, chi2, max(1.0, chi2)) and o_dp == dp
        assert _close(o_Jp, J[1][:, :6]) and _close(o_Jx, J[2]) and np.all(J[1][:, 6] == 0)
        s_r, s_J, s_chi2, s_dp = oracle.xyz_residual(0, CAL_L, CAL_R, T_rl, pose, X, uv, sigma)     # structureOnlyBA's restatement of the same factor
        assert _close(s_r, r) and _close(s_J, J[2]) and s_dp == dp
        # ---- ReprojectionErrorSE3 {pose}, K and the world point in the constructor (ceresPnP)   (:301-358): the same projection
        rp, Jp, chi2p, dpp = ref_eval(ref, 3, [pose], uv, sigma=sigma, K=CAL_L, xyz=X)
        assert _close(o_r, rp) and _close(o_Jp, Jp[0][:, :6]) and dpp == o_dp and np.all(Jp[0][:, 6] == 0)
        # ---- ReprojectionErrorRightCamKSE3XYZ {calib_r, pose, T_rl, X}   (:198-298)
        r, J, chi2, dp = ref_eval(ref, 5, [CAL_R, pose, T_rl, X], uv, sigma=sigma)
        o_r, o_Jp, o_Jx, o_chi2, o_dp = oracle.xyzba_residual(1, CAL_L, CAL_R, T_rl, pose, X, uv, sigma)
        assert _close(o_r, r) and _close(o_chi2, chi2, max(1.0, chi2)) and o_dp == dp
        assert _close(o_Jp, J[1][:, :6]) and _close(o_Jx, J[3]) and np.all(J[2] == 0)
        s_r, s_J, s_chi2, s_dp = oracle.xyz_residual(1, CAL_L, CAL_R, T_rl, pose, X, uv, sigma)
        assert _close(s_r, r) and _close(s_J, J[3]) and s_dp == dp


def test_se3_left_parameterization_matches_the_reference_source(ref, oracle):
    rng = np.random.default_rng(13)
    gs, ls = C.c_int(0), C.c_int(0)
    ref.ref_se3_sizes(C.byref(gs), C.byref(ls))
    assert (gs.value, ls.value) == (7, 6)
    for k in range(300):
        x = _rand_pose(rng)
        d = rng.normal(0, [0.05, 0.3, 1e-6, 1e-12][k % 4], 6)          # down to Sophus' small-angle branch (theta^2 < 1e-20)
        if k % 7 == 0:
            d[3:] = 0.0
        out = np.zeros(7)
        assert ref.ref_se3_plus(_p(x), _p(d), _p(out)) == 0
        o = oracle.se3_left_plus(x, d)
        assert _close(o, out), (k, o, out)
        J = np.full(42, np.nan)
        assert ref.ref_se3_plus_jacobian(_p(x), _p(J)) == 0
        assert np.array_equal(J.reshape(7, 6), np.vstack([np.eye(6), np.zeros((1, 6))]))             # Jacobian of the local map: [I6; 0]
    # the tangent vectors the vendored Sophus tests hold (Thirdparty/Sophus/test/core/test_so3.cpp:53-60: zero, unit axes, pi / 2 pairs,
    # many turns), as rotation parts of the update, on the group elements of :32-51 (near-pi, tiny-angle and exact-pi rotations)
    pi = np.pi
    tangents = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (pi / 2, pi / 2, 0), (-1, 1, 0), (20, -1, 0), (30, 5, -1)]
    quats = [(0., 1., 0., 0.1e-11), (0.00001, 0., 0., -1.), (0., 0., 0., 1.)]       # (x, y, z, w) of test_so3.cpp:32-35 + identity
    for om in [(0.2, 0.5, 0.0), (0.2, 0.5, -1.0), (0., 0., 0.00001), (pi, 0, 0)]:
        th = np.linalg.norm(om)
        quats.append(tuple(np.sin(th / 2) * np.array(om) / th) + (np.cos(th / 2),))
    for q in quats:
        for w in tangents:
            for v in ((0., 0., 0.), (1., -3., 0.5)):
                x = np.array([1., 2., 4.] + list(q)); d = np.array(list(v) + list(w), np.float64)
                out = np.zeros(7)
                assert ref.ref_se3_plus(_p(x), _p(d), _p(out)) == 0
                assert _close(oracle.se3_left_plus(x, d), out), (q, w, v)
    # the parameter-block holders: [tx ty tz qx qy qz qw], inverse depth = 1 / depth
    x = _rand_pose(rng)
    vals, back = np.zeros(7), np.zeros(7)
    ref.ref_pose_block_roundtrip(_p(x), _p(vals), _p(back))
    assert _close(vals, x) and _close(back, x)
    assert ref.ref_invdepth_block(4.0) == 0.25


def test_reference_jacobians_are_the_derivatives_of_the_reference_residuals(ref):
    """independent of the oracle: central differences of the reference's own residual through the reference's own Plus"""
    rng = np.random.default_rng(14)
    h = 1e-6
    for _ in range(40):
        anchor, obs, T_rl, lam, auv, uv, sigma = _case(rng)
        lam = abs(lam)
        blocks = [CAL_L, CAL_R, anchor, obs, T_rl, np.array([lam])]
        r0, J, _, _ = ref_eval(ref, 1, blocks, uv, auv, sigma)
        if np.abs(r0).max() > 1e5:
            continue
        for bi in (2, 3):                                                      # the two pose blocks
            num = np.zeros((2, 6))
            for k in range(6):
                d = np.zeros(6); d[k] = h
                xp, xm = np.zeros(7), np.zeros(7)
                ref.ref_se3_plus(_p(np.ascontiguousarray(blocks[bi])), _p(d), _p(xp)); ref.ref_se3_plus(_p(np.ascontiguousarray(blocks[bi])), _p(-d), _p(xm))
                bp = list(blocks); bp[bi] = xp; bm = list(blocks); bm[bi] = xm
                num[:, k] = (ref_eval(ref, 1, bp, uv, auv, sigma, want_jac=False)[0] - ref_eval(ref, 1, bm, uv, auv, sigma, want_jac=False)[0]) / (2 * h)
            assert np.abs(num - J[bi][:, :6]).max() <= 2e-4 * max(1.0, np.abs(num).max()), (bi, num, J[bi])
        bp = list(blocks); bp[5] = np.array([lam * (1 + h)]); bm = list(blocks); bm[5] = np.array([lam * (1 - h)])
        num = (ref_eval(ref, 1, bp, uv, auv, sigma, want_jac=False)[0] - ref_eval(ref, 1, bm, uv, auv, sigma, want_jac=False)[0]) / (2 * h * lam)
        assert np.abs(num - J[5][:, 0]).max() <= 2e-4 * max(1.0, np.abs(num).max())


def test_sampson_distance_matches_the_reference_source(ref, oracle):
    """MultiViewGeometry::computeSampsonDistance (src/multi_view_geometry.cpp:798-821, the epipolar gate of MapManager::stereoMatching,
    src/map_manager.cpp:595): the reference's own function text, cut out of its file at build time (oracle/ref/Makefile) and compiled against
    the stand-in Eigen, against oracle/stereo.c -- BIT FOR BIT: the function mixes double products with float storage (num, x1, x2, y1, y2,
    den are floats), and the threshold test `<= 2` sits on its float result."""
    ref.ref_sampson_distance.restype = C.c_float
    ref.ref_sampson_distance.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float]
    rng = np.random.default_rng(11)
    K = np.array([[458.654, 0, 367.215], [0, 457.296, 248.375], [0, 0, 1.0]])
    n_close = 0
    for it in range(4000):
        if it % 2 == 0:                                       # a stereo rig's fundamental matrix (frame.cpp:62), small rotation, 11 cm baseline
            w = rng.normal(0, 0.01, 3); th = np.linalg.norm(w); k = w / th
            Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
            t = np.array([-0.11, 0, 0]) + rng.normal(0, 0.002, 3)
            tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
            F = np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)
        else:
            F = rng.normal(0, 1.0, (3, 3)) * 10.0 ** rng.integers(-6, 2)
        l = rng.uniform([0, 0], [752, 480]).astype(np.float32)
        r = (l + rng.normal(0, [20, 1.5])).astype(np.float32)
        Fc = np.ascontiguousarray(F, np.float64)
        a = np.float32(ref.ref_sampson_distance(_p(Fc), l[0], l[1], r[0], r[1]))
        b = np.float32(oracle.sampson_distance(Fc, l, r))
        assert a.view(np.uint32) == b.view(np.uint32) or (np.isnan(a) and np.isnan(b)), (it, a, b)
        n_close += abs(float(a) - 2.0) < 0.5
    assert n_close > 100                                      # the gate's threshold region is exercised


REF_FT_SO = os.path.join(ROOT, "oracle", "_ref", "libref_frontend.so")


@pytest.fixture(scope="module")
def ref_ft(ref, oracle):
    if not os.path.exists(REF_FT_SO):
        pytest.skip("oracle/_ref/libref_frontend.so is absent and /root/reference is not here to build it from")
    oracle.lib()                                              # (its three OpenCV stand-ins resolve into liboracle.so)
    lib = C.CDLL(REF_FT_SO)
    lib.ref_fb_klt.restype = C.c_int
    lib.ref_fb_klt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.ref_line_min_sad.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ref_in_border.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int]
    lib.ref_detect_singlescale.restype = C.c_int
    lib.ref_detect_singlescale.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.ref_detect_grid_fast.restype = C.c_int
    lib.ref_detect_grid_fast.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    return lib


def test_fb_klt_tracking_control_flow_matches_the_reference_source(ref_ft, oracle):
    """FeatureTracker::fbKltTracking (src/feature_tracker.cpp:35-137) AS THE REFERENCE WROTE IT -- the file is compiled from where it lies
    against a stand-in OpenCV whose calcOpticalFlowPyrLK is the oracle's (oracle/ref/standin_cv) -- against the oracle's restatement of the
    whole function (orc_fb_klt): the level clamp for short pyramids, the status / error / inBorder filters after the forward pass, the
    level-0 backward pass from the forward result, the forward-backward distance in double.  Status and float bits must be equal.
    (What this does NOT pin is OpenCV's LK arithmetic itself: both sides run the oracle's.)"""
    from ov2slam_amd import synth
    rng = np.random.default_rng(5)
    for case in range(12):
        w, h = [(752, 480), (640, 400), (321, 243)][case % 3]
        prev, cur, flow = synth.frame_pair(w, h, seed=700 + case)
        levels = [3, 3, 1, 0][case % 4]                       # pyramids of 4 / 4 / 2 / 1 levels
        P, Cc = oracle.Pyramid(prev, 9, levels), oracle.Pyramid(cur, 9, levels)
        n = 300
        kps = np.stack([rng.uniform(-3, w + 3, n), rng.uniform(-3, h + 3, n)], 1).astype(np.float32)      # some outside the image
        gt = flow(np.clip(kps, 0, [w - 1, h - 1]))
        pri = (gt + rng.normal(0, [0.5, 1.5, 6.0][case % 3], gt.shape)).astype(np.float32)
        pri[::7] = kps[::7]                                   # no prior: starts from the keypoint
        for nbpyrlvl in (0, 1, 3, 5):                         # 5: more than any pyramid here holds (:50-52)
            for ferr, fbdist in ((30.0, 0.5), (8.0, 0.25)):
                o_xy, o_st, _ = oracle.fb_klt(P, Cc, 9, nbpyrlvl, ferr, fbdist, kps, pri)
                r_xy = pri.copy(); r_st = np.zeros(n, np.uint8)
                rc = ref_ft.ref_fb_klt(C.addressof(P.p), C.addressof(Cc.p), P.levels, 9, nbpyrlvl, 30, 0.01, ferr, fbdist, _p(kps), _p(r_xy), n, _p(r_st))
                assert rc == 0
                assert np.array_equal(r_st.astype(bool), o_st), (case, nbpyrlvl, int((r_st.astype(bool) != o_st).sum()))
                assert np.array_equal(r_xy.view(np.uint32), o_xy.view(np.uint32)), (case, nbpyrlvl)
                assert 0 < o_st.sum() < n
    # inBorder (:216-221) on the edges
    for x, y, w, h in ((0.999, 5, 100, 50), (1.0, 1.0, 100, 50), (98.999, 48.999, 100, 50), (99.0, 10, 100, 50), (50, 49.0, 100, 50), (-1, -1, 100, 50)):
        assert bool(ref_ft.ref_in_border(x, y, w, h)) == (1.0 <= x < w - 1.0 and 1.0 <= y < h - 1.0)


def test_line_min_sad_matches_the_reference_source(ref_ft, oracle):
    """FeatureTracker::getLineMinSAD (src/feature_tracker.cpp:138-206) as the reference wrote it (cv::getRectSubPix / cv::norm = the oracle's
    restatements) against orc_line_min_sad: the window shrinking near the borders (`int += float`), the scan bounds in both directions, the
    first minimum winning ties, the even-window and the degenerate-window early returns."""
    from ov2slam_amd import synth
    rng = np.random.default_rng(9)
    w, h = 376, 240
    left, _, _ = synth.frame_pair(w, h, seed=41)
    right = np.roll(left, -13, axis=1).copy()
    right[:, -13:] = 7
    pts = np.concatenate([np.stack([rng.uniform(0, w, 150), rng.uniform(0, h, 150)], 1),
                          [[0.2, 0.3], [1.9, 100.0], [w - 1.0, h - 1.0], [w - 2.5, 3.0], [3.0, h - 0.6], [w / 2, 0.0], [5.0, 5.0]]]).astype(np.float32)
    for nwin in (7, 9, 15, 8, 1, 3):
        for go_left in (True, False):
            ox, oe = oracle.line_min_sad(left, right, pts, nwin, go_left)
            for i, (x, y) in enumerate(pts):
                xp = np.zeros(1, np.float32); er = np.zeros(1, np.float32)
                ref_ft.ref_line_min_sad(_p(left), w, _p(right), w, w, h, float(x), float(y), nwin, int(go_left), _p(xp), _p(er))
                assert xp.view(np.uint32)[0] == np.float32(ox[i]).view(np.uint32), (nwin, go_left, i, xp[0], ox[i])
                assert er.view(np.uint32)[0] == np.float32(oe[i]).view(np.uint32), (nwin, go_left, i, er[0], oe[i])


def _detector_inputs(case):
    from ov2slam_amd import synth
    rng = np.random.default_rng(100 + case)
    w, h = [(752, 480), (640, 400), (321, 243), (1241, 376)][case % 4]
    img, _, _ = synth.frame_pair(w, h, seed=900 + case)
    if case % 3 == 1:
        img = (img.astype(np.float32) * 0.35 + 90).astype(np.uint8)                 # low contrast: thresholds adapt downwards
    if case % 5 == 2:
        img[: h // 3] = 120                                                          # a flat band: empty cells
    cell = [35, 45, 50, 35, 25][case % 5]
    ncur = [0, 40, 150, 5][case % 4]
    cur = np.stack([rng.uniform(0, w - 1, ncur), rng.uniform(0, h - 1, ncur)], 1).astype(np.float32).reshape(-1, 2)
    return img, w, h, cell, cur


def test_detect_singlescale_matches_the_reference_source(ref_ft, oracle):
    """FeatureExtractor::detectSingleScale (src/feature_extractor.cpp:288-440) AS THE REFERENCE WROTE IT -- the file is compiled from where it
    lies against the stand-in OpenCV (GaussianBlur + cornerMinEigenVal, circle, minMaxLoc, cornerSubPix = the oracle's restatements; cv::Mat
    with OpenCV's sharing semantics) -- against oracle/detect.c: orc_detect_singlescale.  Pins the first-party code: the cell walk and the
    occupancy table (`px.y / ncellsize` on floats), the in-image test, the float mask and its two masked arg-max passes, the roi test's
    `continue` (which also drops the cell's second detection), the top-up from the secondary detections (`back()`, the count), the quality
    adaptation.  Keypoint lists (float bits, order) and the adapted quality must be equal."""
    for case in range(16):
        img, w, h, cell, cur = _detector_inputs(case)
        roi = [(0, 0, w, h), (5, 5, w - 10, h - 10), (w // 4, h // 4, w // 2, h // 2)][case % 3]
        for q0 in (0.001, 0.02, 1e-5):
            o_pts, o_q = oracle.detect_singlescale(img, cell, cur, roi, q0)
            out = np.zeros((4 * (w // cell) * (h // cell) + 8, 2), np.float32); q = C.c_double(q0)
            roi_a = (C.c_int * 4)(*roi)
            n = ref_ft.ref_detect_singlescale(_p(img), w, h, w, cell, _p(cur) if len(cur) else None, len(cur), roi_a, C.byref(q), _p(out), len(out))
            assert n == len(o_pts), (case, q0, n, len(o_pts))
            assert np.array_equal(out[:n].view(np.uint32), o_pts.view(np.uint32)), (case, q0)
            assert q.value == o_q, (case, q0, q.value, o_q)


def test_detect_grid_fast_matches_the_reference_source(ref_ft, oracle):
    """FeatureExtractor::detectGridFAST (src/feature_extractor.cpp:443-570) as the reference wrote it against orc_detect_grid_fast in the
    AS-EXECUTED mask mode: the reference hands FAST a CV_32F mask, OpenCV's keypoint filter reads it byte-wise
    (`mask.at<uchar>`) -- the stand-in restates that filter on a real float buffer, so the aliasing is executed here, not assumed: a
    keypoint at ROI column x survives iff x mod 4 >= 2 and the float at column x / 4 is 1.0.  Also pinned: occupancy, the in-image test,
    `std::sort` by response and the `>= 20` gate on the best one, the circle written into the mask for the following cells, both
    threshold adaptations (`int *= double`).
    The reference's std::sort is not stable: with more than 16 keypoints left in a cell the winner among EQUAL best responses is the
    standard library's choice.  With libstdc++'s introsort restated (ORC_FAST_TIE_LIBSTDCXX: the oracle's and the HIP kernels' default since
    this was found) the oracle equals the reference's own code, compiled here with g++, on EVERY cell; first-in-scan-order
    (ORC_FAST_TIE_SCAN_ORDER, the default until then) differs exactly where such a tie exists -- counted below."""
    n_tie_cells, n_cells = {7: 0, 20: 0, 45: 0}, {7: 0, 20: 0, 45: 0}
    for case in range(16):
        img, w, h, cell, cur = _detector_inputs(case)
        for th0 in (20, 7, 45):
            out = np.zeros(((w // cell) * (h // cell) + 8, 2), np.float32); th = C.c_int(th0)
            n = ref_ft.ref_detect_grid_fast(_p(img), w, h, w, cell, _p(cur) if len(cur) else None, len(cur), C.byref(th), _p(out), len(out))
            with oracle.fast_tie_mode(oracle.FAST_TIE_LIBSTDCXX):
                s_pts, s_th = oracle.detect_grid_fast(img, cell, cur, th0)
            assert th.value == s_th and n == len(s_pts), (case, th0, th.value, s_th, n, len(s_pts))
            assert np.array_equal(out[:n].view(np.uint32), s_pts.view(np.uint32)), (case, th0)
            # scan-order mode: same count and threshold; points differ only where an equal-score tie exists in a cell with > 16 corners
            with oracle.fast_tie_mode(oracle.FAST_TIE_SCAN_ORDER):
                o_pts, o_th = oracle.detect_grid_fast(img, cell, cur, th0)
            assert o_th == s_th and len(o_pts) == n
            n_cells[th0] += n
            if not np.array_equal(o_pts.view(np.uint32), s_pts.view(np.uint32)):
                with oracle.fast_tie_mode(oracle.FAST_TIE_SCAN_ORDER):
                    o_raw, _ = oracle.detect_grid_fast(img, cell, cur, th0, subpix=False)
                for b in np.nonzero((s_pts != o_pts).any(1))[0]:
                    x0, y0 = int(o_raw[b, 0]) // cell * cell, int(o_raw[b, 1]) // cell * cell
                    xs, ys, sc = oracle.fast9_16(np.ascontiguousarray(img[y0:y0 + cell, x0:x0 + cell]), th0)
                    best = sc[(xs + x0 == int(o_raw[b, 0])) & (ys + y0 == int(o_raw[b, 1]))][0]
                    assert len(xs) > 16 and int((sc == best).sum()) >= 2, (case, th0, b)
                    n_tie_cells[th0] += 1
    assert oracle.fast_tie_sort_fallbacks() == 0
    # dense synthetic texture: ~2 % of the cells at threshold 7, ~0.4 % at the shipped 20, none at 45
    assert n_tie_cells[45] == 0 and n_tie_cells[20] <= 0.01 * n_cells[20] and n_tie_cells[7] <= 0.04 * n_cells[7], (n_tie_cells, n_cells)


@pytest.mark.gpu
def test_device_residuals_match_the_reference_source(ref, gpu_ctx):
    """The DEVICE's evaluation at the initial point (ov2_ba_solve with a zero iteration budget returns the chi2err_ / isdepthpositive_
    of every residual block) against the reference's Evaluate on the same blocks -- no oracle in between."""
    from ov2slam_amd import optimizer, synth
    pb = synth.make_ba_problem(12, 500, 8, stereo=True, seed=21)
    o = optimizer.default_options(gpu_ctx.lib)
    o.max_iter = 0
    g = optimizer.solve(gpu_ctx, pb, o)
    poses = np.asarray(pb["poses"], np.float64).reshape(-1, 7)
    n = 0
    for i in range(0, int(pb["n_res"]), 3):
        t, kf, lm = int(pb["res_type"][i]), int(pb["res_kf"][i]), int(pb["res_lm"][i])
        anchor = poses[int(pb["lm_anchor_kf"][lm])]; lam = np.array([float(pb["invdepth"][lm])])
        auv = np.asarray(pb["lm_anchor_uv"], np.float64).reshape(-1, 2)[lm]; uv = np.asarray(pb["res_uv"], np.float64).reshape(-1, 2)[i]
        sig = float(pb["res_sigma"][i]); cl, cr, T = np.asarray(pb["calib_l"], np.float64), np.asarray(pb["calib_r"], np.float64), np.asarray(pb["T_rl"], np.float64)
        blocks = {0: [cl, anchor, poses[kf], lam], 1: [cl, cr, anchor, poses[kf], T, lam], 2: [cl, cr, T, lam]}[t]
        _, _, chi2, dp = ref_eval(ref, t, blocks, uv, auv, sig, want_jac=False)
        assert abs(g["chi2"][i] - chi2) <= 1e-10 * max(1.0, chi2), (i, t, g["chi2"][i], chi2)
        assert bool(g["depthpos"][i]) == dp
        n += 1
    assert n > 2000
