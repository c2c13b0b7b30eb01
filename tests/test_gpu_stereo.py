"""GPU parity (through the C ABI) of the stereo-matching front half against the oracle: bit-exact."""
import numpy as np
import pytest

import ov2slam_amd
from ov2slam_amd import synth, stereo

pytestmark = pytest.mark.gpu

K = (458.654, 457.296, 367.215, 248.375)
D4 = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)


def _pair(w, h, disp, seed):
    tex = synth.base_texture(max(w, h) + 400, seed)
    l = tex[50:50 + h, 100:100 + w].copy()
    r = tex[50:50 + h, 100 + disp:100 + disp + w].copy()
    return l, r


@pytest.mark.parametrize("wh", [(752, 480), (1241, 376)])
@pytest.mark.parametrize("go_left", [True, False])
def test_line_min_sad_bit_exact(gpu_ctx, oracle, wh, go_left):
    w, h = wh
    l, r = _pair(w, h, 24, w)
    pl = ov2slam_amd.Pyramid(gpu_ctx, w, h, 9, 3).build(l)
    pr = ov2slam_amd.Pyramid(gpu_ctx, w, h, 9, 3).build(r)
    trk = ov2slam_amd.FeatureTracker(gpu_ctx)
    lvl = 3
    lw, lh = pl.level_size(lvl)
    rng = np.random.default_rng(5)
    pts = np.stack([rng.uniform(0, lw - 1, 400), rng.uniform(0, lh - 1, 400)], 1).astype(np.float32)
    pts = np.concatenate([pts, np.array([[0.0, 0.0], [lw - 1, lh - 1], [2.5, 1.5], [lw - 1.5, lh - 1.2], [lw / 2, 0.4], [3.0, lh / 2]], np.float32)])
    xp, err = trk.getLineMinSAD(pl, pr, lvl, pts, 7, go_left)
    il, ir = pl.download(lvl)[0], pr.download(lvl)[0]
    rxp, rerr = oracle.line_min_sad(il, ir, pts, 7, go_left)
    assert np.array_equal(xp, rxp) and np.array_equal(err, rerr)
    if go_left:
        inner = (pts[:, 0] > 12) & (pts[:, 0] < lw - 8) & (pts[:, 1] > 4) & (pts[:, 1] < lh - 5)
        assert np.mean(np.abs((pts[inner, 0] - xp[inner]) - 3.0) <= 1.0) > 0.8       # disparity 24 px = 3 px at level 3
    # level 0 too, and an even window
    pts0 = np.stack([rng.uniform(0, w - 1, 64), rng.uniform(0, h - 1, 64)], 1).astype(np.float32)
    a = trk.getLineMinSAD(pl, pr, 0, pts0, 5, go_left); b = oracle.line_min_sad(l, r, pts0, 5, go_left)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.all(trk.getLineMinSAD(pl, pr, 0, pts0, 6, go_left)[0] == -1)


def test_epipolar_check_bit_exact(gpu_ctx, oracle):
    rng = np.random.default_rng(7)
    n = 500
    lun = np.stack([rng.uniform(0, 752, n), rng.uniform(0, 480, n)], 1).astype(np.float32)
    rk = (lun + np.stack([-rng.uniform(0, 60, n), rng.normal(0, 1.5, n)], 1)).astype(np.float32)
    F = rng.normal(size=(3, 3)) * 1e-4
    F[1, 2], F[2, 1] = -1e-2, 1e-2
    for model, D in (("pinhole", D4), ("pinhole", None), ("fisheye", (-0.02, 0.004, -0.001, 0.0002))):
        cal = ov2slam_amd.CameraCalibration(gpu_ctx, model, *K, D=D)
        for rect in (True, False):
            out, runpx, err, ok = stereo.epipolar_check(gpu_ctx, rect, F, cal, lun, rk)
            ro, rr, re, rok = oracle.stereo_epipolar_check(rect, F, cal.model, K, D, lun, rk)
            if model == "pinhole":
                assert np.array_equal(out, ro) and np.array_equal(runpx, rr) and np.array_equal(err, re) and np.array_equal(ok, rok)
            else:       # tan(): last-bit differences allowed on the undistorted pixel
                assert np.abs(runpx - rr).max() <= 1.3e-4 and np.mean(ok == rok) > 0.995


def test_stereo_matching_flow(gpu_ctx, oracle):
    """MapManager::stereoMatching data path on a synthetic rectified pair: SAD priors -> two fbKlt passes (3-D-prior failures
    retried from the first call's forward result, map_manager.cpp:533-538) -> gate, against oracle.stereo_matching, which
    restates the reference's lists push_back by push_back."""
    w, h, disp = 752, 480, 20
    l, r = _pair(w, h, disp, 11)
    pl = ov2slam_amd.Pyramid(gpu_ctx, w, h, 9, 3).build(l)
    pr = ov2slam_amd.Pyramid(gpu_ctx, w, h, 9, 3).build(r)
    trk = ov2slam_amd.FeatureTracker(gpu_ctx, 30, 0.01)
    cal = ov2slam_amd.CameraCalibration(gpu_ctx, "pinhole", *K, D=None)
    rng = np.random.default_rng(3)
    kps = synth.grid_keypoints(w, h, 35, rng)[:300]
    pri3d = {i: (kps[i, 0] - disp + rng.normal(0, 1.0), kps[i, 1]) for i in range(0, 80)}
    for i in range(0, 80, 3):                                     # a third of the 3-D priors far off: lost on 2 levels, retried on 4
        pri3d[i] = (pri3d[i][0] + rng.uniform(8, 16), pri3d[i][1] - rng.uniform(5, 10))
    ok, right = stereo.stereo_matching(trk, pl, pr, kps, kps, cal, rect=True, priors3d=pri3d)
    O = oracle
    opl, opr = O.Pyramid(l, 9, 3), O.Pyramid(r, 9, 3)
    rok, rright = O.stereo_matching(opl, opr, kps, kps, O.CAM_PINHOLE, K, None, True, priors3d=pri3d)
    assert np.array_equal(ok, rok) and np.array_equal(right.view(np.uint32), rright.view(np.uint32))
    # the retry really happened and really started from the forward result: with the ORIGINAL prior the outcome differs
    p3 = np.array([pri3d[i] for i in sorted(pri3d)], np.float32)
    o3, s3 = O.fb_klt(opl, opr, 9, 1, 30.0, 0.5, kps[:80], p3, 30, 0.01)[:2]
    lost = ~s3.astype(bool)
    assert lost.sum() >= 10
    o_fw = O.fb_klt(opl, opr, 9, 3, 30.0, 0.5, kps[:80][lost], o3[lost], 30, 0.01)[0]
    o_same = O.fb_klt(opl, opr, 9, 3, 30.0, 0.5, kps[:80][lost], p3[lost], 30, 0.01)[0]
    assert not np.array_equal(o_fw.view(np.uint32), o_same.view(np.uint32))
    assert ok.mean() > 0.9
    assert np.abs((kps[ok, 0] - right[ok, 0]) - disp).max() < 0.5
    # and the one-enqueue form agrees with the oracle directly (not only with the call sequence)
    ok_f, right_f = stereo.stereo_matching_fused(trk, pl, pr, kps, kps, cal, rect=True, priors3d=pri3d)
    assert np.array_equal(ok_f, rok) and np.array_equal(right_f.view(np.uint32), rright.view(np.uint32))


@pytest.mark.parametrize("rect", [True, False])
def test_fused_stereo_match_equals_the_call_sequence(gpu_ctx, oracle, rect):
    """ov2_stereo_match (SAD priors + both fbKltTracking calls + retry + gate in one enqueue, one sync) returns what the
    sequence of separate calls returns -- which test_stereo_matching_flow pins to the oracle -- including the retry of failed
    3-D-prior tracks from the first call's forward result and, for a non-rectified pair, the Sampson gate."""
    w, h, disp = 752, 480, 20
    l, r = _pair(w, h, disp, 23)
    pl = ov2slam_amd.Pyramid(gpu_ctx, w, h, 9, 3).build(l)
    pr = ov2slam_amd.Pyramid(gpu_ctx, w, h, 9, 3).build(r)
    trk = ov2slam_amd.FeatureTracker(gpu_ctx, 30, 0.01)
    cal = ov2slam_amd.CameraCalibration(gpu_ctx, "pinhole", *K, D=None)
    rng = np.random.default_rng(8)
    kps = synth.grid_keypoints(w, h, 35, rng)[:300]
    unpx = kps
    pri3d = {i: (kps[i, 0] - disp + rng.normal(0, 1.0), kps[i, 1] + rng.normal(0, 0.3)) for i in range(0, 90)}
    for i in range(0, 90, 4):                                     # a quarter of the 3-D priors far off: lost on 2 levels, retried on 4
        pri3d[i] = (pri3d[i][0] + 14.0, pri3d[i][1] - 9.0)
    # fundamental matrix of a pure x-translation between identical pinhole cameras: rows of equal y are the epipolar lines
    F = None if rect else np.array([[0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]])
    ok_a, right_a = stereo.stereo_matching(trk, pl, pr, kps, unpx, cal, rect=rect, Frl=F, priors3d=pri3d)
    ok_b, right_b = stereo.stereo_matching_fused(trk, pl, pr, kps, unpx, cal, rect=rect, Frl=F, priors3d=pri3d)
    assert np.array_equal(ok_a, ok_b)
    assert np.array_equal(right_a.view(np.uint32), right_b.view(np.uint32))
    assert ok_b.mean() > 0.85
    # the right pyramid may still be building when the call is made (asynchronous build on the same context)
    pr2 = ov2slam_amd.Pyramid(gpu_ctx, w, h, 9, 3)
    pr2.build(r)
    ok_c, right_c = stereo.stereo_matching_fused(trk, pl, pr2, kps, unpx, cal, rect=rect, Frl=F)
    ok_d, right_d = stereo.stereo_matching(trk, pl, pr, kps, unpx, cal, rect=rect, Frl=F)
    assert np.array_equal(ok_c, ok_d) and np.array_equal(right_c.view(np.uint32), right_d.view(np.uint32))

