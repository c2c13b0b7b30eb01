"""GPU parity of the single-sequence tracker (ov2_tracker_*: preprocessImage + kltTracking in one enqueue, both
fbKltTracking calls and the retry of lost prior tracks in ONE LK launch) against the oracle's restatement of
VisualFrontEnd::preprocessImage / kltTracking (/root/reference/src/visual_front_end.cpp:1143-1177, :132-275).
Bar: bit-exact (positions compared as raw float32 bits, status and retry flags equal)."""
import numpy as np
import pytest

import ov2slam_amd
from ov2slam_amd import synth

pytestmark = pytest.mark.gpu

CLIP = 3.0


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _sequence(w, h, nframes, seed):
    rng = np.random.default_rng(seed)
    tex = synth.base_texture(max(w, h) + 500, seed)
    views, offs = [], []
    ox, oy, th = 120.0, 110.0, 0.0
    for _ in range(nframes):
        views.append(synth.warp(tex, w, h, ox, oy, th)); offs.append((ox, oy, th))
        ox += rng.uniform(-5, 5); oy += rng.uniform(-4, 4); th += rng.uniform(-0.006, 0.006)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0

    def flow(pts, a, b):
        (ox0, oy0, t0), (ox1, oy1, t1) = offs[a], offs[b]
        dx, dy = pts[:, 0] - cx, pts[:, 1] - cy
        c, s = np.cos(t0), np.sin(t0)
        tx, ty = c * dx - s * dy + cx + ox0, s * dx + c * dy + cy + oy0
        dx, dy = tx - cx - ox1, ty - cy - oy1
        c, s = np.cos(-t1), np.sin(-t1)
        return np.stack([c * dx - s * dy + cx, s * dx + c * dy + cy], 1)
    return views, flow


def _oracle_frame(O, prev_img, cur_img, kps, pri, hp, use_clahe, w, h, use_prior=True):
    if use_clahe:
        prev_img = O.clahe(prev_img, CLIP, w // 50, h // 50)
        cur_img = O.clahe(cur_img, CLIP, w // 50, h // 50)
    return O.klt_tracking(O.Pyramid(prev_img, 9, 3), O.Pyramid(cur_img, 9, 3), kps, pri, hp, klt_use_prior=use_prior)


def _points(w, h, flow, f, rng, prior_sigma, frac_prior=0.7, bad_frac=0.0, bad_sigma=12.0):
    kps = synth.grid_keypoints(w, h, 35, rng)
    gt = flow(kps.astype(np.float64), f, f + 1)
    hp = rng.uniform(size=len(kps)) < frac_prior
    pri = np.where(hp[:, None], gt + rng.normal(0, prior_sigma, gt.shape), kps).astype(np.float32)
    if bad_frac > 0:                       # some priors far off: lost on 2 levels, recovered (or not) on the full pyramid
        bad = hp & (rng.uniform(size=len(kps)) < bad_frac)
        pri[bad] += rng.normal(0, bad_sigma, (int(bad.sum()), 2)).astype(np.float32)
    return kps, pri, hp.astype(np.uint8)


@pytest.mark.parametrize("impl", ["wave", "row"])
@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("wh,use_clahe", [((752, 480), True), ((1241, 376), True), ((376, 240), False)])
def test_track_frame_matches_oracle(gpu_ctx, oracle, wh, use_clahe, use_graph, impl):
    # impl: the wavefront-per-keypoint LK kernel (k_track_klt_w, the default for the 9 x 9 window) and the row-per-lane one
    # (k_track_klt, the kernel of every other window size); OV2_OPT_TRACK_IMPL is read when the launch is built / captured
    from ov2slam_amd import _lib as L
    with gpu_ctx.options(track_impl=L.OV2_TRACK_IMPL_ROW if impl == "row" else L.OV2_TRACK_IMPL_WAVE):
        _track_frame_case(gpu_ctx, oracle, wh, use_clahe, use_graph)


def _track_frame_case(gpu_ctx, oracle, wh, use_clahe, use_graph, gain=1.0):
    w, h = wh
    views, flow = _sequence(w, h, 5, seed=31)
    if gain != 1.0:
        views = [np.clip((v.astype(np.float32) - 128.0) * gain + 128.0, 0, 255).astype(np.uint8) for v in views]
    rng = np.random.default_rng(5)
    trk = ov2slam_amd.VisualFrontEndTracker(gpu_ctx, w, h, use_clahe=use_clahe, fclahe_val=CLIP, nbmaxkps=512, use_graph=use_graph)
    out, st, p3p = trk.trackFrame(views[0], np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), None)
    assert len(out) == 0
    n_retried = 0
    for f in range(4):
        kps, pri, hp = _points(w, h, flow, f, rng, 1.0, bad_frac=0.25)
        gout, gst, gp3p = trk.trackFrame(views[f + 1], kps, pri, hp)
        rout, rok, rretried, rp3p = _oracle_frame(oracle, views[f], views[f + 1], kps, pri, hp, use_clahe, w, h)
        assert gp3p == rp3p
        assert np.array_equal((gst & 1).astype(bool), rok), "frame %d status" % f
        assert np.array_equal((gst & 2).astype(bool), rretried), "frame %d retry flags" % f
        assert np.array_equal(_bits(gout), _bits(rout)), "frame %d positions" % f
        n_retried += int(rretried.sum())
        assert rok.mean() > 0.6
    assert n_retried > 0                                   # the retry path was exercised
    if use_graph:
        assert trk.uses_graph
    # the tracker's pyramids are the reference's prev_pyr_ / cur_pyr_
    ref = oracle.clahe(views[4], CLIP, w // 50, h // 50) if use_clahe else views[4]
    gi, _ = trk.cur_pyr.download(0)
    assert np.array_equal(gi, ref)
    trk.close()


def test_p3p_rule_and_no_prior_mode(gpu_ctx, oracle):
    """Priors so wrong that fewer than a third survive the 2-level pass: bp3preq_ is raised and the lost tracks restart
    from their keypoints (visual_front_end.cpp:225-230).  klt_use_prior = 0 sends everything through the full pyramid."""
    w, h = 752, 480
    views, flow = _sequence(w, h, 3, seed=77)
    rng = np.random.default_rng(9)
    trk = ov2slam_amd.VisualFrontEndTracker(gpu_ctx, w, h, use_clahe=True, fclahe_val=CLIP, nbmaxkps=400, use_graph=True)
    trk.trackFrame(views[0], np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), None)
    kps, pri, hp = _points(w, h, flow, 0, rng, 1.0, frac_prior=0.8, bad_frac=0.85, bad_sigma=40.0)
    gout, gst, gp3p = trk.trackFrame(views[1], kps, pri, hp)
    rout, rok, rretried, rp3p = _oracle_frame(oracle, views[0], views[1], kps, pri, hp, True, w, h)
    assert rp3p and gp3p
    assert np.array_equal((gst & 1).astype(bool), rok) and np.array_equal((gst & 2).astype(bool), rretried)
    assert np.array_equal(_bits(gout), _bits(rout))
    # split API + klt_use_prior = 0
    kps, pri, hp = _points(w, h, flow, 1, rng, 1.0)
    trk.preprocessImage(views[2])
    gout, gst, gp3p = trk.kltTracking(kps, kps, hp, klt_use_prior=False)
    rout, rok, rretried, rp3p = _oracle_frame(oracle, views[1], views[2], kps, kps, hp, True, w, h, use_prior=False)
    assert not gp3p and not rretried.any() and not (gst & 2).any()
    assert np.array_equal((gst & 1).astype(bool), rok) and np.array_equal(_bits(gout), _bits(rout))
    trk.close()


def test_pinned_image_buffer_and_strided_input(gpu_ctx, oracle):
    w, h = 1241, 376                                      # pitch of the pinned buffer (1248) differs from the width
    views, flow = _sequence(w, h, 3, seed=5)
    rng = np.random.default_rng(2)
    trk = ov2slam_amd.VisualFrontEndTracker(gpu_ctx, w, h, use_clahe=True, fclahe_val=CLIP, nbmaxkps=512, use_graph=True)
    trk.image_buffer[:, :w] = views[0]
    trk.trackFrame(trk.image_buffer, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), None)
    for f in range(2):
        kps, pri, hp = _points(w, h, flow, f, rng, 1.5)
        if f == 0:
            trk.image_buffer[:, :w] = views[f + 1]
            img = trk.image_buffer                        # zero-copy path
        else:
            img = np.zeros((h, w + 37), np.uint8); img[:, :w] = views[f + 1]   # caller stride != width
        gout, gst, _ = trk.trackFrame(img, kps, pri, hp)
        rout, rok, rretried, _ = _oracle_frame(oracle, views[f], views[f + 1], kps, pri, hp, True, w, h)
        assert np.array_equal((gst & 1).astype(bool), rok) and np.array_equal(_bits(gout), _bits(rout))
    trk.close()


def test_cross_context_pyramid_handoff(oracle):
    """A pyramid built on one context's stream and consumed on another's (front-end -> mapper thread hand-off,
    INTEGRATION.md 3): the consumer entry points wait on the producer's `ready` event."""
    a, b = ov2slam_amd.Context(0), ov2slam_amd.Context(0)
    prev, cur, flow = synth.frame_pair(752, 480, seed=8)
    rng = np.random.default_rng(1)
    kps = synth.grid_keypoints(752, 480, 35, rng)
    pri = (flow(kps) + rng.normal(0, 1.0, kps.shape)).astype(np.float32)
    Rp, Rc = oracle.Pyramid(prev, 9, 3), oracle.Pyramid(cur, 9, 3)
    rout, rst, _ = oracle.fb_klt(Rp, Rc, 9, 3, 30., 0.5, kps, pri)
    trk = ov2slam_amd.FeatureTracker(b, 30, 0.01)
    for _ in range(10):                                   # build on `a` (asynchronous), track on `b` straight away
        Gp = ov2slam_amd.Pyramid(a, 752, 480, 9, 3).build(prev)
        Gc = ov2slam_amd.Pyramid(a, 752, 480, 9, 3).build(cur)
        gout, gst = trk.fbKltTracking(Gp, Gc, 9, 3, 30., 0.5, kps, pri)
        assert np.array_equal(gst, rst) and np.array_equal(_bits(gout), _bits(rout))
        a.sync(); Gp.close(); Gc.close()
    a.close(); b.close()


@pytest.mark.parametrize("use_graph", [False, True])
def test_more_keypoints_than_capacity(gpu_ctx, oracle, use_graph):
    """A frame may carry more keypoints than nbmaxkps_ (pruning happens at the next keyframe only,
    /root/reference/src/map_manager.cpp:74): the tracker must track them all, not report "nothing tracked".  Capacity 100,
    273 keypoints -> three launches; identical to the oracle, cross-keypoint p3p rule evaluated over ALL keypoints."""
    w, h = 752, 480
    views, flow = _sequence(w, h, 3, seed=13)
    rng = np.random.default_rng(4)
    trk = ov2slam_amd.VisualFrontEndTracker(gpu_ctx, w, h, use_clahe=True, fclahe_val=CLIP, nbmaxkps=100, use_graph=use_graph)
    trk.trackFrame(views[0], np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), None)
    kps, pri, hp = _points(w, h, flow, 0, rng, 1.0, bad_frac=0.25)
    assert len(kps) > 2 * 100
    gout, gst, gp3p = trk.trackFrame(views[1], kps, pri, hp)
    rout, rok, rretried, rp3p = _oracle_frame(oracle, views[0], views[1], kps, pri, hp, True, w, h)
    assert gp3p == rp3p and np.array_equal((gst & 1).astype(bool), rok) and np.array_equal((gst & 2).astype(bool), rretried)
    assert np.array_equal(_bits(gout), _bits(rout))
    # split API, and the p3p rule with its counts spread over several chunks
    kps, pri, hp = _points(w, h, flow, 1, rng, 1.0, frac_prior=0.8, bad_frac=0.95, bad_sigma=60.0)
    trk.preprocessImage(views[2])
    gout, gst, gp3p = trk.kltTracking(kps, pri, hp)
    rout, rok, rretried, rp3p = _oracle_frame(oracle, views[1], views[2], kps, pri, hp, True, w, h)
    assert rp3p and gp3p
    assert np.array_equal((gst & 1).astype(bool), rok) and np.array_equal((gst & 2).astype(bool), rretried)
    assert np.array_equal(_bits(gout), _bits(rout))
    with pytest.raises(ValueError):
        trk.trackFrame(views[2][:100], kps, pri, hp)                 # wrong image height is refused on the Python side
    with pytest.raises(ValueError):
        trk.kltTracking(kps, pri, hp[:10])
    trk.close()


@pytest.mark.parametrize("use_graph", [False, True])
def test_tracker_computes_keypoints_in_the_same_enqueue(gpu_ctx, oracle, use_graph):
    """ov2_tracker_set_calibration: Frame::computeKeypoint (undistorted pixel + bearing, /root/reference/src/frame.cpp:246-254) of
    every output position inside the per-frame enqueue == ov2_compute_keypoints == the oracle, bit for bit (pinhole model); also
    beyond the launch capacity (chunks) and after the p3p re-run."""
    K = (458.654, 457.296, 367.215, 248.375); D = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)
    w, h = 752, 480
    views, flow = _sequence(w, h, 3, seed=21)
    rng = np.random.default_rng(6)
    for cap in (512, 100):
        trk = ov2slam_amd.VisualFrontEndTracker(gpu_ctx, w, h, use_clahe=True, fclahe_val=CLIP, nbmaxkps=cap, use_graph=use_graph)
        cal = ov2slam_amd.CameraCalibration(gpu_ctx, "pinhole", *K, D=D)
        trk.setCalibration(cal)
        trk.trackFrame(views[0], np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), None)
        kps, pri, hp = _points(w, h, flow, 0, rng, 1.0, bad_frac=0.25)
        out, st, _ = trk.trackFrame(views[1], kps, pri, hp)
        unpx, bv = trk.lastKeypoints(len(out))
        runpx, rbv = oracle.compute_keypoints(oracle.CAM_PINHOLE, K, D, cal.iK, out)
        assert np.array_equal(_bits(unpx), _bits(runpx)) and np.array_equal(bv.view(np.uint64), rbv.view(np.uint64))
        # the p3p rule re-runs lost prior tracks from their keypoints: their keypoints follow the new positions
        kps, pri, hp = _points(w, h, flow, 1, rng, 1.0, frac_prior=0.8, bad_frac=0.95, bad_sigma=60.0)
        out, st, p3p = trk.trackFrame(views[2], kps, pri, hp)
        assert p3p
        unpx, bv = trk.lastKeypoints(len(out))
        runpx, rbv = oracle.compute_keypoints(oracle.CAM_PINHOLE, K, D, cal.iK, out)
        assert np.array_equal(_bits(unpx), _bits(runpx)) and np.array_equal(bv.view(np.uint64), rbv.view(np.uint64))
        # the "LAST call" contract: a frame that tracks fewer keypoints (or none) must not hand out the previous frame's entries
        n_small = 7
        out, st, _ = trk.trackFrame(views[0], kps[:n_small], kps[:n_small], None)
        trk.lastKeypoints(n_small)
        with pytest.raises(ov2slam_amd.Ov2Error):
            trk.lastKeypoints(n_small + 1)
        trk.trackFrame(views[1], np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), None)
        with pytest.raises(ov2slam_amd.Ov2Error):
            trk.lastKeypoints(1)
        trk.close()


@pytest.mark.parametrize("impl", ["wave", "row"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_track_frame_float_accumulators(gpu_ctx, oracle, use_graph, impl):
    """The per-frame tracker with OV2_OPT_LK_ACC = OV2_LK_ACC_FLOAT_UI4 (set before the tracker captures its graphs): both kernels of the
    fused kltTracking launch against the oracle in ORC_LK_ACC_FLOAT_UI4, bit for bit -- the as-executed arithmetic of an x86 OpenCV."""
    from ov2slam_amd import _lib as L
    with gpu_ctx.options(track_impl=L.OV2_TRACK_IMPL_ROW if impl == "row" else L.OV2_TRACK_IMPL_WAVE, lk_acc=L.OV2_LK_ACC_FLOAT_UI4), \
            oracle.lk_acc_mode(oracle.LK_ACC_FLOAT_UI4):
        _track_frame_case(gpu_ctx, oracle, (752, 480), True, use_graph)              # (CLAHE stretches the contrast by itself)
        _track_frame_case(gpu_ctx, oracle, (376, 240), False, use_graph, gain=5.0)
    # the inputs do separate the two modes: the oracle's float and integer accumulators disagree on some of these tracks
    w, h = 376, 240
    views, flow = _sequence(w, h, 2, seed=31)
    views = [np.clip((v.astype(np.float32) - 128.0) * 5.0 + 128.0, 0, 255).astype(np.uint8) for v in views]
    kps, pri, hp = _points(w, h, flow, 0, np.random.default_rng(5), 1.0)
    a = _oracle_frame(oracle, views[0], views[1], kps, pri, hp, False, w, h)[0]
    with oracle.lk_acc_mode(oracle.LK_ACC_FLOAT_UI4):
        b = _oracle_frame(oracle, views[0], views[1], kps, pri, hp, False, w, h)[0]
    assert not np.array_equal(_bits(a), _bits(b))
