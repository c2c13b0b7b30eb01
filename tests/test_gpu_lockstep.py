"""GPU parity of the lock-step tracker (ov2_btracker_*, csrc/trackb.hip: `batch` camera streams advance one frame per call) --
against the single-sequence tracker fed the same frames and keypoints (bit for bit: positions as raw float32 bits, status, retry
flags, p3p request, computeKeypoint outputs, pyramids, keyframe detection, stereo matching on an item view) and against the oracle's
restatement of VisualFrontEnd::preprocessImage / kltTracking (/root/reference/src/visual_front_end.cpp:1143-1177, :132-275)."""
import numpy as np
import pytest

import ov2slam_amd
from ov2slam_amd import synth, stereo
from ov2slam_amd import _lib as L

from tests.test_gpu_tracker import _sequence, _points, _oracle_frame, _bits, CLIP

pytestmark = pytest.mark.gpu

K_EUROC = (458.654, 457.296, 367.215, 248.375)
D_EUROC = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)


def _pack(batch, n_max, per_item):
    """per_item: list of (kps, pri, hp) -> padded (batch, n_max, ...) arrays + counts"""
    kps = np.zeros((batch, n_max, 2), np.float32); pri = np.zeros((batch, n_max, 2), np.float32)
    hp = np.zeros((batch, n_max), np.uint8); n = np.zeros(len(per_item), np.int32)
    for b, (k, p, h) in enumerate(per_item):
        n[b] = len(k); kps[b, :len(k)] = k; pri[b, :len(k)] = p; hp[b, :len(k)] = h
    return kps, pri, hp, n


@pytest.mark.parametrize("impl", ["wave", "row"])
@pytest.mark.parametrize("wh,use_clahe", [((752, 480), True), ((1241, 376), True), ((376, 240), False)])
def test_lockstep_equals_single_trackers(gpu_ctx, oracle, wh, use_clahe, impl):
    w, h = wh
    batch, n_max, nframes = 5, 640, 6
    with gpu_ctx.options(track_impl=L.OV2_TRACK_IMPL_ROW if impl == "row" else L.OV2_TRACK_IMPL_WAVE):
        seqs = [_sequence(w, h, nframes, seed=40 + b) for b in range(batch)]
        length = [6, 6, 5, 4, 3]                                 # longest first: items drop out from the tail
        rngs = [np.random.default_rng(100 + b) for b in range(batch)]
        cal = ov2slam_amd.CameraCalibration(gpu_ctx, "pinhole", *K_EUROC, D=D_EUROC)
        bt = ov2slam_amd.LockstepTracker(gpu_ctx, batch, w, h, use_clahe=use_clahe, fclahe_val=CLIP, nbmaxkps=n_max)
        bt.setCalibration(cal)
        singles = []
        for b in range(batch):
            t = ov2slam_amd.VisualFrontEndTracker(gpu_ctx, w, h, use_clahe=use_clahe, fclahe_val=CLIP, nbmaxkps=n_max, use_graph=False)
            t.setCalibration(cal)
            t.trackFrame(seqs[b][0][0], np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), None)
            singles.append(t)
        z = np.zeros((batch, n_max, 2), np.float32)
        bt.trackFrame([seqs[b][0][0] for b in range(batch)], z, z, None, np.zeros(batch, np.int32))
        checked_oracle = 0
        for f in range(nframes - 1):
            na = sum(1 for b in range(batch) if length[b] > f + 1)
            per = []
            for b in range(na):
                k, p, hp_ = _points(w, h, seqs[b][1], f, rngs[b], 1.0, bad_frac=0.25)
                m = len(k) - 7 * b                               # different counts per item
                per.append((k[:m], p[:m], hp_[:m]))
            if f == 2:
                per[1] = (per[1][0][:0], per[1][1][:0], per[1][2][:0])      # an item with nothing to track this frame
            kps, pri, hp, n = _pack(batch, n_max, per)
            imgs = [seqs[b][0][f + 1] for b in range(na)]
            if f % 2 == 1:                                        # frames written into the pinned slots + look-ahead upload / pre-processing
                which = f % 3
                for b in range(na):
                    bt.image_buffers[which][b][:, :w] = imgs[b]
                bt.upload(which, na)
                if f == 3:
                    bt.prepare(which, na)
                imgs = [bt.image_buffers[which][b] for b in range(na)]
            out, st, p3p = bt.trackFrame(imgs, kps, pri, hp, n)
            for b in range(na):
                k, p, hp_ = per[b]
                so, ss, sp = singles[b].trackFrame(seqs[b][0][f + 1], k, p, hp_)
                m = len(k)
                assert np.array_equal(_bits(out[b, :m]), _bits(so)), "frame %d item %d positions" % (f, b)
                assert np.array_equal(st[b, :m], ss), "frame %d item %d status" % (f, b)
                assert bool(p3p[b]) == sp
                if m:
                    bu, bb = bt.lastKeypoints(b, m)
                    su, sb = singles[b].lastKeypoints(m)
                    assert np.array_equal(_bits(bu), _bits(su)) and np.array_equal(bb.view(np.uint64), sb.view(np.uint64))
                for lvl in range(4):
                    gi, _ = bt.cur_item(b).download(lvl)
                    si, _ = singles[b].cur_pyr.download(lvl)
                    assert np.array_equal(gi, si), "frame %d item %d pyramid level %d" % (f, b, lvl)
                if b == na - 1 and m:                             # and against the oracle itself (one item per frame: CPU time)
                    rout, rok, rretried, rp3p = _oracle_frame(oracle, seqs[b][0][f], seqs[b][0][f + 1], k, p, hp_, use_clahe, w, h)
                    assert np.array_equal(_bits(out[b, :m]), _bits(rout)) and np.array_equal((st[b, :m] & 1).astype(bool), rok)
                    assert np.array_equal((st[b, :m] & 2).astype(bool), rretried) and bool(p3p[b]) == rp3p
                    checked_oracle += 1
        assert checked_oracle >= 3
        # keyframe: detection on the current frames of the still active items, one call
        na = sum(1 for b in range(batch) if length[b] >= nframes)
        roi = (5, 5, w - 10, h - 10)
        cur = np.zeros((batch, n_max, 2), np.float32); ncur = np.zeros(na, np.int32)
        for b in range(na):
            good = out[b, :n[b]][(st[b, :n[b]] & 1) > 0][:20 + 5 * b]
            cur[b, :len(good)] = good; ncur[b] = len(good)
        q = np.array([0.001, 0.004, 0.0005, 0.001, 0.001][:na], np.float64)
        det = bt.detectSingleScale(na, 35, cur, ncur, roi, q)
        th = np.array([10, 20, 7, 10, 10][:na], np.int32)
        detf = bt.detectGridFAST(na, 35, cur, ncur, th)
        for b in range(na):
            fx = ov2slam_amd.FeatureExtractor(gpu_ctx, nfast_th=[10, 20, 7, 10, 10][b], dmaxquality=[0.001, 0.004, 0.0005, 0.001, 0.001][b])
            s1 = fx.detectSingleScalePyr(singles[b].cur_pyr, 35, cur[b, :ncur[b]], roi)
            assert np.array_equal(_bits(det[b]), _bits(s1)) and len(s1) > 5, "item %d detectSingleScale" % b
            assert q[b] == fx.dmaxquality_
            s0 = fx.detectGridFASTPyr(singles[b].cur_pyr, 35, cur[b, :ncur[b]])
            assert np.array_equal(_bits(detf[b]), _bits(s0)), "item %d detectGridFAST" % b
            assert th[b] == fx.nfast_th_
            # single-item entry points on the item view give the same again
            fx2 = ov2slam_amd.FeatureExtractor(gpu_ctx, dmaxquality=[0.001, 0.004, 0.0005, 0.001, 0.001][b])
            s2 = fx2.detectSingleScalePyr(bt.cur_item(b), 35, cur[b, :ncur[b]], roi)
            assert np.array_equal(_bits(s2), _bits(s1))
        for t in singles:
            t.close()
        bt.close()


@pytest.mark.parametrize("split", [False, True])
def test_lockstep_pipeline(gpu_ctx, split):
    """The three-stage look-ahead schedules of tools/lockstep_driver.cpp -- upload(f + 2), prepare(f + 1), track_frame(f) with two prepared
    frames in flight; or (split) track_frame_begin(f), upload(f + 2), prepare(f + 1), track_frame_end(f): the enqueue of the frames to come
    beside the tracking kernels -- give the in-order results; a track_frame on other frames than the prepared ones is refused."""
    w, h, batch, n_max, nframes = 752, 480, 4, 400, 9
    seqs = [_sequence(w, h, nframes, seed=90 + b) for b in range(batch)]
    length = [9, 9, 7, 5]
    rngs = [np.random.default_rng(300 + b) for b in range(batch)]
    cal = ov2slam_amd.CameraCalibration(gpu_ctx, "pinhole", *K_EUROC, D=D_EUROC)
    bt = ov2slam_amd.LockstepTracker(gpu_ctx, batch, w, h, fclahe_val=CLIP, nbmaxkps=n_max)
    bt.setCalibration(cal)
    singles = []
    for b in range(batch):
        t = ov2slam_amd.VisualFrontEndTracker(gpu_ctx, w, h, fclahe_val=CLIP, nbmaxkps=n_max, use_graph=False)
        t.setCalibration(cal)
        singles.append(t)
    na_at = lambda f: sum(1 for b in range(batch) if length[b] > f)

    def fill(f):
        for b in range(na_at(f)):
            bt.image_buffers[f % 3][b][:, :w] = seqs[b][0][f]
    for f in range(3):
        fill(f)
    bt.upload(0, na_at(0)); bt.prepare(0, na_at(0)); bt.upload(1, na_at(1))
    z = np.zeros((batch, n_max, 2), np.float32)

    def look_ahead(f):
        if f + 2 < nframes and na_at(f + 2):
            bt.upload((f + 2) % 3, na_at(f + 2))
        if f + 1 < nframes and na_at(f + 1):
            bt.prepare((f + 1) % 3, na_at(f + 1))
    for f in range(nframes):
        na = na_at(f)
        if not split or f == 0:
            look_ahead(f)
        imgs = [bt.image_buffers[f % 3][b] for b in range(na)]
        if f == 0:
            bt.trackFrame(imgs, z, z, None, np.zeros(na, np.int32))
            for b in range(batch):
                singles[b].trackFrame(seqs[b][0][0], np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), None)
        else:
            per = [_points(w, h, seqs[b][1], f - 1, rngs[b], 1.0, bad_frac=0.2) for b in range(na)]
            kps, pri, hp, n = _pack(batch, n_max, per)
            if f == 4:                                            # not the prepared frames: refused, nothing consumed
                with pytest.raises(ov2slam_amd.Ov2Error):
                    bt.trackFrame([seqs[b][0][f] for b in range(na)], kps, pri, hp, n)
            if split:
                bt.trackFrameBegin(imgs, kps, pri, hp, n)
                with pytest.raises(ov2slam_amd.Ov2Error):         # one step at a time
                    bt.trackFrameBegin(imgs, kps, pri, hp, n)
                look_ahead(f)
                out, st, p3p = bt.trackFrameEnd()
            else:
                out, st, p3p = bt.trackFrame(imgs, kps, pri, hp, n)
            for b in range(na):
                so, ss, sp = singles[b].trackFrame(seqs[b][0][f], *per[b])
                m = len(per[b][0])
                assert np.array_equal(_bits(out[b, :m]), _bits(so)) and np.array_equal(st[b, :m], ss) and bool(p3p[b]) == sp, "frame %d item %d" % (f, b)
                bu, bb = bt.lastKeypoints(b, m)
                su, sb = singles[b].lastKeypoints(m)
                assert np.array_equal(_bits(bu), _bits(su)) and np.array_equal(bb.view(np.uint64), sb.view(np.uint64))
                gi, _ = bt.cur_item(b).download(1)
                si, _ = singles[b].cur_pyr.download(1)
                assert np.array_equal(gi, si)
        if f + 3 < nframes:
            fill(f + 3)                                           # staging set f % 3 is free again
    for t in singles:
        t.close()
    bt.close()


def test_lockstep_p3p_rule(gpu_ctx):
    """Priors so wrong in ONE item that fewer than a third survive the 2-level pass: that item's bp3preq_ is raised and its lost
    tracks restart from their keypoints (visual_front_end.cpp:225-230); the other items are untouched."""
    w, h, batch, n_max = 752, 480, 3, 400
    seqs = [_sequence(w, h, 2, seed=70 + b) for b in range(batch)]
    rng = np.random.default_rng(9)
    cal = ov2slam_amd.CameraCalibration(gpu_ctx, "pinhole", *K_EUROC)
    bt = ov2slam_amd.LockstepTracker(gpu_ctx, batch, w, h, fclahe_val=CLIP, nbmaxkps=n_max)
    bt.setCalibration(cal)
    z = np.zeros((batch, n_max, 2), np.float32)
    bt.trackFrame([s[0][0] for s in seqs], z, z, None, np.zeros(batch, np.int32))
    per = [_points(w, h, seqs[b][1], 0, rng, 1.0, frac_prior=0.8, bad_frac=0.85 if b == 1 else 0.1, bad_sigma=40.0) for b in range(batch)]
    kps, pri, hp, n = _pack(batch, n_max, per)
    out, st, p3p = bt.trackFrame([s[0][1] for s in seqs], kps, pri, hp, n)
    assert list(p3p) == [False, True, False]
    for b in range(batch):
        t = ov2slam_amd.VisualFrontEndTracker(gpu_ctx, w, h, fclahe_val=CLIP, nbmaxkps=n_max, use_graph=False)
        t.setCalibration(cal)
        t.trackFrame(seqs[b][0][0], np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), None)
        so, ss, sp = t.trackFrame(seqs[b][0][1], *per[b])
        assert sp == bool(p3p[b])
        assert np.array_equal(_bits(out[b, :n[b]]), _bits(so)) and np.array_equal(st[b, :n[b]], ss)
        bu, bb = bt.lastKeypoints(b, int(n[b]))
        su, sb = t.lastKeypoints(int(n[b]))
        assert np.array_equal(_bits(bu), _bits(su)) and np.array_equal(bb.view(np.uint64), sb.view(np.uint64))
        t.close()
    bt.close()


def test_item_view_in_stereo_matching(gpu_ctx):
    """ov2_stereo_match with `left` = an item view of the lock-step tracker's pyramid (what the mapper context of a lock-step rank
    passes) equals the call on a stand-alone pyramid of the same frame; a view refuses to be built into."""
    w, h, batch = 752, 480, 3
    tex = synth.base_texture(1400, 1234)
    lefts = [synth.warp(tex, w, h, 200 + 30 * b, 150 + 20 * b, 0.002 * b) for b in range(batch)]
    rights = [synth.warp(tex, w, h, 220 + 30 * b, 150 + 20 * b, 0.002 * b) for b in range(batch)]
    bt = ov2slam_amd.LockstepTracker(gpu_ctx, batch, w, h, fclahe_val=CLIP, nbmaxkps=400)
    z = np.zeros((batch, 400, 2), np.float32)
    bt.trackFrame(lefts, z, z, None, np.zeros(batch, np.int32))
    ctxB = ov2slam_amd.Context(0)                                  # the mapper's context: cross-stream hand-off through the parent's event
    ftrk = ov2slam_amd.FeatureTracker(ctxB, 30, 0.01)
    cal = ov2slam_amd.CameraCalibration(ctxB, "pinhole", *K_EUROC)
    rng = np.random.default_rng(3)
    for b in range(batch):
        kps = synth.grid_keypoints(w, h, 35, rng)
        unpx, _ = cal.computeKeypoints(kps)
        hp = (rng.uniform(size=len(kps)) < 0.5).astype(np.uint8)
        p3 = kps.copy(); p3[:, 0] -= 20.0; p3 += rng.normal(0, 1.0, p3.shape).astype(np.float32)
        pr = ov2slam_amd.Pyramid(ctxB, w, h, 9, 3).build_clahe(rights[b], CLIP, w // 50, h // 50)
        pl = ov2slam_amd.Pyramid(ctxB, w, h, 9, 3).build_clahe(lefts[b], CLIP, w // 50, h // 50)
        ok1, r1 = stereo.stereo_match_arrays(ftrk, bt.cur_item(b), pr, kps, unpx, p3, hp, cal, rect=True)
        ok2, r2 = stereo.stereo_match_arrays(ftrk, pl, pr, kps, unpx, p3, hp, cal, rect=True)
        assert np.array_equal(ok1, ok2) and np.array_equal(_bits(r1), _bits(r2)) and ok1.mean() > 0.5
        pl.close(); pr.close()
    # the whole batch in ONE call (ov2_pyr_build_clahe_hb + ov2_stereo_match_batch: the mapper of a lock-step rank): per item the
    # results of the single-item entry point, for both LK kernels, with different counts per item and an item that has no keypoint
    n_max = 400
    per = []
    for b in range(batch):
        kps = synth.grid_keypoints(w, h, 35, rng)[:273 - 31 * b]
        if b == 1:
            kps = kps[:0]
        unpx, _ = cal.computeKeypoints(kps) if len(kps) else (kps.copy(), None)
        hp = (rng.uniform(size=len(kps)) < 0.5).astype(np.uint8)
        p3 = kps.copy(); p3[:, 0] -= 20.0; p3 += rng.normal(0, 1.0, p3.shape).astype(np.float32)
        per.append((kps, unpx, p3, hp))
    K = np.zeros((batch, n_max, 2), np.float32); U = K.copy(); P = K.copy(); H = np.zeros((batch, n_max), np.uint8); nn = np.zeros(batch, np.int32)
    for b, (kps, unpx, p3, hp) in enumerate(per):
        m = len(kps); nn[b] = m; K[b, :m] = kps; U[b, :m] = unpx; P[b, :m] = p3; H[b, :m] = hp
    prb = ov2slam_amd.Pyramid(ctxB, w, h, 9, 3, batch=batch)
    for impl in (L.OV2_TRACK_IMPL_WAVE, L.OV2_TRACK_IMPL_ROW):
        with ctxB.options(track_impl=impl):
            prb.build_clahe_batch(rights, CLIP, w // 50, h // 50)
            okb, rb = stereo.stereo_match_batch_arrays(ftrk, bt.cur_pyr, prb, batch, n_max, K, U, P, H, nn, cal, rect=True)
            for b, (kps, unpx, p3, hp) in enumerate(per):
                if not len(kps):
                    continue
                pr = ov2slam_amd.Pyramid(ctxB, w, h, 9, 3).build_clahe(rights[b], CLIP, w // 50, h // 50)
                ok1, r1 = stereo.stereo_match_arrays(ftrk, bt.cur_item(b), pr, kps, unpx, p3, hp, cal, rect=True)
                assert np.array_equal(okb[b, :len(kps)], ok1) and np.array_equal(_bits(rb[b, :len(kps)]), _bits(r1)), "item %d" % b
                gi, _ = prb.download(2, b=b)
                si, _ = pr.download(2)
                assert np.array_equal(gi, si)
                pr.close()
    prb.close()
    view = bt.cur_item(0)
    with pytest.raises(ov2slam_amd.Ov2Error):
        L.check(gpu_ctx.lib.ov2_pyr_build_h(gpu_ctx.h, view.h_pyr, lefts[0].ctypes.data, w, 0))
    ctxB.close()
    bt.close()


def test_lockstep_argument_checks(gpu_ctx):
    bt = ov2slam_amd.LockstepTracker(gpu_ctx, 2, 376, 240, nbmaxkps=64)
    img = np.zeros((240, 376), np.uint8)
    z = np.zeros((2, 64, 2), np.float32)
    with pytest.raises(ov2slam_amd.Ov2Error):                     # more keypoints than slots
        L.check(bt.lib.ov2_btracker_track_frame(bt.h_trk, 1, (L.C.c_void_p * 1)(img.ctypes.data), 376, z.ctypes.data, z.ctypes.data, None,
                                                np.array([65], np.int32).ctypes.data, 1, z.ctypes.data, np.zeros(128, np.uint8).ctypes.data, None))
    with pytest.raises(ov2slam_amd.Ov2Error):                     # n_active beyond the batch
        bt.upload(0, 3)
    with pytest.raises(ov2slam_amd.Ov2Error):                     # detection before any frame
        bt.detectSingleScale(1, 35, z, np.zeros(1, np.int32), (5, 5, 366, 230), np.array([0.001]))
    bt.close()
