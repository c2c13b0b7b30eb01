"""CPU tests of the oracle's pyramid / LK restatement against independent numpy/scipy
implementations and analytic known answers.  (The reference ships no golden vectors
for the front-end -- SURVEY.md 4 / 8c -- so these pin the oracle's self-consistency.)"""
import numpy as np
import pytest
from scipy import ndimage


def np_pyr_down(img):
    """Independent pyrDown: float64 5x5 binomial with 'mirror' (= REFLECT_101), stride 2."""
    k = np.array([1, 4, 6, 4, 1], np.float64)
    a = ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror")
    a = ndimage.correlate1d(a, k, axis=0, mode="mirror")
    a = a[::2, ::2]
    return np.floor((a + 128) / 256).astype(np.uint8)


def np_scharr(img):
    a = img.astype(np.int64)
    sm = np.array([3, 10, 3]); df = np.array([-1, 0, 1])
    dx = ndimage.correlate1d(ndimage.correlate1d(a, sm, axis=0, mode="mirror"), df, axis=1, mode="mirror")
    dy = ndimage.correlate1d(ndimage.correlate1d(a, df, axis=0, mode="mirror"), sm, axis=1, mode="mirror")
    return np.stack([dx, dy], -1).astype(np.int16)


@pytest.mark.parametrize("shape", [(480, 752), (376, 1241), (61, 95), (33, 32)])
def test_pyr_down_matches_numpy(oracle, shape):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(oracle.pyr_down(img), np_pyr_down(img))


@pytest.mark.parametrize("shape", [(480, 752), (47, 156), (12, 11)])
def test_scharr_matches_numpy(oracle, shape):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(oracle.scharr(img), np_scharr(img))


def test_scharr_known_answer_ramp(oracle):
    # horizontal ramp of slope 2: dx = 16*2*2 = 64 away from the border, dy = 0
    img = np.tile((np.arange(40) * 2).astype(np.uint8), (20, 1))
    d = oracle.scharr(img)
    assert np.all(d[:, 1:-1, 0] == 64) and np.all(d[..., 1] == 0)
    assert np.all(d[:, 0, 0] == 0) and np.all(d[:, -1, 0] == 0)   # REFLECT_101 -> symmetric -> 0


@pytest.mark.parametrize("wh", [(752, 480), (1241, 376)])
def test_pyramid_structure(oracle, wh):
    w, h = wh
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    P = oracle.Pyramid(img, 9, 3)
    assert P.levels == 4
    sizes = [P.level_size(l) for l in range(4)]
    exp = [(w, h)]
    for _ in range(3):
        exp.append(((exp[-1][0] + 1) // 2, (exp[-1][1] + 1) // 2))
    assert sizes == exp
    cur = img
    for l in range(4):
        im, der = P.level(l)
        assert np.array_equal(im, cur)
        assert np.array_equal(der, np_scharr(cur))
        imp, derp = P.level(l, padded=True)
        assert np.array_equal(imp, np.pad(cur, 9, mode="reflect"))          # REFLECT_101
        assert np.array_equal(derp, np.pad(np_scharr(cur), ((9, 9), (9, 9), (0, 0))))  # CONSTANT 0
        cur = np_pyr_down(cur)


def test_pyramid_stops_early(oracle):
    img = np.zeros((40, 40), np.uint8)
    P = oracle.Pyramid(img, 9, 3)          # 40 -> 20 -> 10 -> (5 <= 9 stops)
    assert P.levels == 3


def np_lk_single(prev_img, next_img, pt, guess, win=9, iters=30, eps=0.01):
    """Independent float64 single-level LK (no fixed point) for a sanity envelope."""
    from scipy.ndimage import map_coordinates
    I = prev_img.astype(np.float64); J = next_img.astype(np.float64)
    d = np_scharr(prev_img).astype(np.float64) / 32.0
    hw = (win - 1) / 2
    ys, xs = np.mgrid[0:win, 0:win]
    px, py = xs + pt[0] - hw, ys + pt[1] - hw
    Iw = map_coordinates(I, [py, px], order=1, mode="mirror")
    gx = map_coordinates(d[..., 0], [py, px], order=1, mode="constant")
    gy = map_coordinates(d[..., 1], [py, px], order=1, mode="constant")
    A = np.array([[np.sum(gx * gx), np.sum(gx * gy)], [np.sum(gx * gy), np.sum(gy * gy)]])
    n = np.array(guess, np.float64)
    for _ in range(iters):
        Jw = map_coordinates(J, [ys + n[1] - hw, xs + n[0] - hw], order=1, mode="mirror")
        bvec = np.array([np.sum((Jw - Iw) * gx), np.sum((Jw - Iw) * gy)])
        delta = -np.linalg.solve(A, bvec)
        n += delta
        if delta @ delta <= eps * eps:
            break
    return n


def test_lk_recovers_known_shift(oracle, euroc_pair):
    d = euroc_pair
    P, Cq = oracle.Pyramid(d["prev"]), oracle.Pyramid(d["cur"])
    out, st, stats = oracle.fb_klt(P, Cq, 9, 3, 30., 0.5, d["kps"], d["pri"])
    assert st.mean() > 0.9
    e = np.linalg.norm(out - d["gt"], axis=1)[st]
    assert e.mean() < 0.08 and e.max() < 0.6
    assert stats[0] > 0 and stats[1] >= len(st)


def test_lk_level0_close_to_float_lk(oracle, euroc_pair):
    d = euroc_pair
    P, Cq = oracle.Pyramid(d["prev"], 9, 0), oracle.Pyramid(d["cur"], 9, 0)
    kps = d["kps"][::9]; gt = d["gt"][::9]
    guess = (gt + 0.4).astype(np.float32)
    out, st, err, iters = oracle.lk_track(P, Cq, kps, guess, 9, 0)
    for i in range(len(kps)):
        if not st[i]:
            continue
        ref = np_lk_single(d["prev"], d["cur"], kps[i], guess[i])
        assert np.linalg.norm(out[i] - ref) < 0.05, (i, out[i], ref)


def test_lk_status_semantics(oracle):
    # flat image: min eigenvalue 0 -> status 0, err (min eig) == 0
    flat = np.full((120, 160), 77, np.uint8)
    P, Cq = oracle.Pyramid(flat, 9, 1), oracle.Pyramid(flat, 9, 1)
    pts = np.array([[50, 50], [80.5, 60.25]], np.float32)
    out, st, err, _ = oracle.lk_track(P, Cq, pts, pts, 9, 1)
    assert not st.any() and np.all(err == 0)
    # a point whose window origin falls outside (-win) at level 0 -> status 0
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (120, 160), dtype=np.uint8)
    P = oracle.Pyramid(img, 9, 0)
    pts = np.array([[-6.0, 50.0], [50.0, 50.0]], np.float32)      # ipx = floor(-10) < -9
    out, st, err, _ = oracle.lk_track(P, P, pts, pts, 9, 0)
    assert st[0] == 0 and st[1] == 1
    # identical images: converges at the initial guess with exactly one iteration
    assert np.allclose(out[1], pts[1], atol=1e-3)


def test_fbklt_empty_and_threads(oracle, euroc_pair):
    d = euroc_pair
    P, Cq = oracle.Pyramid(d["prev"]), oracle.Pyramid(d["cur"])
    out, st, _ = oracle.fb_klt(P, Cq, 9, 3, 30., 0.5, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32))
    assert out.shape == (0, 2) and st.shape == (0,)
    a = oracle.fb_klt(P, Cq, 9, 3, 30., 0.5, d["kps"], d["pri"], nthreads=1)
    b = oracle.fb_klt(P, Cq, 9, 3, 30., 0.5, d["kps"], d["pri"], nthreads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # nbpyrlvl larger than the pyramid is clamped (feature_tracker.cpp:50-52)
    c = oracle.fb_klt(P, Cq, 9, 7, 30., 0.5, d["kps"], d["pri"])
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])


def test_pool_and_rebuild_do_not_change_results(oracle):
    """The persistent worker pool (oracle/pool.c, stand-in for cv::parallel_for_) and the buffer-reusing pyramid rebuild
    are pure scheduling: CLAHE, pyramid and LK results are identical for 1 and 5 threads."""
    from ov2slam_amd import synth
    prev, cur, flow = synth.frame_pair(376, 240, seed=4, shift=(2.0, -1.0), theta=0.002)
    rng = np.random.default_rng(3)
    kps = synth.grid_keypoints(376, 240, 35, rng)
    pri = (flow(kps) + rng.normal(0, 1.0, kps.shape)).astype(np.float32)
    res = []
    for nt in (1, 5):
        assert oracle.set_num_threads(nt) == nt
        eq = oracle.clahe(cur, 3.0, 7, 4)
        P0 = oracle.Pyramid(oracle.clahe(prev, 3.0, 7, 4), 9, 3)
        P1 = oracle.Pyramid(prev, 9, 3).rebuild(eq)                    # buffers re-used, other content before
        out, st, stats = oracle.fb_klt(P0, P1, 9, 3, 30., 0.5, kps, pri, nthreads=nt)
        res.append((eq, [P1.level(l, padded=True) for l in range(P1.levels)], out, st, stats))
    oracle.set_num_threads(1)
    a, b = res
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    assert np.array_equal(a[3], b[3]) and a[4] == b[4]
    for (ia, da), (ib, db) in zip(a[1], b[1]):
        assert np.array_equal(ia, ib) and np.array_equal(da, db)
    fresh = oracle.Pyramid(a[0], 9, 3)
    for l in range(fresh.levels):
        fi, fd = fresh.level(l, padded=True)
        assert np.array_equal(fi, a[1][l][0]) and np.array_equal(fd, a[1][l][1])


def test_float_accumulator_variants_stay_within_a_hundredth_of_a_pixel(oracle):
    """The HIP kernels are bit-exact against the oracle's int64-accumulator LK; stock OpenCV builds accumulate in float, in a
    build-dependent order (scalar / 3.4 SSE2 intrinsics / 4.x universal intrinsics, restated in oracle/frontend.c).  This bounds
    the distance on the config-2 input (CLAHE'd EuRoC-sized frames) and on a worst case of hard-edged binary blocks: no status
    flip, sub-0.02-px positions -- and the variants are really different code paths (they do differ in the last bits)."""
    from ov2slam_amd import synth
    O = oracle
    w, h = 752, 480
    prev, cur, flow = synth.frame_pair(w, h, seed=3, shift=(3.8, -1.7), theta=0.004)
    prev, cur = O.clahe(prev, 3.0, w // 50, h // 50), O.clahe(cur, 3.0, w // 50, h // 50)
    rng = np.random.default_rng(3)
    kps = synth.grid_keypoints(w, h, 35, rng)
    pri = (flow(kps) + rng.normal(0, 1.5, kps.shape)).astype(np.float32)
    rep = O.lk_acc_mode_report(O.Pyramid(prev, 9, 3), O.Pyramid(cur, 9, 3), kps, pri)
    assert rep["tracked_int64"] > 0.9 * rep["points"]
    for name, m in rep["modes"].items():
        assert m["status_flips"] == 0, name
        assert m["max_abs_dpx"] < 2e-3, name
        assert m["bit_identical_positions"] > 0.8 * rep["tracked_int64"], name
    # worst case: binary 7x7 blocks (Scharr responses up to +-4080, sums far beyond 2^24)
    big = np.kron((np.random.default_rng(9).integers(0, 2, (120, 140)) * 255).astype(np.uint8), np.ones((7, 7), np.uint8))
    prev, cur = big[100:100 + h, 100:100 + w].copy(), big[98:98 + h, 96:96 + w].copy()
    pri = (kps + np.array([4, 2], np.float32) + rng.normal(0, 1.0, kps.shape)).astype(np.float32)
    rep = O.lk_acc_mode_report(O.Pyramid(prev, 9, 3), O.Pyramid(cur, 9, 3), kps, pri)
    assert rep["tracked_int64"] > 0.5 * rep["points"]
    differing = 0
    for name, m in rep["modes"].items():
        assert m["status_flips"] <= 2 and m["max_abs_dpx"] < 0.02, (name, m)
        differing += rep["tracked_int64"] - m["bit_identical_positions"]
    assert differing > 0
    assert O.lib().orc_get_lk_acc_mode() == O.LK_ACC_INT64          # the context manager restored the canonical mode
