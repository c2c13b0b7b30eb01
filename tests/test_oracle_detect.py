"""CPU tests of the oracle's detector restatement against independent numpy/scipy
implementations and analytic known answers (no reference golden vectors exist)."""
import numpy as np
import pytest
from scipy import ndimage

from ov2slam_amd import synth

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
          (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def np_fast(img, t, nonmax=True):
    """Brute-force FAST-9/16 from the textbook definition (Rosten & Drummond) with the
    OpenCV score = max threshold for which the pixel is still a corner."""
    img = img.astype(np.int32)
    h, w = img.shape
    score = np.zeros((h, w), np.int32)
    corner = np.zeros((h, w), bool)

    def is_corner(ring, v, th):
        for sign in (1, -1):
            ok = (sign * (ring - v)) > th
            ok2 = np.concatenate([ok, ok[:8]])
            run = 0
            for k in range(24):
                run = run + 1 if ok2[k] else 0
                if run >= 9:
                    return True
        return False

    for y in range(3, h - 3):
        for x in range(3, w - 3):
            ring = np.array([img[y + dy, x + dx] for dx, dy in CIRCLE])
            v = img[y, x]
            if is_corner(ring, v, t):
                corner[y, x] = True
                th = t
                while th < 255 and is_corner(ring, v, th + 1):
                    th += 1
                score[y, x] = th
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if not corner[y, x]:
                continue
            s = score[y, x]
            nb = score[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if not nonmax or np.all(s > nb):
                out.append((x, y, s))
    return out


@pytest.mark.parametrize("seed,t", [(0, 10), (1, 20), (2, 6)])
def test_fast_matches_bruteforce(oracle, seed, t):
    tex = synth.base_texture(300, 40 + seed, nblobs=300)
    img = synth.warp(tex, 50, 50, 60 + 37 * seed, 80)
    xs, ys, sc = oracle.fast9_16(img, t)
    ref = np_fast(img, t)
    assert len(ref) > 0
    assert [(int(a), int(b), int(c)) for a, b, c in zip(xs, ys, sc)] == [(int(a), int(b), int(c)) for a, b, c in ref]


def test_fast_no_nms_and_flat(oracle):
    assert len(oracle.fast9_16(np.full((35, 35), 100, np.uint8), 10)[0]) == 0
    img = np.full((35, 35), 50, np.uint8)
    img[10:20, 10:20] = 200                                 # bright square: its 4 corners fire
    xs, ys, sc = oracle.fast9_16(img, 10, nonmax=False)
    pts = set(zip(xs.tolist(), ys.tolist()))
    assert {(10, 10), (19, 10), (10, 19), (19, 19)} <= pts
    for p in pts:
        assert min(abs(p[0] - 10), abs(p[0] - 19)) <= 2 and min(abs(p[1] - 10), abs(p[1] - 19)) <= 2
    assert np.all(sc == 149)                                # 200 - 50 - 1
    # all scores tie -> the strict 3x3 NMS suppresses every one of them (OpenCV semantics)
    assert len(oracle.fast9_16(img, 10, nonmax=True)[0]) == 0


def np_circle(h, w, cx, cy, r):
    """Independent midpoint-circle fill written from the algorithm description in SURVEY.md C4."""
    m = np.ones((h, w), np.uint8)
    err, dx, dy, plus, minus = 0, r, 0, 1, 2 * r - 1
    while dx >= dy:
        for (yy, half) in ((cy - dy, dx), (cy + dy, dx), (cy - dx, dy), (cy + dx, dy)):
            if 0 <= yy < h:
                m[yy, max(0, cx - half):max(0, min(w, cx + half + 1))] = 0
        dy += 1; err += plus; plus += 2
        if err > 0:
            err -= minus; dx -= 1; minus -= 2
    return m


@pytest.mark.parametrize("c", [(30, 30, 8), (3, 5, 12), (59, 59, 8), (-3, 20, 8), (30, 70, 11), (200, 200, 8)])
def test_circle(oracle, c):
    cx, cy, r = c
    m = oracle.circle_fill0(np.ones((64, 60), np.uint8), cx, cy, r)
    assert np.array_equal(m, np_circle(64, 60, cx, cy, r))
    # every zeroed pixel lies within r + 0.5 of the centre, every pixel within r - 1 is zeroed
    ys, xs = np.mgrid[0:64, 0:60]
    d = np.hypot(xs - cx, ys - cy)
    assert np.all(d[m == 0] <= r + 0.71)
    assert np.all(m[d <= r - 1] == 0)


def np_mineig(img, x0, y0, cs):
    """Independent float64 version of blur -> Sobel/3060 -> 3x3 box -> min eigenvalue."""
    k = np.array([1, 2, 1], np.int64)
    b = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.int64), k, axis=0, mode="mirror"), k, axis=1, mode="mirror")
    b = ((b + 8) >> 4)[y0:y0 + cs, x0:x0 + cs].astype(np.float64)
    s = 1.0 / 3060.0
    dx = ndimage.correlate1d(ndimage.correlate1d(b, [1, 2, 1], axis=0, mode="mirror"), [-1, 0, 1], axis=1, mode="mirror") * s
    dy = ndimage.correlate1d(ndimage.correlate1d(b, [-1, 0, 1], axis=0, mode="mirror"), [1, 2, 1], axis=1, mode="mirror") * s
    box = lambda a: ndimage.uniform_filter(a, 3, mode="mirror") * 9
    a, bb, c = box(dx * dx) * 0.5, box(dx * dy), box(dy * dy) * 0.5
    return (a + c) - np.sqrt((a - c) ** 2 + bb * bb)


@pytest.mark.parametrize("x0,y0,cs", [(0, 0, 35), (70, 105, 35), (700, 420, 35), (100, 50, 50)])
def test_cell_mineig_close_to_float64(oracle, x0, y0, cs):
    prev, _, _ = synth.frame_pair(752, 480, seed=5)
    h = oracle.cell_mineig(prev, x0, y0, cs)
    ref = np_mineig(prev, x0, y0, cs)
    assert h.shape == (cs, cs)
    assert np.allclose(h, ref, rtol=2e-4, atol=2e-9)
    assert ref.max() > 1e-4


def test_corner_subpix_finds_synthetic_corner(oracle):
    # anti-aliased quadrant corner at a known sub-pixel location
    cx, cy = 40.37, 33.81
    ys, xs = np.mgrid[0:80, 0:96].astype(np.float64)
    sx = np.clip((xs - cx) + 0.5, 0, 1); sy = np.clip((ys - cy) + 0.5, 0, 1)
    img = (40 + 170 * sx * sy)
    img = ndimage.gaussian_filter(img, 1.0)
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    out = oracle.corner_subpix(img, [[41.0, 34.0], [39.0, 33.0]])
    assert np.all(np.abs(out - [cx, cy]) < 0.25), out
    # far from any structure (flat): determinant ~ 0 -> point unchanged
    out2 = oracle.corner_subpix(img, [[10.0, 10.0]])
    assert np.array_equal(out2, np.array([[10.0, 10.0]], np.float32))


def test_detect_grid_fast_properties(oracle):
    prev, _, _ = synth.frame_pair(752, 480, seed=8)
    for mode in (oracle.MASK_AS_EXECUTED, oracle.MASK_INTENDED):
        pts, th = oracle.detect_grid_fast(prev, 50, np.zeros((0, 2), np.float32), 10, mode, subpix=False)
        assert 20 < len(pts) <= 135
        cells = (pts[:, 1] // 50).astype(int) * 15 + (pts[:, 0] // 50).astype(int)
        assert len(set(cells.tolist())) == len(pts)            # at most one per cell
        lx = pts[:, 0].astype(int) % 50
        assert np.all((lx >= 3) & (lx <= 46))
        if mode == oracle.MASK_AS_EXECUTED:
            assert np.all(lx % 4 >= 2)
        # pairwise distance > radius 12 in INTENDED mode (exclusion discs)
        if mode == oracle.MASK_INTENDED:
            d = np.linalg.norm(pts[:, None] - pts[None], axis=2) + np.eye(len(pts)) * 1e3
            assert d.min() > 12 - 1.5
        assert th in (6, 10, 15)
    # occupied cells are skipped
    cur = np.array([[25.0, 25.0], [420.0, 260.0]], np.float32)
    pts2, _ = oracle.detect_grid_fast(prev, 50, cur, 10, oracle.MASK_INTENDED, subpix=False)
    cells2 = set(((pts2[:, 1] // 50).astype(int) * 15 + (pts2[:, 0] // 50).astype(int)).tolist())
    assert 0 not in cells2 and (5 * 15 + 8) not in cells2
    # empty-ish image: nothing found and threshold decays 10 -> 6
    flat = np.full((480, 752), 128, np.uint8)
    pts3, th3 = oracle.detect_grid_fast(flat, 50, np.zeros((0, 2), np.float32), 10)
    assert len(pts3) == 0 and th3 == 6


def test_detect_singlescale_properties(oracle):
    prev, _, _ = synth.frame_pair(752, 480, seed=9)
    roi = (5, 5, 742, 470)
    pts, q = oracle.detect_singlescale(prev, 35, np.zeros((0, 2), np.float32), roi, 0.001, subpix=False)
    assert 100 < len(pts) <= 2 * 273
    assert np.all((pts[:, 0] >= 5) & (pts[:, 0] < 747) & (pts[:, 1] >= 5) & (pts[:, 1] < 475))
    d = np.linalg.norm(pts[:, None] - pts[None], axis=2) + np.eye(len(pts)) * 1e3
    assert d.min() > 8 - 1.5                                   # exclusion discs of radius cell/4
    assert q in (0.0005, 0.001, 0.0015)
    # with every cell occupied nothing is detected
    cur = synth.grid_keypoints(752, 480, 35, np.random.default_rng(0), jitter=0.1)
    pts2, _ = oracle.detect_singlescale(prev, 35, cur, roi, 0.001, subpix=False)
    assert len(pts2) == 0
    # sub-pixel refinement moves points by less than the 3-px window
    pts3, _ = oracle.detect_singlescale(prev, 35, np.zeros((0, 2), np.float32), roi, 0.001, subpix=True)
    assert len(pts3) == len(pts) and np.abs(pts3 - pts).max() <= 3.0 + 1e-6


def test_sobel_dy_order_switch(oracle):
    """cv::Sobel(dx=0, dy=1, scale) scales the smoothing kernel: OpenCV's row-filter order is the default, round 1's
    exact-sum order the alternative; they agree to <= 1e-6 relative on the min-eigenvalue map (and normally select the
    same keypoints), and each is reproducible."""
    from ov2slam_amd import synth
    prev, _, _ = synth.frame_pair(376, 240, seed=6)
    assert oracle.lib().orc_get_sobel_dy_order() == oracle.SOBEL_DY_OPENCV_ROWFILTER
    a = oracle.cell_mineig(prev, 70, 35, 35)
    assert oracle.set_sobel_dy_order(oracle.SOBEL_DY_EXACT_SUM) == oracle.SOBEL_DY_OPENCV_ROWFILTER
    try:
        b = oracle.cell_mineig(prev, 70, 35, 35)
    finally:
        oracle.set_sobel_dy_order(oracle.SOBEL_DY_OPENCV_ROWFILTER)
    assert not np.array_equal(a.view(np.uint32), b.view(np.uint32))          # the orders are really different ...
    assert np.abs(a - b).max() <= 2e-6 * np.abs(a).max()                      # ... by rounding only
    assert np.array_equal(a, oracle.cell_mineig(prev, 70, 35, 35))
    # dy of a horizontal step edge: both orders must see the same sign / magnitude to 1e-6
    img = np.zeros((35, 35), np.uint8); img[18:] = 200
    m = oracle.cell_mineig(img, 0, 0, 35)
    assert np.isfinite(m).all()
