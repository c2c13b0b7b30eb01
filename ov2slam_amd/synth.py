"""Deterministic synthetic inputs shaped like BASELINE.json's configs.
numpy only; used by tests/ and bench.py (no dataset is available offline)."""
import numpy as np


def _gauss_blur(a, sigma):
    r = max(1, int(3 * sigma + 0.5))
    x = np.arange(-r, r + 1, dtype=np.float64)
    k = np.exp(-0.5 * (x / sigma) ** 2)
    k /= k.sum()
    a = np.apply_along_axis(lambda v: np.convolve(np.pad(v, r, mode="reflect"), k, mode="valid"), 1, a)
    a = np.apply_along_axis(lambda v: np.convolve(np.pad(v, r, mode="reflect"), k, mode="valid"), 0, a)
    return a


def base_texture(size=1536, seed=1234, nblobs=600):
    """Band-limited noise + high-contrast blobs/corners, float64 in [0,255]."""
    rng = np.random.default_rng(seed)
    a = _gauss_blur(rng.uniform(0, 1, (size, size)), 1.5)
    a += 0.6 * _gauss_blur(rng.uniform(0, 1, (size, size)), 6.0)
    a = (a - a.min()) / (a.max() - a.min())
    for _ in range(nblobs):
        cx, cy = rng.integers(8, size - 24, 2)
        bw, bh = rng.integers(4, 16, 2)
        a[cy:cy + bh, cx:cx + bw] = rng.uniform(0, 1)
    a = _gauss_blur(a, 0.7)
    a = (a - a.min()) / (a.max() - a.min())
    return a * 255.0


def warp(tex, w, h, tx, ty, theta=0.0, scale=1.0):
    """Sample a w x h view of `tex` (bilinear) under a similarity transform about the
    view centre; returns uint8."""
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    c, s = np.cos(theta) * scale, np.sin(theta) * scale
    u = c * (xs - cx) - s * (ys - cy) + cx + tx
    v = s * (xs - cx) + c * (ys - cy) + cy + ty
    u = np.clip(u, 0, tex.shape[1] - 1.001)
    v = np.clip(v, 0, tex.shape[0] - 1.001)
    x0, y0 = np.floor(u).astype(int), np.floor(v).astype(int)
    a, b = u - x0, v - y0
    out = (tex[y0, x0] * (1 - a) * (1 - b) + tex[y0, x0 + 1] * a * (1 - b)
           + tex[y0 + 1, x0] * (1 - a) * b + tex[y0 + 1, x0 + 1] * a * b)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def frame_pair(w=752, h=480, seed=1234, shift=(3.3, -2.1), theta=0.004, tex=None):
    """(prev, cur, flow_fn): two views of one texture; flow_fn maps prev pixel coords to
    cur pixel coords (ground truth)."""
    if tex is None:
        tex = base_texture(max(w, h) + 256, seed)
    ox, oy = 100.0, 100.0
    prev = warp(tex, w, h, ox, oy)
    cur = warp(tex, w, h, ox + shift[0], oy + shift[1], theta)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0

    def flow(pts):
        pts = np.asarray(pts, np.float64)
        # texture coord of prev pixel p: p + (ox,oy).  cur pixel q maps to R(q-c)+c+o+shift.
        tx_, ty_ = pts[:, 0] + ox, pts[:, 1] + oy
        dx, dy = tx_ - cx - ox - shift[0], ty_ - cy - oy - shift[1]
        c, s = np.cos(-theta), np.sin(-theta)
        return np.stack([c * dx - s * dy + cx, s * dx + c * dy + cy], 1)

    return prev, cur, flow


def grid_keypoints(w, h, cell, rng, jitter=0.45):
    """One keypoint per cell (like the grid detectors' output), float32 (n,2)."""
    xs = (np.arange(w // cell) + 0.5) * cell
    ys = (np.arange(h // cell) + 0.5) * cell
    g = np.stack(np.meshgrid(xs, ys), -1).reshape(-1, 2)
    g += rng.uniform(-jitter, jitter, g.shape) * cell
    g[:, 0] = np.clip(g[:, 0], 6, w - 7)
    g[:, 1] = np.clip(g[:, 1], 6, h - 7)
    return g.astype(np.float32)
