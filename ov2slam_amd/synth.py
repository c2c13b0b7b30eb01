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


# ----------------------------------------------------------------------------- local BA
def _quat_from_R(R):
    """rotation matrix -> [x y z w] (Eigen coefficient order)"""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0, 0, 0, 0]
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    q = np.array(q)
    return q / np.linalg.norm(q)


def _R_from_quat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _so3_exp(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def make_ba_problem(n_kf=50, n_lm=10000, obs_per_lm=30, stereo=False, seed=42, px_noise=1.0,
                    outlier_frac=0.02, pose_noise=(0.02, np.deg2rad(0.5)), invdepth_noise=0.05):
    """Synthetic anchored-inverse-depth local-BA problem: KFs on a 10 m arc
    looking inward (3 deg apart), landmarks in the viewed volume, each seen by a contiguous range
    of `obs_per_lm` KFs starting at its anchor.  Returns a dict of flat numpy arrays in the layout
    of ov2_ba_problem (include/ov2slam_hip.h) plus the ground truth."""
    rng = np.random.default_rng(seed)
    fx = fy = 458.654; cx, cy = 367.215, 248.375
    K = np.array([fx, fy, cx, cy])
    obs_per_lm = min(obs_per_lm, n_kf)
    poses_gt = np.zeros((n_kf, 7))
    Rs, ts = [], []
    for k in range(n_kf):
        th = np.deg2rad(3.0) * k
        c = 10.0 * np.array([np.cos(th), np.sin(th), 0.0])
        zc = -np.array([np.cos(th), np.sin(th), 0.0]); yc = np.array([0, 0, -1.0]); xc = np.cross(yc, zc)
        R = np.stack([xc, yc, zc], 1)
        Rs.append(R); ts.append(c)
        poses_gt[k, :3] = c; poses_gt[k, 3:] = _quat_from_R(R)
    baseline = 0.11
    T_rl = np.array([-baseline, 0, 0, 0, 0, 0, 1.0])

    def project(k, X, right=False):
        pc = Rs[k].T @ (X - ts[k])
        if right:
            pc = pc + T_rl[:3]
        return np.array([fx * pc[0] / pc[2] + cx, fy * pc[1] / pc[2] + cy]), pc[2]

    anchors = rng.integers(0, n_kf - obs_per_lm + 1, n_lm)
    X = np.stack([rng.uniform(-3, 3, n_lm), rng.uniform(-3, 3, n_lm), rng.uniform(-2, 2, n_lm)], 1)
    lm_anchor_uv = np.zeros((n_lm, 2)); invdepth_gt = np.zeros(n_lm)
    res_type, res_kf, res_lm, res_uv = [], [], [], []
    is_outlier = []
    for l in range(n_lm):
        a = int(anchors[l])
        uv, z = project(a, X[l])
        lm_anchor_uv[l] = uv
        invdepth_gt[l] = 1.0 / z
        if stereo:
            uvr, _ = project(a, X[l], True)
            res_type.append(2); res_kf.append(a); res_lm.append(l); res_uv.append(uvr + rng.normal(0, px_noise, 2)); is_outlier.append(False)
        for k in range(a + 1, a + obs_per_lm):
            uvk, _ = project(k, X[l])
            out = rng.uniform() < outlier_frac
            n = rng.uniform(-50, 50, 2) if out else rng.normal(0, px_noise, 2)
            res_type.append(0); res_kf.append(k); res_lm.append(l); res_uv.append(uvk + n); is_outlier.append(out)
            if stereo:
                uvr, _ = project(k, X[l], True)
                res_type.append(1); res_kf.append(k); res_lm.append(l); res_uv.append(uvr + rng.normal(0, px_noise, 2)); is_outlier.append(False)
    poses0 = poses_gt.copy()
    for k in range(n_kf):
        dt = rng.normal(0, pose_noise[0], 3); dw = rng.normal(0, pose_noise[1], 3)
        R = _so3_exp(dw) @ Rs[k]
        poses0[k, :3] = ts[k] + dt; poses0[k, 3:] = _quat_from_R(R)
    kf_const = np.zeros(n_kf, np.uint8); kf_const[0] = 1
    if not stereo and n_kf > 1:
        kf_const[1] = 1
    poses0[kf_const == 1] = poses_gt[kf_const == 1]
    invdepth0 = invdepth_gt * (1 + rng.normal(0, invdepth_noise, n_lm))
    n_res = len(res_type)
    return dict(n_kf=n_kf, n_lm=n_lm, n_res=n_res,
                poses=np.ascontiguousarray(poses0), kf_const=kf_const,
                invdepth=np.ascontiguousarray(invdepth0), lm_anchor_kf=anchors.astype(np.int32),
                lm_anchor_uv=np.ascontiguousarray(lm_anchor_uv),
                res_type=np.array(res_type, np.uint8), res_kf=np.array(res_kf, np.int32), res_lm=np.array(res_lm, np.int32),
                res_uv=np.ascontiguousarray(np.array(res_uv, np.float64).reshape(-1, 2)), res_sigma=np.ones(n_res),
                calib_l=K.copy(), calib_r=K.copy(), T_rl=T_rl,
                poses_gt=poses_gt, invdepth_gt=invdepth_gt, is_outlier=np.array(is_outlier, bool))
