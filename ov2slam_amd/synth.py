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
    of `obs_per_lm` KFs starting at its anchor (the anchor's own left observation carries no factor,
    src/optimizer.cpp:269-289).  Returns a dict of flat numpy arrays in the layout of ov2_ba_problem
    (include/ov2slam_hip.h) plus the ground truth.  Residual blocks are emitted landmark by landmark:
    [RIGHT_ANCH] then for every observer [LEFT, RIGHT]."""
    rng = np.random.default_rng(seed)
    fx = fy = 458.654; cx, cy = 367.215, 248.375
    K = np.array([fx, fy, cx, cy])
    obs_per_lm = min(obs_per_lm, n_kf)
    th = np.deg2rad(3.0) * np.arange(n_kf)
    ts = 10.0 * np.stack([np.cos(th), np.sin(th), np.zeros(n_kf)], 1)
    zc = -np.stack([np.cos(th), np.sin(th), np.zeros(n_kf)], 1)
    yc = np.tile(np.array([0, 0, -1.0]), (n_kf, 1))
    xc = np.cross(yc, zc)
    Rs = np.stack([xc, yc, zc], 2)                      # (n_kf,3,3) Rwc, columns = camera axes
    poses_gt = np.zeros((n_kf, 7))
    poses_gt[:, :3] = ts
    for k in range(n_kf):
        poses_gt[k, 3:] = _quat_from_R(Rs[k])
    baseline = 0.11
    T_rl = np.array([-baseline, 0, 0, 0, 0, 0, 1.0])

    def project(kf, X, right=False):
        pc = np.einsum("nji,nj->ni", Rs[kf], X - ts[kf])             # Rcw (X - t)
        if right:
            pc = pc + T_rl[:3]
        return np.stack([fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy], 1), pc[:, 2]

    anchors = rng.integers(0, n_kf - obs_per_lm + 1, n_lm)
    X = np.stack([rng.uniform(-3, 3, n_lm), rng.uniform(-3, 3, n_lm), rng.uniform(-2, 2, n_lm)], 1)
    lm_anchor_uv, z_a = project(anchors, X)
    invdepth_gt = 1.0 / z_a
    n_obs = obs_per_lm - 1
    lm_rep = np.repeat(np.arange(n_lm), n_obs)
    kf_rep = (anchors[:, None] + 1 + np.arange(n_obs)[None, :]).reshape(-1)
    uv_l, _ = project(kf_rep, X[lm_rep])
    out_l = rng.uniform(size=len(lm_rep)) < outlier_frac
    noise_l = np.where(out_l[:, None], rng.uniform(-50, 50, uv_l.shape), rng.normal(0, px_noise, uv_l.shape))
    uv_l = uv_l + noise_l
    if stereo:
        uv_ra, _ = project(anchors, X, True)
        uv_ra = uv_ra + rng.normal(0, px_noise, uv_ra.shape)
        uv_r, _ = project(kf_rep, X[lm_rep], True)
        uv_r = uv_r + rng.normal(0, px_noise, uv_r.shape)
        per = 1 + 2 * n_obs
        n_res = n_lm * per
        res_type = np.zeros((n_lm, per), np.uint8); res_kf = np.zeros((n_lm, per), np.int32)
        res_uv = np.zeros((n_lm, per, 2)); is_out = np.zeros((n_lm, per), bool)
        res_type[:, 0] = 2; res_kf[:, 0] = anchors; res_uv[:, 0] = uv_ra
        res_type[:, 1::2] = 0; res_type[:, 2::2] = 1
        res_kf[:, 1::2] = kf_rep.reshape(n_lm, n_obs); res_kf[:, 2::2] = kf_rep.reshape(n_lm, n_obs)
        res_uv[:, 1::2] = uv_l.reshape(n_lm, n_obs, 2); res_uv[:, 2::2] = uv_r.reshape(n_lm, n_obs, 2)
        is_out[:, 1::2] = out_l.reshape(n_lm, n_obs)
        res_lm = np.repeat(np.arange(n_lm, dtype=np.int32), per)
        res_type = res_type.reshape(-1); res_kf = res_kf.reshape(-1); res_uv = res_uv.reshape(-1, 2); is_out = is_out.reshape(-1)
    else:
        n_res = len(lm_rep)
        res_type = np.zeros(n_res, np.uint8); res_kf = kf_rep.astype(np.int32); res_lm = lm_rep.astype(np.int32)
        res_uv = uv_l; is_out = out_l
    poses0 = poses_gt.copy()
    for k in range(n_kf):
        dt = rng.normal(0, pose_noise[0], 3); dw = rng.normal(0, pose_noise[1], 3)
        poses0[k, :3] = ts[k] + dt; poses0[k, 3:] = _quat_from_R(_so3_exp(dw) @ Rs[k])
    kf_const = np.zeros(n_kf, np.uint8); kf_const[0] = 1
    if not stereo and n_kf > 1:
        kf_const[1] = 1
    poses0[kf_const == 1] = poses_gt[kf_const == 1]
    invdepth0 = invdepth_gt * (1 + rng.normal(0, invdepth_noise, n_lm))
    return dict(n_kf=n_kf, n_lm=n_lm, n_res=int(n_res),
                poses=np.ascontiguousarray(poses0), kf_const=kf_const,
                invdepth=np.ascontiguousarray(invdepth0), lm_anchor_kf=anchors.astype(np.int32),
                lm_anchor_uv=np.ascontiguousarray(lm_anchor_uv),
                res_type=np.ascontiguousarray(res_type), res_kf=np.ascontiguousarray(res_kf, np.int32),
                res_lm=np.ascontiguousarray(res_lm, np.int32),
                res_uv=np.ascontiguousarray(res_uv, np.float64), res_sigma=np.ones(int(n_res)),
                calib_l=K.copy(), calib_r=K.copy(), T_rl=T_rl,
                poses_gt=poses_gt, invdepth_gt=invdepth_gt, is_outlier=np.ascontiguousarray(is_out))


def make_pnp_problem(n_pts=300, seed=7, px_noise=1.0, outlier_frac=0.05, pose_noise=(0.05, np.deg2rad(1.0))):
    """Motion-only BA (MultiViewGeometry::ceresPnP, src/multi_view_geometry.cpp:492-586): one free pose, `n_pts`
    fixed world points observed in the left camera.  Same flat layout as make_ba_problem (n_lm = 0, residual
    type 3 = OV2_RES_PNP with res_xyz)."""
    rng = np.random.default_rng(seed)
    fx = fy = 458.654; cx, cy = 367.215, 248.375
    R = _so3_exp(rng.normal(0, 0.3, 3)); t = rng.normal(0, 1.0, 3)
    pose_gt = np.concatenate([t, _quat_from_R(R)])
    pc = np.stack([rng.uniform(-3, 3, n_pts), rng.uniform(-2, 2, n_pts), rng.uniform(2, 12, n_pts)], 1)   # camera frame
    X = (R @ pc.T).T + t
    uv = np.stack([fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy], 1)
    out = rng.uniform(size=n_pts) < outlier_frac
    uv = uv + np.where(out[:, None], rng.uniform(-60, 60, uv.shape), rng.normal(0, px_noise, uv.shape))
    R0 = _so3_exp(rng.normal(0, pose_noise[1], 3)) @ R
    pose0 = np.concatenate([t + rng.normal(0, pose_noise[0], 3), _quat_from_R(R0)])
    K = np.array([fx, fy, cx, cy])
    return dict(n_kf=1, n_lm=0, n_res=n_pts, poses=pose0[None].copy(), kf_const=np.zeros(1, np.uint8),
                invdepth=np.zeros(0), lm_anchor_kf=np.zeros(0, np.int32), lm_anchor_uv=np.zeros((0, 2)),
                res_type=np.full(n_pts, 3, np.uint8), res_kf=np.zeros(n_pts, np.int32), res_lm=np.full(n_pts, -1, np.int32),
                res_uv=np.ascontiguousarray(uv), res_sigma=np.ones(n_pts), res_xyz=np.ascontiguousarray(X),
                calib_l=K.copy(), calib_r=K.copy(), T_rl=np.array([0, 0, 0, 0, 0, 0, 1.0]),
                poses_gt=pose_gt[None].copy(), is_outlier=out)


def make_structure_problem(n_kf=12, n_pts=500, obs_per_pt=5, stereo=True, seed=7, px_noise=1.0, outlier_frac=0.02, xyz_noise=0.15):
    """Synthetic Optimizer::structureOnlyBA problem (src/optimizer.cpp:2594-2781): constant keyframes on an arc,
    3-D points seen by `obs_per_pt` consecutive keyframes (left, plus right when stereo), perturbed initial
    points.  Flat arrays in the layout of ov2_sba_problem."""
    rng = np.random.default_rng(seed)
    fx = fy = 458.654; cx, cy = 367.215, 248.375
    K = np.array([fx, fy, cx, cy])
    obs_per_pt = min(obs_per_pt, n_kf)
    th = np.deg2rad(3.0) * np.arange(n_kf)
    poses = np.zeros((n_kf, 7)); Rs = []
    for k in range(n_kf):
        c, s = np.cos(th[k]), np.sin(th[k])
        R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])           # camera-to-world
        t = np.array([10 * np.sin(th[k]), 0.05 * k, 10 - 10 * np.cos(th[k])])
        w = np.sqrt(max(0.0, 1 + np.trace(R))) / 2
        if w > 0.1:
            q = np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])
        else:                                                       # near half a turn (keyframe 60 of the arc): half-angle form
            q = np.array([0.0, np.sin(0.5 * th[k]), 0.0, np.cos(0.5 * th[k])])
        poses[k] = np.concatenate([t, q]); Rs.append(R)
    T_rl = np.array([-0.11, 0, 0, 0, 0, 0, 1.0])
    xyz_gt = np.zeros((n_pts, 3)); first = rng.integers(0, n_kf - obs_per_pt + 1, n_pts)
    rt, rk, rp, ruv, rs, is_out = [], [], [], [], [], []
    for i in range(n_pts):
        a = first[i] + obs_per_pt // 2
        pc = np.array([rng.uniform(-2.5, 2.5), rng.uniform(-1.5, 1.5), rng.uniform(4, 12)])
        xyz_gt[i] = Rs[a] @ pc + poses[a, :3]
        for k in range(first[i], first[i] + obs_per_pt):
            c = Rs[k].T @ (xyz_gt[i] - poses[k, :3])
            for typ in ((0, 1) if stereo else (0,)):
                cc = c + T_rl[:3] if typ == 1 else c
                uv = np.array([fx * cc[0] / cc[2] + cx, fy * cc[1] / cc[2] + cy]) + rng.normal(0, px_noise, 2)
                out = rng.random() < outlier_frac
                if out:
                    uv += rng.uniform(-40, 40, 2)
                rt.append(typ); rk.append(k); rp.append(i); ruv.append(uv); rs.append(2.0 ** rng.integers(0, 2)); is_out.append(out)
    xyz0 = xyz_gt + rng.normal(0, xyz_noise, xyz_gt.shape)
    return dict(n_kf=n_kf, n_pts=n_pts, n_res=len(rt), poses=poses, xyz=xyz0, xyz_gt=xyz_gt, res_type=np.array(rt, np.uint8),
                res_kf=np.array(rk, np.int32), res_pt=np.array(rp, np.int32), res_uv=np.array(ruv), res_sigma=np.array(rs, np.float64),
                calib_l=K, calib_r=K.copy(), T_rl=T_rl, is_outlier=np.array(is_out))


def make_xyz_ba_problem(n_kf=12, n_pts=500, obs_per_pt=5, stereo=True, seed=7, px_noise=1.0, outlier_frac=0.02, xyz_noise=0.1,
                        pose_noise=(0.02, np.deg2rad(0.5))):
    """Bundle adjustment over 3-D points AND keyframe poses: the `buse_inv_depth: 0` branch of Optimizer::localBA
    (src/optimizer.cpp:207-209, :333-384).  make_structure_problem's scene with perturbed, variable poses (keyframe 0 --
    and keyframe 1 for mono -- constant, :397-407).  Layout = ov2_xyzba_problem (ov2_sba_problem + kf_const)."""
    pb = make_structure_problem(n_kf, n_pts, obs_per_pt, stereo, seed, px_noise, outlier_frac, xyz_noise)
    rng = np.random.default_rng(seed + 1000)
    poses_gt = pb["poses"].copy()
    poses0 = poses_gt.copy()
    kf_const = np.zeros(n_kf, np.uint8); kf_const[0] = 1
    if not stereo and n_kf > 1:
        kf_const[1] = 1
    for k in range(n_kf):
        if kf_const[k]:
            continue
        R = _R_from_quat(poses_gt[k, 3:])
        poses0[k, :3] = poses_gt[k, :3] + rng.normal(0, pose_noise[0], 3)
        poses0[k, 3:] = _quat_from_R(_so3_exp(rng.normal(0, pose_noise[1], 3)) @ R)
    pb.update(poses=np.ascontiguousarray(poses0), poses_gt=poses_gt, kf_const=kf_const)
    return pb
