#!/usr/bin/env python3
"""A pose trajectory through the hot path, GPU against the oracle (VERDICT r5 "missing" 4: "at matched ATE ... never demonstrated on a
pose trajectory").  No dataset exists offline, so the trajectory is a synthetic one with exact ground truth: a fronto-parallel textured
plane at the depth a rectified stereo pair with baseline B sees at DISP pixels of disparity, a pinhole camera translating parallel to
it and rolling about its optical axis (batch.SyntheticSequence renders exactly that: view v = texture under Rz(th_v), offset o_v, so
T_wc(v) = [Rz(th_v) | (o_v Z0 / f, 0)]).  A minimal stereo visual odometry in the reference's schedule runs on it, every per-frame
arithmetic step of it one of SURVEY.md section 8's functions:
    per frame      VisualFrontEnd::preprocessImage + kltTracking (visual_front_end.cpp:132-275), then the pose by
                   MultiViewGeometry::ceresPnP on the tracked map points (multi_view_geometry.cpp:492-586, as computePose calls it);
    every 5th      keyframe: detectSingleScale tops the keypoints up (feature_extractor.cpp:288-440), stereoMatching on the right image
                   (map_manager.cpp:367-611), new map points = depth from the measured disparity, placed with the ESTIMATED pose
                   (so errors accumulate the way they do in the reference).
Run once with the product (libov2slam_hip.so) and once with the oracle (CPU restatement), each carrying its own state; reported: the
absolute trajectory error of both (RMSE of the camera centres after a rigid alignment to the ground truth), and how far the two
estimated trajectories are apart.  The glue (map-point bookkeeping, alignment) is this script; nothing of it is product code.
Usage: ate_synthetic.py [n_frames] [n_views] [out.json]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ov2slam_amd
from ov2slam_amd import batch, stereo, optimizer
from oracle import oracle as O

F, B, DISP = 458.654, 0.11, 20.0
Z0 = F * B / DISP
CLIP, WIN, LEVELS, CELL, NKPS, KF_EVERY = 3.0, 9, 3, 35, 308, 5


def gt_pose(seq, f):
    ox, oy, th = seq.offs[seq.view_index(f)]
    c, s = np.cos(th), np.sin(th)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    return R, np.array([ox * Z0 / F, oy * Z0 / F, 0.0])


def quat_from_R(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])


def R_from_quat(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class GpuBackend:
    name = "gpu"

    def __init__(self, seq, K):
        self.seq, self.K = seq, K
        self.ctx = ov2slam_amd.Context(0)
        w, h = seq.w, seq.h
        self.trk = ov2slam_amd.VisualFrontEndTracker(self.ctx, w, h, use_clahe=True, fclahe_val=CLIP, nbmaxkps=2 * NKPS, use_graph=True)
        self.cal = ov2slam_amd.CameraCalibration(self.ctx, "pinhole", *K, D=(0.0, 0.0, 0.0, 0.0))
        self.trk.setCalibration(self.cal)
        self.fx = ov2slam_amd.FeatureExtractor(self.ctx, dmaxquality=0.001)
        self.ftrk = ov2slam_amd.FeatureTracker(self.ctx, 30, 0.01)
        self.pyrR = ov2slam_amd.Pyramid(self.ctx, w, h, WIN, LEVELS)
        self.mvg = optimizer.MultiViewGeometry(self.ctx)
        self.opt = optimizer.Optimizer(self.ctx)

    def track(self, f, kps, pri, hp):
        out, sb, _ = self.trk.trackFrame(self.seq.frame(f), kps, pri, hp)
        unpx, _ = self.trk.lastKeypoints(len(out)) if len(out) else (np.zeros((0, 2), np.float32), None)
        return out, (sb & 1).astype(bool), unpx

    def detect(self, f, kps, roi):
        return self.fx.detectSingleScalePyr(self.trk.cur_pyr, CELL, kps, roi)

    def stereo(self, f, kps):
        unpx, _ = self.cal.computeKeypoints(kps, want_bv=True)
        self.pyrR.build_clahe(self.seq.right_frame(f), CLIP, self.seq.w // 50, self.seq.h // 50)
        z = np.zeros(len(kps), np.uint8)
        ok, right = stereo.stereo_match_arrays(self.ftrk, self.trk.cur_pyr, self.pyrR, kps, unpx, kps, z, self.cal, rect=True)
        return ok, right, unpx

    def pnp(self, unpx, wpts, Twc):
        return self.mvg.ceresPnP(unpx, wpts, np.zeros(len(unpx)), Twc, 5, 5.9915, True, True, *self.K)

    def local_ba(self, prob):
        return self.opt.localBA(prob)

    def close(self):
        self.trk.close(); self.pyrR.close()


class OracleBackend:
    name = "oracle"

    def __init__(self, seq, K):
        self.seq, self.K = seq, K
        self.cache, self.q = {}, 0.001
        self.iK = np.linalg.inv(np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1.0]]))
        self.D = (0.0, 0.0, 0.0, 0.0)

        def solver(prob, res_active, chi2_init, depthpos_init, **kw):
            return O.ba_solve(prob, O.ba_default_options(**kw), res_active, chi2_init, depthpos_init)
        self.mvg = optimizer.MultiViewGeometry(None, solver=solver)
        self.opt = optimizer.Optimizer(None, solver=solver)

    def pre(self, f, right=False):
        key = (self.seq.view_index(f), right)
        if key not in self.cache:
            img = O.clahe(self.seq.right_frame(f) if right else self.seq.frame(f), CLIP, self.seq.w // 50, self.seq.h // 50)
            self.cache[key] = (img, O.Pyramid(img, WIN, LEVELS))
        return self.cache[key]

    def track(self, f, kps, pri, hp):
        if f == 0 or len(kps) == 0:
            self.pre(f)
            return np.zeros((0, 2), np.float32), np.zeros(0, bool), np.zeros((0, 2), np.float32)
        out, ok, _, _ = O.klt_tracking(self.pre(f - 1)[1], self.pre(f)[1], kps, pri, hp)
        unpx, _ = O.compute_keypoints(O.CAM_PINHOLE, self.K, self.D, self.iK, out)
        return out, ok, unpx

    def detect(self, f, kps, roi):
        new, self.q = O.detect_singlescale(self.pre(f)[0], CELL, kps, roi, self.q, True)
        return new

    def stereo(self, f, kps):
        unpx, _ = O.compute_keypoints(O.CAM_PINHOLE, self.K, self.D, self.iK, kps)
        ok, right = O.stereo_matching(self.pre(f)[1], self.pre(f, right=True)[1], kps, unpx, O.CAM_PINHOLE, self.K, self.D, True)
        return ok, right, unpx

    def pnp(self, unpx, wpts, Twc):
        return self.mvg.ceresPnP(unpx, wpts, np.zeros(len(unpx)), Twc, 5, 5.9915, True, True, *self.K)

    def local_ba(self, prob):
        r = self.opt.localBA(prob)
        its = (r["pass1"]["iterations"], r["pass2"]["iterations"] if r["l2_done"] else 0)
        return dict(poses=r["poses"], invdepth=r["invdepth"], bad_obs=r["bad_obs"], iterations=its)

    def close(self):
        pass


def run(backend, seq, n_frames, with_ba=False, window=8):
    """the odometry loop; returns (poses (n,7) [t, q], counters).  with_ba: every keyframe also runs Optimizer::localBA (both passes, outlier
    removal) over the last `window` keyframes -- inverse-depth landmarks anchored in their first keyframe, left / right / right-at-anchor
    blocks, the oldest keyframe of the window constant -- and the odometry continues from the refined poses and depths."""
    w, h = seq.w, seq.h
    K = backend.K
    roi = (5, 5, w - 10, h - 10)
    empty = np.zeros((0, 2), np.float32)
    R0, t0 = gt_pose(seq, 0)
    Twc = np.concatenate([t0, quat_from_R(R0)])                       # the first pose is given (the reference starts at identity)
    poses = [Twc.copy()]
    kps, lm = empty, np.zeros(0, np.int64)                             # current keypoints and their landmark ids (-1: none)
    L_akf, L_auv, L_lam = [], [], []                                   # landmarks: anchor keyframe, anchor pixel, inverse depth in the anchor frame
    kf_pose, kf_obs = [], []                                           # keyframes: pose, [(landmark, left uv, right uv or None)]
    cnt = dict(tracked=0, attempted=0, pnp_points=0, pnp_outliers=0, stereo_ok=0, keyframes=0, pnp_failed=0, depth_gated=0,
               ba_solves=0, ba_iterations=0, ba_blocks=0, ba_bad_obs=0)

    def world_points(ids):
        out = np.zeros((len(ids), 3))
        for n, i in enumerate(ids):
            P = kf_pose[L_akf[i]]
            z = 1.0 / L_lam[i]
            pc = np.array([(L_auv[i][0] - K[2]) * z / K[0], (L_auv[i][1] - K[3]) * z / K[1], z])
            out[n] = R_from_quat(P[3:]) @ pc + P[:3]
        return out

    def local_ba(k):
        k0 = max(0, k - window + 1)
        if k == k0:
            return
        # the window's keyframes are optimised; the anchor keyframes of the map points they see enter as CONSTANT keyframes when they are
        # older than the window (optimizer.cpp:128-407 walks the covisible map the same way: what is outside the window is fixed)
        lms = sorted({o[0] for j in range(k0, k + 1) for o in kf_obs[j]})
        kfs = sorted(set(range(k0, k + 1)) | {L_akf[g] for g in lms})
        kidx = {j: n for n, j in enumerate(kfs)}
        idx = {g: n for n, g in enumerate(lms)}
        rt, rk, rl, ruv, where = [], [], [], [], []
        for j in kfs:
            for n, (g, uvl, uvr) in enumerate(kf_obs[j]):
                if g not in idx:
                    continue
                if j == L_akf[g]:
                    if uvr is not None:
                        rt.append(2); rk.append(kidx[j]); rl.append(idx[g]); ruv.append(uvr); where.append((j, n, 1))
                elif j >= k0:
                    rt.append(0); rk.append(kidx[j]); rl.append(idx[g]); ruv.append(uvl); where.append((j, n, 0))
                    if uvr is not None:
                        rt.append(1); rk.append(kidx[j]); rl.append(idx[g]); ruv.append(uvr); where.append((j, n, 1))
        if not rt:
            return
        kc = np.array([1 if j < k0 else 0 for j in kfs], np.uint8)
        if not kc.any():
            kc[0] = 1
        prob = dict(n_kf=len(kfs), n_lm=len(lms), n_res=len(rt), poses=np.array([kf_pose[j] for j in kfs]), kf_const=kc,
                    invdepth=np.array([L_lam[g] for g in lms]), lm_anchor_kf=np.array([kidx[L_akf[g]] for g in lms], np.int32),
                    lm_anchor_uv=np.array([L_auv[g] for g in lms], np.float64), res_type=np.array(rt, np.uint8), res_kf=np.array(rk, np.int32),
                    res_lm=np.array(rl, np.int32), res_uv=np.array(ruv, np.float64), res_sigma=np.ones(len(rt)),
                    calib_l=np.array(K, np.float64), calib_r=np.array(K, np.float64), T_rl=np.array([-B, 0, 0, 0, 0, 0, 1.0]))
        r = backend.local_ba(prob)
        cnt["ba_solves"] += 1; cnt["ba_blocks"] += len(rt); cnt["ba_bad_obs"] += int(r["bad_obs"].sum())
        cnt["ba_iterations"] += int(r["iterations"][0]) + int(r["iterations"][1])
        for j in kfs:
            if j >= k0:
                kf_pose[j] = np.asarray(r["poses"][kidx[j]], np.float64).copy()
        for n, g in enumerate(lms):
            if r["invdepth"][n] > 0:
                L_lam[g] = float(r["invdepth"][n])
        # outlier blocks leave the map (optimizer.cpp:741-883 removes the observation); a left block of the newest keyframe takes the
        # current keypoint's landmark with it
        drop = {}
        for b in np.nonzero(r["bad_obs"])[0]:
            j, n, side = where[b]
            drop.setdefault(j, {})[n] = max(drop.get(j, {}).get(n, 0), 1 if side == 0 else 0)
        for j, d in drop.items():
            keep = []
            for n, o in enumerate(kf_obs[j]):
                if n in d:
                    if d[n]:                                           # the left observation is bad: the whole observation goes
                        if j == k:
                            lm[lm == o[0]] = -1
                        continue
                    o = (o[0], o[1], None)                             # only the right one
                keep.append(o)
            kf_obs[j] = keep

    def keyframe(f, kps, lm, Twc):
        new = backend.detect(f, kps, roi)[:max(0, NKPS - len(kps))]
        kps = np.concatenate([kps, new]); lm = np.concatenate([lm, np.full(len(new), -1, np.int64)])
        ok, right, unpx = backend.stereo(f, kps)
        cnt["stereo_ok"] += int(ok.sum()); cnt["keyframes"] += 1
        k = len(kf_pose)
        kf_pose.append(Twc.copy()); obs = []
        for i in range(len(kps)):
            uvl = (float(unpx[i, 0]), float(unpx[i, 1]))
            uvr = (float(right[i, 0]), float(right[i, 1])) if ok[i] else None
            if lm[i] >= 0:
                obs.append((int(lm[i]), uvl, uvr))
            elif ok[i]:
                # a new map point: depth from the MEASURED disparity (rectified pair), anchored in this keyframe (placed with its ESTIMATED pose)
                d = uvl[0] - uvr[0]
                if d <= 1.0:
                    continue
                z = K[0] * B / d
                if not 0.5 <= z <= 40.0:                               # a rectified pair makes ANY disparity geometrically consistent: a wrong match along the
                    cnt["depth_gated"] += 1                             # line survives stereoMatching's gates and must be caught by the triangulation's plausibility
                    continue                                           # check (one such point at 8 cm depth bends every PnP it takes part in)
                lm[i] = len(L_akf)
                L_akf.append(k); L_auv.append(uvl); L_lam.append(1.0 / z)
                obs.append((int(lm[i]), uvl, uvr))
        kf_obs.append(obs)
        return kps, lm

    backend.track(0, empty, empty, None)
    kps, lm = keyframe(0, kps, lm, Twc)
    for f in range(1, n_frames):
        hp = np.zeros(len(kps), np.uint8)                              # constant-position priors: nothing pose-dependent enters the tracker
        out, ok, unpx = backend.track(f, kps, kps.copy(), hp)
        cnt["attempted"] += len(kps); cnt["tracked"] += int(ok.sum())
        inside = ok & (out[:, 0] > 8) & (out[:, 0] < w - 9) & (out[:, 1] > 8) & (out[:, 1] < h - 9)
        kps, unpx, lm = out[inside], unpx[inside], lm[inside]
        m = np.nonzero(lm >= 0)[0]
        if len(m) >= 6:
            success, pose, vout = backend.pnp(unpx[m].astype(np.float64), world_points(lm[m]), Twc)
            cnt["pnp_points"] += len(m); cnt["pnp_outliers"] += len(vout)
            if success:
                Twc = np.asarray(pose, np.float64).copy()
                lm[m[vout]] = -1                                       # the reference removes the outliers' observations (visual_front_end.cpp:812-840)
            else:
                cnt["pnp_failed"] += 1
        else:
            cnt["pnp_failed"] += 1
        if f % KF_EVERY == 0:
            kps, lm = keyframe(f, kps, lm, Twc)
            if with_ba:
                local_ba(len(kf_pose) - 1)
                Twc = kf_pose[-1].copy()
        poses.append(Twc.copy())
    return np.array(poses), cnt


def ate(est_t, gt_t):
    """RMSE of the camera centres after the rigid alignment (Horn / Umeyama without scale) of est to gt"""
    a, b = est_t - est_t.mean(0), gt_t - gt_t.mean(0)
    U, _, Vt = np.linalg.svd(a.T @ b)
    S = np.diag([1, 1, np.sign(np.linalg.det(U @ Vt))])
    R = (U @ S @ Vt).T
    al = (R @ a.T).T + gt_t.mean(0)
    return float(np.sqrt(((al - gt_t) ** 2).sum(1).mean())), float(np.sqrt(((est_t - gt_t) ** 2).sum(1).mean()))


def main(n_frames=200, n_views=50, out_path=None, backends=("gpu", "oracle"), modes=("pnp", "pnp_and_local_ba")):
    seq = batch.SyntheticSequence("ATE", n_frames, seed=91, n_views=n_views, stereo=True, disparity=DISP)
    K = (F, F, (seq.w - 1) / 2.0, (seq.h - 1) / 2.0)
    gt = np.array([gt_pose(seq, f)[1] for f in range(n_frames)])
    res = dict(what="synthetic stereo visual odometry on a fronto-parallel plane (tools/ate_synthetic.py): per frame preprocessImage + kltTracking + ceresPnP, "
                    "every 5th frame detectSingleScale + stereoMatching + new map points anchored with the estimated pose; mode pnp_and_local_ba: plus "
                    "Optimizer::localBA (both passes) over the last 8 keyframes at every keyframe",
               frames=n_frames, distinct_views=n_views, plane_depth_m=Z0, path_length_m=float(np.linalg.norm(np.diff(gt, axis=0), axis=1).sum()),
               trajectory_extent_m=float(np.linalg.norm(gt.max(0) - gt.min(0))))
    trajs = {}
    same = ("tracked", "attempted", "pnp_points", "pnp_outliers", "stereo_ok", "keyframes", "pnp_failed", "depth_gated", "ba_solves", "ba_iterations", "ba_blocks", "ba_bad_obs")
    for mode in modes:
        out, traj = {}, {}
        for name in backends:
            be = GpuBackend(seq, K) if name == "gpu" else OracleBackend(seq, K)
            t = time.time()
            poses, cnt = run(be, seq, n_frames, with_ba=(mode == "pnp_and_local_ba"))
            be.close()
            a_al, a_raw = ate(poses[:, :3], gt)
            out[name] = dict(ate_rmse_m=a_al, ate_rmse_unaligned_m=a_raw, seconds=time.time() - t, final_position_error_m=float(np.linalg.norm(poses[-1, :3] - gt[-1])), **cnt)
            traj[name] = poses
        if len(traj) == 2:
            g, o = traj["gpu"], traj["oracle"]
            qd = np.minimum(np.abs(g[:, 3:] - o[:, 3:]).max(1), np.abs(g[:, 3:] + o[:, 3:]).max(1))
            out["gpu_vs_oracle"] = dict(max_position_difference_m=float(np.abs(g[:, :3] - o[:, :3]).max()), max_quaternion_difference=float(qd.max()),
                                        ate_difference_m=abs(out["gpu"]["ate_rmse_m"] - out["oracle"]["ate_rmse_m"]),
                                        same_counters=all(out["gpu"][k] == out["oracle"][k] for k in same))
        res[mode] = out; trajs[mode] = traj
    print(json.dumps(res))
    if out_path:
        open(out_path, "w").write(json.dumps(res, indent=1) + "\n")
    return res, trajs


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 50, sys.argv[3] if len(sys.argv) > 3 else None)
