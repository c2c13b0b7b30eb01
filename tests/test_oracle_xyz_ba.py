"""CPU tests pinning the XYZ-landmark / variable-pose BA oracle (oracle/xyz_ba.c: the buse_inv_depth: 0 branch of
Optimizer::localBA, /root/reference/src/optimizer.cpp:207-209, :333-384; factors ceres_parametrization.cpp:107-298):
finite-difference Jacobians, equality with the structure-only solver when every pose is constant, and an independent
dense numpy Levenberg-Marquardt (numeric Jacobians, dense normal equations) that must reproduce its iterations."""
import numpy as np
import pytest

from ov2slam_amd import synth
from tests.test_oracle_ba import np_plus, np_T


def np_xyz_residual(rtype, K_l, K_r, T_rl, pose, X, uv, sigma=1.0):
    pc = np.linalg.inv(np_T(pose)) @ np.append(X, 1)
    K = K_l
    if rtype == 1:
        pc = np_T(T_rl) @ pc
        K = K_r
    return np.array([K[0] * pc[0] / pc[2] + K[2] - uv[0], K[1] * pc[1] / pc[2] + K[3] - uv[1]]) / sigma, pc[2] > 0


@pytest.mark.parametrize("rtype", [0, 1])
def test_xyz_residual_and_jacobians(oracle, rtype):
    pb = synth.make_xyz_ba_problem(6, 20, 4, stereo=True, seed=3)
    T_rl = np.array([-0.11, 0.01, 0.02, 0.01, -0.02, 0.005, 1.0]); T_rl[3:] /= np.linalg.norm(T_rl[3:])
    for i in np.nonzero(pb["res_type"] == rtype)[0][:8]:
        pose, X, uv, sg = pb["poses"][pb["res_kf"][i]], pb["xyz"][pb["res_pt"][i]], pb["res_uv"][i], pb["res_sigma"][i]
        r, Jp, Jx, chi2, dp = oracle.xyzba_residual(rtype, pb["calib_l"], pb["calib_r"], T_rl, pose, X, uv, sg)
        r_np, dp_np = np_xyz_residual(rtype, pb["calib_l"], pb["calib_r"], T_rl, pose, X, uv, sg)
        assert np.allclose(r, r_np, atol=1e-9) and dp == dp_np and abs(chi2 - r @ r) < 1e-12
        f = lambda p, x: np_xyz_residual(rtype, pb["calib_l"], pb["calib_r"], T_rl, p, x, uv, sg)[0]
        eps = 1e-6
        for c in range(6):
            d = np.zeros(6); d[c] = eps
            assert np.allclose(Jp[:, c], (f(np_plus(pose, d), X) - f(np_plus(pose, -d), X)) / (2 * eps), atol=2e-5, rtol=1e-5)
        for c in range(3):
            d = np.zeros(3); d[c] = eps
            assert np.allclose(Jx[:, c], (f(pose, X + d) - f(pose, X - d)) / (2 * eps), atol=2e-5, rtol=1e-5)


def test_constant_poses_reduce_to_structure_only_ba(oracle):
    pb = synth.make_xyz_ba_problem(8, 80, 4, stereo=True, seed=9)
    pb["kf_const"] = np.ones(8, np.uint8)
    for kw in (dict(max_iter=10, function_tolerance=1e-3), dict(max_iter=8, huber_delta=-1.0, function_tolerance=1e-9)):
        a = oracle.xyz_ba_solve(pb, oracle.ba_default_options(**kw))
        b = oracle.structure_ba(pb, oracle.ba_default_options(**kw))
        assert a["iterations"] == b["iterations"] and a["termination"] == b["termination"]
        assert np.allclose(a["xyz"], b["xyz"], rtol=1e-10, atol=1e-12) and np.array_equal(a["poses"], pb["poses"])
        assert abs(a["final_cost"] - b["final_cost"]) <= 1e-10 * b["final_cost"]


def dense_lm_xyz(pb, max_iter, huber, ftol):
    """Ceres' TR-LM with dense algebra and numeric Jacobians over [variable poses (6 each), points (3 each)]."""
    n_kf, n_pts, n_res = pb["n_kf"], pb["n_pts"], pb["n_res"]
    var = [k for k in range(n_kf) if not pb["kf_const"][k]]
    col = {k: 6 * i for i, k in enumerate(var)}
    nf = 6 * len(var)
    poses = pb["poses"].copy(); X = pb["xyz"].copy()

    def rho(s):
        if huber > 0 and s > huber * huber:
            r = np.sqrt(s)
            return 2 * huber * r - huber * huber, huber / r
        return s, 1.0

    def raw(poses, X):
        return np.array([np_xyz_residual(pb["res_type"][i], pb["calib_l"], pb["calib_r"], pb["T_rl"], poses[pb["res_kf"][i]],
                                         X[pb["res_pt"][i]], pb["res_uv"][i], pb["res_sigma"][i])[0] for i in range(n_res)])

    def cost_of(poses, X):
        return 0.5 * sum(rho(r @ r)[0] for r in raw(poses, X))

    def linearize(poses, X):
        R = raw(poses, X)
        w = np.array([np.sqrt(rho(r @ r)[1]) for r in R])
        J = np.zeros((2 * n_res, nf + 3 * n_pts))
        eps = 1e-6
        corr = lambda p2, x2: (raw(p2, x2) * w[:, None]).reshape(-1)
        for k in var:
            for c in range(6):
                d = np.zeros(6); d[c] = eps
                pp = poses.copy(); pm = poses.copy()
                pp[k] = np_plus(poses[k], d); pm[k] = np_plus(poses[k], -d)
                J[:, col[k] + c] = (corr(pp, X) - corr(pm, X)) / (2 * eps)
        for l in range(n_pts):
            for c in range(3):
                xp = X.copy(); xm = X.copy(); xp[l, c] += eps; xm[l, c] -= eps
                J[:, nf + 3 * l + c] = (corr(poses, xp) - corr(poses, xm)) / (2 * eps)
        return (R * w[:, None]).reshape(-1), J, 0.5 * sum(rho(r @ r)[0] for r in R)

    r, J, cost = linearize(poses, X)
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))
    J = J * scale
    radius, nu, reuse, diag = 1e4, 2.0, False, None
    costs = [cost]; iters = 0; x_norm = -1.0
    for it in range(max_iter):
        if not reuse:
            diag = np.clip((J * J).sum(0), 1e-6, 1e32)
        iters += 1
        H = J.T @ J + np.diag(diag / radius)
        y = -np.linalg.solve(H, J.T @ r)
        reuse = True
        Jy = J @ y
        mcc = -(Jy @ (r + Jy / 2))
        if not mcc > 0:
            radius /= nu; nu *= 2
            continue
        delta = y * scale
        pc = poses.copy()
        for k in var:
            pc[k] = np_plus(poses[k], delta[col[k]:col[k] + 6])
        Xc = X + delta[nf:].reshape(-1, 3)
        cand = cost_of(pc, Xc)
        if abs(cost - cand) <= ftol * cost:
            break
        rel = (cost - cand) / mcc
        if rel > 1e-3:
            poses, X = pc, Xc
            r, J, cost = linearize(poses, X)
            J = J * scale
            radius = min(1e16, radius / max(1.0 / 3.0, 1 - (2 * rel - 1) ** 3)); nu = 2.0; reuse = False
            costs.append(cost)
        else:
            radius /= nu; nu *= 2
    return poses, X, costs, iters


@pytest.mark.parametrize("stereo,huber", [(True, np.sqrt(5.9915)), (False, -1.0)])
def test_xyz_ba_matches_dense_numpy_lm(oracle, stereo, huber):
    pb = synth.make_xyz_ba_problem(5, 14, 4, stereo=stereo, seed=11, outlier_frac=0.05)
    o = oracle.xyz_ba_solve(pb, oracle.ba_default_options(max_iter=6, huber_delta=float(huber), function_tolerance=1e-6))
    poses, X, costs, iters = dense_lm_xyz(pb, 6, huber, 1e-6)
    assert o["iterations"] == iters
    assert abs(o["initial_cost"] - costs[0]) <= 1e-9 * costs[0]
    assert abs(o["final_cost"] - min(costs)) <= 1e-6 * min(costs)
    assert np.allclose(o["poses"][:, :3], poses[:, :3], atol=2e-6)
    assert np.allclose(o["xyz"], X, atol=2e-5)
    assert o["final_cost"] < 0.7 * o["initial_cost"]


def test_xyz_ba_recovers_ground_truth_and_respects_masks(oracle):
    pb = synth.make_xyz_ba_problem(10, 300, 5, stereo=True, seed=5, outlier_frac=0.0, px_noise=0.2)
    o = oracle.xyz_ba_solve(pb, oracle.ba_default_options(max_iter=20, function_tolerance=1e-9))
    e0 = np.abs(pb["poses"][:, :3] - pb["poses_gt"][:, :3]).max(); e1 = np.abs(o["poses"][:, :3] - pb["poses_gt"][:, :3]).max()
    assert e1 < 0.25 * e0 and np.array_equal(o["poses"][0], pb["poses"][0])          # constant keyframe untouched
    # inactive residual blocks keep the caller's chi2 / depth flags (N4) and do not influence the solve
    act = np.ones(pb["n_res"], np.uint8); act[::7] = 0
    chi0 = np.full(pb["n_res"], 123.0); dp0 = np.full(pb["n_res"], 1, np.uint8)
    a = oracle.xyz_ba_solve(pb, None, act, chi0, dp0)
    assert np.all(a["chi2"][act == 0] == 123.0) and np.all(a["chi2"][act == 1] != 123.0)
