"""GPU parity tests (through the C ABI) of the HIP LM solver against the oracle.
Bar (BASELINE.json north_star): BA pose outputs within 1e-4 relative on identical inputs; here the
fp64 device solver is required to agree far tighter (1e-7), with identical iteration counts,
termination reasons and outlier sets."""
import numpy as np
import pytest

import ov2slam_amd
from ov2slam_amd import synth, optimizer

pytestmark = pytest.mark.gpu

POSE_RTOL = 1e-4          # the bar of BASELINE.json
TIGHT = 1e-7              # what fp64 on both sides actually delivers


def _cmp(g, r, pb, tight=TIGHT):
    assert g["iterations"] == r["iterations"] and g["termination"] == r["termination"]
    assert g["num_successful_steps"] == r["num_successful_steps"]
    assert abs(g["initial_cost"] - r["initial_cost"]) <= 1e-10 * abs(r["initial_cost"]) + 1e-12
    assert abs(g["final_cost"] - r["final_cost"]) <= 1e-8 * abs(r["final_cost"]) + 1e-12
    scale = np.abs(r["poses"][:, :3]).max()
    assert np.abs(g["poses"][:, :3] - r["poses"][:, :3]).max() <= POSE_RTOL * scale
    assert np.abs(g["poses"][:, :3] - r["poses"][:, :3]).max() <= tight * scale
    qg = g["poses"][:, 3:] * np.sign((g["poses"][:, 3:] * r["poses"][:, 3:]).sum(1))[:, None]
    assert np.abs(qg - r["poses"][:, 3:]).max() <= tight
    assert np.allclose(g["invdepth"], r["invdepth"], rtol=1e-6, atol=1e-12)
    m = np.isfinite(r["chi2"])
    assert np.array_equal(np.isfinite(g["chi2"]), m)
    assert np.allclose(g["chi2"][m], r["chi2"][m], rtol=1e-6, atol=1e-9)
    assert np.array_equal(g["depthpos"], r["depthpos"])
    assert np.array_equal(g["chi2"][m] > 5.9915, r["chi2"][m] > 5.9915)       # identical outlier sets


@pytest.mark.parametrize("n_kf,n_lm,obs,stereo,seed", [(6, 40, 4, False, 1), (12, 400, 8, False, 2), (12, 400, 8, True, 3),
                                                      (25, 3000, 12, True, 4), (50, 2000, 30, False, 5)])
def test_single_solve_matches_oracle(gpu_ctx, oracle, n_kf, n_lm, obs, stereo, seed):
    pb = synth.make_ba_problem(n_kf, n_lm, obs, stereo=stereo, seed=seed)
    for kw in (dict(), dict(max_iter=10, huber_delta=-1.0), dict(max_iter=12, function_tolerance=1e-9)):
        g = optimizer.solve(gpu_ctx, pb, optimizer.default_options(gpu_ctx.lib, **kw))
        r = oracle.ba_solve(pb, oracle.ba_default_options(**kw))
        _cmp(g, r, pb)
        assert g["final_cost"] < g["initial_cost"]


def test_localba_protocol_matches_oracle(gpu_ctx, oracle):
    """Optimizer.localBA -- ONE ov2_local_ba call: robust pass -> outlier test + block removal on the device -> L2 pass -> second
    test, the problem resident in HBM throughout -- against the same protocol run as two solver calls on the oracle, and against
    the two-call form on the GPU (two ov2_ba_solve calls, host-side removal)."""
    def oracle_solver(prob, res_active, chi2_init, depthpos_init, **kw):
        return oracle.ba_solve(prob, oracle.ba_default_options(**kw), res_active, chi2_init, depthpos_init)
    for stereo in (True, False):
        pb = synth.make_ba_problem(15, 800, 8, stereo=stereo, seed=7)
        g = ov2slam_amd.Optimizer(gpu_ctx).localBA(pb)
        g2 = ov2slam_amd.Optimizer(gpu_ctx).localBA_two_calls(pb)
        r = ov2slam_amd.Optimizer(None, solver=oracle_solver).localBA(pb)
        assert g["l2_done"] and r["l2_done"] and g2["l2_done"]
        assert np.array_equal(g["bad_after_pass1"], r["bad_after_pass1"])
        assert np.array_equal(g["bad_obs"], r["bad_obs"]) and np.array_equal(g2["bad_obs"], r["bad_obs"])
        _cmp(g2["pass1"], r["pass1"], pb)
        _cmp(g2["pass2"], r["pass2"], pb)
        assert g["iterations"] == (r["pass1"]["iterations"], r["pass2"]["iterations"])
        assert g["termination"] == (r["pass1"]["termination"], r["pass2"]["termination"])
        # the final state: pass 2's result; chi2 / depth: pass-2 values for the blocks still in the problem, the values cached by
        # pass 1 for the removed ones (N4)
        _cmp(dict(g, iterations=g["iterations"][1], termination=g["termination"][1], final_cost=g["final_cost"][1],
                  initial_cost=g["initial_cost"][1], num_successful_steps=g["num_successful_steps"][1]), r["pass2"], pb)
        assert np.allclose(g["poses"], r["poses"], rtol=0, atol=1e-7 * max(1.0, np.abs(r["poses"]).max()))
        # injected gross outliers are (almost) all caught
        assert g["bad_obs"][pb["is_outlier"]].mean() > 0.9
    # the protocol's switches: no robust cost -> one pass; apply_l2_after_robust off -> verdicts of the first test only; stop flag
    pb = synth.make_ba_problem(12, 400, 8, stereo=True, seed=3)
    for kw, robust, stop in ((dict(), False, False), (dict(apply_l2_after_robust=False), True, False), (dict(), True, True)):
        og, orr = ov2slam_amd.Optimizer(gpu_ctx, **kw), ov2slam_amd.Optimizer(None, solver=oracle_solver, **kw)
        if stop:
            og.signalStopLocalBA(); orr.signalStopLocalBA()
        g, r = og.localBA(pb, robust), orr.localBA(pb, robust)
        assert not g["l2_done"] and not r["l2_done"]
        assert np.array_equal(g["bad_obs"], r["bad_obs"]) and g["iterations"][0] == r["pass1"]["iterations"]
        assert np.allclose(g["poses"], r["poses"], rtol=0, atol=1e-7 * max(1.0, np.abs(r["poses"]).max()))
    # a mono problem keeps Huber in the second pass (:606-608): exercised by stereo=False above; want_chi2=False skips the arrays
    g = ov2slam_amd.Optimizer(gpu_ctx).localBA(pb, want_chi2=False)
    assert "chi2" not in g and g["l2_done"]


def test_resident_problem_is_repeatable(gpu_ctx, oracle):
    pb = synth.make_ba_problem(12, 400, 8, stereo=True, seed=9)
    rp = optimizer.ResidentProblem(gpu_ctx, pb)
    a = rp.solve(); pa = a["poses"].copy(); ca = a["final_cost"]
    b = rp.solve()
    assert np.allclose(pa, b["poses"], atol=1e-12) and abs(ca - b["final_cost"]) <= 1e-9 * ca
    r = oracle.ba_solve(pb)
    _cmp(b, r, pb)
    rp.close()


def test_ba_edge_cases(gpu_ctx, oracle):
    pb = synth.make_ba_problem(4, 10, 3, seed=2)
    g = optimizer.solve(gpu_ctx, pb, optimizer.default_options(gpu_ctx.lib, max_iter=0))
    assert g["iterations"] == 0 and g["termination"] == 0 and np.allclose(g["poses"], pb["poses"])
    pbc = dict(pb); pbc["kf_const"] = np.ones(4, np.uint8)           # structure-only
    g = optimizer.solve(gpu_ctx, pbc); r = oracle.ba_solve(pbc)
    _cmp(g, r, pbc)
    none = np.zeros(pb["n_res"], np.uint8)                            # every residual block removed
    g = optimizer.solve(gpu_ctx, pb, res_active=none, chi2_init=np.arange(pb["n_res"], dtype=np.float64))
    assert g["initial_cost"] == 0.0 and np.array_equal(g["chi2"], np.arange(pb["n_res"], dtype=np.float64))
    # invalid input is rejected, not executed
    bad = dict(pb); bad["res_lm"] = pb["res_lm"].copy(); bad["res_lm"][0] = 10**6
    with pytest.raises(ov2slam_amd.Ov2Error):
        optimizer.solve(gpu_ctx, bad)


def test_pnp_matches_oracle(gpu_ctx, oracle):
    """OV2_RES_PNP (ReprojectionErrorSE3, ceresPnP) through the device solver vs the oracle."""
    for n, seed in ((30, 1), (300, 2), (2000, 3)):
        pb = synth.make_pnp_problem(n, seed=seed)
        for kw in (dict(), dict(huber_delta=-1.0, max_iter=10), dict(max_iter=12, function_tolerance=1e-9)):
            g = optimizer.solve(gpu_ctx, pb, optimizer.default_options(gpu_ctx.lib, **kw))
            r = oracle.ba_solve(pb, oracle.ba_default_options(**kw))
            _cmp(g, r, pb)


def test_ceres_pnp_protocol_matches_oracle(gpu_ctx, oracle):
    def oracle_solver(prob, res_active, chi2_init, depthpos_init, **kw):
        return oracle.ba_solve(prob, oracle.ba_default_options(**kw), res_active, chi2_init, depthpos_init)
    pb = synth.make_pnp_problem(300, seed=9)
    K = pb["calib_l"]
    args = (pb["res_uv"], pb["res_xyz"], np.zeros(300), pb["poses"][0], 5, 5.9915, True, True) + tuple(K)
    gok, gT, gout = ov2slam_amd.MultiViewGeometry(gpu_ctx).ceresPnP(*args)
    rok, rT, rout = ov2slam_amd.MultiViewGeometry(None, solver=oracle_solver).ceresPnP(*args)
    assert gok == rok and np.array_equal(gout, rout)
    assert np.abs(gT[:3] - rT[:3]).max() <= 1e-7 * max(1.0, np.abs(rT[:3]).max())
    assert np.abs(gT[3:] * np.sign(gT[3:] @ rT[3:]) - rT[3:]).max() <= 1e-7


def test_mixed_problem_with_pose_only_blocks(gpu_ctx, oracle):
    """Landmark factors and pose-only factors in one problem (rows with and without an e-block)."""
    pb = synth.make_ba_problem(8, 200, 5, stereo=True, seed=12)
    pn = synth.make_pnp_problem(60, seed=4)
    n0, n1 = pb["n_res"], pn["n_res"]
    # attach the PnP observations to keyframe 3 of the BA problem (world points expressed for that pose)
    from tests.test_oracle_ba import np_T
    T3 = np_T(pb["poses_gt"][3]) @ np.linalg.inv(np_T(pn["poses_gt"][0]))
    X = (T3[:3, :3] @ pn["res_xyz"].T).T + T3[:3, 3]
    mix = dict(pb)
    mix["n_res"] = n0 + n1
    mix["res_type"] = np.concatenate([pb["res_type"], pn["res_type"]])
    mix["res_kf"] = np.concatenate([pb["res_kf"], np.full(n1, 3, np.int32)])
    mix["res_lm"] = np.concatenate([pb["res_lm"], pn["res_lm"]])
    mix["res_uv"] = np.concatenate([pb["res_uv"], pn["res_uv"]])
    mix["res_sigma"] = np.concatenate([pb["res_sigma"], pn["res_sigma"]])
    mix["res_xyz"] = np.concatenate([np.zeros((n0, 3)), X])
    g = optimizer.solve(gpu_ctx, mix); r = oracle.ba_solve(mix)
    _cmp(g, r, mix)


def test_looseba_fullba_protocols_match_oracle(gpu_ctx, oracle):
    """Optimizer::looseBA (5 it, ftol 1e-4, one pass) and fullBA (100 it, Ceres default tolerances, optional L2 pass) on
    the device vs the identical protocol on the oracle: same iteration counts / terminations, poses within 1e-6."""
    def oracle_solver(prob, res_active, chi2_init, depthpos_init, **kw):
        return oracle.ba_solve(prob, oracle.ba_default_options(**kw), res_active, chi2_init, depthpos_init)

    for seed, stereo in ((3, True), (4, False)):
        pb = synth.make_ba_problem(n_kf=12, n_lm=400, obs_per_lm=6, stereo=stereo, outlier_frac=0.04, seed=seed,
                                    pose_noise=(0.15, np.deg2rad(3.0)), invdepth_noise=0.3)
        for name in ("looseBA", "fullBA"):
            g = getattr(ov2slam_amd.Optimizer(gpu_ctx), name)(pb)
            r = getattr(ov2slam_amd.Optimizer(None, solver=oracle_solver), name)(pb)
            assert g["pass1"]["iterations"] == r["pass1"]["iterations"], (name, seed)
            assert g["pass1"]["termination"] == r["pass1"]["termination"]
            assert np.array_equal(g["bad_obs"], r["bad_obs"]), (name, seed)
            assert g.get("l2_done", False) == r.get("l2_done", False)
            if g.get("l2_done"):
                assert g["pass2"]["iterations"] == r["pass2"]["iterations"]
            scale = max(1.0, np.abs(r["poses"]).max())
            assert np.abs(g["poses"] - r["poses"]).max() <= 1e-6 * scale, (name, seed, np.abs(g["poses"] - r["poses"]).max())
            assert np.abs(g["invdepth"] - r["invdepth"]).max() <= 1e-6 * max(1.0, np.abs(r["invdepth"]).max())
        assert g["pass1"]["iterations"] > 5, g["pass1"]["iterations"]       # fullBA (last in the loop): the long run is exercised


@pytest.mark.parametrize("n_kf,n_pts,obs,stereo,seed", [(6, 30, 3, False, 1), (12, 500, 5, True, 2), (20, 3000, 8, True, 3), (10, 1500, 6, False, 4)])
def test_structure_only_ba_matches_oracle(gpu_ctx, oracle, n_kf, n_pts, obs, stereo, seed):
    """Optimizer::structureOnlyBA (3-D points, constant poses) on the device vs the oracle: same LM trajectory."""
    pb = synth.make_structure_problem(n_kf, n_pts, obs, stereo=stereo, seed=seed)
    for kw in (dict(max_iter=10, function_tolerance=1e-3, huber_delta=float(np.sqrt(5.9915))),
               dict(max_iter=30, function_tolerance=1e-9, huber_delta=-1.0)):
        g = optimizer.structure_only_ba(gpu_ctx, pb, optimizer.default_options(gpu_ctx.lib, **kw))
        r = oracle.structure_ba(pb, oracle.ba_default_options(**kw))
        assert g["iterations"] == r["iterations"] and g["termination"] == r["termination"], (g["iterations"], r["iterations"])
        assert g["num_successful_steps"] == r["num_successful_steps"]
        assert abs(g["initial_cost"] - r["initial_cost"]) <= 1e-10 * abs(r["initial_cost"])
        assert abs(g["final_cost"] - r["final_cost"]) <= 1e-8 * abs(r["final_cost"])
        assert np.abs(g["xyz"] - r["xyz"]).max() <= 1e-7 * max(1.0, np.abs(r["xyz"]).max())
        assert np.allclose(g["chi2"], r["chi2"], rtol=1e-6, atol=1e-9) and np.array_equal(g["depthpos"], r["depthpos"])
    o = ov2slam_amd.Optimizer(gpu_ctx).structureOnlyBA(pb)
    assert o["final_cost"] < o["initial_cost"]


def test_structure_only_ba_edge_cases(gpu_ctx, oracle):
    pb = synth.make_structure_problem(8, 100, 4, seed=5)
    act = np.ones(pb["n_res"], np.uint8); act[pb["res_pt"] == 7] = 0; act[::9] = 0
    g = optimizer.structure_only_ba(gpu_ctx, pb, None, act); r = oracle.structure_ba(pb, None, act)
    assert g["iterations"] == r["iterations"] and np.abs(g["xyz"] - r["xyz"]).max() < 1e-8
    assert np.array_equal(g["xyz"][7], pb["xyz"][7]) and np.array_equal(np.isnan(g["chi2"]), np.isnan(r["chi2"]))
    e = dict(pb); e.update(n_res=0, res_type=np.zeros(0, np.uint8), res_kf=np.zeros(0, np.int32), res_pt=np.zeros(0, np.int32),
                           res_uv=np.zeros((0, 2)), res_sigma=np.zeros(0))
    g0 = optimizer.structure_only_ba(gpu_ctx, e)
    assert g0["iterations"] == 0 and np.array_equal(g0["xyz"], pb["xyz"])
    bad = dict(pb); bad["res_pt"] = pb["res_pt"].copy(); bad["res_pt"][3] = 10 ** 6
    with pytest.raises(ov2slam_amd.Ov2Error):
        optimizer.structure_only_ba(gpu_ctx, bad)


def test_config4_full_size_matches_oracle(gpu_ctx, oracle):
    """BASELINE.json configs[3] at its stated size: 50 KF x 10000 landmarks x 30 observations (290000 residual blocks),
    robust pass of optimizer.cpp:436-485 -- the problem bench.py times -- against the oracle (~0.5 s of CPU)."""
    pb = synth.make_ba_problem(50, 10000, 30, stereo=False, seed=42)
    assert pb["n_res"] == 290000
    g = optimizer.solve(gpu_ctx, pb)
    r = oracle.ba_solve(pb)
    _cmp(g, r, pb)
    assert g["final_cost"] < 0.5 * g["initial_cost"]


def test_config4_stereo_full_size_matches_oracle(gpu_ctx, oracle):
    """The stereo variant of BASELINE.json configs[3] at its stated size -- 290000 left + 290000 right + 10000 right-anchor
    residual blocks (590000), the problem bench.py times as `config3_stereo` and `localba_two_pass_stereo` -- robust pass
    against the oracle (~1 s of CPU): same iterations / termination / N4 outputs, poses within 1e-7."""
    pb = synth.make_ba_problem(50, 10000, 30, stereo=True, seed=42)
    assert pb["n_res"] == 590000
    g = optimizer.solve(gpu_ctx, pb)
    r = oracle.ba_solve(pb)
    _cmp(g, r, pb)
    assert g["final_cost"] < 0.5 * g["initial_cost"]


def test_long_budget_stops_enqueuing_after_convergence(gpu_ctx, oracle):
    """fullBA-style budget (100 iterations) on a problem that converges in a few: the chunked enqueue must return the same
    result as the oracle (and as a run whose budget equals the iterations actually needed)."""
    pb = synth.make_ba_problem(12, 400, 8, stereo=True, seed=3)
    kw = dict(max_iter=100, function_tolerance=1e-6)
    g = optimizer.solve(gpu_ctx, pb, optimizer.default_options(gpu_ctx.lib, **kw))
    r = oracle.ba_solve(pb, oracle.ba_default_options(**kw))
    _cmp(g, r, pb)
    assert g["iterations"] < 40
    g2 = optimizer.solve(gpu_ctx, pb, optimizer.default_options(gpu_ctx.lib, max_iter=g["iterations"], function_tolerance=1e-6))
    assert np.allclose(g2["poses"], g["poses"], atol=1e-12)


def test_large_problem_path_matches_oracle(gpu_ctx, oracle):
    """More optimised keyframes than the LDS-resident reduced system holds (~95): the sparse-W / HBM-Cholesky path (BADev::big)
    takes over -- a loop-closure fullBA is the realistic case.  It must follow the oracle like the small path does; it is also
    forced onto small problems (OV2_OPT_BA_FORCE_LARGE) so that both paths are compared on the same inputs."""
    pb = synth.make_ba_problem(120, 1500, 8, stereo=True, seed=1)
    g = optimizer.solve(gpu_ctx, pb, optimizer.default_options(gpu_ctx.lib, max_iter=6))
    r = oracle.ba_solve(pb, oracle.ba_default_options(max_iter=6))
    _cmp(g, r, pb)
    assert g["final_cost"] < 0.5 * g["initial_cost"]
    with gpu_ctx.options(ba_force_large=1):
        for n_kf, n_lm, obs, stereo, seed in ((6, 40, 4, False, 1), (12, 400, 8, True, 3), (40, 1200, 20, False, 5)):
            pb = synth.make_ba_problem(n_kf, n_lm, obs, stereo=stereo, seed=seed)
            for kw in (dict(), dict(max_iter=10, huber_delta=-1.0)):
                g = optimizer.solve(gpu_ctx, pb, optimizer.default_options(gpu_ctx.lib, **kw))
                r = oracle.ba_solve(pb, oracle.ba_default_options(**kw))
                _cmp(g, r, pb)
        pb = synth.make_ba_problem(15, 800, 8, stereo=True, seed=7)               # the whole localBA protocol on the big path
        g = ov2slam_amd.Optimizer(gpu_ctx).localBA(pb)
    s = ov2slam_amd.Optimizer(gpu_ctx).localBA(pb)
    assert np.array_equal(g["bad_obs"], s["bad_obs"]) and np.allclose(g["poses"], s["poses"], atol=1e-9)


def test_stop_flag_raised_during_pass_1_skips_the_l2_pass(gpu_ctx):
    """The reference evaluates !stopLocalBA() AFTER its first ceres::Solve (src/optimizer.cpp:603-604) and Estimator::addNewKf raises
    the flag from another thread while pass 1 runs.  ov2_local_ba reads the LIVE flag (ov2_local_ba_options::stop_flag) right before
    it decides on pass 2: a flag raised after the call has started (here: 0.3 ms into a ~5 ms call) must skip the L2 pass, and the
    adapter clears it when the call is over (:896)."""
    import threading
    import time
    pb = synth.make_ba_problem(50, 10000, 30, stereo=False, seed=11)
    opt = ov2slam_amd.Optimizer(gpu_ctx)
    base = opt.localBA(pb, want_chi2=False)
    assert base["l2_done"] and base["pass2_error"] == 0
    started = threading.Event()

    def raiser():
        started.wait()
        time.sleep(3e-4)
        opt.signalStopLocalBA()
    th = threading.Thread(target=raiser)
    th.start()
    started.set()
    g = opt.localBA(pb, want_chi2=False)
    th.join()
    assert not g["l2_done"] and g["iterations"][0] == base["iterations"][0]
    assert np.array_equal(g["bad_obs"], g["bad_after_pass1"])
    assert not opt.stopLocalBA()                                   # cleared at the end of localBA
    assert opt.localBA(pb, want_chi2=False)["l2_done"]              # and the next keyframe's localBA is unaffected


def _mixed_with_pnp(pb, pn, kf):
    """the landmark problem `pb` plus the pose-only blocks of `pn` attached to keyframe `kf`"""
    from tests.test_oracle_ba import np_T
    n0, n1 = pb["n_res"], pn["n_res"]
    T = np_T(pb["poses_gt"][kf]) @ np.linalg.inv(np_T(pn["poses_gt"][0]))
    X = (T[:3, :3] @ pn["res_xyz"].T).T + T[:3, 3]
    mix = dict(pb)
    mix["n_res"] = n0 + n1
    mix["res_type"] = np.concatenate([pb["res_type"], pn["res_type"]])
    mix["res_kf"] = np.concatenate([pb["res_kf"], np.full(n1, kf, np.int32)])
    mix["res_lm"] = np.concatenate([pb["res_lm"], pn["res_lm"]])
    mix["res_uv"] = np.concatenate([pb["res_uv"], pn["res_uv"]])
    mix["res_sigma"] = np.concatenate([pb["res_sigma"], pn["res_sigma"]])
    mix["res_xyz"] = np.concatenate([np.zeros((n0, 3)), X])
    return mix


def test_large_problem_path_variants(gpu_ctx, oracle):
    """Round 3 extensions of the large-problem path, each forced onto small problems so that the oracle stays cheap:
    pose-only (OV2_RES_PNP) blocks on the sparse-W path -- mixed with landmarks AND alone (round 4: the path's accumulators are
    cleared by k_ba_zero_lin whenever anything is linearised) --, the lineariser without LDS pre-aggregation of the observer blocks
    (what runs beyond ~570 optimised keyframes) and the column-chunked sparse Schur complement (beyond 341 keyframes)."""
    mix = _mixed_with_pnp(synth.make_ba_problem(8, 200, 5, stereo=True, seed=12), synth.make_pnp_problem(60, seed=4), 3)
    pnp = synth.make_pnp_problem(120, seed=9)
    cases = ((12, 400, 8, True, 3), (40, 1200, 20, False, 5))
    with gpu_ctx.options(ba_force_large=1):
        _cmp(optimizer.solve(gpu_ctx, mix), oracle.ba_solve(mix), mix)
        # pose-only blocks ALONE on the large path, through the multi-kernel loop (several re-linearisations of H and F^T b)
        with gpu_ctx.options(ba_pose_only_fused=0):
            for kw in (dict(), dict(max_iter=10, huber_delta=-1.0)):
                g = optimizer.solve(gpu_ctx, pnp, optimizer.default_options(gpu_ctx.lib, **kw))
                r = oracle.ba_solve(pnp, oracle.ba_default_options(**kw))
                assert g["iterations"] == r["iterations"] >= 2 and g["termination"] == r["termination"]
                assert np.allclose(g["poses"], r["poses"], rtol=0, atol=1e-7 * max(1.0, np.abs(r["poses"]).max()))
        for opts in (dict(ba_lin_direct=1), dict(ba_schur_chunk=36), dict(ba_lin_direct=1, ba_schur_chunk=96)):
            with gpu_ctx.options(**opts):
                for n_kf, n_lm, obs, stereo, seed in cases:
                    pb = synth.make_ba_problem(n_kf, n_lm, obs, stereo=stereo, seed=seed)
                    for kw in (dict(), dict(max_iter=10, huber_delta=-1.0)):
                        g = optimizer.solve(gpu_ctx, pb, optimizer.default_options(gpu_ctx.lib, **kw))
                        r = oracle.ba_solve(pb, oracle.ba_default_options(**kw))
                        _cmp(g, r, pb)
                _cmp(optimizer.solve(gpu_ctx, mix), oracle.ba_solve(mix), mix)
    assert gpu_ctx.get_option(ov2slam_amd._lib.OV2_OPT_BA_FORCE_LARGE) == 0


def test_more_than_341_keyframes_match_oracle(gpu_ctx, oracle):
    """A loop-closure / offline fullBA over 360 keyframes (359 optimised: reduced system 2154 x 2154, beyond the 2048 columns one
    LDS row block holds -- rounds 1-2 returned OV2_EUNSUPPORTED here): three LM iterations against the oracle (~4 s of CPU).
    Problems with pose-only blocks take the same path (no longer capped at ~90 keyframes)."""
    pb = synth.make_ba_problem(360, 3000, 8, stereo=True, seed=1)
    g = optimizer.solve(gpu_ctx, pb, optimizer.default_options(gpu_ctx.lib, max_iter=3))
    r = oracle.ba_solve(pb, oracle.ba_default_options(max_iter=3))
    _cmp(g, r, pb)
    assert g["final_cost"] < 0.5 * g["initial_cost"]
    mix = _mixed_with_pnp(synth.make_ba_problem(120, 900, 6, stereo=False, seed=2), synth.make_pnp_problem(80, seed=5), 60)
    _cmp(optimizer.solve(gpu_ctx, mix, optimizer.default_options(gpu_ctx.lib, max_iter=4)), oracle.ba_solve(mix, oracle.ba_default_options(max_iter=4)), mix)


def test_too_many_keyframes_is_reported_for_point_landmarks(gpu_ctx):
    """The 3-D-point parameterisation keeps W dense (3 rows per wavefront in LDS): beyond ~450 optimised keyframes
    ov2_xyz_ba_solve returns OV2_EUNSUPPORTED with a message, never a silent skip (up to there: tests/test_gpu_xyz_ba.py)."""
    pb = synth.make_xyz_ba_problem(500, 600, 4, stereo=False, seed=1)
    with pytest.raises(ov2slam_amd.Ov2Error) as e:
        optimizer.solve_xyz(gpu_ctx, pb)
    assert e.value.code == -4 and "keyframes" in str(e.value)


def test_max_solver_time_stops_between_chunks(gpu_ctx, oracle):
    """max_solver_time_s (Ceres max_solver_time_in_seconds, optimizer.cpp:464-468): an impossible budget stops the solve after
    the first chunk of iterations with NO_CONVERGENCE and a state that is one of the oracle's accepted iterates; no limit
    (the default) reproduces the oracle."""
    pb = synth.make_ba_problem(12, 400, 8, stereo=True, seed=3)
    kw = dict(max_iter=50, function_tolerance=1e-12)
    full = optimizer.solve(gpu_ctx, pb, optimizer.default_options(gpu_ctx.lib, **kw))
    cut = optimizer.solve(gpu_ctx, pb, optimizer.default_options(gpu_ctx.lib, max_solver_time_s=1e-9, **kw))
    assert cut["termination"] == 0 and 1 <= cut["iterations"] <= 4 and cut["iterations"] < full["iterations"]
    ref = oracle.ba_solve(pb, oracle.ba_default_options(max_iter=cut["iterations"], function_tolerance=1e-12))
    assert np.allclose(cut["poses"], ref["poses"], atol=1e-9) and abs(cut["final_cost"] - ref["final_cost"]) <= 1e-8 * ref["final_cost"]
    assert cut["final_cost"] <= cut["initial_cost"] and full["final_cost"] <= cut["final_cost"]




def test_deterministic_mode_is_bit_identical_from_run_to_run(gpu_ctx, oracle):
    """OV2_OPT_BA_DETERMINISTIC: every sum the default accumulates with fp64 atomics in arrival order goes through per-work-group
    copies added up in a fixed order -- two solves return the same BITS (poses, inverse depths, costs, chi2), the result matches the
    oracle like the default path's, ov2_local_ba's two passes included; forms the mode does not cover answer OV2_EUNSUPPORTED."""
    cases = [synth.make_ba_problem(12, 400, 8, stereo=False, seed=2), synth.make_ba_problem(25, 3000, 12, stereo=True, seed=4)]
    # landmark factors and pose-only factors in one problem
    pb = synth.make_ba_problem(8, 200, 5, stereo=True, seed=12)
    pn = synth.make_pnp_problem(60, seed=4)
    from tests.test_oracle_ba import np_T
    T3 = np_T(pb["poses_gt"][3]) @ np.linalg.inv(np_T(pn["poses_gt"][0]))
    mix = dict(pb)
    mix["n_res"] = pb["n_res"] + pn["n_res"]
    for k in ("res_type", "res_lm", "res_uv", "res_sigma"): mix[k] = np.concatenate([pb[k], pn[k]])
    mix["res_kf"] = np.concatenate([pb["res_kf"], np.full(pn["n_res"], 3, np.int32)])
    mix["res_xyz"] = np.concatenate([np.zeros((pb["n_res"], 3)), (T3[:3, :3] @ pn["res_xyz"].T).T + T3[:3, 3]])
    cases.append(mix)
    with gpu_ctx.options(ba_deterministic=1):
        for pb in cases:
            runs = [optimizer.solve(gpu_ctx, pb, optimizer.default_options(gpu_ctx.lib, max_iter=8)) for _ in range(4)]
            for g in runs[1:]:
                for k in ("poses", "invdepth", "chi2", "depthpos"):
                    assert np.array_equal(np.asarray(g[k]), np.asarray(runs[0][k])), k
                assert g["final_cost"] == runs[0]["final_cost"] and g["initial_cost"] == runs[0]["initial_cost"] and g["iterations"] == runs[0]["iterations"]
            _cmp(runs[0], oracle.ba_solve(pb, oracle.ba_default_options(max_iter=8)), pb)
        pb = synth.make_ba_problem(15, 800, 8, stereo=True, seed=7)
        la = [ov2slam_amd.Optimizer(gpu_ctx).localBA(pb) for _ in range(3)]
        for g in la[1:]:
            assert np.array_equal(g["bad_obs"], la[0]["bad_obs"])
            for k in ("poses", "invdepth"):
                assert np.array_equal(np.asarray(g[k]), np.asarray(la[0][k])), k
        # the 3-D point form is not covered: the call says so instead of answering with an unordered sum
        px = synth.make_xyz_ba_problem(6, 60, 4, stereo=False, seed=1) if hasattr(synth, "make_xyz_ba_problem") else None
        if px is not None:
            with pytest.raises(Exception):
                optimizer.solve_xyz(gpu_ctx, px)
    # and the default path on the same context is back to normal
    pb = cases[0]
    _cmp(optimizer.solve(gpu_ctx, pb), oracle.ba_solve(pb), pb)
