"""GPU parity tests of ov2_local_ba_batch (the estimator side of the lock-step batch of sequences, BASELINE configs[4]): n local-BA
problems share every launch of the solver (grid.z = problem).  Per problem the result must be what ov2_local_ba returns for that
problem alone -- same protocol decisions, iteration counts, terminations and outlier sets; parameters within the BA bar (1e-7, the
sums over work-groups are grouped by the batch's grid) -- and what the oracle's two-call protocol returns."""
import numpy as np
import pytest

import ov2slam_amd
from ov2slam_amd import synth

pytestmark = pytest.mark.gpu


def _same(b, a, tight=1e-7):
    assert b["l2_done"] == a["l2_done"] and b["iterations"] == a["iterations"] and b["termination"] == a["termination"]
    assert b["num_successful_steps"] == a["num_successful_steps"]
    assert np.array_equal(b["bad_after_pass1"], a["bad_after_pass1"]) and np.array_equal(b["bad_obs"], a["bad_obs"])
    for q in range(2):
        assert abs(b["initial_cost"][q] - a["initial_cost"][q]) <= 1e-9 * abs(a["initial_cost"][q]) + 1e-12
        assert abs(b["final_cost"][q] - a["final_cost"][q]) <= 1e-8 * abs(a["final_cost"][q]) + 1e-12
    assert np.abs(b["poses"] - a["poses"]).max() <= tight * max(1.0, np.abs(a["poses"]).max())
    assert np.allclose(b["invdepth"], a["invdepth"], rtol=1e-6, atol=1e-12)
    if "chi2" in a:
        m = np.isfinite(a["chi2"])
        assert np.array_equal(np.isfinite(b["chi2"]), m)
        assert np.allclose(b["chi2"][m], a["chi2"][m], rtol=1e-6, atol=1e-9)
        assert np.array_equal(b["depthpos"], a["depthpos"])


def _problems():
    # different window sizes (optimised keyframes 10 .. 25, so different reduced-system sizes / tile counts), mono and stereo,
    # with and without injected outliers
    return [synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7), synth.make_ba_problem(12, 400, 8, stereo=True, seed=3),
            synth.make_ba_problem(15, 800, 8, stereo=False, seed=7), synth.make_ba_problem(20, 1500, 10, stereo=True, seed=11),
            synth.make_ba_problem(6, 40, 4, stereo=False, seed=1), synth.make_ba_problem(25, 3000, 12, stereo=True, seed=8)]


def test_batch_matches_single_problem_calls_and_oracle(gpu_ctx, oracle):
    pbs = _problems()
    opt = ov2slam_amd.Optimizer(gpu_ctx)
    res, nb = opt.localBA_batch(pbs)
    assert nb == len(pbs)
    def oracle_solver(prob, res_active, chi2_init, depthpos_init, **kw):
        return oracle.ba_solve(prob, oracle.ba_default_options(**kw), res_active, chi2_init, depthpos_init)
    for i, pb in enumerate(pbs):
        a = ov2slam_amd.Optimizer(gpu_ctx).localBA(pb)
        _same(res[i], a)
        if i in (1, 2):
            r = ov2slam_amd.Optimizer(None, solver=oracle_solver).localBA(pb)
            assert res[i]["l2_done"] == r["l2_done"] and np.array_equal(res[i]["bad_obs"], r["bad_obs"])
            assert res[i]["iterations"] == (r["pass1"]["iterations"], r["pass2"]["iterations"] if r["l2_done"] else 0)
            assert np.allclose(res[i]["poses"], r["poses"], rtol=0, atol=1e-7 * max(1.0, np.abs(r["poses"]).max()))
    # the same batch again on the warm blocks, and a batch of one
    res2, _ = opt.localBA_batch(pbs)
    for a, b in zip(res, res2):
        _same(b, a)
    one, nb = opt.localBA_batch(pbs[1:2])
    assert nb == 1
    _same(one[0], res[1])


def test_batch_protocol_switches(gpu_ctx):
    pbs = _problems()[:4]
    # a stop request per problem: those problems keep the result of pass 1 (optimizer.cpp:603-604), the others take pass 2
    stop = [False, True, False, True]
    res, nb = ov2slam_amd.Optimizer(gpu_ctx).localBA_batch(pbs, stop=stop)
    assert nb == 4
    for i, pb in enumerate(pbs):
        o = ov2slam_amd.Optimizer(gpu_ctx)
        if stop[i]:
            o.signalStopLocalBA()
        a = o.localBA(pb)
        assert res[i]["l2_done"] == (not stop[i]) or not a["l2_done"]
        _same(res[i], a)
    # no robust cost -> one pass for everybody; apply_l2_after_robust off -> verdicts of the first test only
    for kw, robust in ((dict(), False), (dict(apply_l2_after_robust=False), True)):
        res, _ = ov2slam_amd.Optimizer(gpu_ctx, **kw).localBA_batch(pbs, buse_robust_cost=robust, want_chi2=False)
        for i, pb in enumerate(pbs):
            a = ov2slam_amd.Optimizer(gpu_ctx, **kw).localBA(pb, robust, want_chi2=False)
            assert not res[i]["l2_done"]
            _same(res[i], a)


def test_batch_falls_back_for_problems_it_does_not_cover(gpu_ctx):
    small = synth.make_ba_problem(12, 400, 8, stereo=True, seed=3)
    large = synth.make_ba_problem(120, 1500, 10, stereo=True, seed=5)          # more optimised keyframes than the LDS-resident path holds
    res, nb = ov2slam_amd.Optimizer(gpu_ctx).localBA_batch([small, large, small])
    assert nb == 2
    _same(res[0], ov2slam_amd.Optimizer(gpu_ctx).localBA(small))
    _same(res[1], ov2slam_amd.Optimizer(gpu_ctx).localBA(large), tight=1e-6)
    _same(res[2], res[0])
    # the deterministic mode is a one-problem-at-a-time mode: the batch call falls back for every problem and returns its bit patterns
    from ov2slam_amd import _lib as L
    gpu_ctx.set_option(L.OV2_OPT_BA_DETERMINISTIC, 1)
    try:
        res, nb = ov2slam_amd.Optimizer(gpu_ctx).localBA_batch([small, small])
        one = ov2slam_amd.Optimizer(gpu_ctx).localBA(small)
    finally:
        gpu_ctx.set_option(L.OV2_OPT_BA_DETERMINISTIC, 0)
    assert nb == 0 and np.array_equal(res[0]["poses"], one["poses"]) and np.array_equal(res[1]["poses"], one["poses"])
    # an empty batch is a no-op; invalid input is rejected
    res, nb = ov2slam_amd.Optimizer(gpu_ctx).localBA_batch([])
    assert res == [] and nb == 0
    bad = dict(small); bad["res_lm"] = small["res_lm"].copy(); bad["res_lm"][0] = 10**6
    with pytest.raises(ov2slam_amd.Ov2Error):
        ov2slam_amd.Optimizer(gpu_ctx).localBA_batch([small, bad])


def test_batch_reports_a_status_per_problem(gpu_ctx):
    """a left-out problem that ov2_local_ba rejects (OV2_RES_PNP blocks: OV2_EUNSUPPORTED) must neither hide the valid
    results of the others nor stop the left-out problems after it (include/ov2slam_hip.h ov2_local_ba_result::status)."""
    from ov2slam_amd import _lib as L
    small = synth.make_ba_problem(12, 400, 8, stereo=True, seed=3)
    large = synth.make_ba_problem(120, 1500, 10, stereo=True, seed=5)
    pnp = synth.make_pnp_problem(100, seed=2)
    with pytest.raises(ov2slam_amd.Ov2Error):
        ov2slam_amd.Optimizer(gpu_ctx).localBA_batch([small, pnp, large])
    res, nb = ov2slam_amd.Optimizer(gpu_ctx).localBA_batch([small, pnp, large], raise_on_error=False)
    assert nb == 1
    assert [r["status"] for r in res] == [L.OV2_OK, L.OV2_EUNSUPPORTED, L.OV2_OK]
    _same(res[0], ov2slam_amd.Optimizer(gpu_ctx).localBA(small))
    _same(res[2], ov2slam_amd.Optimizer(gpu_ctx).localBA(large), tight=1e-6)
