"""The LM iteration of the oracle (oracle/ba.c orc_ba_solve) and of the device (ov2_ba_solve) against CERES' OWN LOOP, EXECUTED:
/root/reference/Thirdparty/ceres-solver/internal/ceres/{trust_region_minimizer, trust_region_step_evaluator, levenberg_marquardt_strategy,
corrector, loss_function, minimizer, ...}.cc are compiled from where they lie, unchanged, against Ceres' own headers and a stand-in
Eigen (oracle/ref/standin_dyn: Eigen is absent from this image) into oracle/_ref/libref_trlm.so (recipe: oracle/ref/Makefile).  Ceres'
TrustRegionMinimizer::Minimize() (trust_region_minimizer.cc:67-829) then drives the reference's own factors and SE(3) parameterisation
(libref_factors.so = src/ceres_parametrization.cpp compiled in place) through the evaluator / Jacobian / Schur solver of
oracle/ref/trlm_capi.cpp, with the options of src/optimizer.cpp:436-467.

What is compared, per problem: iteration count (trust-region steps computed), termination type, successful-step count, the whole
per-iteration sequence Ceres records in Solver::Summary::iterations -- step valid / accepted flags, cost, cost change, gradient max norm,
step norm, relative decrease, trust-region radius -- the final parameters and the factors' chi2err_ / isdepthpositive_ after their LAST
Evaluate.  Decisions must be identical; numbers agree to rounding (1e-9 relative: the stand-in Eigen, the harness' Schur
solve and the oracle sum in different orders; the bar for poses is north_star's 1e-4).

The library is built here (where /root/reference exists) and travels to the GPU box for the device leg."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

from ov2slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRLM_SO = os.path.join(ROOT, "oracle", "_ref", "libref_trlm.so")
NUM_FIELDS = ("cost", "cost_change", "gradient_max_norm", "step_norm", "relative_decrease", "trust_region_radius")
TERM = dict(NO_CONVERGENCE=0, FUNCTION_TOL=1, PARAMETER_TOL=2, GRADIENT_TOL=3, MIN_RADIUS=4)


class RefIter(C.Structure):
    _fields_ = [("iteration", C.c_int), ("step_is_valid", C.c_int), ("step_is_successful", C.c_int), ("linear_solver_iterations", C.c_int),
                ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double), ("gradient_norm", C.c_double),
                ("step_norm", C.c_double), ("relative_decrease", C.c_double), ("trust_region_radius", C.c_double), ("eta", C.c_double)]


@pytest.fixture(scope="module")
def ceres():
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref")])
    if not os.path.exists(TRLM_SO):
        pytest.skip("oracle/_ref/libref_trlm.so is absent and /root/reference is not here to build it from")
    return C.CDLL(TRLM_SO)


def ceres_solve(lib, oracle, prob, opts, res_active=None, chi2_init=None, depthpos_init=None):
    """ref_trlm_solve: Ceres' own Minimize() on the flat problem -> dict like oracle.ba_solve(trace=True) (+ counts, message)"""
    P, R, out, keep = oracle.pack_ba(prob, res_active, chi2_init, depthpos_init)
    buf = (RefIter * 64)(); n = C.c_int(0); cnt = (C.c_int * 4)(); msg = C.create_string_buffer(512)
    rc = lib.ref_trlm_solve(C.byref(P), C.byref(opts), C.byref(R), buf, 64, C.byref(n), cnt, msg, 512)
    assert rc == 0
    d = oracle.unpack_ba(R, out)
    d["trace"] = [{f: getattr(buf[i], f) for f in oracle.TRACE_FIELDS} for i in range(min(n.value, 64))]
    d["counts"] = dict(successful=cnt[0], unsuccessful=cnt[1], cost_evaluations=cnt[2], jacobian_evaluations=cnt[3])
    d["message"] = msg.value.decode()
    return d


def pattern(tr):
    return "".join("S" if t["step_is_successful"] else ("r" if t["step_is_valid"] else "i") for t in tr)


def same_solve(a, b, rtol, what, gradient_norm=True):
    """a (oracle or device) against b (Ceres' loop): identical decisions, numbers within rtol"""
    assert a["iterations"] == b["iterations"], (what, a["iterations"], b["iterations"])
    assert a["termination"] == b["termination"], (what, a["termination"], b["termination"], b.get("message"))
    assert a["num_successful_steps"] == b["num_successful_steps"], what
    assert pattern(a["trace"]) == pattern(b["trace"]), (what, pattern(a["trace"]), pattern(b["trace"]))
    worst = 0.0
    for ta, tb in zip(a["trace"], b["trace"]):
        assert ta["iteration"] == tb["iteration"]
        for f in NUM_FIELDS + (("gradient_norm",) if gradient_norm else ()):
            scale = max(abs(tb[f]), 1.0 if f == "relative_decrease" else 1e-300)
            if f in ("cost_change",):
                scale = max(scale, 1e-12 * abs(tb["cost"]))         # a difference of two costs: relative to the costs
            err = abs(ta[f] - tb[f]) / scale if tb[f] != 0 or ta[f] != 0 else 0.0
            worst = max(worst, err)
            assert err <= rtol, (what, tb["iteration"], f, ta[f], tb[f], err)
    for k in ("initial_cost", "final_cost"):
        assert abs(a[k] - b[k]) <= rtol * abs(b[k]), (what, k)
    return worst


def same_point(a, b, tol, what):
    assert np.abs(a["poses"] - b["poses"]).max() <= tol * max(1.0, np.abs(b["poses"]).max()), what
    assert np.allclose(a["invdepth"], b["invdepth"], rtol=10 * tol, atol=1e-12), what
    m = np.isfinite(b["chi2"])
    assert np.array_equal(np.isfinite(a["chi2"]), m), what
    assert np.allclose(a["chi2"][m], b["chi2"][m], rtol=100 * tol, atol=1e-9), what
    assert np.array_equal(a["depthpos"][m], b["depthpos"][m]), what


def problems(oracle):
    """(name, problem, options, expected termination or None, substring the accept / reject pattern must contain or None)"""
    O = oracle.ba_default_options
    huber = math.sqrt(5.9915)
    out = []
    # pass 1 of localBA as the reference runs it (5 iterations, function tolerance 1e-3, Huber): stereo and mono windows
    out.append(("pass1 stereo", synth.make_ba_problem(8, 200, 6, stereo=True, seed=5), O(), TERM["FUNCTION_TOL"], None))
    out.append(("pass1 mono", synth.make_ba_problem(12, 400, 8, stereo=False, seed=3), O(), TERM["FUNCTION_TOL"], None))
    out.append(("pass1 25 kf", synth.make_ba_problem(25, 1500, 10, stereo=True, seed=11), O(), None, None))
    # pass 2: trivial loss, 10 iterations
    out.append(("pass2 l2", synth.make_ba_problem(10, 300, 6, stereo=True, seed=2), O(huber_delta=-1.0, max_iter=10), None, None))
    # the iteration budget ends the solve: NO_CONVERGENCE after two steps
    out.append(("budget", synth.make_ba_problem(10, 300, 6, stereo=True, seed=9), O(max_iter=2, function_tolerance=1e-12), TERM["NO_CONVERGENCE"], None))
    # far from the optimum: rejected steps (radius halved, the diagonal reused), with and without the robustifier
    far = dict(pose_noise=(0.5, np.deg2rad(8)), invdepth_noise=0.8)
    out.append(("rejected l2", synth.make_ba_problem(10, 300, 6, stereo=True, seed=1, pose_noise=(1.0, np.deg2rad(15)), invdepth_noise=0.8),
                O(huber_delta=-1.0, max_iter=12, function_tolerance=1e-6), None, "rrrr"))
    out.append(("rejected huber", synth.make_ba_problem(10, 300, 6, stereo=False, seed=4, **far), O(huber_delta=huber, max_iter=12, function_tolerance=1e-6), None, "rrrr"))
    out.append(("rejected first", synth.make_ba_problem(10, 300, 6, stereo=True, seed=3, pose_noise=(1.0, np.deg2rad(15)), invdepth_noise=0.8),
                O(huber_delta=-1.0, max_iter=6, function_tolerance=1e-6), None, "Sr"))
    # the radius test ends the solve before any step
    out.append(("min radius", synth.make_ba_problem(8, 200, 6, stereo=True, seed=6), O(min_radius=2e4), TERM["MIN_RADIUS"], None))
    # motion-only BA (ceresPnP): one pose, pose-only blocks, no landmark columns
    out.append(("pnp", synth.make_pnp_problem(120, seed=2), O(max_iter=10, function_tolerance=1e-6), None, None))
    return out


def test_ceres_own_loop_agrees_with_the_oracle(ceres, oracle):
    seen_reject = seen_ftol_before_step = 0
    worst_all = 0.0
    for name, pb, opts, term, pat in problems(oracle):
        a = oracle.ba_solve(pb, opts, trace=True)
        b = ceres_solve(ceres, oracle, pb, opts)
        worst_all = max(worst_all, same_solve(a, b, 1e-9 if "rejected" not in name else 1e-6, name))
        same_point(a, b, 1e-9 if "rejected" not in name else 1e-7, name)
        if term is not None:
            assert b["termination"] == term, (name, b["message"])
        if pat is not None:
            assert pat in pattern(b["trace"]), (name, pattern(b["trace"]))
        seen_reject += "r" in pattern(b["trace"])
        # FUNCTION_TOLERANCE is tested before the step is taken (:729-748): the last step was computed, not recorded, not applied
        if b["termination"] == TERM["FUNCTION_TOL"]:
            seen_ftol_before_step += 1
            assert b["iterations"] == len(b["trace"]) and b["counts"]["successful"] == len(b["trace"])
        # Ceres' bookkeeping of its own run: every trust-region step costs one linear solve and one cost-only evaluation
        assert b["counts"]["successful"] + b["counts"]["unsuccessful"] == len(b["trace"])
    assert seen_reject >= 3 and seen_ftol_before_step >= 3
    print("worst relative difference of any recorded quantity: %.2e" % worst_all)


def test_gradient_and_parameter_tolerance_exits(ceres, oracle):
    pb = synth.make_ba_problem(10, 300, 6, stereo=True, seed=7)
    base = oracle.ba_solve(pb, oracle.ba_default_options(max_iter=10, function_tolerance=1e-9), trace=True)
    g = [t["gradient_max_norm"] for t in base["trace"] if t["step_is_successful"]]
    assert len(g) >= 4 and g[2] < g[1]
    # gradient tolerance between the norms of two accepted points: the solve ends right after the second of them (:660-678)
    opts = oracle.ba_default_options(max_iter=10, function_tolerance=1e-9, gradient_tolerance=math.sqrt(g[1] * g[2]))
    a = oracle.ba_solve(pb, opts, trace=True); b = ceres_solve(ceres, oracle, pb, opts)
    assert b["termination"] == TERM["GRADIENT_TOL"] and len(b["trace"]) == 3
    same_solve(a, b, 1e-9, "gradient tolerance"); same_point(a, b, 1e-9, "gradient tolerance")
    # parameter tolerance: x_norm is "invalid" (-1) until the first accepted step (:185), so the test cannot fire on the first step
    opts = oracle.ba_default_options(max_iter=10, function_tolerance=1e-12, parameter_tolerance=1e-3)
    a = oracle.ba_solve(pb, opts, trace=True); b = ceres_solve(ceres, oracle, pb, opts)
    assert b["termination"] == TERM["PARAMETER_TOL"] and len(b["trace"]) >= 2
    same_solve(a, b, 1e-9, "parameter tolerance"); same_point(a, b, 1e-9, "parameter tolerance")


def test_removed_blocks_keep_their_cached_chi2(ceres, oracle):
    """Pass 2 of localBA: the residual blocks removed after pass 1 keep the chi2err_ of their last evaluation, the live ones
    carry the value of the LAST point Ceres evaluated -- which is the rejected / not-taken candidate, not the returned solution."""
    pb = synth.make_ba_problem(10, 300, 6, stereo=True, seed=12, outlier_frac=0.1)
    o1 = oracle.ba_default_options()
    a1 = oracle.ba_solve(pb, o1, trace=True); b1 = ceres_solve(ceres, oracle, pb, o1)
    same_solve(a1, b1, 1e-9, "pass 1"); same_point(a1, b1, 1e-9, "pass 1")
    bad = (b1["chi2"] > 5.9915) | (b1["depthpos"] == 0)
    assert np.array_equal(bad, (a1["chi2"] > 5.9915) | (a1["depthpos"] == 0)) and 10 < bad.sum() < bad.size // 2
    pb2 = dict(pb); pb2["poses"] = b1["poses"]; pb2["invdepth"] = b1["invdepth"]
    o2 = oracle.ba_default_options(max_iter=10, huber_delta=-1.0)
    a2 = oracle.ba_solve(pb2, o2, res_active=~bad, chi2_init=b1["chi2"], depthpos_init=b1["depthpos"], trace=True)
    b2 = ceres_solve(ceres, oracle, pb2, o2, res_active=~bad, chi2_init=b1["chi2"], depthpos_init=b1["depthpos"])
    same_solve(a2, b2, 1e-9, "pass 2"); same_point(a2, b2, 1e-9, "pass 2")
    assert np.array_equal(b2["chi2"][bad], b1["chi2"][bad])
    # the last evaluated point is not the returned one: re-evaluating at the solution gives different chi2 values
    at_solution = oracle.ba_solve(dict(pb2, poses=b2["poses"], invdepth=b2["invdepth"]), oracle.ba_default_options(max_iter=0), res_active=~bad)
    live = ~bad
    assert np.abs(at_solution["chi2"][live] - b2["chi2"][live]).max() > 1e-6


@pytest.mark.gpu
def test_device_trace_agrees_with_ceres_own_loop(ceres, oracle, gpu_ctx):
    """The same sequences from the device: ov2_ba_solve with OV2_OPT_BA_TRACE against Ceres' own Minimize() (and the oracle)."""
    from ov2slam_amd import optimizer
    worst = 0.0
    for name, pb, opts, term, pat in problems(oracle):
        gopts = optimizer.default_options(gpu_ctx.lib, **{f: getattr(opts, f) for f, _ in opts._fields_})
        g = optimizer.solve(gpu_ctx, pb, gopts, trace=True)
        b = ceres_solve(ceres, oracle, pb, opts)
        a = oracle.ba_solve(pb, opts, trace=True)
        loose = "rejected" in name
        worst = max(worst, same_solve(g, b, 1e-5 if loose else 1e-8, name + " (device vs Ceres)", gradient_norm=False))
        same_solve(g, a, 1e-5 if loose else 1e-8, name + " (device vs oracle)", gradient_norm=False)
        same_point(g, b, 1e-6 if loose else 1e-8, name)
        assert all(math.isnan(t["gradient_norm"]) for t in g["trace"])
    print("device vs Ceres' own loop, worst relative difference of any recorded quantity: %.2e" % worst)
    # the trace is off by default and a solve without it leaves an empty trace behind
    optimizer.solve(gpu_ctx, problems(oracle)[0][1])
    assert optimizer.last_trace(gpu_ctx) == []
