"""Host-side mirror of Optimizer::localBA's solve stage (/root/reference/src/optimizer.cpp:436-627)
on the flat problem layout of include/ov2slam_hip.h.  The map walk that builds the problem
(:43-430) and the write-back (:741-883) stay in the caller, exactly as the C++ adapter described
in INTEGRATION.md does; everything Ceres did runs on the GPU through ov2_ba_solve."""
import ctypes as C
import math

import numpy as np

from . import _lib as L

RES_LEFT, RES_RIGHT, RES_RIGHT_ANCH, RES_PNP = 0, 1, 2, 3
TERMINATION = {0: "NO_CONVERGENCE", 1: "FUNCTION_TOLERANCE", 2: "PARAMETER_TOLERANCE", 3: "GRADIENT_TOLERANCE",
               4: "MIN_RADIUS", 5: "INVALID_STEPS", 6: "FAILURE"}


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def pack_problem(prob, res_active=None):
    """dict of arrays (see ov2slam_amd.synth.make_ba_problem) -> (BAProblem, keep-alive list)"""
    keep = []

    def arr(name, dt):
        a = np.ascontiguousarray(prob[name], dt)
        keep.append(a)
        return a

    P = L.BAProblem()
    P.n_kf, P.n_lm, P.n_res = int(prob["n_kf"]), int(prob["n_lm"]), int(prob["n_res"])
    P.poses = _dp(arr("poses", np.float64)); P.kf_const = _u8p(arr("kf_const", np.uint8))
    P.invdepth = _dp(arr("invdepth", np.float64)); P.lm_anchor_kf = _ip(arr("lm_anchor_kf", np.int32))
    P.lm_anchor_uv = _dp(arr("lm_anchor_uv", np.float64))
    P.res_type = _u8p(arr("res_type", np.uint8)); P.res_kf = _ip(arr("res_kf", np.int32)); P.res_lm = _ip(arr("res_lm", np.int32))
    P.res_uv = _dp(arr("res_uv", np.float64)); P.res_sigma = _dp(arr("res_sigma", np.float64))
    if res_active is not None:
        ra = np.ascontiguousarray(res_active, np.uint8)
        keep.append(ra)
        P.res_active = _u8p(ra)
    if prob.get("res_xyz") is not None:
        P.res_xyz = _dp(arr("res_xyz", np.float64))
    for i in range(4):
        P.calib_l[i] = float(prob["calib_l"][i]); P.calib_r[i] = float(prob["calib_r"][i])
    for i in range(7):
        P.T_rl[i] = float(prob["T_rl"][i])
    return P, keep


def default_options(lib, **kw):
    o = L.BAOptions()
    lib.ov2_ba_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


TRACE_FIELDS = ("iteration", "step_is_valid", "step_is_successful", "cost", "cost_change", "gradient_max_norm", "gradient_norm",
                "step_norm", "relative_decrease", "trust_region_radius")


def last_trace(ctx):
    """ov2_ba_get_trace: the iteration summaries of the context's last one-problem solve (OV2_OPT_BA_TRACE must have been on)"""
    buf = (L.BAIter * 64)(); n = C.c_int(0)
    L.check(ctx.lib.ov2_ba_get_trace(ctx.h, buf, 64, C.byref(n)))
    return [{f: getattr(buf[i], f) for f in TRACE_FIELDS} for i in range(min(n.value, 64))]


def solve(ctx, prob, opts=None, res_active=None, chi2_init=None, depthpos_init=None, trace=False):
    """One ceres::Solve (ov2_ba_solve).  Returns a dict like oracle.ba_solve (trace: with the iteration summaries under "trace")."""
    if trace:
        prev = ctx.get_option(L.OV2_OPT_BA_TRACE)
        ctx.set_option(L.OV2_OPT_BA_TRACE, 1)
        try:
            d = solve(ctx, prob, opts, res_active, chi2_init, depthpos_init)
            d["trace"] = last_trace(ctx)
        finally:
            ctx.set_option(L.OV2_OPT_BA_TRACE, prev)
        return d
    lib = ctx.lib
    opts = opts or default_options(lib)
    P, keep = pack_problem(prob, res_active)
    poses = np.zeros((P.n_kf, 7)); lam = np.zeros(max(1, P.n_lm))
    chi2 = np.full(max(1, P.n_res), np.nan) if chi2_init is None else np.array(chi2_init, np.float64, copy=True)
    dpos = np.zeros(max(1, P.n_res), np.uint8) if depthpos_init is None else np.array(depthpos_init, np.uint8, copy=True)
    R = L.BAResult()
    R.poses_out = _dp(poses); R.invdepth_out = _dp(lam); R.chi2_last_eval = _dp(chi2); R.depthpos_last_eval = _u8p(dpos)
    L.check(lib.ov2_ba_solve(ctx.h, C.byref(P), C.byref(opts), C.byref(R)))
    return dict(poses=poses, invdepth=lam[:P.n_lm], chi2=chi2[:P.n_res], depthpos=dpos[:P.n_res], iterations=R.iterations,
                num_successful_steps=R.num_successful_steps, initial_cost=R.initial_cost, final_cost=R.final_cost,
                termination=R.termination, solve_ms=R.solve_ms)


def is_xyz_problem(prob):
    """3-D point landmarks with variable poses (ov2slam_amd.synth.make_xyz_ba_problem layout: `xyz` + `res_pt` + `kf_const`)"""
    return "res_pt" in prob and "kf_const" in prob


def solve_xyz(ctx, prob, opts=None, res_active=None, chi2_init=None, depthpos_init=None):
    """One ceres::Solve of the buse_inv_depth: 0 branch (ov2_xyz_ba_solve).  Returns a dict like oracle.xyz_ba_solve."""
    lib = ctx.lib
    opts = opts or default_options(lib)
    keep = []

    def arr(name, dt, ct):
        a = np.ascontiguousarray(prob[name], dt); keep.append(a)
        return a.ctypes.data_as(C.POINTER(ct))

    P = L.XYZBAProblem()
    P.n_kf, P.n_pts, P.n_res = int(prob["n_kf"]), int(prob["n_pts"]), int(prob["n_res"])
    P.poses = arr("poses", np.float64, C.c_double); P.xyz = arr("xyz", np.float64, C.c_double); P.kf_const = arr("kf_const", np.uint8, C.c_uint8)
    P.res_type = arr("res_type", np.uint8, C.c_uint8); P.res_kf = arr("res_kf", np.int32, C.c_int); P.res_pt = arr("res_pt", np.int32, C.c_int)
    P.res_uv = arr("res_uv", np.float64, C.c_double); P.res_sigma = arr("res_sigma", np.float64, C.c_double)
    if res_active is not None:
        ra = np.ascontiguousarray(res_active, np.uint8); keep.append(ra)
        P.res_active = _u8p(ra)
    for i in range(4):
        P.calib_l[i] = float(prob["calib_l"][i]); P.calib_r[i] = float(prob["calib_r"][i])
    for i in range(7):
        P.T_rl[i] = float(prob["T_rl"][i])
    poses = np.zeros((P.n_kf, 7)); xyz = np.zeros((max(1, P.n_pts), 3))
    chi2 = np.full(max(1, P.n_res), np.nan) if chi2_init is None else np.array(chi2_init, np.float64, copy=True)
    dpos = np.zeros(max(1, P.n_res), np.uint8) if depthpos_init is None else np.array(depthpos_init, np.uint8, copy=True)
    R = L.XYZBAResult()
    R.poses_out = _dp(poses); R.xyz_out = _dp(xyz); R.chi2_last_eval = _dp(chi2); R.depthpos_last_eval = _u8p(dpos)
    L.check(lib.ov2_xyz_ba_solve(ctx.h, C.byref(P), C.byref(opts), C.byref(R)))
    return dict(poses=poses, xyz=xyz[:P.n_pts], chi2=chi2[:P.n_res], depthpos=dpos[:P.n_res], iterations=R.iterations,
                num_successful_steps=R.num_successful_steps, initial_cost=R.initial_cost, final_cost=R.final_cost,
                termination=R.termination, solve_ms=R.solve_ms)


def structure_only_ba(ctx, prob, opts=None, res_active=None):
    """One ceres::Solve of Optimizer::structureOnlyBA (src/optimizer.cpp:2594-2781) through ov2_structure_ba.
    prob: dict in the layout of ov2slam_amd.synth.make_structure_problem.  Default options = the reference's
    (10 iterations, function_tolerance 1e-3, Huber sqrt(5.9915))."""
    lib = ctx.lib
    opts = opts or default_options(lib, max_iter=10, function_tolerance=1e-3)
    keep = []

    def arr(name, dt, ct):
        a = np.ascontiguousarray(prob[name], dt); keep.append(a)
        return a.ctypes.data_as(C.POINTER(ct))

    P = L.SBAProblem()
    P.n_kf, P.n_pts, P.n_res = int(prob["n_kf"]), int(prob["n_pts"]), int(prob["n_res"])
    P.poses = arr("poses", np.float64, C.c_double); P.xyz = arr("xyz", np.float64, C.c_double)
    P.res_type = arr("res_type", np.uint8, C.c_uint8); P.res_kf = arr("res_kf", np.int32, C.c_int); P.res_pt = arr("res_pt", np.int32, C.c_int)
    P.res_uv = arr("res_uv", np.float64, C.c_double); P.res_sigma = arr("res_sigma", np.float64, C.c_double)
    if res_active is not None:
        ra = np.ascontiguousarray(res_active, np.uint8); keep.append(ra)
        P.res_active = ra.ctypes.data_as(C.POINTER(C.c_uint8))
    for i in range(4):
        P.calib_l[i] = float(prob["calib_l"][i]); P.calib_r[i] = float(prob["calib_r"][i])
    for i in range(7):
        P.T_rl[i] = float(prob["T_rl"][i])
    xyz = np.zeros((max(1, P.n_pts), 3)); chi2 = np.full(max(1, P.n_res), np.nan); dpos = np.zeros(max(1, P.n_res), np.uint8)
    R = L.SBAResult()
    R.xyz_out = _dp(xyz); R.chi2_last_eval = _dp(chi2); R.depthpos_last_eval = _u8p(dpos)
    L.check(lib.ov2_structure_ba(ctx.h, C.byref(P), C.byref(opts), C.byref(R)))
    return dict(xyz=xyz[:P.n_pts], chi2=chi2[:P.n_res], depthpos=dpos[:P.n_res], iterations=R.iterations,
                num_successful_steps=R.num_successful_steps, initial_cost=R.initial_cost, final_cost=R.final_cost,
                termination=R.termination, solve_ms=R.solve_ms)


class ResidentProblem:
    """ov2_ba_create / ov2_ba_solve_resident: the problem stays in HBM between solves."""

    def __init__(self, ctx, prob):
        self.ctx, self.lib = ctx, ctx.lib
        P, self._keep = pack_problem(prob)
        self.n_kf, self.n_lm, self.n_res = P.n_kf, P.n_lm, P.n_res
        h = C.c_void_p()
        L.check(self.lib.ov2_ba_create(ctx.h, C.byref(P), C.byref(h)))
        self.h = h
        self.poses = np.zeros((P.n_kf, 7)); self.lam = np.zeros(max(1, P.n_lm))
        self.chi2 = np.zeros(max(1, P.n_res)); self.dpos = np.zeros(max(1, P.n_res), np.uint8)

    def solve(self, opts=None):
        opts = opts or default_options(self.lib)
        R = L.BAResult()
        R.poses_out = _dp(self.poses); R.invdepth_out = _dp(self.lam)
        R.chi2_last_eval = _dp(self.chi2); R.depthpos_last_eval = _u8p(self.dpos)
        L.check(self.lib.ov2_ba_solve_resident(self.ctx.h, self.h, C.byref(opts), C.byref(R)))
        return dict(poses=self.poses, invdepth=self.lam[:self.n_lm], chi2=self.chi2[:self.n_res], depthpos=self.dpos[:self.n_res],
                    iterations=R.iterations, num_successful_steps=R.num_successful_steps, initial_cost=R.initial_cost,
                    final_cost=R.final_cost, termination=R.termination, solve_ms=R.solve_ms)

    def close(self):
        if getattr(self, "h", None):
            self.lib.ov2_ba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Optimizer:
    """Mirror of /root/reference/include/optimizer.hpp:36-60: localBA, looseBA, fullBA (structureOnlyBA: see
    structure_only_ba below; the two pose-graph solvers are out of scope).

    localBA(problem, buse_robust_cost) runs optimizer.cpp:436-627 on a flat problem:
      pass 1  Huber(sqrt(robust_mono_th)), 5 iterations, function_tolerance 1e-3          (:436-485)
      outlier test  chi2err_ > robust_mono_th or depth <= 0 on the values cached by the last
                    Evaluate of pass 1; residual blocks removed if
                    apply_l2_after_robust                                                (:492-594)
      pass 2  only if outliers were found: loss reset to L2 when both the left and the right
              residual lists are still non-empty (mono runs keep Huber!), 10 iterations  (:603-627)
    `solver` is injectable so that the tests can run the identical protocol on the oracle."""

    def __init__(self, ctx=None, robust_mono_th=5.9915, apply_l2_after_robust=True, solver=None):
        self.ctx = ctx
        self.robust_mono_th = float(robust_mono_th)
        self.apply_l2_after_robust = bool(apply_l2_after_robust)
        self._solver = solver
        # Optimizer::bstop_localba_ (include/optimizer.hpp:64): raised by the estimator thread while a localBA runs; the
        # library reads this very int between the two passes (ov2_local_ba_options::stop_flag), like the reference's
        # !stopLocalBA() after its first ceres::Solve (src/optimizer.cpp:603-604)
        self._stop_flag = C.c_int(0)
        # Ceres' max_solver_time_in_seconds of pass 1: the reference uses 0.2 s, 0.4 s unless force_realtime (:463-467), and
        # half of it for the L2 pass (:612).  0 = no limit (default: results independent of machine load)
        self.max_solver_time_s = 0.0

    def signalStopLocalBA(self):          # optimizer.hpp:48
        self._stop_flag.value = 1

    def stopLocalBA(self):                # optimizer.hpp:49
        return bool(self._stop_flag.value)

    def _solve(self, prob, res_active, chi2_init, depthpos_init, **opt_kw):
        if self._solver is not None:
            return self._solver(prob, res_active, chi2_init, depthpos_init, **opt_kw)
        opts = default_options(self.ctx.lib, **opt_kw)
        if is_xyz_problem(prob):                              # buse_inv_depth: 0 (optimizer.cpp:207-209, :333-384)
            return solve_xyz(self.ctx, prob, opts, res_active, chi2_init, depthpos_init)
        return solve(self.ctx, prob, opts, res_active, chi2_init, depthpos_init)

    def localBA(self, prob, buse_robust_cost=True, want_chi2=True):
        """Optimizer::localBA's solve stage.  Inverse-depth problems on the GPU go through ov2_local_ba: ONE call, the
        problem resident in HBM between the two passes, outlier tests and block removal on the device (want_chi2=False
        skips the download of the per-block chi2 / depth arrays, which the reference's write-back does not need).
        Everything else -- 3-D point problems, an injected solver (the oracle in the tests) -- runs the same protocol
        as two solver calls: localBA_two_calls."""
        if self._solver is None and not is_xyz_problem(prob) and (prob.get("res_xyz") is None):
            return self._local_ba_resident(prob, buse_robust_cost, want_chi2)
        return self.localBA_two_calls(prob, buse_robust_cost)

    def _local_ba_resident(self, prob, buse_robust_cost, want_chi2):
        lib = self.ctx.lib
        P, keep = pack_problem(prob)
        O = L.LocalBAOptions()
        lib.ov2_local_ba_default_options(C.byref(O))
        O.robust_mono_th = self.robust_mono_th; O.use_robust_cost = int(bool(buse_robust_cost))
        O.apply_l2_after_robust = int(self.apply_l2_after_robust); O.stop_requested = 0
        O.stop_flag = C.pointer(self._stop_flag)                                  # the LIVE flag, polled after pass 1
        O.pass1.max_solver_time_s = self.max_solver_time_s; O.pass2.max_solver_time_s = 0.5 * self.max_solver_time_s
        n_res = P.n_res
        poses = np.zeros((P.n_kf, 7)); lam = np.zeros(max(1, P.n_lm))
        bad = np.zeros(max(1, n_res), np.uint8); bad1 = np.zeros(max(1, n_res), np.uint8)
        R = L.LocalBAResult()
        R.poses_out = _dp(poses); R.invdepth_out = _dp(lam); R.bad_obs = _u8p(bad); R.bad_after_pass1 = _u8p(bad1)
        chi2 = dpos = None
        if want_chi2:
            chi2 = np.full(max(1, n_res), np.nan); dpos = np.zeros(max(1, n_res), np.uint8)
            R.chi2_last_eval = _dp(chi2); R.depthpos_last_eval = _u8p(dpos)
        try:
            L.check(lib.ov2_local_ba(self.ctx.h, C.byref(P), C.byref(O), C.byref(R)))
        finally:
            self._stop_flag.value = 0                                             # src/optimizer.cpp:896
        out = dict(poses=poses, invdepth=lam[:P.n_lm], bad_obs=bad[:n_res].astype(bool), bad_after_pass1=bad1[:n_res].astype(bool),
                   l2_done=bool(R.l2_done), pass2_error=int(R.pass2_error), iterations=(R.iterations[0], R.iterations[1]),
                   num_successful_steps=(R.num_successful_steps[0], R.num_successful_steps[1]), termination=(R.termination[0], R.termination[1]),
                   initial_cost=(R.initial_cost[0], R.initial_cost[1]), final_cost=(R.final_cost[0], R.final_cost[1]),
                   solve_ms=(R.solve_ms[0], R.solve_ms[1]))
        if want_chi2:
            out.update(chi2=chi2[:n_res], depthpos=dpos[:n_res])
        return out

    def localBA_batch(self, probs, buse_robust_cost=True, want_chi2=True, stop=None, raise_on_error=True):
        """Optimizer::localBA's solve stage for a lock-step batch of sequences (BASELINE configs[4]): one ov2_local_ba_batch call,
        every kernel of the solver launched once for all problems.  `probs`: inverse-depth problems (make_ba_problem layout);
        `stop`: per-problem stop requests (the estimators' bstop_localba_, read after the first pass) or None.
        Returns (list of per-problem dicts as localBA returns them, number of problems that shared the launches)."""
        lib = self.ctx.lib
        n = len(probs)
        PA = (L.BAProblem * n)(); OA = (L.LocalBAOptions * n)(); RA = (L.LocalBAResult * n)()
        keep, outs = [], []
        for i, prob in enumerate(probs):
            P, k = pack_problem(prob)
            keep.append((P, k)); PA[i] = P
            O = OA[i]
            lib.ov2_local_ba_default_options(C.byref(O))
            O.robust_mono_th = self.robust_mono_th; O.use_robust_cost = int(bool(buse_robust_cost))
            O.apply_l2_after_robust = int(self.apply_l2_after_robust)
            O.stop_requested = int(bool(stop[i])) if stop is not None else 0
            O.pass1.max_solver_time_s = self.max_solver_time_s; O.pass2.max_solver_time_s = 0.5 * self.max_solver_time_s
            n_res = P.n_res
            o = dict(poses=np.zeros((P.n_kf, 7)), _lam=np.zeros(max(1, P.n_lm)), _bad=np.zeros(max(1, n_res), np.uint8), _bad1=np.zeros(max(1, n_res), np.uint8),
                     _n_res=n_res, _n_lm=P.n_lm)
            R = RA[i]
            R.poses_out = _dp(o["poses"]); R.invdepth_out = _dp(o["_lam"]); R.bad_obs = _u8p(o["_bad"]); R.bad_after_pass1 = _u8p(o["_bad1"])
            if want_chi2:
                o["_chi2"] = np.full(max(1, n_res), np.nan); o["_dpos"] = np.zeros(max(1, n_res), np.uint8)
                R.chi2_last_eval = _dp(o["_chi2"]); R.depthpos_last_eval = _u8p(o["_dpos"])
            outs.append(o)
        nb = C.c_int(0)
        rc_all = lib.ov2_local_ba_batch(self.ctx.h, n, PA, OA, RA, C.byref(nb))
        if raise_on_error:
            L.check(rc_all)         # otherwise: every problem was attempted, the dicts carry `status` (0 = that result is valid)
        res = []
        for i, o in enumerate(outs):
            R = RA[i]; n_res = o["_n_res"]
            d = dict(poses=o["poses"], invdepth=o["_lam"][:o["_n_lm"]], bad_obs=o["_bad"][:n_res].astype(bool), bad_after_pass1=o["_bad1"][:n_res].astype(bool),
                     l2_done=bool(R.l2_done), pass2_error=int(R.pass2_error), status=int(R.status), iterations=(R.iterations[0], R.iterations[1]),
                     num_successful_steps=(R.num_successful_steps[0], R.num_successful_steps[1]), termination=(R.termination[0], R.termination[1]),
                     initial_cost=(R.initial_cost[0], R.initial_cost[1]), final_cost=(R.final_cost[0], R.final_cost[1]),
                     solve_ms=(R.solve_ms[0], R.solve_ms[1]), n_bad=(R.n_bad_pass1, R.n_bad_total))
            if want_chi2:
                d.update(chi2=o["_chi2"][:n_res], depthpos=o["_dpos"][:n_res])
            res.append(d)
        return res, int(nb.value)

    def localBA_two_calls(self, prob, buse_robust_cost=True):
        """Inverse-depth problems (make_ba_problem layout) or, with buse_inv_depth: 0, 3-D point problems
        (make_xyz_ba_problem layout): the protocol is the same, the landmark state is `invdepth` resp. `xyz`."""
        lmk = "xyz" if is_xyz_problem(prob) else "invdepth"
        th = self.robust_mono_th
        huber = math.sqrt(th) if buse_robust_cost else -1.0
        n_res = int(prob["n_res"])
        p1 = self._solve(prob, None, None, None, max_iter=5, function_tolerance=1e-3, huber_delta=huber)
        bad = (p1["chi2"] > th) | (p1["depthpos"] == 0)
        rtype = np.asarray(prob["res_type"])
        nbbadobs = int(bad.sum())
        out = dict(pass1=p1, bad_after_pass1=bad.copy(), l2_done=False)
        poses, lam, chi2, dpos = p1["poses"], p1[lmk], p1["chi2"], p1["depthpos"]
        active = np.ones(n_res, np.uint8)
        if self.apply_l2_after_robust:
            active[bad] = 0
        if self.apply_l2_after_robust and buse_robust_cost and not self.stopLocalBA() and nbbadobs > 0:
            left_remaining = bool(((rtype == RES_LEFT) & ~bad).any())
            right_remaining = bool(((rtype == RES_RIGHT) & ~bad).any())
            huber2 = -1.0 if (left_remaining and right_remaining) else huber       # :606-608
            prob2 = dict(prob)
            prob2["poses"] = poses; prob2[lmk] = lam
            p2 = self._solve(prob2, active, chi2, dpos, max_iter=10, function_tolerance=1e-3, huber_delta=huber2)
            out["pass2"] = p2; out["l2_done"] = True
            poses, lam, chi2, dpos = p2["poses"], p2[lmk], p2["chi2"], p2["depthpos"]
            # second outlier test on the residual blocks that are still in the problem (:637-735)
            bad2 = (active == 1) & ((chi2 > th) | (dpos == 0))
            bad = bad | bad2
        out.update(poses=poses, chi2=chi2, depthpos=dpos, bad_obs=bad)
        out[lmk] = lam
        return out


    def structureOnlyBA(self, prob):
        """Optimizer::structureOnlyBA (src/optimizer.cpp:2594-2781): Huber(sqrt(robust_mono_th)), 10 iterations,
        function_tolerance 1e-3; the map points take the optimised positions (:2768-2779), no outlier handling."""
        return structure_only_ba(self.ctx, prob, default_options(self.ctx.lib, max_iter=10, function_tolerance=1e-3,
                                                                 huber_delta=math.sqrt(self.robust_mono_th)))

    def looseBA(self, prob, buse_robust_cost=True):
        """Optimizer::looseBA (src/optimizer.cpp:900-1672) on a flat problem -- the loop-closure BA over the KFs
        between the loop KF and the new KF.  Same residual blocks as localBA (buse_inv_depth_: 1 in every shipped
        parameter file), ONE solve: Huber(sqrt(robust_mono_th)) unless !buse_robust_cost, 5 iterations,
        function_tolerance 1e-4, no time limit (:1297-1310); then the chi2 / depth outlier test over the left,
        right and anchor-right residual lists (:1327-1430)."""
        th = self.robust_mono_th
        huber = math.sqrt(th) if buse_robust_cost else -1.0
        p1 = self._solve(prob, None, None, None, max_iter=5, function_tolerance=1e-4, huber_delta=huber)
        bad = (p1["chi2"] > th) | (p1["depthpos"] == 0)
        return dict(pass1=p1, poses=p1["poses"], invdepth=p1["invdepth"], chi2=p1["chi2"], depthpos=p1["depthpos"], bad_obs=bad)

    def fullBA(self, prob, buse_robust_cost=True):
        """Optimizer::fullBA (src/optimizer.cpp:1674-2332) on a flat problem -- every KF and map point, run offline
        after the sequence (src/mapper.cpp:780).
          pass 1  Huber unless !buse_robust_cost, max 100 iterations, Ceres' default tolerances (function 1e-6,
                  gradient 1e-10, parameter 1e-8)                                                   (:2055-2061)
          outliers  chi2 / depth test over the left and right-camera lists (the anchor-right blocks are never
                    tested here); residual blocks removed if apply_l2_after_robust                   (:2067-2138)
          pass 2  if apply_l2_after_robust and outliers were found: loss reset to L2 when the left list is
                  non-empty, same options; then the outlier test again                              (:2143-2232)"""
        th = self.robust_mono_th
        huber = math.sqrt(th) if buse_robust_cost else -1.0
        n_res = int(prob["n_res"])
        rtype = np.asarray(prob["res_type"])
        tested = (rtype == RES_LEFT) | (rtype == RES_RIGHT)
        kw = dict(max_iter=100, function_tolerance=1e-6)
        p1 = self._solve(prob, None, None, None, huber_delta=huber, **kw)
        bad = tested & ((p1["chi2"] > th) | (p1["depthpos"] == 0))
        out = dict(pass1=p1, bad_after_pass1=bad.copy(), l2_done=False)
        poses, lam, chi2, dpos = p1["poses"], p1["invdepth"], p1["chi2"], p1["depthpos"]
        active = np.ones(n_res, np.uint8)
        if self.apply_l2_after_robust:
            active[bad] = 0
        if self.apply_l2_after_robust and int(bad.sum()) > 0:
            left_remaining = bool(((rtype == RES_LEFT) & ~bad).any())
            huber2 = -1.0 if left_remaining else huber                             # :2145-2147
            prob2 = dict(prob)
            prob2["poses"] = poses; prob2["invdepth"] = lam
            p2 = self._solve(prob2, active, chi2, dpos, huber_delta=huber2, **kw)
            out["pass2"] = p2; out["l2_done"] = True
            poses, lam, chi2, dpos = p2["poses"], p2["invdepth"], p2["chi2"], p2["depthpos"]
        bad2 = tested & (active == 1) & ((chi2 > th) | (dpos == 0))                # :2155-2232 (runs in both cases)
        bad = bad | bad2
        out.update(poses=poses, invdepth=lam, chi2=chi2, depthpos=dpos, bad_obs=bad)
        return out


class MultiViewGeometry:
    """Mirror of MultiViewGeometry::ceresPnP (/root/reference/src/multi_view_geometry.cpp:492-586), the per-frame
    motion-only BA called from VisualFrontEnd::computePose (src/visual_front_end.cpp:788-801):
      pass 1  Huber(sqrt(chi2th)) if buse_robust, nmaxiter iterations, function_tolerance 1e-3   (:519-544)
      outliers  chi2err_ > chi2th or depth <= 0 (values cached by the last Evaluate, N4); residual blocks removed
                if bapply_l2_after_robust; returns False when every observation is bad                (:547-565)
      pass 2  loss reset to L2, same options, only if outliers were found                           (:567-570)
    `solver` is injectable so that the tests can run the identical protocol on the oracle."""

    def __init__(self, ctx=None, solver=None):
        self.ctx, self._solver = ctx, solver

    def _solve(self, prob, res_active, chi2_init, depthpos_init, **kw):
        if self._solver is not None:
            return self._solver(prob, res_active, chi2_init, depthpos_init, **kw)
        return solve(self.ctx, prob, default_options(self.ctx.lib, **kw), res_active, chi2_init, depthpos_init)

    def ceresPnP(self, vunkps, vwpts, vscales, Twc, nmaxiter, chi2th, buse_robust, bapply_l2_after_robust, fx, fy, cx, cy):
        """Twc: [tx ty tz qx qy qz qw].  Returns (success, Twc_out, voutliersidx)."""
        n = len(vunkps)
        K = np.array([fx, fy, cx, cy], np.float64)
        prob = dict(n_kf=1, n_lm=0, n_res=n, poses=np.asarray(Twc, np.float64).reshape(1, 7), kf_const=np.zeros(1, np.uint8),
                    invdepth=np.zeros(0), lm_anchor_kf=np.zeros(0, np.int32), lm_anchor_uv=np.zeros((0, 2)),
                    res_type=np.full(n, RES_PNP, np.uint8), res_kf=np.zeros(n, np.int32), res_lm=np.full(n, -1, np.int32),
                    res_uv=np.asarray(vunkps, np.float64).reshape(n, 2), res_sigma=np.power(2.0, np.asarray(vscales, np.float64)),
                    res_xyz=np.asarray(vwpts, np.float64).reshape(n, 3), calib_l=K, calib_r=K, T_rl=np.array([0, 0, 0, 0, 0, 0, 1.0]))
        huber = math.sqrt(chi2th) if buse_robust else -1.0
        p1 = self._solve(prob, None, None, None, max_iter=int(nmaxiter), function_tolerance=1e-3, huber_delta=huber)
        bad = (p1["chi2"] > chi2th) | (p1["depthpos"] == 0)
        vout = np.nonzero(bad)[0]
        pose, last = p1["poses"][0], p1
        if bad.all():
            return False, pose, vout
        if bapply_l2_after_robust and len(vout):
            prob2 = dict(prob); prob2["poses"] = p1["poses"]
            last = self._solve(prob2, (~bad).astype(np.uint8), p1["chi2"], p1["depthpos"], max_iter=int(nmaxiter),
                               function_tolerance=1e-3, huber_delta=-1.0)
            pose = last["poses"][0]
        usable = last["termination"] in (0, 1, 2, 3, 4)      # Summary::IsSolutionUsable: CONVERGENCE / NO_CONVERGENCE
        return bool(usable), pose, vout
