"""ctypes wrapper of oracle/liboracle.so -- CPU ORACLE, test infrastructure only.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

ORC_MAX_LEVELS = 8
LK_USE_INITIAL_FLOW, LK_GET_MIN_EIGENVALS = 4, 8
MASK_AS_EXECUTED, MASK_INTENDED = 0, 1


class _Level(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("pad", C.c_int), ("img_pitch", C.c_int), ("der_pitch", C.c_int),
                ("img", C.c_void_p), ("der", C.c_void_p)]


class _Pyr(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("win", C.c_int), ("lv", _Level * ORC_MAX_LEVELS)]


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_fb_klt.restype = C.c_int
        _lib.orc_lk_track.restype = C.c_int
    return _lib


def use_native():
    """Switch this module to oracle/_native/liboracle.so, built here and now with -O3 -march=native (`make native`):
    the CPU baseline of bench.py runs the same restatement compiled for the host it is timed on.
    Returns the compiler flags used, or None when the native build failed (the portable build stays loaded)."""
    global _lib
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        nl = C.CDLL(os.path.join(_HERE, "_native", "liboracle.so"))
    except Exception:
        return None
    nl.orc_fb_klt.restype = C.c_int
    nl.orc_lk_track.restype = C.c_int
    _lib = nl
    return "-O3 -march=native -ffp-contract=off"


def use_fast():
    """Switch this module to oracle/_fast/liboracle.so (`make fast`): the relaxed-floating-point, vectorisation-friendly build of
    the same restatement.  A SPEED baseline (bench.py `cpu_baseline_simd`), never the checker: results agree with the strict
    build to rounding only.  Returns the flags, or None when the build failed."""
    global _lib
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "fast"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        nl = C.CDLL(os.path.join(_HERE, "_fast", "liboracle.so"))
    except Exception:
        return None
    nl.orc_fb_klt.restype = C.c_int
    nl.orc_lk_track.restype = C.c_int
    _lib = nl
    return "-O3 -march=native -ffast-math -ffp-contract=fast -funroll-loops -DORC_FAST (upper-triangle block Schur update)"


def set_num_threads(n):
    """Threads of the persistent pool behind CLAHE / pyrDown / Scharr / LK (cv::setNumThreads)."""
    lib().orc_set_num_threads(int(n))
    return lib().orc_get_num_threads()


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Pyramid:
    def __init__(self, img, win=9, max_level=3):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        self.p = _Pyr()
        self.win = win
        rc = lib().orc_pyr_build(_p(img), w, h, w, win, max_level, C.byref(self.p))
        if rc != 0:
            raise RuntimeError("orc_pyr_build failed: %d" % rc)

    def rebuild(self, img):
        """cv::buildOpticalFlowPyramid into the existing Mats (same geometry): what the reference does every frame."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        rc = lib().orc_pyr_rebuild(_p(img), w, h, w, C.byref(self.p))
        assert rc == 0, rc
        return self

    @property
    def levels(self):
        return self.p.n_levels

    def level_size(self, l):
        return self.p.lv[l].w, self.p.lv[l].h

    def level(self, l, padded=False):
        w, h = self.level_size(l)
        if padded:
            w, h = w + 2 * self.win, h + 2 * self.win
        img = np.empty((h, w), np.uint8)
        der = np.empty((h, w, 2), np.int16)
        fn = lib().orc_pyr_copy_level_padded if padded else lib().orc_pyr_copy_level
        assert fn(C.byref(self.p), l, _p(img), _p(der)) == 0
        return img, der

    def __del__(self):
        try:
            lib().orc_pyr_free(C.byref(self.p))
        except Exception:
            pass


def pyr_down(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    dw, dh = (w + 1) // 2, (h + 1) // 2
    out = np.empty((dh, dw), np.uint8)
    lib().orc_pyr_down_u8(_p(img), w, h, w, _p(out), dw, dh, dw)
    return out


def scharr(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty((h, w, 2), np.int16)
    lib().orc_scharr_u8(_p(img), w, h, w, _p(out), 2 * w)
    return out


def lk_track(prev, nxt, prev_xy, next_xy, win=9, max_level=3, max_count=30, eps=0.01,
             flags=LK_USE_INITIAL_FLOW | LK_GET_MIN_EIGENVALS, nthreads=1):
    p0 = np.ascontiguousarray(prev_xy, np.float32).reshape(-1, 2)
    p1 = np.array(next_xy, np.float32, copy=True).reshape(-1, 2)
    n = p0.shape[0]
    status = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    iters = np.zeros(n, np.int32)
    rc = lib().orc_lk_track(C.byref(prev.p), C.byref(nxt.p), _p(p0), _p(p1), n, _p(status), _p(err),
                            win, max_level, max_count, C.c_double(float(np.float32(eps))), flags, C.c_double(1e-4),
                            _p(iters), nthreads)
    assert rc == 0, rc
    return p1, status, err, iters


def fb_klt(prev, cur, win, nbpyrlvl, ferr, fbdist, kps, priors, max_iter=30, eps=0.01, nthreads=1):
    k = np.ascontiguousarray(kps, np.float32).reshape(-1, 2)
    pr = np.array(priors, np.float32, copy=True).reshape(-1, 2)
    n = k.shape[0]
    status = np.zeros(n, np.uint8)
    it, vis = C.c_longlong(0), C.c_longlong(0)
    rc = lib().orc_fb_klt(C.byref(prev.p), C.byref(cur.p), win, nbpyrlvl, max_iter, C.c_float(eps),
                          C.c_float(ferr), C.c_float(fbdist), _p(k), _p(pr), n, _p(status),
                          C.byref(it), C.byref(vis), nthreads)
    assert rc == 0, rc
    return pr, status.astype(bool), (it.value, vis.value)


LK_ACC_INT64, LK_ACC_FLOAT_SCALAR, LK_ACC_FLOAT_SSE34, LK_ACC_FLOAT_UI4 = 0, 1, 2, 3
LK_ACC_NAMES = {0: "int64 (canonical, what the HIP kernels implement)", 1: "float, scalar raster order (no SIMD build)",
                2: "float, OpenCV 3.4 SSE2 intrinsics order", 3: "float, OpenCV 4.x universal-intrinsics order"}


class lk_acc_mode:
    """with lk_acc_mode(LK_ACC_FLOAT_UI4): ...  -- accumulator variant of calcOpticalFlowPyrLK (frontend.c)."""

    def __init__(self, mode):
        self.mode = int(mode)

    def __enter__(self):
        f = lib().orc_get_lk_acc_mode
        f.restype = C.c_int
        self.prev = f()
        lib().orc_set_lk_acc_mode(self.mode)
        return self

    def __exit__(self, *a):
        lib().orc_set_lk_acc_mode(self.prev)
        return False


def lk_acc_mode_report(prev, cur, kps, priors, win=9, nbpyrlvl=3, ferr=30., fbdist=0.5, max_iter=30, eps=0.01):
    """fbKltTracking with the canonical int64 accumulators against the three float-accumulator orders of stock OpenCV builds on
    the same inputs: per float mode the number of status flips and the largest position difference among keypoints tracked by
    both.  This is the measured distance between what the HIP kernels compute bit-exactly and what a real OpenCV build would."""
    ref_xy, ref_st, _ = fb_klt(prev, cur, win, nbpyrlvl, ferr, fbdist, kps, priors, max_iter, eps)
    out = {"points": int(len(ref_st)), "tracked_int64": int(ref_st.sum()), "modes": {}}
    for mode in (LK_ACC_FLOAT_SCALAR, LK_ACC_FLOAT_SSE34, LK_ACC_FLOAT_UI4):
        with lk_acc_mode(mode):
            xy, st, _ = fb_klt(prev, cur, win, nbpyrlvl, ferr, fbdist, kps, priors, max_iter, eps)
        both = ref_st & st
        d = np.abs(xy[both].astype(np.float64) - ref_xy[both].astype(np.float64)).max(axis=1) if both.any() else np.zeros(0)
        out["modes"][LK_ACC_NAMES[mode]] = {
            "status_flips": int((ref_st != st).sum()),
            "bit_identical_positions": int((xy[both].view(np.uint32) == ref_xy[both].view(np.uint32)).all(axis=1).sum()),
            "max_abs_dpx": float(d.max()) if len(d) else 0.0,
            "p99_abs_dpx": float(np.quantile(d, 0.99)) if len(d) else 0.0,
            "above_0.01px": int((d > 0.01).sum()),
        }
    return out


# ---------------------------------------------------------------- detection
def klt_tracking(prev, cur, kps, priors, has_prior, win=9, nklt_pyr_lvl=3, ferr=30., fbdist=0.5, klt_use_prior=True,
                 max_iter=30, eps=0.01):
    """Control flow of VisualFrontEnd::kltTracking (/root/reference/src/visual_front_end.cpp:132-275) on top of
    fb_klt: keypoints with a 3-D prior are tracked on 2 levels first (:186-199), the lost ones join the second call
    with the first call's forward result as prior (:213-217), and when fewer than a third were good every prior of
    the second call is reset to its keypoint (:225-230).  Test infrastructure (the reference composes its list by
    push_back; keypoints are independent inside fbKltTracking, so results are reported per input index).
    Returns (px (n,2) float32: value handed to updateKeypoint / last forward result, ok (n,) bool,
             retried (n,) bool: went through the second call after losing the first, bp3preq)."""
    kps = np.ascontiguousarray(kps, np.float32).reshape(-1, 2)
    pri = np.array(priors, np.float32, copy=True).reshape(-1, 2)
    n = len(kps)
    hp = np.zeros(n, bool) if (has_prior is None or not klt_use_prior) else np.asarray(has_prior).astype(bool)
    out = pri.copy()
    ok = np.zeros(n, bool)
    retried = np.zeros(n, bool)
    p3p = False
    ia = np.nonzero(hp)[0]
    ib = list(np.nonzero(~hp)[0])
    pri_b = {int(i): pri[i].copy() for i in ib}
    if len(ia):
        o, st, _ = fb_klt(prev, cur, win, 1, ferr, fbdist, kps[ia], pri[ia], max_iter, eps)
        st = np.asarray(st).astype(bool)
        for j, i in enumerate(ia):
            out[i] = o[j]
            if st[j]:
                ok[i] = True
            else:
                ib.append(int(i)); pri_b[int(i)] = o[j].copy(); retried[i] = True
        if st.sum() < 0.33 * len(ia):
            p3p = True
            for i in ib:
                pri_b[i] = kps[i].copy()
    if len(ib):
        ib = np.array(ib)
        o, st, _ = fb_klt(prev, cur, win, nklt_pyr_lvl, ferr, fbdist, kps[ib], np.stack([pri_b[int(i)] for i in ib]), max_iter, eps)
        st = np.asarray(st).astype(bool)
        out[ib] = o
        ok[ib] = st
    return out, ok, retried, p3p


def fast9_16(img, threshold, nonmax=True):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = w * h
    xs = np.zeros(cap, np.int32); ys = np.zeros(cap, np.int32); sc = np.zeros(cap, np.int32)
    n = lib().orc_fast9_16(_p(img), w, h, w, int(threshold), int(nonmax), _p(xs), _p(ys), _p(sc), cap)
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def circle_fill0(mask, cx, cy, r):
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    lib().orc_circle_fill0(_p(mask), w, h, int(cx), int(cy), int(r))
    return mask


SOBEL_DY_OPENCV_ROWFILTER, SOBEL_DY_EXACT_SUM = 0, 1


def set_sobel_dy_order(order):
    """Evaluation order of cv::Sobel(dx=0, dy=1, scale) inside cornerMinEigenVal (oracle/detect.c); returns the previous one."""
    prev = lib().orc_get_sobel_dy_order()
    lib().orc_set_sobel_dy_order(int(order))
    return prev


def cell_mineig(img, x0, y0, cell):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    hmap = np.zeros((cell, cell), np.float32)
    lib().orc_cell_mineig(_p(img), w, h, w, int(x0), int(y0), int(cell), _p(hmap))
    return hmap


def corner_subpix(img, pts, half_win=3, max_iter=30, eps=0.01):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    p = np.array(pts, np.float32, copy=True).reshape(-1, 2)
    lib().orc_corner_subpix(_p(img), w, h, w, _p(p), p.shape[0], int(half_win), int(max_iter), C.c_double(eps))
    return p


def detect_grid_fast(img, cell, cur_kps, fast_th, mask_mode=MASK_AS_EXECUTED, subpix=True):
    """returns (points (n,2) float32, new fast threshold)"""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cur = np.ascontiguousarray(cur_kps, np.float32).reshape(-1, 2)
    out = np.zeros((max(1, (w // cell) * (h // cell)), 2), np.float32)
    n = C.c_int(0); th = C.c_int(int(fast_th))
    rc = lib().orc_detect_grid_fast(_p(img), w, h, w, int(cell), _p(cur), cur.shape[0], C.byref(th), int(mask_mode),
                                    int(bool(subpix)), _p(out), C.byref(n))
    assert rc == 0, rc
    return out[:n.value].copy(), th.value


FAST_TIE_SCAN_ORDER, FAST_TIE_LIBSTDCXX = 0, 1


class fast_tie_mode:
    """with fast_tie_mode(FAST_TIE_LIBSTDCXX): ...  -- which of several equal best FAST responses of a cell wins in detect_grid_fast: the
    first in scan order (canonical, the HIP kernels) or the one libstdc++'s std::sort leaves in front (the reference as built with g++)."""

    def __init__(self, mode):
        self.mode = int(mode)

    def __enter__(self):
        L = lib()
        L.orc_get_fast_tie_mode.restype = C.c_int
        self.prev = L.orc_get_fast_tie_mode()
        L.orc_set_fast_tie_mode(self.mode)
        return self

    def __exit__(self, *a):
        lib().orc_set_fast_tie_mode(self.prev)


def fast_tie_sort_fallbacks():
    f = lib().orc_fast_tie_sort_fallbacks
    f.restype = C.c_int
    return int(f())


BLUR_FIXED, BLUR_HALF_EVEN = 0, 1
SUBPIX_FAST, SUBPIX_GENERIC, SUBPIX_FLOAT_ACC = 0, 1, 2


class detect_variant:
    """with detect_variant(blur=BLUR_HALF_EVEN, subpix=SUBPIX_GENERIC): ...  -- version-dependent arithmetic of the detector's
    OpenCV calls (detect.c); the canonical choices (what the HIP kernels implement) are the defaults."""

    def __init__(self, blur=BLUR_FIXED, subpix=SUBPIX_FAST):
        self.blur, self.subpix = int(blur), int(subpix)

    def __enter__(self):
        L = lib()
        L.orc_get_blur_mode.restype = C.c_int; L.orc_get_subpix_mode.restype = C.c_int
        self.prev = (L.orc_get_blur_mode(), L.orc_get_subpix_mode())
        L.orc_set_blur_mode(self.blur); L.orc_set_subpix_mode(self.subpix)
        return self

    def __exit__(self, *a):
        lib().orc_set_blur_mode(self.prev[0]); lib().orc_set_subpix_mode(self.prev[1])


def detect_singlescale(img, cell, cur_kps, roi, quality, subpix=True):
    """returns (points (n,2) float32, new dmaxquality)"""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cur = np.ascontiguousarray(cur_kps, np.float32).reshape(-1, 2)
    out = np.zeros((max(1, 2 * (w // cell) * (h // cell)), 2), np.float32)
    n = C.c_int(0); q = C.c_double(float(quality))
    roi_a = (C.c_int * 4)(*[int(v) for v in roi])
    rc = lib().orc_detect_singlescale(_p(img), w, h, w, int(cell), _p(cur), cur.shape[0], roi_a, C.byref(q),
                                      int(bool(subpix)), _p(out), C.byref(n))
    assert rc == 0, rc
    return out[:n.value].copy(), q.value


# ---------------------------------------------------------------- local BA
class _BAProblem(C.Structure):
    _fields_ = [
        ("n_kf", C.c_int), ("poses", C.c_void_p), ("kf_const", C.c_void_p),
        ("n_lm", C.c_int), ("invdepth", C.c_void_p), ("lm_anchor_kf", C.c_void_p), ("lm_anchor_uv", C.c_void_p),
        ("n_res", C.c_int), ("res_type", C.c_void_p), ("res_kf", C.c_void_p), ("res_lm", C.c_void_p),
        ("res_uv", C.c_void_p), ("res_sigma", C.c_void_p), ("res_active", C.c_void_p), ("res_xyz", C.c_void_p),
        ("calib_l", C.c_double * 4), ("calib_r", C.c_double * 4), ("T_rl", C.c_double * 7),
    ]


class BAOptions(C.Structure):
    _fields_ = [
        ("max_iter", C.c_int), ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double), ("huber_delta", C.c_double), ("initial_radius", C.c_double),
        ("max_radius", C.c_double), ("min_radius", C.c_double), ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double), ("min_relative_decrease", C.c_double), ("jacobi_scaling", C.c_int),
        ("max_consecutive_invalid_steps", C.c_int),
    ]


class _BAResult(C.Structure):
    _fields_ = [
        ("poses_out", C.c_void_p), ("invdepth_out", C.c_void_p), ("chi2_last_eval", C.c_void_p),
        ("depthpos_last_eval", C.c_void_p), ("iterations", C.c_int), ("num_successful_steps", C.c_int),
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("termination", C.c_int),
    ]


def ba_default_options(**kw):
    o = BAOptions()
    lib().orc_ba_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class BAIter(C.Structure):
    """orc_ba_iter / ref_trlm_iter share the leading layout (ints, then doubles)"""
    _fields_ = [("iteration", C.c_int), ("step_is_valid", C.c_int), ("step_is_successful", C.c_int), ("pad_", C.c_int),
                ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double), ("gradient_norm", C.c_double),
                ("step_norm", C.c_double), ("relative_decrease", C.c_double), ("trust_region_radius", C.c_double)]


TRACE_FIELDS = ("iteration", "step_is_valid", "step_is_successful", "cost", "cost_change", "gradient_max_norm", "gradient_norm",
                "step_norm", "relative_decrease", "trust_region_radius")


def pack_ba(prob, res_active=None, chi2_init=None, depthpos_init=None):
    """-> (orc_ba_problem, orc_ba_result, output arrays, keep-alive dict) for a dict in the layout of ov2slam_amd.synth.make_ba_problem"""
    keep = {}

    def arr(name, dt):
        a = np.ascontiguousarray(prob[name], dt)
        keep[name] = a
        return a.ctypes.data

    P = _BAProblem()
    P.n_kf, P.n_lm, P.n_res = int(prob["n_kf"]), int(prob["n_lm"]), int(prob["n_res"])
    P.poses = arr("poses", np.float64); P.kf_const = arr("kf_const", np.uint8)
    P.invdepth = arr("invdepth", np.float64); P.lm_anchor_kf = arr("lm_anchor_kf", np.int32)
    P.lm_anchor_uv = arr("lm_anchor_uv", np.float64)
    P.res_type = arr("res_type", np.uint8); P.res_kf = arr("res_kf", np.int32); P.res_lm = arr("res_lm", np.int32)
    P.res_uv = arr("res_uv", np.float64); P.res_sigma = arr("res_sigma", np.float64)
    if res_active is not None:
        ra = np.ascontiguousarray(res_active, np.uint8); keep["ra"] = ra
        P.res_active = ra.ctypes.data
    if prob.get("res_xyz") is not None:
        P.res_xyz = arr("res_xyz", np.float64)
    for i in range(4):
        P.calib_l[i] = float(prob["calib_l"][i]); P.calib_r[i] = float(prob["calib_r"][i])
    for i in range(7):
        P.T_rl[i] = float(prob["T_rl"][i])
    poses_out = np.zeros((P.n_kf, 7)); lam_out = np.zeros(P.n_lm)
    chi2 = np.full(P.n_res, np.nan) if chi2_init is None else np.array(chi2_init, np.float64, copy=True)
    dpos = np.zeros(P.n_res, np.uint8) if depthpos_init is None else np.array(depthpos_init, np.uint8, copy=True)
    R = _BAResult()
    R.poses_out = poses_out.ctypes.data; R.invdepth_out = lam_out.ctypes.data
    R.chi2_last_eval = chi2.ctypes.data; R.depthpos_last_eval = dpos.ctypes.data
    return P, R, dict(poses=poses_out, invdepth=lam_out, chi2=chi2, depthpos=dpos), keep


def unpack_ba(R, out):
    return dict(out, iterations=R.iterations, num_successful_steps=R.num_successful_steps, initial_cost=R.initial_cost,
                final_cost=R.final_cost, termination=R.termination)


def trace_rows(buf, n):
    return [{f: getattr(buf[i], f) for f in TRACE_FIELDS} for i in range(n)]


def ba_solve(prob, opts=None, res_active=None, chi2_init=None, depthpos_init=None, trace=False):
    """prob: dict in the layout produced by ov2slam_amd.synth.make_ba_problem.  trace: also return the per-iteration records
    (orc_ba_set_trace) under "trace"."""
    opts = opts or ba_default_options()
    P, R, out, keep = pack_ba(prob, res_active, chi2_init, depthpos_init)
    L = lib()
    if trace:
        buf = (BAIter * 64)(); n = C.c_int(0)
        L.orc_ba_set_trace(buf, 64, C.byref(n))
    try:
        rc = L.orc_ba_solve(C.byref(P), C.byref(opts), C.byref(R))
    finally:
        if trace:
            L.orc_ba_set_trace(None, 0, None)
    assert rc == 0, rc
    d = unpack_ba(R, out)
    if trace:
        d["trace"] = trace_rows(buf, min(n.value, 64))
    return d


def huber(a, s):
    rho = (C.c_double * 3)()
    lib().orc_huber(C.c_double(a), C.c_double(s), rho)
    return [rho[0], rho[1], rho[2]]


def corrector(sq_norm, rho, residuals, jacobian):
    r = np.array(residuals, np.float64, copy=True).reshape(-1)
    J = np.array(jacobian, np.float64, copy=True)
    J2 = J.reshape(r.shape[0], -1)
    rho_a = (C.c_double * 3)(*rho)
    lib().orc_corrector(C.c_double(sq_norm), rho_a, r.shape[0], J2.shape[1], _p(r), _p(J2))
    return r, J2


def lm_radius_sequence(initial, max_radius, events):
    rad, dec = C.c_double(initial), C.c_double(2.0)
    out = []
    for kind, q in events:
        if kind == "reject":
            lib().orc_lm_step_rejected(C.byref(rad), C.byref(dec))
        else:
            lib().orc_lm_step_accepted(C.c_double(q), C.byref(rad), C.byref(dec), C.c_double(max_radius))
        out.append(rad.value)
    return out


def se3_left_plus(pose, delta):
    p = np.ascontiguousarray(pose, np.float64); d = np.ascontiguousarray(delta, np.float64)
    out = np.zeros(7)
    lib().orc_se3_left_plus(_p(p), _p(d), _p(out))
    return out


def ba_residual(rtype, calib_l, calib_r, T_rl, anchor_pose, obs_pose, invdepth, anchor_uv, uv, sigma=1.0):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    cl, cr, trl, ap, op, auv, uvv = f(calib_l), f(calib_r), f(T_rl), f(anchor_pose), f(obs_pose), f(anchor_uv), f(uv)
    r = np.zeros(2); Ja = np.zeros(12); Jo = np.zeros(12); Jl = np.zeros(2); chi2 = C.c_double(0)
    lib().orc_ba_residual.restype = C.c_int
    dp = lib().orc_ba_residual(int(rtype), _p(cl), _p(cr), _p(trl), _p(ap), _p(op), C.c_double(invdepth), _p(auv), _p(uvv),
                               C.c_double(sigma), _p(r), _p(Ja), _p(Jo), _p(Jl), C.byref(chi2))
    return r, Ja.reshape(2, 6), Jo.reshape(2, 6), Jl, chi2.value, bool(dp)


# ---------------------------------------------------------------- CLAHE
def clahe(img, clip_limit, tiles_x, tiles_y):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty_like(img)
    rc = lib().orc_clahe(_p(img), w, h, w, C.c_double(clip_limit), int(tiles_x), int(tiles_y), _p(out), w)
    assert rc == 0, rc
    return out


# ---- per-keypoint undistortion + bearing (undistort.c) -------------------------------------------
CAM_PINHOLE, CAM_FISHEYE = 0, 1


def compute_keypoints(model, K, D, iK, px):
    """Frame::computeKeypoint for an (n,2) float32 array: returns (unpx (n,2) float32, bv (n,3) float64)."""
    px = np.ascontiguousarray(px, dtype=np.float32).reshape(-1, 2)
    K = np.ascontiguousarray(K, dtype=np.float64); iK = np.ascontiguousarray(iK, dtype=np.float64).reshape(9)
    D = np.ascontiguousarray(D if D is not None else [], dtype=np.float64)
    n = len(px)
    unpx = np.empty((n, 2), np.float32); bv = np.empty((n, 3), np.float64)
    f = lib().orc_compute_keypoints
    f.restype = None
    f(int(model), _p(K), _p(D) if len(D) else None, int(len(D)), _p(iK), _p(px), n, _p(unpx), _p(bv))
    return unpx, bv


# ---- stereo matching front half (stereo.c) --------------------------------------------------------
def get_rect_subpix_8u(img, pw, ph, cx, cy):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty((ph, pw), np.uint8)
    f = lib().orc_get_rect_subpix_8u
    f.restype = None
    f(_p(img), w, w, h, _p(out), int(pw), int(ph), C.c_float(cx), C.c_float(cy))
    return out


def line_min_sad(iml, imr, pts, nwinsize=7, go_left=True):
    """FeatureTracker::getLineMinSAD for an (n,2) array of points: (xprior (n,), l1err (n,))."""
    iml = np.ascontiguousarray(iml, dtype=np.uint8); imr = np.ascontiguousarray(imr, dtype=np.uint8)
    h, w = iml.shape
    assert imr.shape == iml.shape
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
    n = len(pts)
    xp = np.empty(n, np.float32); err = np.empty(n, np.float32)
    f = lib().orc_line_min_sad_batch
    f.restype = None
    f(_p(iml), w, _p(imr), w, w, h, _p(pts), n, int(nwinsize), int(bool(go_left)), _p(xp), _p(err))
    return xp, err


def sampson_distance(F, l, r):
    F = np.ascontiguousarray(F, dtype=np.float64).reshape(9)
    f = lib().orc_sampson_distance
    f.restype = C.c_float
    return float(f(_p(F), C.c_float(l[0]), C.c_float(l[1]), C.c_float(r[0]), C.c_float(r[1])))


def stereo_epipolar_check(rect, Frl, model, K, D, lunpx, rkps):
    """returns (rkps_out, runpx, epi_err, ok) like the loop at map_manager.cpp:568-590."""
    Frl = np.ascontiguousarray(Frl, dtype=np.float64).reshape(9)
    K = np.ascontiguousarray(K, dtype=np.float64)
    D = np.ascontiguousarray(D if D is not None else [], dtype=np.float64)
    lunpx = np.ascontiguousarray(lunpx, dtype=np.float32).reshape(-1, 2)
    rk = np.array(rkps, dtype=np.float32).reshape(-1, 2).copy()
    n = len(rk)
    runpx = np.empty((n, 2), np.float32); err = np.empty(n, np.float32); ok = np.empty(n, np.uint8)
    f = lib().orc_stereo_epipolar_check
    f.restype = None
    f(int(bool(rect)), _p(Frl), int(model), _p(K), _p(D) if len(D) else None, int(len(D)), _p(lunpx), _p(rk), n, _p(runpx), _p(err), _p(ok))
    return rk, runpx, err, ok.astype(bool)


def stereo_matching(leftpyr, rightpyr, kps_px, kps_unpx, model, K, D, rect, Frl=None, win=9, nklt_pyr_lvl=3, ferr=30., fbdist=0.5,
                    priors3d=None, max_iter=30, eps=0.01):
    """Data path of MapManager::stereoMatching (/root/reference/src/map_manager.cpp:367-611) restated list by list, with the
    reference's own push_back order (test infrastructure; the map look-ups that produce `priors3d` stay outside):
      :421-439  keypoints without a 3-D prior: prior = keypoint, x replaced by getLineMinSAD on level nklt_pyr_lvl when
                0 <= xprior <= kp.x (rectified pairs only);
      :497-541  fbKltTracking(left, right, nbpyrlvl 1) on the 3-D-prior list; a failure is appended to the 2-D list with
                v3dpriors.at(i) -- the vector fbKltTracking has just overwritten with its forward result (feature_tracker.cpp:66);
      :544-565  fbKltTracking on the 2-D list with the full pyramid;
      :568-590  epipolar gate on the good tracks.
    Returns (stereo_ok (n,) bool, right_px (n,2) float32)."""
    kps_px = np.ascontiguousarray(kps_px, np.float32).reshape(-1, 2)
    kps_unpx = np.ascontiguousarray(kps_unpx, np.float32).reshape(-1, 2)
    n = len(kps_px)
    priors3d = priors3d or {}
    v3dkpids, v3dkps, v3dpriors, vkpids, vkps, vpriors = [], [], [], [], [], []
    up = np.float32(2.0 ** nklt_pyr_lvl); down = np.float32(1.0) / up
    xp_all = None
    if rect and n:                                                         # getLineMinSAD is per keypoint (:431): one batched call here
        xp_all, _ = line_min_sad(leftpyr.level(nklt_pyr_lvl)[0], rightpyr.level(nklt_pyr_lvl)[0], kps_px * down, 7, True)
    for i in range(n):                                                     # :392-489 (kps in frame order)
        if i in priors3d:
            v3dkpids.append(i); v3dkps.append(kps_px[i]); v3dpriors.append(np.asarray(priors3d[i], np.float32))
            continue
        pr = kps_px[i].copy()
        if rect:
            x = np.float32(xp_all[i]) * up
            if x >= 0 and x <= kps_px[i, 0]:
                pr[0] = x
        vkpids.append(i); vkps.append(kps_px[i]); vpriors.append(pr)
    goodids, goodr = [], []
    if v3dpriors:
        out, st, _ = fb_klt(leftpyr, rightpyr, win, 1, ferr, fbdist, np.array(v3dkps, np.float32), np.array(v3dpriors, np.float32), max_iter, eps)
        v3dpriors = list(out)                                              # updated in place by calcOpticalFlowPyrLK
        for j in range(len(v3dkpids)):
            if st[j]:
                goodr.append(v3dpriors[j]); goodids.append(v3dkpids[j])
            else:
                vkpids.append(v3dkpids[j]); vkps.append(v3dkps[j]); vpriors.append(v3dpriors[j])
    if vkps:
        out, st, _ = fb_klt(leftpyr, rightpyr, win, nklt_pyr_lvl, ferr, fbdist, np.array(vkps, np.float32), np.array(vpriors, np.float32), max_iter, eps)
        for j in range(len(vkpids)):
            if st[j]:
                goodr.append(out[j]); goodids.append(vkpids[j])
    ok = np.zeros(n, bool); right = np.zeros((n, 2), np.float32)
    if goodids:
        gi = np.array(goodids, np.int64)
        rk, _, _, eok = stereo_epipolar_check(rect, Frl if Frl is not None else np.zeros(9), model, K, D, kps_unpx[gi], np.array(goodr, np.float32))
        ok[gi] = eok; right[gi] = rk
    return ok, right


# ---- Optimizer::structureOnlyBA (struct_ba.c) -------------------------------------------------------
XYZ_LEFT, XYZ_RIGHT = 0, 1


class _SBAProblem(C.Structure):
    _fields_ = [("n_kf", C.c_int), ("poses", C.c_void_p), ("n_pts", C.c_int), ("xyz", C.c_void_p), ("n_res", C.c_int),
                ("res_type", C.c_void_p), ("res_kf", C.c_void_p), ("res_pt", C.c_void_p), ("res_uv", C.c_void_p),
                ("res_sigma", C.c_void_p), ("res_active", C.c_void_p), ("calib_l", C.c_double * 4), ("calib_r", C.c_double * 4),
                ("T_rl", C.c_double * 7)]


class _SBAResult(C.Structure):
    _fields_ = [("xyz_out", C.c_void_p), ("chi2_last_eval", C.c_void_p), ("depthpos_last_eval", C.c_void_p),
                ("iterations", C.c_int), ("num_successful_steps", C.c_int), ("initial_cost", C.c_double),
                ("final_cost", C.c_double), ("termination", C.c_int)]


def structure_ba(prob, opts=None, res_active=None):
    """prob: dict from ov2slam_amd.synth.make_structure_problem.  One ceres::Solve of structureOnlyBA."""
    opts = opts or ba_default_options(max_iter=10, function_tolerance=1e-3)
    keep = {}

    def arr(name, dt):
        a = np.ascontiguousarray(prob[name], dt); keep[name] = a
        return a.ctypes.data

    P = _SBAProblem()
    P.n_kf, P.n_pts, P.n_res = int(prob["n_kf"]), int(prob["n_pts"]), int(prob["n_res"])
    P.poses = arr("poses", np.float64); P.xyz = arr("xyz", np.float64)
    P.res_type = arr("res_type", np.uint8); P.res_kf = arr("res_kf", np.int32); P.res_pt = arr("res_pt", np.int32)
    P.res_uv = arr("res_uv", np.float64); P.res_sigma = arr("res_sigma", np.float64)
    if res_active is not None:
        ra = np.ascontiguousarray(res_active, np.uint8); keep["ra"] = ra
        P.res_active = ra.ctypes.data
    for i in range(4):
        P.calib_l[i] = float(prob["calib_l"][i]); P.calib_r[i] = float(prob["calib_r"][i])
    for i in range(7):
        P.T_rl[i] = float(prob["T_rl"][i])
    xyz = np.zeros((P.n_pts, 3)); chi2 = np.full(P.n_res, np.nan); dpos = np.zeros(P.n_res, np.uint8)
    R = _SBAResult()
    R.xyz_out = xyz.ctypes.data; R.chi2_last_eval = chi2.ctypes.data; R.depthpos_last_eval = dpos.ctypes.data
    rc = lib().orc_structure_ba(C.byref(P), C.byref(opts), C.byref(R))
    assert rc == 0, rc
    return dict(xyz=xyz, chi2=chi2, depthpos=dpos, iterations=R.iterations, num_successful_steps=R.num_successful_steps,
                initial_cost=R.initial_cost, final_cost=R.final_cost, termination=R.termination)


def xyz_residual(rtype, calib_l, calib_r, T_rl, pose, X, uv, sigma, want_jac=True):
    r = np.zeros(2); J = np.zeros((2, 3)); chi2 = C.c_double(0)
    f = lib().orc_xyz_residual
    f.restype = C.c_int
    a = lambda v: np.ascontiguousarray(v, np.float64)
    cl, cr, T, p, x, u = a(calib_l), a(calib_r), a(T_rl), a(pose), a(X), a(uv)
    dp = f(int(rtype), _p(cl), _p(cr), _p(T), _p(p), _p(x), _p(u), C.c_double(sigma), _p(r), _p(J) if want_jac else None, C.byref(chi2))
    return r, J, chi2.value, bool(dp)


# ---------------------------------------------------------------- BA over 3-D points with variable poses (xyz_ba.c)
class _XYZBAProblem(C.Structure):
    _fields_ = [("n_kf", C.c_int), ("poses", C.c_void_p), ("kf_const", C.c_void_p), ("n_pts", C.c_int), ("xyz", C.c_void_p),
                ("n_res", C.c_int), ("res_type", C.c_void_p), ("res_kf", C.c_void_p), ("res_pt", C.c_void_p),
                ("res_uv", C.c_void_p), ("res_sigma", C.c_void_p), ("res_active", C.c_void_p),
                ("calib_l", C.c_double * 4), ("calib_r", C.c_double * 4), ("T_rl", C.c_double * 7)]


class _XYZBAResult(C.Structure):
    _fields_ = [("poses_out", C.c_void_p), ("xyz_out", C.c_void_p), ("chi2_last_eval", C.c_void_p), ("depthpos_last_eval", C.c_void_p),
                ("iterations", C.c_int), ("num_successful_steps", C.c_int), ("initial_cost", C.c_double),
                ("final_cost", C.c_double), ("termination", C.c_int)]


def xyz_ba_solve(prob, opts=None, res_active=None, chi2_init=None, depthpos_init=None):
    """prob: dict from ov2slam_amd.synth.make_xyz_ba_problem.  One ceres::Solve of the buse_inv_depth: 0 branch."""
    opts = opts or ba_default_options()
    keep = {}

    def arr(name, dt):
        a = np.ascontiguousarray(prob[name], dt); keep[name] = a
        return a.ctypes.data

    P = _XYZBAProblem()
    P.n_kf, P.n_pts, P.n_res = int(prob["n_kf"]), int(prob["n_pts"]), int(prob["n_res"])
    P.poses = arr("poses", np.float64); P.xyz = arr("xyz", np.float64); P.kf_const = arr("kf_const", np.uint8)
    P.res_type = arr("res_type", np.uint8); P.res_kf = arr("res_kf", np.int32); P.res_pt = arr("res_pt", np.int32)
    P.res_uv = arr("res_uv", np.float64); P.res_sigma = arr("res_sigma", np.float64)
    if res_active is not None:
        ra = np.ascontiguousarray(res_active, np.uint8); keep["ra"] = ra
        P.res_active = ra.ctypes.data
    for i in range(4):
        P.calib_l[i] = float(prob["calib_l"][i]); P.calib_r[i] = float(prob["calib_r"][i])
    for i in range(7):
        P.T_rl[i] = float(prob["T_rl"][i])
    poses = np.zeros((P.n_kf, 7)); xyz = np.zeros((P.n_pts, 3))
    chi2 = np.full(P.n_res, np.nan) if chi2_init is None else np.array(chi2_init, np.float64, copy=True)
    dpos = np.zeros(P.n_res, np.uint8) if depthpos_init is None else np.array(depthpos_init, np.uint8, copy=True)
    R = _XYZBAResult()
    R.poses_out = poses.ctypes.data; R.xyz_out = xyz.ctypes.data; R.chi2_last_eval = chi2.ctypes.data; R.depthpos_last_eval = dpos.ctypes.data
    rc = lib().orc_xyzba_solve(C.byref(P), C.byref(opts), C.byref(R))
    assert rc == 0, rc
    return dict(poses=poses, xyz=xyz, chi2=chi2, depthpos=dpos, iterations=R.iterations, num_successful_steps=R.num_successful_steps,
                initial_cost=R.initial_cost, final_cost=R.final_cost, termination=R.termination)


def xyzba_residual(rtype, calib_l, calib_r, T_rl, pose, X, uv, sigma):
    """-> (r (2,), Jp (2,6), Jx (2,3), chi2, depth positive)"""
    r = np.zeros(2); Jp = np.zeros((2, 6)); Jx = np.zeros((2, 3)); chi2 = C.c_double(0)
    f = lib().orc_xyzba_residual
    f.restype = C.c_int
    a = lambda v: np.ascontiguousarray(v, np.float64)
    cl, cr, T, p, x, u = a(calib_l), a(calib_r), a(T_rl), a(pose), a(X), a(uv)
    dp = f(int(rtype), _p(cl), _p(cr), _p(T), _p(p), _p(x), _p(u), C.c_double(sigma), _p(r), _p(Jp), _p(Jx), C.byref(chi2))
    return r, Jp, Jx, chi2.value, bool(dp)
