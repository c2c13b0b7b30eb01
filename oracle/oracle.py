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


# ---------------------------------------------------------------- detection
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
