"""Host-side mirror of the reference's front-end operator interface, on top of the
C ABI (include/ov2slam_hip.h).  Names, argument meaning and failure behaviour follow
/root/reference/include/feature_tracker.hpp and feature_extractor.hpp so that the
parity tests read like calls into the reference:

    reference (C++ / OpenCV)                              here
    ----------------------------------------------------  ---------------------------------
    cv::buildOpticalFlowPyramid(img, pyr, win, lvl)       Pyramid(ctx, w, h, win, lvl).build(img)
    FeatureTracker::fbKltTracking(prevpyr, curpyr, ...)   FeatureTracker.fbKltTracking(...)
    FeatureExtractor::detectGridFAST(im, cell, kps, roi)  FeatureExtractor.detectGridFAST(...)
    FeatureExtractor::detectSingleScale(im, cell, ...)    FeatureExtractor.detectSingleScale(...)

numpy arrays are the host buffers; nothing here computes on the CPU.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib as L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One HIP stream + scratch (ov2_ctx).  One per calling thread."""

    def __init__(self, device=0, stream=None):
        self.lib = L.load()
        h = C.c_void_p()
        if stream is None:
            L.check(self.lib.ov2_ctx_create(device, C.byref(h)))
        else:
            L.check(self.lib.ov2_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(h)))
        self.h = h
        self.device = device
        # objects that hold device memory / streams of this context: closed before the context itself
        self._children = weakref.WeakSet()

    def sync(self):
        L.check(self.lib.ov2_ctx_sync(self.h))

    def set_option(self, option, value):
        """ov2_ctx_set_option, e.g. (L.OV2_OPT_SOBEL_DY_ORDER, L.OV2_SOBEL_DY_EXACT_SUM)."""
        L.check(self.lib.ov2_ctx_set_option(self.h, int(option), int(value)))

    def get_option(self, option):
        v = C.c_int(0)
        L.check(self.lib.ov2_ctx_get_option(self.h, int(option), C.byref(v)))
        return v.value

    def options(self, **kw):
        """Context manager: pin kernel paths for a block (tests, A/B runs), e.g. ctx.options(lk_impl=L.OV2_LK_IMPL_LANE3);
        names are the OV2_OPT_* suffixes in lower case.  The previous values come back on exit."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            ids = {k: getattr(L, "OV2_OPT_" + k.upper()) for k in kw}
            old = {k: self.get_option(ids[k]) for k in kw}
            try:
                for k, v in kw.items():
                    self.set_option(ids[k], v)
                yield self
            finally:
                for k, v in old.items():
                    self.set_option(ids[k], v)
        return _cm()

    @property
    def stream(self):
        return self.lib.ov2_ctx_stream(self.h)

    def close(self):
        if getattr(self, "h", None):
            for c in list(self._children):
                c.close()
            self.lib.ov2_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pyramid:
    """Device-resident result of cv::buildOpticalFlowPyramid (image + Scharr
    derivative per level, padded by `win`), for `batch` images of equal size."""

    def __init__(self, ctx, w, h, win=9, max_level=3, batch=1):
        self.ctx, self.lib = ctx, ctx.lib
        self.w, self.h, self.win, self.max_level, self.batch = w, h, win, max_level, batch
        hp = C.c_void_p()
        L.check(self.lib.ov2_pyr_create(ctx.h, w, h, win, max_level, batch, C.byref(hp)))
        self.h_pyr = hp
        ctx._children.add(self)

    @property
    def levels(self):
        return self.lib.ov2_pyr_levels(self.h_pyr)

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        L.check(self.lib.ov2_pyr_level_size(self.h_pyr, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def build(self, img):
        """img: (h,w) or (batch,h,w) uint8 numpy array (host).  Asynchronous."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        if img.ndim == 2:
            img = img[None]
        assert img.shape == (self.batch, self.h, self.w), img.shape
        self._keep = img
        L.check(self.lib.ov2_pyr_build_h(self.ctx.h, self.h_pyr, _ptr(img), self.w, self.w * self.h))
        return self

    def build_from_device(self, dev_ptr, stride=None, batch_stride=None):
        stride = stride or self.w
        batch_stride = batch_stride or stride * self.h
        L.check(self.lib.ov2_pyr_build_d(self.ctx.h, self.h_pyr, C.c_void_p(dev_ptr), stride, batch_stride))
        return self

    def build_clahe(self, img, clip_limit, tiles_x, tiles_y):
        """preprocessImage from a host image (batch 1): CLAHE written straight into level 0 + coarser levels."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        assert img.shape == (self.h, self.w) and self.batch == 1
        self._keep = img
        L.check(self.lib.ov2_pyr_build_clahe_h(self.ctx.h, self.h_pyr, _ptr(img), self.w, float(clip_limit), int(tiles_x), int(tiles_y)))
        return self

    def build_clahe_batch(self, imgs, clip_limit, tiles_x, tiles_y):
        """preprocessImage of len(imgs) host images into items [0, len(imgs)) of this batch pyramid in one enqueue (ov2_pyr_build_clahe_hb);
        clip_limit < 0: no CLAHE"""
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in imgs]
        for im in imgs:
            assert im.shape == (self.h, self.w)
        self._keep = imgs
        ptrs = (C.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs])
        L.check(self.lib.ov2_pyr_build_clahe_hb(self.ctx.h, self.h_pyr, len(imgs), ptrs, self.w, float(clip_limit), int(tiles_x), int(tiles_y)))
        return self

    def build_clahe_from_device(self, dev_ptr, clip_limit, tiles_x, tiles_y, stride=None, batch_stride=None):
        """preprocessImage: CLAHE written straight into level 0, then the coarser levels (one call)."""
        stride = stride or self.w
        batch_stride = batch_stride or stride * self.h
        L.check(self.lib.ov2_pyr_build_clahe_d(self.ctx.h, self.h_pyr, C.c_void_p(dev_ptr), stride, batch_stride,
                                               float(clip_limit), int(tiles_x), int(tiles_y)))
        return self

    def download(self, level, b=0, padded=False):
        w, h = self.level_size(level)
        if padded:
            w, h = w + 2 * self.win, h + 2 * self.win
        img = np.empty((h, w), np.uint8)
        der = np.empty((h, w, 2), np.int16)
        fn = self.lib.ov2_pyr_download_padded if padded else self.lib.ov2_pyr_download
        L.check(fn(self.ctx.h, self.h_pyr, b, level, _ptr(img), _ptr(der)))
        return img, der

    @property
    def algorithmic_bytes(self):
        return self.lib.ov2_pyr_algorithmic_bytes(self.h_pyr)

    def close(self):
        if getattr(self, "h_pyr", None):
            self.lib.ov2_pyr_destroy(self.h_pyr)
            self.h_pyr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CLAHE:
    """Mirror of the cv::CLAHE handle the reference creates at src/ov2slam.cpp:85-89 and applies at
    src/visual_front_end.cpp:1159 / src/mapper.cpp:76:  createCLAHE(clipLimit, tileGridSize).apply(src)."""

    def __init__(self, ctx, clipLimit=3.0, tileGridSize=(15, 9)):
        self.ctx, self.lib = ctx, ctx.lib
        self.clipLimit = float(clipLimit)
        self.tiles_x, self.tiles_y = int(tileGridSize[0]), int(tileGridSize[1])

    def apply(self, src):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        h, w = src.shape
        dst = np.empty_like(src)
        L.check(self.lib.ov2_clahe_h(self.ctx.h, _ptr(src), w, h, w, self.clipLimit, self.tiles_x, self.tiles_y, _ptr(dst), w))
        return dst


class FeatureTracker:
    """Mirror of /root/reference/include/feature_tracker.hpp:36-56.

    FeatureTracker(nmax_iter, fmax_px_precision) stores the KLT convergence criteria
    (cv::TermCriteria(COUNT+EPS, nmax_iter, fmax_px_precision), :39-40)."""

    def __init__(self, ctx, nmax_iter=30, fmax_px_precision=0.01):
        self.ctx, self.lib = ctx, ctx.lib
        self.nmax_iter = int(nmax_iter)
        self.fmax_px_precision = float(np.float32(fmax_px_precision))

    def fbKltTracking(self, vprevpyr, vcurpyr, nwinsize, nbpyrlvl, ferr, fmax_fbklt_dist, vkps, vpriorkps,
                      return_stats=False):
        """src/feature_tracker.cpp:35-137.  vkps / vpriorkps: (n,2) float32.
        Returns (vpriorkps_out, vkpstatus) -- the reference updates vpriorkps in place
        and push_back()s into vkpstatus; with no keypoints it returns untouched
        outputs (:43-46), here an empty status list."""
        kps = np.ascontiguousarray(vkps, dtype=np.float32).reshape(-1, 2)
        pri = np.array(vpriorkps, dtype=np.float32, copy=True).reshape(-1, 2)
        n = kps.shape[0]
        assert pri.shape[0] == n
        status = np.zeros(n, np.uint8)
        stats = (C.c_longlong * 2)(0, 0)
        L.check(self.lib.ov2_fb_klt(self.ctx.h, vprevpyr.h_pyr, vcurpyr.h_pyr, int(nwinsize), int(nbpyrlvl),
                                    self.nmax_iter, self.fmax_px_precision, float(ferr), float(fmax_fbklt_dist),
                                    _ptr(kps), _ptr(pri), n, _ptr(status), stats))
        if return_stats:
            return pri, status.astype(bool), (int(stats[0]), int(stats[1]))
        return pri, status.astype(bool)

    def getLineMinSAD(self, leftpyr, rightpyr, level, pts, nwinsize=7, bgoleft=True):
        """src/feature_tracker.cpp:138-206 for an (n,2) array of points already scaled to `level`
        (the reference passes vleftpyr.at(2 * nklt_pyr_lvl_) and kp.px_ * downpyrcoef, map_manager.cpp:427-431).
        Returns (xprior (n,) float32, -1 = none; l1err (n,) float32)."""
        pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
        n = len(pts)
        xp = np.full(n, -1, np.float32); err = np.full(n, 255, np.float32)
        L.check(self.lib.ov2_line_min_sad(self.ctx.h, leftpyr.h_pyr, rightpyr.h_pyr, int(level), int(nwinsize), int(bool(bgoleft)),
                                          _ptr(pts), n, _ptr(xp), _ptr(err)))
        return xp, err

    def calcOpticalFlowPyrLK(self, prevpyr, nextpyr, prevpts, nextpts, nwinsize, maxlevel,
                             flags=L.OV2_LK_USE_INITIAL_FLOW | L.OV2_LK_GET_MIN_EIGENVALS):
        """One cv::calcOpticalFlowPyrLK call (src/feature_tracker.cpp:66-69)."""
        p0 = np.ascontiguousarray(prevpts, dtype=np.float32).reshape(-1, 2)
        p1 = np.array(nextpts, dtype=np.float32, copy=True).reshape(-1, 2)
        n = p0.shape[0]
        status = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        iters = np.zeros(n, np.int32)
        L.check(self.lib.ov2_lk_track(self.ctx.h, prevpyr.h_pyr, nextpyr.h_pyr, int(nwinsize), int(maxlevel),
                                      self.nmax_iter, self.fmax_px_precision, int(flags),
                                      _ptr(p0), _ptr(p1), n, _ptr(status), _ptr(err), _ptr(iters)))
        return p1, status, err, iters


class _PyrView:
    """Borrowed ov2_pyr handle (owned by a Tracker): usable wherever a Pyramid is expected."""

    def __init__(self, ctx, h_pyr, w, h, win):
        self.ctx, self.lib, self.h_pyr, self.w, self.h, self.win, self.batch = ctx, ctx.lib, h_pyr, w, h, win, 1

    levels = Pyramid.levels
    level_size = Pyramid.level_size
    download = Pyramid.download


class VisualFrontEndTracker:
    """The per-frame GPU half of the reference's VisualFrontEnd (src/visual_front_end.cpp): it owns prev_pyr_ / cur_pyr_
    and performs preprocessImage (:1143-1177) + kltTracking (:132-275) per frame -- one H2D of the frame, one H2D of the
    keypoint block, ONE LK launch for both fbKltTracking calls and the retry of lost prior tracks, one D2H, one sync
    (ov2_tracker_*).  Parameters are the SlamParams fields of the same names."""

    def __init__(self, ctx, w, h, nklt_win_size=9, nklt_pyr_lvl=3, nmax_iter=30, fmax_px_precision=0.01, nklt_err=30.0,
                 fmax_fbklt_dist=0.5, use_clahe=True, fclahe_val=3.0, nbmaxkps=512, use_graph=True, prior_pyr_lvl=1):
        self.ctx, self.lib = ctx, ctx.lib
        self.w, self.h, self.win = int(w), int(h), int(nklt_win_size)
        cfg = L.TrackerConfig(self.w, self.h, self.win, int(nklt_pyr_lvl), int(prior_pyr_lvl), int(nmax_iter),
                              float(np.float32(fmax_px_precision)), float(nklt_err), float(fmax_fbklt_dist),
                              int(bool(use_clahe)), float(fclahe_val), self.w // 50, self.h // 50, int(nbmaxkps),
                              int(bool(use_graph)))
        ht = C.c_void_p()
        L.check(self.lib.ov2_tracker_create(ctx.h, C.byref(cfg), C.byref(ht)))
        self.h_trk = ht
        ctx._children.add(self)
        self.nbmaxkps = int(nbmaxkps)
        stride = C.c_int()
        ptr = self.lib.ov2_tracker_image_buffer(ht, C.byref(stride))
        self.stride = stride.value
        # numpy view of the pinned staging image: write the next frame here to skip the host-side copy
        self.image_buffer = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(self.h, self.stride))

    def _pts(self, vkps, vpriors, vhasprior):
        kps = np.ascontiguousarray(vkps, dtype=np.float32).reshape(-1, 2)
        pri = np.ascontiguousarray(vpriors, dtype=np.float32).reshape(-1, 2)
        n = len(kps)
        if len(pri) != n:
            raise ValueError("vkps and vpriors differ in length")
        hp = None if vhasprior is None else np.ascontiguousarray(vhasprior, dtype=np.uint8).reshape(-1)
        if hp is not None and len(hp) != n:
            raise ValueError("vhasprior has %d entries for %d keypoints" % (len(hp), n))
        return kps, pri, hp, n

    def _img(self, img_raw):
        img = img_raw if img_raw is self.image_buffer else np.ascontiguousarray(img_raw, dtype=np.uint8)
        if img.ndim != 2 or img.shape[0] != self.h or img.shape[1] < self.w:       # the C side reads h rows of w bytes
            raise ValueError("image shape %s does not match the tracker's %dx%d" % (img.shape, self.w, self.h))
        return img

    def preprocessImage(self, img_raw):
        """Asynchronous (returns after the enqueue)."""
        img = self._img(img_raw)
        L.check(self.lib.ov2_tracker_preprocess(self.h_trk, _ptr(img), img.strides[0]))

    def kltTracking(self, vkps, vpriors, vhasprior, klt_use_prior=True):
        """-> (tracked px (n,2), status bits (n,) uint8 [bit0 tracked, bit1 re-tracked on the full pyramid], bp3preq)."""
        kps, pri, hp, n = self._pts(vkps, vpriors, vhasprior)
        out = np.zeros((n, 2), np.float32); st = np.zeros(n, np.uint8); p3p = C.c_int(0)
        L.check(self.lib.ov2_tracker_klt(self.h_trk, _ptr(kps), _ptr(pri), _ptr(hp), n, int(bool(klt_use_prior)),
                                         _ptr(out), _ptr(st), C.byref(p3p)))
        return out, st, bool(p3p.value)

    def trackFrame(self, img_raw, vkps, vpriors, vhasprior, klt_use_prior=True):
        """preprocessImage + kltTracking in one enqueue (graph replay when enabled)."""
        img = self._img(img_raw)
        kps, pri, hp, n = self._pts(vkps, vpriors, vhasprior)
        out = np.zeros((n, 2), np.float32); st = np.zeros(n, np.uint8); p3p = C.c_int(0)
        L.check(self.lib.ov2_tracker_track_frame(self.h_trk, _ptr(img), img.strides[0], _ptr(kps), _ptr(pri), _ptr(hp), n,
                                                 int(bool(klt_use_prior)), _ptr(out), _ptr(st), C.byref(p3p)))
        return out, st, bool(p3p.value)

    def setCalibration(self, calib):
        """calib: CameraCalibration.  From now on every kltTracking / trackFrame also computes Frame::computeKeypoint (undistorted
        pixel + bearing vector) of its output positions in the same enqueue: lastKeypoints().  Call right after construction."""
        L.check(self.lib.ov2_tracker_set_calibration(self.h_trk, calib.model, _ptr(calib.K), _ptr(calib.D) if calib.D is not None else None,
                                                     0 if calib.D is None else len(calib.D), _ptr(calib.iK)))

    def lastKeypoints(self, n, want_bv=True):
        """(unpx (n,2) float32, bv (n,3) float64 or None) of the n keypoints of the last kltTracking / trackFrame call."""
        unpx = np.empty((n, 2), np.float32)
        bv = np.empty((n, 3), np.float64) if want_bv else None
        L.check(self.lib.ov2_tracker_last_keypoints(self.h_trk, int(n), _ptr(unpx), _ptr(bv) if want_bv else None))
        return unpx, bv

    @property
    def cur_pyr(self):
        return _PyrView(self.ctx, C.c_void_p(self.lib.ov2_tracker_cur_pyr(self.h_trk)), self.w, self.h, self.win)

    @property
    def prev_pyr(self):
        return _PyrView(self.ctx, C.c_void_p(self.lib.ov2_tracker_prev_pyr(self.h_trk)), self.w, self.h, self.win)

    @property
    def uses_graph(self):
        return bool(self.lib.ov2_tracker_uses_graph(self.h_trk))

    def close(self):
        if getattr(self, "h_trk", None):
            self.image_buffer = None
            self.lib.ov2_tracker_destroy(self.h_trk)
            self.h_trk = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LockstepTracker:
    """`batch` camera streams in lock-step (ov2_btracker_*, csrc/trackb.hip): the offline batch-of-sequences form of
    VisualFrontEndTracker.  Every call advances items [0, n_active) by one frame with ONE enqueue over all of them; results per
    item are bit-identical to a VisualFrontEndTracker fed the same frames and keypoints.  Point arrays are (batch, n_max, 2) with
    n[b] valid rows per item."""

    def __init__(self, ctx, batch, w, h, nklt_win_size=9, nklt_pyr_lvl=3, nmax_iter=30, fmax_px_precision=0.01, nklt_err=30.0,
                 fmax_fbklt_dist=0.5, use_clahe=True, fclahe_val=3.0, nbmaxkps=512, prior_pyr_lvl=1):
        self.ctx, self.lib = ctx, ctx.lib
        self.batch, self.w, self.h, self.win, self.n_max = int(batch), int(w), int(h), int(nklt_win_size), int(nbmaxkps)
        cfg = L.TrackerConfig(self.w, self.h, self.win, int(nklt_pyr_lvl), int(prior_pyr_lvl), int(nmax_iter),
                              float(np.float32(fmax_px_precision)), float(nklt_err), float(fmax_fbklt_dist),
                              int(bool(use_clahe)), float(fclahe_val), self.w // 50, self.h // 50, self.n_max, 0)
        ht = C.c_void_p()
        L.check(self.lib.ov2_btracker_create(ctx.h, C.byref(cfg), self.batch, C.byref(ht)))
        self.h_trk = ht
        ctx._children.add(self)
        stride = C.c_int()
        self.image_buffers = []                       # [which][item] -> numpy view of the pinned slot
        for which in range(3):
            row = []
            for b in range(self.batch):
                ptr = self.lib.ov2_btracker_image_buffer(ht, which, b, C.byref(stride))
                row.append(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(self.h, stride.value)))
            self.image_buffers.append(row)
        self.stride = stride.value

    def setCalibration(self, calib):
        L.check(self.lib.ov2_btracker_set_calibration(self.h_trk, calib.model, _ptr(calib.K), _ptr(calib.D) if calib.D is not None else None,
                                                      0 if calib.D is None else len(calib.D), _ptr(calib.iK)))

    def upload(self, which, n_active):
        """start the H2D of staging set `which` (already filled through image_buffers[which]) on the copy stream"""
        L.check(self.lib.ov2_btracker_upload(self.h_trk, int(which), int(n_active)))

    def prepare(self, which, n_active):
        """preprocessImage of staging set `which` for the NEXT trackFrame, on the tracker's prep stream"""
        L.check(self.lib.ov2_btracker_prepare(self.h_trk, int(which), int(n_active)))

    def trackFrame(self, imgs, kps, pri, hasprior, n, klt_use_prior=True):
        """imgs: list of n_active (h, >=w) uint8 arrays of one row pitch (or pinned slots of one staging set); kps / pri:
        (batch, n_max, 2) float32; hasprior: (batch, n_max) uint8 or None; n: per-item counts (len n_active).
        -> (out (batch, n_max, 2), status bits (batch, n_max), p3p_req (n_active,) bool)"""
        na = len(imgs)
        imgs = [im if im.dtype == np.uint8 and im.strides[1] == 1 else np.ascontiguousarray(im, np.uint8) for im in imgs]
        stride = imgs[0].strides[0]
        for im in imgs:
            if im.ndim != 2 or im.shape[0] != self.h or im.shape[1] < self.w or im.strides[0] != stride:
                raise ValueError("frames must be %d rows of >= %d bytes with one common row pitch" % (self.h, self.w))
        ptrs = (C.c_void_p * na)(*[im.ctypes.data for im in imgs])
        kps = np.ascontiguousarray(kps, np.float32).reshape(self.batch, self.n_max, 2)
        pri = np.ascontiguousarray(pri, np.float32).reshape(self.batch, self.n_max, 2)
        hp = None if hasprior is None else np.ascontiguousarray(hasprior, np.uint8).reshape(self.batch, self.n_max)
        nn = np.ascontiguousarray(n, np.int32)
        if len(nn) != na:
            raise ValueError("one keypoint count per active item")
        out = np.zeros((self.batch, self.n_max, 2), np.float32); st = np.zeros((self.batch, self.n_max), np.uint8)
        p3p = np.zeros(na, np.int32)
        L.check(self.lib.ov2_btracker_track_frame(self.h_trk, na, ptrs, stride, _ptr(kps), _ptr(pri), _ptr(hp), _ptr(nn),
                                                  int(bool(klt_use_prior)), _ptr(out), _ptr(st), _ptr(p3p)))
        return out, st, p3p.astype(bool)

    def trackFrameBegin(self, imgs, kps, pri, hasprior, n, klt_use_prior=True):
        """first half of trackFrame: enqueue the step and return (upload / prepare of the frames to come go between the halves)"""
        na = len(imgs)
        imgs = [im if im.dtype == np.uint8 and im.strides[1] == 1 else np.ascontiguousarray(im, np.uint8) for im in imgs]
        stride = imgs[0].strides[0]
        ptrs = (C.c_void_p * na)(*[im.ctypes.data for im in imgs])
        kps = np.ascontiguousarray(kps, np.float32).reshape(self.batch, self.n_max, 2)
        pri = np.ascontiguousarray(pri, np.float32).reshape(self.batch, self.n_max, 2)
        hp = None if hasprior is None else np.ascontiguousarray(hasprior, np.uint8).reshape(self.batch, self.n_max)
        nn = np.ascontiguousarray(n, np.int32)
        self._pending = (imgs, kps, pri, hp, nn)                                  # (kps / hasprior are re-read by trackFrameEnd: keep them alive)
        L.check(self.lib.ov2_btracker_track_frame_begin(self.h_trk, na, ptrs, stride, _ptr(kps), _ptr(pri), _ptr(hp), _ptr(nn), int(bool(klt_use_prior))))

    def trackFrameEnd(self):
        """second half: wait, results, p3p rule -> (out, status bits, p3p_req) as trackFrame"""
        na = len(self._pending[0])
        out = np.zeros((self.batch, self.n_max, 2), np.float32); st = np.zeros((self.batch, self.n_max), np.uint8)
        p3p = np.zeros(na, np.int32)
        L.check(self.lib.ov2_btracker_track_frame_end(self.h_trk, _ptr(out), _ptr(st), _ptr(p3p)))
        self._pending = None
        return out, st, p3p.astype(bool)

    def lastKeypoints(self, item, n, want_bv=True):
        unpx = np.empty((n, 2), np.float32)
        bv = np.empty((n, 3), np.float64) if want_bv else None
        L.check(self.lib.ov2_btracker_last_keypoints(self.h_trk, int(item), int(n), _ptr(unpx), _ptr(bv) if want_bv else None))
        return unpx, bv

    def detectSingleScale(self, n_active, ncellsize, cur, ncur, roi, quality, subpix=True):
        """MapManager::extractKeypoints of items [0, n_active) on their current frames.  cur: (batch, n_max, 2); ncur: counts;
        quality: float64 array, one dmaxquality_ per item, updated in place.  -> list of (k, 2) arrays"""
        cap = max(1, 2 * (self.w // ncellsize) * (self.h // ncellsize))
        cur = np.ascontiguousarray(cur, np.float32).reshape(self.batch, self.n_max, 2)
        nc = np.ascontiguousarray(ncur, np.int32)
        q = np.ascontiguousarray(quality, np.float64)
        assert q is quality or np.shares_memory(q, quality), "quality must be a contiguous float64 array (updated in place)"
        out = np.zeros((n_active, cap, 2), np.float32); on = np.zeros(n_active, np.int32)
        r = (C.c_int * 4)(*[int(v) for v in roi])
        L.check(self.lib.ov2_btracker_detect_singlescale(self.h_trk, int(n_active), int(ncellsize), _ptr(cur), _ptr(nc), r, _ptr(q),
                                                         1 if subpix else 0, _ptr(out), cap, _ptr(on)))
        return [out[b, :on[b]].copy() for b in range(n_active)]

    def detectGridFAST(self, n_active, ncellsize, cur, ncur, fast_th, mask_mode=L.OV2_MASK_AS_EXECUTED, subpix=True):
        cap = max(1, (self.w // ncellsize) * (self.h // ncellsize))
        cur = np.ascontiguousarray(cur, np.float32).reshape(self.batch, self.n_max, 2)
        nc = np.ascontiguousarray(ncur, np.int32)
        t = np.ascontiguousarray(fast_th, np.int32)
        assert t is fast_th or np.shares_memory(t, fast_th), "fast_th must be a contiguous int32 array (updated in place)"
        out = np.zeros((n_active, cap, 2), np.float32); on = np.zeros(n_active, np.int32)
        L.check(self.lib.ov2_btracker_detect_grid_fast(self.h_trk, int(n_active), int(ncellsize), _ptr(cur), _ptr(nc), _ptr(t), int(mask_mode),
                                                       1 if subpix else 0, _ptr(out), cap, _ptr(on)))
        return [out[b, :on[b]].copy() for b in range(n_active)]

    @property
    def cur_pyr(self):
        """the current frame's pyramids, all items (e.g. `left` of stereo.stereo_match_batch_arrays)"""
        v = _PyrView(self.ctx, C.c_void_p(self.lib.ov2_btracker_cur_pyr(self.h_trk)), self.w, self.h, self.win)
        v.batch = self.batch
        return v

    def cur_item(self, item):
        return _PyrView(self.ctx, C.c_void_p(self.lib.ov2_btracker_cur_item(self.h_trk, int(item))), self.w, self.h, self.win)

    def prev_item(self, item):
        return _PyrView(self.ctx, C.c_void_p(self.lib.ov2_btracker_prev_item(self.h_trk, int(item))), self.w, self.h, self.win)

    def close(self):
        if getattr(self, "h_trk", None):
            self.image_buffers = None
            self.lib.ov2_btracker_destroy(self.h_trk)
            self.h_trk = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FeatureExtractor:
    """Mirror of /root/reference/include/feature_extractor.hpp:30-54: holds the
    adaptive thresholds nfast_th_ and dmaxquality_ that the two grid detectors update."""

    def __init__(self, ctx, nfast_th=10, dmaxquality=0.001, mask_mode=L.OV2_MASK_AS_EXECUTED):
        self.ctx, self.lib = ctx, ctx.lib
        self.nfast_th_ = int(nfast_th)
        self.dmaxquality_ = float(dmaxquality)
        self.mask_mode = mask_mode

    def detectGridFAST(self, im, ncellsize, vcurkps, roi=None, subpix=True):
        """src/feature_extractor.cpp:443-570 (roi is unused there too)."""
        im = np.ascontiguousarray(im, dtype=np.uint8)
        if im.size == 0:
            return np.zeros((0, 2), np.float32)               # :446-449
        h, w = im.shape
        cur = np.ascontiguousarray(vcurkps, dtype=np.float32).reshape(-1, 2)
        cap = max(1, (w // ncellsize) * (h // ncellsize))
        out = np.zeros((cap, 2), np.float32)
        n = C.c_int(0)
        th = C.c_int(self.nfast_th_)
        L.check(self.lib.ov2_detect_grid_fast(self.ctx.h, _ptr(im), w, h, w, int(ncellsize), _ptr(cur), cur.shape[0],
                                              C.byref(th), self.mask_mode, int(bool(subpix)), _ptr(out), C.byref(n)))
        self.nfast_th_ = th.value
        return out[:n.value].copy()

    def detectSingleScale(self, im, ncellsize, vcurkps, roi, subpix=True):
        """src/feature_extractor.cpp:288-440.  roi = (x, y, width, height)."""
        im = np.ascontiguousarray(im, dtype=np.uint8)
        if im.size == 0:
            return np.zeros((0, 2), np.float32)               # :291-294
        h, w = im.shape
        cur = np.ascontiguousarray(vcurkps, dtype=np.float32).reshape(-1, 2)
        cap = max(1, 2 * (w // ncellsize) * (h // ncellsize))
        out = np.zeros((cap, 2), np.float32)
        n = C.c_int(0)
        q = C.c_double(self.dmaxquality_)
        roi_a = (C.c_int * 4)(*[int(v) for v in roi])
        L.check(self.lib.ov2_detect_singlescale(self.ctx.h, _ptr(im), w, h, w, int(ncellsize), _ptr(cur), cur.shape[0],
                                                roi_a, C.byref(q), int(bool(subpix)), _ptr(out), C.byref(n)))
        self.dmaxquality_ = q.value
        return out[:n.value].copy()

    def detectSingleScalePyr(self, pyr, ncellsize, vcurkps, roi, subpix=True, item=0):
        """detectSingleScale on level 0 of a device-resident pyramid (e.g. VisualFrontEndTracker.cur_pyr: the CLAHE'd
        cur_img_ of the keyframe, src/map_manager.cpp:312-320) -- no image upload."""
        cur = np.ascontiguousarray(vcurkps, dtype=np.float32).reshape(-1, 2)
        w, h = pyr.level_size(0)
        out = np.zeros((max(1, 2 * (w // ncellsize) * (h // ncellsize)), 2), np.float32)
        n = C.c_int(0); q = C.c_double(self.dmaxquality_)
        roi_a = (C.c_int * 4)(*[int(v) for v in roi])
        L.check(self.lib.ov2_detect_singlescale_d(self.ctx.h, pyr.h_pyr, int(item), int(ncellsize), _ptr(cur), cur.shape[0],
                                                  roi_a, C.byref(q), int(bool(subpix)), _ptr(out), C.byref(n)))
        self.dmaxquality_ = q.value
        return out[:n.value].copy()

    @staticmethod
    def detectSingleScaleBatch(ctx, pyr, ncellsize, cur_xy_d, cur_cap, ncur_d, roi, quality, out_xy_d, out_cap, subpix=True):
        """ov2_detect_singlescale_batch_d: detectSingleScale on EVERY batch item of `pyr` in one call.  cur_xy_d / ncur_d /
        out_xy_d are device addresses (ints; 0 = NULL for the first two), `quality` a float64 array with one dmaxquality_ per
        item (updated in place like the member).  Returns the per-item point counts (int32 array)."""
        q = np.ascontiguousarray(quality, np.float64)
        assert q is quality or np.shares_memory(q, quality), "quality must be a contiguous float64 array (updated in place)"
        n = np.zeros(pyr.batch, np.int32)
        r = (C.c_int * 4)(*[int(v) for v in roi])
        L.check(ctx.lib.ov2_detect_singlescale_batch_d(ctx.h, pyr.h_pyr, int(ncellsize), C.c_void_p(cur_xy_d or None), int(cur_cap),
                                                       C.c_void_p(ncur_d or None), r, q.ctypes.data_as(C.c_void_p), 1 if subpix else 0,
                                                       C.c_void_p(out_xy_d), int(out_cap), n.ctypes.data_as(C.c_void_p)))
        return n

    @staticmethod
    def detectGridFASTBatch(ctx, pyr, ncellsize, cur_xy_d, cur_cap, ncur_d, fast_th, out_xy_d, out_cap, mask_mode=L.OV2_MASK_AS_EXECUTED, subpix=True):
        """ov2_detect_grid_fast_batch_d; `fast_th` an int32 array with one nfast_th_ per item (updated in place)."""
        t = np.ascontiguousarray(fast_th, np.int32)
        assert t is fast_th or np.shares_memory(t, fast_th), "fast_th must be a contiguous int32 array (updated in place)"
        n = np.zeros(pyr.batch, np.int32)
        L.check(ctx.lib.ov2_detect_grid_fast_batch_d(ctx.h, pyr.h_pyr, int(ncellsize), C.c_void_p(cur_xy_d or None), int(cur_cap),
                                                     C.c_void_p(ncur_d or None), t.ctypes.data_as(C.c_void_p), int(mask_mode), 1 if subpix else 0,
                                                     C.c_void_p(out_xy_d), int(out_cap), n.ctypes.data_as(C.c_void_p)))
        return n

    def detectGridFASTPyr(self, pyr, ncellsize, vcurkps, subpix=True, item=0):
        """detectGridFAST on level 0 of a device-resident pyramid."""
        cur = np.ascontiguousarray(vcurkps, dtype=np.float32).reshape(-1, 2)
        w, h = pyr.level_size(0)
        out = np.zeros((max(1, (w // ncellsize) * (h // ncellsize)), 2), np.float32)
        n = C.c_int(0); th = C.c_int(self.nfast_th_)
        L.check(self.lib.ov2_detect_grid_fast_d(self.ctx.h, pyr.h_pyr, int(item), int(ncellsize), _ptr(cur), cur.shape[0],
                                                C.byref(th), self.mask_mode, int(bool(subpix)), _ptr(out), C.byref(n)))
        self.nfast_th_ = th.value
        return out[:n.value].copy()

    def cornerSubPix(self, im, pts, half_win=3, max_iter=30, eps=0.01):
        im = np.ascontiguousarray(im, dtype=np.uint8)
        h, w = im.shape
        p = np.array(pts, dtype=np.float32, copy=True).reshape(-1, 2)
        if p.shape[0]:
            L.check(self.lib.ov2_corner_subpix(self.ctx.h, _ptr(im), w, h, w, _ptr(p), p.shape[0], half_win, max_iter, eps))
        return p


class CameraCalibration:
    """Mirror of the reference's CameraCalibration / Frame::computeKeypoint pair
    (/root/reference/src/camera_calibration.cpp:313-333, src/frame.cpp:246-254) for arrays of keypoints.

    model: "pinhole" | "fisheye"; K = (fx, fy, cx, cy); D = distortion coefficients or None
    (`Dcv_.empty()` -> undistortImagePoint returns the point unchanged)."""

    MODELS = {"pinhole": L.OV2_CAM_PINHOLE, "fisheye": L.OV2_CAM_FISHEYE}

    def __init__(self, ctx, model, fx, fy, cx, cy, D=None):
        self.ctx, self.lib = ctx, ctx.lib
        self.model = self.MODELS[model]
        self.K = np.array([fx, fy, cx, cy], np.float64)
        self.D = None if D is None or len(D) == 0 else np.ascontiguousarray(D, dtype=np.float64)
        Km = np.array([[fx, 0., cx], [0., fy, cy], [0., 0., 1.]])
        self.iK = np.ascontiguousarray(np.linalg.inv(Km))            # the reference's iK_ = K_.inverse()

    def computeKeypoints(self, px, want_bv=True):
        """px (n,2) float32 -> (unpx (n,2) float32, bv (n,3) float64 or None)."""
        px = np.ascontiguousarray(px, dtype=np.float32).reshape(-1, 2)
        n = len(px)
        unpx = np.empty((n, 2), np.float32)
        bv = np.empty((n, 3), np.float64) if want_bv else None
        L.check(self.lib.ov2_compute_keypoints(self.ctx.h, self.model, _ptr(self.K), _ptr(self.D) if self.D is not None else None,
                                               0 if self.D is None else len(self.D), _ptr(self.iK), _ptr(px), n, _ptr(unpx),
                                               _ptr(bv) if want_bv else None))
        return unpx, bv

    def undistortImagePoint(self, pt):
        return self.computeKeypoints(np.asarray(pt, np.float32).reshape(1, 2), want_bv=False)[0][0]
