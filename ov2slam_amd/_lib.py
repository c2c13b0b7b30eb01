"""ctypes binding of libov2slam_hip.so (the C ABI declared in include/ov2slam_hip.h).

The product path has no CPU fallback: if the HIP library is missing this module
raises at import of the symbol table, and if no GPU is visible every compute entry
point returns OV2_ENODEVICE which is surfaced as Ov2Error.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OV2SLAM_HIP_LIB") or os.path.join(_HERE, "libov2slam_hip.so")   # override: A/B builds

OV2_OK = 0
OV2_EINVAL, OV2_EHIP, OV2_ENOMEM, OV2_EUNSUPPORTED, OV2_ENODEVICE = -1, -2, -3, -4, -5
OV2_LK_USE_INITIAL_FLOW = 4
OV2_LK_GET_MIN_EIGENVALS = 8
OV2_MASK_AS_EXECUTED, OV2_MASK_INTENDED = 0, 1
OV2_CAM_PINHOLE, OV2_CAM_FISHEYE = 0, 1
OV2_OPT_SOBEL_DY_ORDER = 1
OV2_SOBEL_DY_OPENCV_ROWFILTER, OV2_SOBEL_DY_EXACT_SUM = 0, 1
# path selection / test forcing (include/ov2slam_hip.h): per context, never through the environment
OV2_OPT_LK_IMPL, OV2_OPT_TRACK_IMPL, OV2_OPT_CLAHE_STRIPS, OV2_OPT_BA_FORCE_LARGE, OV2_OPT_BA_LIN_DIRECT = 2, 3, 4, 5, 6
OV2_OPT_BA_SCHUR_CHUNK, OV2_OPT_BA_XYZ_LIN_WAVES, OV2_OPT_BA_POSE_ONLY_FUSED, OV2_OPT_BA_DETERMINISTIC, OV2_OPT_DEBUG = 7, 8, 9, 10, 11
OV2_OPT_FAST_TIE = 12
OV2_OPT_BA_TRACE = 13
OV2_OPT_LK_ACC = 14
OV2_OPT_DETECT_STRIP = 15
OV2_LK_ACC_INT64, OV2_LK_ACC_FLOAT_UI4 = 0, 1
OV2_FAST_TIE_SCAN_ORDER, OV2_FAST_TIE_LIBSTDCXX = 0, 1
OV2_LK_IMPL_AUTO, OV2_LK_IMPL_ROW, OV2_LK_IMPL_LANE3 = 0, 1, 2
OV2_TRACK_IMPL_WAVE, OV2_TRACK_IMPL_ROW = 0, 1
OV2_RES_LEFT, OV2_RES_RIGHT, OV2_RES_RIGHT_ANCH, OV2_RES_PNP = 0, 1, 2, 3


class Ov2Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ov2slam_hip error %d: %s" % (code, msg))
        self.code = code


class BAProblem(C.Structure):
    _fields_ = [
        ("n_kf", C.c_int), ("poses", C.POINTER(C.c_double)), ("kf_const", C.POINTER(C.c_uint8)),
        ("n_lm", C.c_int), ("invdepth", C.POINTER(C.c_double)), ("lm_anchor_kf", C.POINTER(C.c_int)),
        ("lm_anchor_uv", C.POINTER(C.c_double)),
        ("n_res", C.c_int), ("res_type", C.POINTER(C.c_uint8)), ("res_kf", C.POINTER(C.c_int)),
        ("res_lm", C.POINTER(C.c_int)), ("res_uv", C.POINTER(C.c_double)), ("res_sigma", C.POINTER(C.c_double)),
        ("res_active", C.POINTER(C.c_uint8)), ("res_xyz", C.POINTER(C.c_double)),
        ("calib_l", C.c_double * 4), ("calib_r", C.c_double * 4), ("T_rl", C.c_double * 7),
    ]


class BAOptions(C.Structure):
    _fields_ = [
        ("max_iter", C.c_int), ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double), ("huber_delta", C.c_double), ("initial_radius", C.c_double),
        ("max_radius", C.c_double), ("min_radius", C.c_double), ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double), ("min_relative_decrease", C.c_double), ("jacobi_scaling", C.c_int),
        ("max_consecutive_invalid_steps", C.c_int), ("max_solver_time_s", C.c_double),
    ]


class BAResult(C.Structure):
    _fields_ = [
        ("poses_out", C.POINTER(C.c_double)), ("invdepth_out", C.POINTER(C.c_double)),
        ("chi2_last_eval", C.POINTER(C.c_double)), ("depthpos_last_eval", C.POINTER(C.c_uint8)),
        ("iterations", C.c_int), ("num_successful_steps", C.c_int),
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("termination", C.c_int),
        ("solve_ms", C.c_double),
    ]


class LocalBAOptions(C.Structure):
    _fields_ = [
        ("robust_mono_th", C.c_double), ("use_robust_cost", C.c_int), ("apply_l2_after_robust", C.c_int),
        ("stop_requested", C.c_int), ("stop_flag", C.POINTER(C.c_int)), ("pass1", BAOptions), ("pass2", BAOptions),
    ]


class LocalBAResult(C.Structure):
    _fields_ = [
        ("poses_out", C.POINTER(C.c_double)), ("invdepth_out", C.POINTER(C.c_double)), ("bad_obs", C.POINTER(C.c_uint8)),
        ("bad_after_pass1", C.POINTER(C.c_uint8)), ("chi2_last_eval", C.POINTER(C.c_double)),
        ("depthpos_last_eval", C.POINTER(C.c_uint8)), ("l2_done", C.c_int), ("pass2_error", C.c_int), ("n_bad_pass1", C.c_int), ("n_bad_total", C.c_int),
        ("iterations", C.c_int * 2), ("num_successful_steps", C.c_int * 2), ("termination", C.c_int * 2), ("initial_cost", C.c_double * 2),
        ("final_cost", C.c_double * 2), ("solve_ms", C.c_double * 2), ("status", C.c_int),
    ]


class BAIter(C.Structure):
    """ov2_ba_iter"""
    _fields_ = [("iteration", C.c_int), ("step_is_valid", C.c_int), ("step_is_successful", C.c_int), ("reserved_", C.c_int),
                ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double), ("gradient_norm", C.c_double),
                ("step_norm", C.c_double), ("relative_decrease", C.c_double), ("trust_region_radius", C.c_double)]


class SBAProblem(C.Structure):
    _fields_ = [
        ("n_kf", C.c_int), ("poses", C.POINTER(C.c_double)), ("n_pts", C.c_int), ("xyz", C.POINTER(C.c_double)),
        ("n_res", C.c_int), ("res_type", C.POINTER(C.c_uint8)), ("res_kf", C.POINTER(C.c_int)), ("res_pt", C.POINTER(C.c_int)),
        ("res_uv", C.POINTER(C.c_double)), ("res_sigma", C.POINTER(C.c_double)), ("res_active", C.POINTER(C.c_uint8)),
        ("calib_l", C.c_double * 4), ("calib_r", C.c_double * 4), ("T_rl", C.c_double * 7),
    ]


class SBAResult(C.Structure):
    _fields_ = [
        ("xyz_out", C.POINTER(C.c_double)), ("chi2_last_eval", C.POINTER(C.c_double)), ("depthpos_last_eval", C.POINTER(C.c_uint8)),
        ("iterations", C.c_int), ("num_successful_steps", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("termination", C.c_int), ("solve_ms", C.c_double),
    ]


class TrackerConfig(C.Structure):
    _fields_ = [
        ("w", C.c_int), ("h", C.c_int), ("win", C.c_int), ("nklt_pyr_lvl", C.c_int), ("prior_pyr_lvl", C.c_int),
        ("max_iter", C.c_int), ("eps", C.c_float), ("err_th", C.c_float), ("fb_dist", C.c_float),
        ("use_clahe", C.c_int), ("clahe_clip", C.c_double), ("tiles_x", C.c_int), ("tiles_y", C.c_int),
        ("n_max", C.c_int), ("use_graph", C.c_int),
    ]


class XYZBAProblem(C.Structure):
    _fields_ = [
        ("n_kf", C.c_int), ("poses", C.POINTER(C.c_double)), ("kf_const", C.POINTER(C.c_uint8)), ("n_pts", C.c_int),
        ("xyz", C.POINTER(C.c_double)), ("n_res", C.c_int), ("res_type", C.POINTER(C.c_uint8)), ("res_kf", C.POINTER(C.c_int)),
        ("res_pt", C.POINTER(C.c_int)), ("res_uv", C.POINTER(C.c_double)), ("res_sigma", C.POINTER(C.c_double)),
        ("res_active", C.POINTER(C.c_uint8)), ("calib_l", C.c_double * 4), ("calib_r", C.c_double * 4), ("T_rl", C.c_double * 7),
    ]


class XYZBAResult(C.Structure):
    _fields_ = [
        ("poses_out", C.POINTER(C.c_double)), ("xyz_out", C.POINTER(C.c_double)), ("chi2_last_eval", C.POINTER(C.c_double)),
        ("depthpos_last_eval", C.POINTER(C.c_uint8)), ("iterations", C.c_int), ("num_successful_steps", C.c_int),
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("termination", C.c_int), ("solve_ms", C.c_double),
    ]


_vp, _i, _f, _d = C.c_void_p, C.c_int, C.c_float, C.c_double
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes).  Must list every symbol include/ov2slam_hip.h declares
# (tests/test_abi.py cross-checks this table against the header).
SIGNATURES = {
    "ov2_version": (_i, []),
    "ov2_last_error": (C.c_char_p, []),
    "ov2_ctx_create": (_i, [_i, _pp]),
    "ov2_ctx_create_with_priority": (_i, [_i, _i, _pp]),
    "ov2_ctx_create_on_stream": (_i, [_i, _vp, _pp]),
    "ov2_ctx_destroy": (None, [_vp]),
    "ov2_ctx_sync": (_i, [_vp]),
    "ov2_ctx_stream": (_vp, [_vp]),
    "ov2_ctx_set_option": (_i, [_vp, _i, _i]),
    "ov2_ctx_get_option": (_i, [_vp, _i, C.POINTER(C.c_int)]),
    "ov2_pyr_create": (_i, [_vp, _i, _i, _i, _i, _i, _pp]),
    "ov2_pyr_destroy": (None, [_vp]),
    "ov2_pyr_levels": (_i, [_vp]),
    "ov2_pyr_level_size": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    "ov2_pyr_batch": (_i, [_vp]),
    "ov2_pyr_item_view": (_i, [_vp, _i, _pp]),
    "ov2_pyr_build_h": (_i, [_vp, _vp, _vp, _i, C.c_size_t]),
    "ov2_pyr_build_d": (_i, [_vp, _vp, _vp, _i, C.c_size_t]),
    "ov2_pyr_download": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "ov2_pyr_download_padded": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "ov2_pyr_algorithmic_bytes": (C.c_size_t, [_vp]),
    "ov2_clahe_h": (_i, [_vp, _vp, _i, _i, _i, _d, _i, _i, _vp, _i]),
    "ov2_clahe_d": (_i, [_vp, _vp, _i, _i, _i, C.c_size_t, _i, _d, _i, _i, _vp, _i, C.c_size_t]),
    "ov2_compute_keypoints": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "ov2_compute_keypoints_d": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "ov2_line_min_sad": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "ov2_stereo_epipolar_check": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "ov2_stereo_match": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "ov2_pyr_build_clahe_d": (_i, [_vp, _vp, _vp, _i, C.c_size_t, _d, _i, _i]),
    "ov2_pyr_build_clahe_h": (_i, [_vp, _vp, _vp, _i, _d, _i, _i]),
    "ov2_pyr_build_clahe_hb": (_i, [_vp, _vp, _i, _vp, _i, _d, _i, _i]),
    "ov2_stereo_match_batch": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ov2_tracker_create": (_i, [_vp, C.POINTER(TrackerConfig), _pp]),
    "ov2_tracker_destroy": (None, [_vp]),
    "ov2_tracker_image_buffer": (_vp, [_vp, C.POINTER(_i)]),
    "ov2_tracker_preprocess": (_i, [_vp, _vp, _i]),
    "ov2_tracker_klt": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, C.POINTER(_i)]),
    "ov2_tracker_track_frame": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, C.POINTER(_i)]),
    "ov2_tracker_set_calibration": (_i, [_vp, _i, _vp, _vp, _i, _vp]),
    "ov2_tracker_last_keypoints": (_i, [_vp, _i, _vp, _vp]),
    "ov2_tracker_cur_pyr": (_vp, [_vp]),
    "ov2_tracker_prev_pyr": (_vp, [_vp]),
    "ov2_tracker_frames": (_i, [_vp]),
    "ov2_tracker_uses_graph": (_i, [_vp]),
    "ov2_btracker_create": (_i, [_vp, C.POINTER(TrackerConfig), _i, _pp]),
    "ov2_btracker_destroy": (None, [_vp]),
    "ov2_btracker_batch": (_i, [_vp]),
    "ov2_btracker_frames": (_i, [_vp]),
    "ov2_btracker_image_buffer": (_vp, [_vp, _i, _i, C.POINTER(_i)]),
    "ov2_btracker_upload": (_i, [_vp, _i, _i]),
    "ov2_btracker_prepare": (_i, [_vp, _i, _i]),
    "ov2_btracker_set_calibration": (_i, [_vp, _i, _vp, _vp, _i, _vp]),
    "ov2_btracker_track_frame": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "ov2_btracker_track_frame_begin": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i]),
    "ov2_btracker_track_frame_end": (_i, [_vp, _vp, _vp, _vp]),
    "ov2_btracker_last_keypoints": (_i, [_vp, _i, _i, _vp, _vp]),
    "ov2_btracker_detect_singlescale": (_i, [_vp, _i, _i, _vp, _vp, C.POINTER(_i), _vp, _i, _vp, _i, _vp]),
    "ov2_btracker_detect_grid_fast": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "ov2_btracker_pyramid_sets": (_i, [_vp]),
    "ov2_btracker_cur_pyr": (_vp, [_vp]),
    "ov2_btracker_prev_pyr": (_vp, [_vp]),
    "ov2_btracker_cur_item": (_vp, [_vp, _i]),
    "ov2_btracker_prev_item": (_vp, [_vp, _i]),
    "ov2_lk_track": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "ov2_fb_klt": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _vp, _vp, _i, _vp, _vp]),
    "ov2_fb_klt_d": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _vp, _vp, _i, _vp, _vp, _vp]),
    "ov2_detect_grid_fast": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, C.POINTER(_i), _i, _i, _vp, C.POINTER(_i)]),
    "ov2_detect_singlescale": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, C.POINTER(_i), C.POINTER(_d), _i, _vp, C.POINTER(_i)]),
    "ov2_detect_grid_fast_d": (_i, [_vp, _vp, _i, _i, _vp, _i, C.POINTER(_i), _i, _i, _vp, C.POINTER(_i)]),
    "ov2_detect_singlescale_d": (_i, [_vp, _vp, _i, _i, _vp, _i, C.POINTER(_i), C.POINTER(_d), _i, _vp, C.POINTER(_i)]),
    "ov2_detect_singlescale_batch_d": (_i, [_vp, _vp, _i, _vp, _i, _vp, C.POINTER(_i), _vp, _i, _vp, _i, _vp]),
    "ov2_detect_grid_fast_batch_d": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "ov2_corner_subpix": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _d]),
    "ov2_ba_default_options": (None, [C.POINTER(BAOptions)]),
    "ov2_structure_ba": (_i, [_vp, C.POINTER(SBAProblem), C.POINTER(BAOptions), C.POINTER(SBAResult)]),
    "ov2_xyz_ba_solve": (_i, [_vp, C.POINTER(XYZBAProblem), C.POINTER(BAOptions), C.POINTER(XYZBAResult)]),
    "ov2_ba_solve": (_i, [_vp, C.POINTER(BAProblem), C.POINTER(BAOptions), C.POINTER(BAResult)]),
    "ov2_ba_get_trace": (_i, [_vp, C.POINTER(BAIter), _i, C.POINTER(_i)]),
    "ov2_ba_create": (_i, [_vp, C.POINTER(BAProblem), _pp]),
    "ov2_ba_solve_resident": (_i, [_vp, _vp, C.POINTER(BAOptions), C.POINTER(BAResult)]),
    "ov2_ba_destroy": (None, [_vp]),
    "ov2_local_ba_default_options": (None, [C.POINTER(LocalBAOptions)]),
    "ov2_local_ba": (_i, [_vp, C.POINTER(BAProblem), C.POINTER(LocalBAOptions), C.POINTER(LocalBAResult)]),
    "ov2_local_ba_batch": (_i, [_vp, _i, C.POINTER(BAProblem), C.POINTER(LocalBAOptions), C.POINTER(LocalBAResult), C.POINTER(_i)]),
}

OV2_ABI_VERSION = 600          # include/ov2slam_hip.h

_lib = None


def load():
    """Load libov2slam_hip.so and bind every symbol.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ov2slam_amd/csrc` (hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.ov2_version() != OV2_ABI_VERSION:    # struct layouts below are those of include/ov2slam_hip.h at this version
        raise ImportError("%s reports ABI version %d, these bindings were written for %d: rebuild the library"
                          % (LIB_PATH, lib.ov2_version(), OV2_ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != OV2_OK:
        raise Ov2Error(rc, load().ov2_last_error().decode("utf-8", "replace"))
