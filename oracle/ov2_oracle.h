/*
 * ov2_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the arithmetic on OV2SLAM's front-end + local-BA hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (ov2slam_amd/) never links or calls it.
 *
 * PARITY STATUS
 *   - front-end (pyramid / LK / FAST / min-eig / subpix): "parity unpinned".
 *     The reference delegates these to OpenCV (unvendored, version unpinned,
 *     absent from this container) and ships no tests or golden vectors for
 *     them.  The oracle restates the public OpenCV 3.4/4.x
 *     algorithms named at each function and is pinned only by self-consistency
 *     tests + independent numpy/scipy cross-checks.
 *     Round 5: the reference's first-party control flow (src/feature_tracker.cpp, src/feature_extractor.cpp compiled in place, oracle/ref)
 *     runs on top of these restatements and agrees with them; that pins the walk between the OpenCV calls, not the calls.
 *     ORC_LK_ACC_FLOAT_* restate the float-accumulator orders of stock x86 builds (the device implements FLOAT_UI4 too, round 6).
 *   - local BA: PINNED against the reference's own sources executed here (oracle/ref, oracle/_ref/*.so):
 *     every factor / the SE(3) parameterisation / the parameter blocks against src/ceres_parametrization.cpp compiled in place
 *     (tests/test_reference_factors.py, <= 1e-11), and -- round 6 -- the whole Levenberg-Marquardt iteration against Ceres' own
 *     trust_region_minimizer.cc, trust_region_step_evaluator.cc, levenberg_marquardt_strategy.cc, corrector.cc, loss_function.cc
 *     compiled in place and driving those factors (tests/test_reference_trlm.py: identical decisions, traces within 2e-9).
 *     Both against stand-in Eigen / Sophus headers (absent from the image): rounding-level agreement, not bit-exact, and not a
 *     "reference build" in the sense of a CPU baseline.  Sub-steps additionally by the known-answer tests vendored under
 *     /root/reference/Thirdparty/ceres-solver/internal/ceres/ (the _test.cc files).  Optimizer::localBA outputs end to end have no
 *     reference golden vectors (tools/ref_capture would produce them on a box with Eigen + Ceres).
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef OV2_ORACLE_H
#define OV2_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* Pyramid (cv::buildOpticalFlowPyramid, called at                     */
/* src/visual_front_end.cpp:1172, :53 and src/mapper.cpp:81)           */
/* ------------------------------------------------------------------ */
#define ORC_MAX_LEVELS 8

typedef struct {
    int w, h;            /* un-padded level size                               */
    int pad;             /* padding on every side (= win)                      */
    int img_pitch;       /* bytes per padded image row                         */
    int der_pitch;       /* int16 elements per padded derivative row           */
    uint8_t *img;        /* (h+2pad) x img_pitch, REFLECT_101 border           */
    int16_t *der;        /* (h+2pad) x der_pitch, (dx,dy) interleaved, 0 border */
} orc_level;

typedef struct {
    int n_levels;        /* levels actually built (maxLevel+1 or fewer)        */
    int win;
    orc_level lv[ORC_MAX_LEVELS];
} orc_pyr;

/* returns 0 on success. img is u8 row-major, `stride` bytes per row. */
/* ---- worker pool (pool.c): stand-in for cv::parallel_for_ -------------------------------------------
 * orc_set_num_threads(n): threads used by the parallel loops of CLAHE / pyrDown / Scharr / LK (default 1);
 * loops are split into ranges whose results do not depend on the split.                                */
typedef void (*orc_range_fn)(int begin, int end, void *ctx);
void orc_set_num_threads(int n);
int  orc_get_num_threads(void);
void orc_parallel_for(int n, orc_range_fn fn, void *ctx, int min_grain);

int  orc_pyr_build(const uint8_t *img, int w, int h, int stride, int win,
                   int max_level, orc_pyr *out);
int  orc_pyr_rebuild(const uint8_t *img, int w, int h, int stride, orc_pyr *p);   /* same geometry, buffers re-used */
void orc_pyr_free(orc_pyr *p);
/* accessors used by the python tests (copy out the un-padded ROI) */
int  orc_pyr_level_size(const orc_pyr *p, int level, int *w, int *h);
int  orc_pyr_copy_level(const orc_pyr *p, int level, uint8_t *img_out, int16_t *der_out);
/* copy out a padded level (pad pixels on each side) for border checks */
int  orc_pyr_copy_level_padded(const orc_pyr *p, int level, uint8_t *img_out, int16_t *der_out);

/* single building blocks, exposed for unit tests */
void orc_pyr_down_u8(const uint8_t *src, int sw, int sh, int sstride,
                     uint8_t *dst, int dw, int dh, int dstride);
void orc_scharr_u8(const uint8_t *src, int w, int h, int sstride,
                   int16_t *dst, int dstride_elems);

/* cv::CLAHE::apply (CV_8UC1): src/ov2slam.cpp:85-89 (createCLAHE(fclahe_val, Size(w/50, h/50))),
 * src/visual_front_end.cpp:1159, src/mapper.cpp:76.  dst may alias src only if strides match. */
int orc_clahe(const uint8_t *src, int w, int h, int stride, double clip_limit, int tiles_x, int tiles_y,
              uint8_t *dst, int dst_stride);

/* ------------------------------------------------------------------ */
/* LK (cv::calcOpticalFlowPyrLK, called at src/feature_tracker.cpp:66, */
/* :113) and FeatureTracker::fbKltTracking (src/feature_tracker.cpp:35)*/
/* ------------------------------------------------------------------ */
#define ORC_LK_USE_INITIAL_FLOW   4   /* cv::OPTFLOW_USE_INITIAL_FLOW    */
#define ORC_LK_GET_MIN_EIGENVALS  8   /* cv::OPTFLOW_LK_GET_MIN_EIGENVALS */

/* One calcOpticalFlowPyrLK call on pre-built pyramids.
 * next_xy is in/out (initial flow when USE_INITIAL_FLOW).  iters_out (may be
 * NULL) receives, per point, the total number of Gauss-Newton iterations that
 * were executed over all levels (used for the algorithmic-bytes figure).
 * nthreads > 1 splits the points over pthreads (mirrors cv::parallel_for_).  */
/* LK accumulator variant (frontend.c): INT64 = the canonical, order-independent variant the HIP kernels implement;
 * FLOAT_* = float accumulators in the summation orders of stock OpenCV builds (restated from the public lkpyramid.cpp, parity
 * unpinned): scalar raster order (no SIMD), the 3.4 SSE2 intrinsics, the 4.x universal intrinsics (SSE / AVX2 baselines, no FMA). */
#define ORC_LK_ACC_INT64        0
#define ORC_LK_ACC_FLOAT_SCALAR 1
#define ORC_LK_ACC_FLOAT_SSE34  2
#define ORC_LK_ACC_FLOAT_UI4    3
void orc_set_lk_acc_mode(int mode);
/* diagnostics: histogram (64 bins, last = 63 and more) of Gauss-Newton trips per level visit of every orc_lk_track since enable(1) */
void orc_lk_trip_hist_enable(int on);
void orc_lk_trip_hist_get(int *out64);
int orc_get_lk_acc_mode(void);

int orc_lk_track(const orc_pyr *prev, const orc_pyr *next,
                 const float *prev_xy, float *next_xy, int n,
                 uint8_t *status, float *err,
                 int win, int max_level, int max_count, double epsilon,
                 int flags, double min_eig_threshold,
                 int *iters_out, int nthreads);

/* FeatureTracker::fbKltTracking.  status_out has n entries (0/1).
 * Returns 0; n==0 is a no-op like the reference (:43-46).
 * iters_total (may be NULL): sum of GN iterations (fwd+bwd) for byte counting;
 * lvl_visits (may be NULL): number of (point,level) patch builds.            */
int orc_fb_klt(const orc_pyr *prevpyr, const orc_pyr *curpyr,
               int win, int nbpyrlvl, int max_iter, float eps_px,
               float ferr, float fmax_fbklt_dist,
               const float *kps_xy, float *prior_xy_inout, int n,
               uint8_t *status_out, long long *iters_total, long long *lvl_visits,
               int nthreads);

/* ------------------------------------------------------------------ */
/* Detection (src/feature_extractor.cpp:288-570)                       */
/* ------------------------------------------------------------------ */
#define ORC_MASK_AS_EXECUTED 0  /* CV_32F mask read through at<uchar> */
#define ORC_MASK_INTENDED    1  /* true per-pixel float mask                       */

/* FAST-9/16 + score + 3x3 NMS on a standalone w x h u8 image (cv::FAST).
 * Outputs raster-ordered keypoints; returns count (<= cap).                */
int orc_fast9_16(const uint8_t *img, int w, int h, int stride, int threshold,
                 int nonmax, int *xs, int *ys, int *scores, int cap);

/* FeatureExtractor::detectGridFAST (:443-570), serial raster order (N2).
 * out_xy capacity: (w/cell)*(h/cell) points.  do_subpix=0 skips cornerSubPix. */
int orc_detect_grid_fast(const uint8_t *img, int w, int h, int stride, int cell,
                         const float *cur_xy, int ncur, int *fast_th_inout,
                         int mask_mode, int do_subpix,
                         float *out_xy, int *out_n);

/* FeatureExtractor::detectSingleScale (:288-440), serial raster order.
 * roi = {x, y, width, height}.  out_xy capacity: 2*(w/cell)*(h/cell).        */
int orc_detect_singlescale(const uint8_t *img, int w, int h, int stride, int cell,
                           const float *cur_xy, int ncur, const int roi[4],
                           double *quality_inout, int do_subpix,
                           float *out_xy, int *out_n);

/* min-eigenvalue response map of one cell (blur w/ parent pixels + isolated
 * cornerMinEigenVal); exposed for unit tests. hmap is cell*cell floats.      */
enum { ORC_SOBEL_DY_OPENCV_ROWFILTER = 0, ORC_SOBEL_DY_EXACT_SUM = 1 };
enum { ORC_BLUR_FIXED = 0, ORC_BLUR_HALF_EVEN = 1 };
enum { ORC_SUBPIX_FAST = 0, ORC_SUBPIX_GENERIC = 1, ORC_SUBPIX_FLOAT_ACC = 2 };
void orc_set_blur_mode(int mode);            /* GaussianBlur 3x3 rounding inside detectSingleScale (detect.c) */
int orc_get_blur_mode(void);
void orc_set_subpix_mode(int mode);          /* getRectSubPix path / accumulator type inside cornerSubPix (detect.c) */
int orc_get_subpix_mode(void);
/* detectGridFAST: which of several EQUAL best FAST responses of a cell wins.  The reference sorts the cell's keypoints with std::sort
 * (src/feature_extractor.cpp:518, not stable) and takes the first: with more than 16 of them the winner among ties is the standard
 * library's choice.  LIBSTDCXX (canonical: what the HIP kernels implement by default, OV2_OPT_FAST_TIE): libstdc++'s introsort restated
 * (bits/stl_algo.h) -- with it the oracle equals the reference's own code compiled with g++ on every cell
 * (tests/test_reference_factors.py).  SCAN_ORDER: the first in scan order (what a stable sort, and every insertion-sort-sized cell,
 * gives); the two differ on 0.4 % of the cells of a dense synthetic texture at threshold 20, none at 30. */
enum { ORC_FAST_TIE_SCAN_ORDER = 0, ORC_FAST_TIE_LIBSTDCXX = 1 };
void orc_set_fast_tie_mode(int mode);
int orc_get_fast_tie_mode(void);
int orc_fast_tie_sort_fallbacks(void);      /* times the emulated introsort ran out of depth (its heap-sort branch is not restated): must stay 0 */
void orc_set_sobel_dy_order(int order);      /* evaluation order of cv::Sobel(dx=0, dy=1, scale) in the min-eigenvalue map */
int  orc_get_sobel_dy_order(void);
void orc_cell_mineig(const uint8_t *img, int w, int h, int stride,
                     int x0, int y0, int cell, float *hmap);

/* cv::cornerSubPix(win=(3,3), zeroZone=(-1,-1), 30 it / 0.01) in place.      */
void orc_corner_subpix(const uint8_t *img, int w, int h, int stride,
                       float *xy_inout, int n, int half_win, int max_iter, double eps);

/* cv::circle(mask, c, r, 0, FILLED) on a w x h byte mask (1 = free).          */
void orc_circle_fill0(uint8_t *mask, int w, int h, int cx, int cy, int r);

/* ------------------------------------------------------------------ */
/* Local BA (src/optimizer.cpp:34-897 + Ceres 2.0.0 TR-LM / Schur)     */
/* ------------------------------------------------------------------ */
enum {
    ORC_RES_LEFT        = 0, /* ReprojectionErrorKSE3AnchInvDepth        ceres_parametrization.cpp:361 */
    ORC_RES_RIGHT       = 1, /* ReprojectionErrorRightCamKSE3AnchInvDepth  :579 */
    ORC_RES_RIGHT_ANCH  = 2, /* ReprojectionErrorRightAnchCamKSE3AnchInvDepth :476 */
    ORC_RES_PNP         = 3, /* DirectLeftSE3::ReprojectionErrorSE3 (fixed world point), ceres_parametrization.cpp:301-358;
                                used by MultiViewGeometry::ceresPnP, src/multi_view_geometry.cpp:492-586 */
};

typedef struct {
    int n_kf;               /* poses, Twc, [tx ty tz qx qy qz qw]  */
    const double *poses;    /* 7*n_kf                                         */
    const uint8_t *kf_const;/* n_kf                                           */
    int n_lm;
    const double *invdepth; /* n_lm                                           */
    const int *lm_anchor_kf;/* n_lm                                           */
    const double *lm_anchor_uv; /* 2*n_lm  anchor (undistorted) pixel          */
    int n_res;
    const uint8_t *res_type;/* n_res  ORC_RES_*                               */
    const int *res_kf;      /* n_res  observing KF (ignored for RIGHT_ANCH)   */
    const int *res_lm;      /* n_res                                          */
    const double *res_uv;   /* 2*n_res observed pixel                         */
    const double *res_sigma;/* n_res                                          */
    const uint8_t *res_active; /* n_res or NULL (all active)                   */
    const double *res_xyz;  /* 3*n_res world point of ORC_RES_PNP blocks (ignored for the others); NULL if none */
    double calib_l[4];      /* fx fy cx cy                                    */
    double calib_r[4];
    double T_rl[7];         /* [t, q] of Trl (right <- left)                  */
} orc_ba_problem;

typedef struct {
    int max_iter;
    double function_tolerance;   /* 1e-3 in localBA (optimizer.cpp:462)          */
    double gradient_tolerance;   /* ceres default 1e-10                          */
    double parameter_tolerance;  /* ceres default 1e-8                           */
    double huber_delta;          /* sqrt(5.9915); <= 0 -> trivial loss           */
    double initial_radius;       /* 1e4                                          */
    double max_radius;           /* 1e16                                         */
    double min_radius;           /* 1e-32                                        */
    double min_lm_diagonal;      /* 1e-6                                         */
    double max_lm_diagonal;      /* 1e32                                         */
    double min_relative_decrease;/* 1e-3                                         */
    int jacobi_scaling;          /* 1                                            */
    int max_consecutive_invalid_steps; /* 5                                      */
} orc_ba_options;

enum {
    ORC_TERM_NO_CONVERGENCE = 0,   /* hit max_iter                               */
    ORC_TERM_FUNCTION_TOL   = 1,
    ORC_TERM_PARAMETER_TOL  = 2,
    ORC_TERM_GRADIENT_TOL   = 3,
    ORC_TERM_MIN_RADIUS     = 4,
    ORC_TERM_INVALID_STEPS  = 5,
    ORC_TERM_FAILURE        = 6,
};

typedef struct {
    double *poses_out;        /* 7*n_kf  */
    double *invdepth_out;     /* n_lm    */
    double *chi2_last_eval;   /* n_res   (value at last evaluated point) */
    uint8_t *depthpos_last_eval; /* n_res */
    int iterations;           /* ceres "num_iterations" = LM iterations run (excluding iter 0) */
    int num_successful_steps;
    double initial_cost, final_cost;
    int termination;
} orc_ba_result;

void orc_ba_default_options(orc_ba_options *o);
int  orc_ba_solve(const orc_ba_problem *p, const orc_ba_options *o, orc_ba_result *r);
/* Per-iteration trace of the NEXT orc_ba_solve calls of the calling thread (tests only): one entry per iteration Ceres records in
 * Solver::Summary::iterations (trust_region_minimizer.cc:313-337; entry 0 = the starting point; an iteration that ends the solve inside
 * the loop -- parameter / function tolerance -- is not recorded, as in Ceres).  *n counts every entry, buf keeps the first cap.
 * orc_ba_set_trace(NULL, 0, NULL) switches it off.  Checked against Ceres' own loop by tests/test_reference_trlm.py.               */
typedef struct {
    int iteration, step_is_valid, step_is_successful, pad_;
    double cost, cost_change, gradient_max_norm, gradient_norm, step_norm, relative_decrease, trust_region_radius;
} orc_ba_iter;
void orc_ba_set_trace(orc_ba_iter *buf, int cap, int *n);

/* building blocks exposed for the known-answer tests */
void orc_huber(double a, double s, double rho[3]);                 /* loss_function.cc:48-62  */
/* corrector.cc:42-156; residuals (n_rows) and jacobian (n_rows x n_cols,
 * row-major) corrected in place                                               */
void orc_corrector(double sq_norm, const double rho[3], int n_rows, int n_cols,
                   double *residuals, double *jacobian);
/* levenberg_marquardt_strategy.cc:147-160 */
void orc_lm_step_accepted(double step_quality, double *radius, double *decrease_factor, double max_radius);
void orc_lm_step_rejected(double *radius, double *decrease_factor);
/* D_i = sqrt(clamp(diag(J^T J)_i, min, max) / radius), levenberg_marquardt_strategy.cc:76-88 (the routine orc_ba_solve uses) */
void orc_lm_diagonal(double *jtj_diag, int n, double radius, double min_diagonal, double max_diagonal, int clamp, double *D_out);
/* dense restatement of SchurEliminator::Eliminate / BackSubstitute for scalar e-blocks (schur_eliminator_impl.h:179-377) */
int  orc_schur_eliminate_dense(const double *J, const double *b, const double *D, int m, int n, int n_e, const int *row_e,
                               double *lhs, double *rhs, double *sol);
/* Sophus SE3 exp (se3.hpp:763-784), tangent [v, w] -> (t, q=[x y z w])          */
void orc_se3_exp(const double tangent[6], double t_out[3], double q_out[4]);
/* left-multiplicative update  T' = Exp(delta) * T  (se3left_parametrization.hpp:45-57) */
void orc_se3_left_plus(const double pose[7], const double delta[6], double pose_out[7]);
/* one residual block: residual (2), jacobians wrt anchor pose (2x6 local),
 * obs pose (2x6 local), inverse depth (2x1); un-robustified.
 * Returns depth-positive flag; chi2 in *chi2.                                 */
int orc_ba_residual(int type, const double calib_l[4], const double calib_r[4], const double T_rl[7],
                    const double anchor_pose[7], const double obs_pose[7], double invdepth,
                    const double anchor_uv[2], const double uv[2], double sigma,
                    double r[2], double J_anchor[12], double J_obs[12], double J_lambda[2],
                    double *chi2);

/* ------------------------------------------------------------------ */
/* Per-keypoint undistortion + bearing vector (Frame::computeKeypoint,  */
/* src/frame.cpp:246-254; CameraCalibration::undistortImagePoint,       */
/* src/camera_calibration.cpp:313-333) -- see undistort.c               */
/* ------------------------------------------------------------------ */
#define ORC_CAM_PINHOLE 0
#define ORC_CAM_FISHEYE 1
/* K = (fx, fy, cx, cy); D = distortion coefficients (nD = 4, 5, 8, 12 or 14 for pinhole; 4 for fisheye) */
void orc_undistort_pinhole(const double K[4], const double *D, int nD, const float *px, int n, float *out);
void orc_undistort_fisheye(const double K[4], const double D[4], const float *px, int n, float *out);
/* iK row-major 3x3 (the reference's K_.inverse()); bv: 3 doubles per point */
void orc_compute_keypoints(int model, const double K[4], const double *D, int nD, const double iK[9],
                           const float *px, int n, float *unpx, double *bv);

/* ------------------------------------------------------------------ */
/* Stereo matching front half (MapManager::stereoMatching,              */
/* src/map_manager.cpp:367-611) -- see stereo.c                         */
/* ------------------------------------------------------------------ */
void orc_get_rect_subpix_8u(const uint8_t *src, int src_step, int sw, int sh, uint8_t *dst, int pw, int ph, float cx, float cy);
void orc_line_min_sad(const uint8_t *iml, int lstride, const uint8_t *imr, int rstride, int w, int h,
                      float x, float y, int nwinsize, int go_left, float *xprior, float *l1err);
void orc_line_min_sad_batch(const uint8_t *iml, int lstride, const uint8_t *imr, int rstride, int w, int h,
                            const float *xy, int n, int nwinsize, int go_left, float *xprior, float *l1err);
float orc_sampson_distance(const double F[9], float lx, float ly, float rx, float ry);
void orc_stereo_epipolar_check(int rect, const double Frl[9], int model, const double K[4], const double *D, int nD,
                               const float *lunpx, float *rkps, int n, float *runpx, float *epi_err, uint8_t *ok);

/* ------------------------------------------------------------------ */
/* Optimizer::structureOnlyBA (src/optimizer.cpp:2594-2781): 3-D points, */
/* every pose constant -- see struct_ba.c                               */
/* ------------------------------------------------------------------ */
enum {
    ORC_XYZ_LEFT  = 0,   /* DirectLeftSE3::ReprojectionErrorKSE3XYZ          */
    ORC_XYZ_RIGHT = 1,   /* DirectLeftSE3::ReprojectionErrorRightCamKSE3XYZ  */
};
typedef struct {
    int n_kf;
    const double *poses;    /* 7*n_kf  [t, q(x,y,z,w)] of Twc, all constant   */
    int n_pts;
    const double *xyz;      /* 3*n_pts world points (initial values)          */
    int n_res;
    const uint8_t *res_type;/* ORC_XYZ_*                                      */
    const int *res_kf, *res_pt;
    const double *res_uv;   /* 2*n_res */
    const double *res_sigma;
    const uint8_t *res_active; /* or NULL */
    double calib_l[4], calib_r[4], T_rl[7];
} orc_sba_problem;
typedef struct {
    double *xyz_out;          /* 3*n_pts */
    double *chi2_last_eval;   /* n_res   */
    uint8_t *depthpos_last_eval;
    int iterations, num_successful_steps;
    double initial_cost, final_cost;
    int termination;
} orc_sba_result;
/* ---- BA over 3-D points with variable poses (buse_inv_depth: 0, src/optimizer.cpp:207-209, :333-384) -- xyz_ba.c ---- */
typedef struct {
    int n_kf;
    const double *poses;       /* 7*n_kf  [t, q(x,y,z,w)] of Twc, initial values          */
    const uint8_t *kf_const;   /* n_kf    1 = constant block (NULL: all variable)          */
    int n_pts;
    const double *xyz;         /* 3*n_pts world points, initial values                     */
    int n_res;
    const uint8_t *res_type;   /* ORC_XYZ_LEFT / ORC_XYZ_RIGHT                             */
    const int *res_kf, *res_pt;
    const double *res_uv;      /* 2*n_res */
    const double *res_sigma;
    const uint8_t *res_active; /* or NULL */
    double calib_l[4], calib_r[4], T_rl[7];
} orc_xyzba_problem;
typedef struct {
    double *poses_out;         /* 7*n_kf  */
    double *xyz_out;           /* 3*n_pts */
    double *chi2_last_eval;    /* n_res (N4) */
    uint8_t *depthpos_last_eval;
    int iterations, num_successful_steps;
    double initial_cost, final_cost;
    int termination;
} orc_xyzba_result;
int orc_xyzba_residual(int type, const double calib_l[4], const double calib_r[4], const double T_rl[7], const double pose[7],
                       const double X[3], const double uv[2], double sigma, double r[2], double *Jp, double *Jx, double *chi2);
int orc_xyzba_solve(const orc_xyzba_problem *p, const orc_ba_options *o, orc_xyzba_result *r);

int orc_xyz_residual(int type, const double calib_l[4], const double calib_r[4], const double T_rl[7], const double pose[7],
                     const double X[3], const double uv[2], double sigma, double r[2], double *J, double *chi2);
int orc_structure_ba(const orc_sba_problem *p, const orc_ba_options *o, orc_sba_result *r);

#ifdef __cplusplus
}
#endif
#endif /* OV2_ORACLE_H */
