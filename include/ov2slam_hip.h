/*
 * ov2slam_hip.h -- C ABI of libov2slam_hip.so (MI355X / gfx950 HIP kernels).
 *
 * Drop-in boundary for OV2SLAM's front-end + local-BA hot path.  The reference
 * has no FFI layer: the boundary is three C++ classes (FeatureExtractor,
 * FeatureTracker, Optimizer) that call OpenCV / Ceres.  A thin C++ adapter with
 * the reference's own signatures (the headers in ov2slam_amd/host/)
 * forwards to the entry points below.  Plain pointers and sizes only; no
 * OpenCV / Eigen / torch types.  All citations are relative to /root/reference.
 *
 * Conventions
 *   - every function returns 0 (OV2_OK) on success, a negative OV2_E* otherwise,
 *     and never throws or aborts; on error no output buffer is modified unless
 *     stated (the adapter maps errors to the reference's "nothing tracked /
 *     BA skipped" behaviour, SURVEY.md 5 "failure detection").
 *   - `*_h` pointers are host memory, `*_d` pointers are device (HBM) memory.
 *   - an ov2_ctx owns one HIP stream + scratch; one ctx per calling thread
 *     (fbKltTracking is called concurrently from the SLAM and mapper threads,
 *     src/visual_front_end.cpp:196 vs src/map_manager.cpp:510).  The library is
 *     re-entrant across contexts.
 *   - keypoints are float2 (x,y) interleaved, like std::vector<cv::Point2f>.
 */
#ifndef OV2SLAM_HIP_H
#define OV2SLAM_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OV2_OK            0
#define OV2_EINVAL       -1   /* bad argument                                  */
#define OV2_EHIP         -2   /* HIP runtime error (see ov2_last_error)        */
#define OV2_ENOMEM       -3
#define OV2_EUNSUPPORTED -4   /* e.g. LK window size without a kernel instance */
#define OV2_ENODEVICE    -5   /* no gfx950 device visible                      */

typedef struct ov2_ctx ov2_ctx;
typedef struct ov2_pyr ov2_pyr;

/* ---- context ------------------------------------------------------- */
/* ABI version of THIS header: bumped whenever a struct passed by pointer grows or an entry point changes its signature
 * (round 2 added ov2_ba_options::max_solver_time_s).  ov2_version() returns the value the library was built with; a caller
 * must refuse to run when the two differ (the C++ adapters' ov2::Context and ov2slam_amd/_lib.py do): a shorter options
 * struct from an older header would otherwise be read past its end.                                                    */
#define OV2_ABI_VERSION 600
int  ov2_version(void);
/* last error message of the calling thread ("" if none); never NULL */
const char *ov2_last_error(void);
/* creates a context with its own non-blocking HIP stream on `device` */
int  ov2_ctx_create(int device, ov2_ctx **out);
/* same with a stream priority: > 0 the device's highest, < 0 its lowest, 0 = ov2_ctx_create.  A host that runs several contexts on one
 * GPU states what is latency-critical: the reference's SLAM thread is real time while its estimator thread works on whatever
 * keyframe is newest when it gets round to it (src/estimator.cpp:195-205) -- tools/lockstep_driver.cpp gives the tracking context the
 * high priority and the localBA contexts the low one, so that a dozen concurrent solves do not stretch the per-frame enqueue */
int  ov2_ctx_create_with_priority(int device, int priority, ov2_ctx **out);
/* same, but enqueue on an existing hipStream_t (e.g. torch's current stream) */
int  ov2_ctx_create_on_stream(int device, void *hip_stream, ov2_ctx **out);
void ov2_ctx_destroy(ov2_ctx *ctx);
int  ov2_ctx_sync(ov2_ctx *ctx);
void *ov2_ctx_stream(ov2_ctx *ctx);        /* the hipStream_t, for event timing */
/* Per-context options.
 * OV2_OPT_SOBEL_DY_ORDER: evaluation order of cv::Sobel(dx = 0, dy = 1, scale) inside cv::cornerMinEigenVal
 * (detectSingleScale, src/feature_extractor.cpp:354).  OpenCV multiplies the SMOOTHING kernel by the scale when dx == 0,
 * so its 8U -> 32F row pass rounds ((p[x-1] k0 + p[x] k1) + p[x+1] k2) per pixel and the column pass subtracts two
 * rounded rows: OV2_SOBEL_DY_OPENCV_ROWFILTER, the default.  OV2_SOBEL_DY_EXACT_SUM scales the exact integer difference
 * instead (<= 1 ulp apart per pixel; can flip arg-max ties).  Neither is pinned against a real OpenCV build.           */
#define OV2_OPT_SOBEL_DY_ORDER         1
#define OV2_SOBEL_DY_OPENCV_ROWFILTER  0
#define OV2_SOBEL_DY_EXACT_SUM         1
/* Path selection.  Every kernel choice the library makes by itself can be pinned per context -- the parity tests run every
 * path on the same inputs this way, A/B measurements use it.  The library reads no environment variable after ov2_ctx_create
 * (its entry points are called from several threads of a host process that may call setenv concurrently).
 * OV2_OPT_LK_IMPL           ov2_fb_klt* / ov2_lk_track with the reference's window (9): AUTO picks the 3-lanes-per-keypoint kernel
 *                           (lk3.hip) from 65536 points per call on, the row-per-lane kernel (lk.hip) below
 * OV2_OPT_TRACK_IMPL        ov2_tracker_* / ov2_stereo_match, window 9: wavefront per keypoint (lkw.hip, default) or row per lane
 * OV2_OPT_CLAHE_STRIPS      ov2_pyr_build_clahe_*: the one-walk strip kernel (CLAHE apply + level 1 + borders): -1 auto (batch x
 *                           strips >= 1024: the fused form), 0 never, 1 whenever the geometry allows, after the LUT kernel,
 *                           2 whenever the geometry allows, fused with the LUT computation (one launch, LUTs stay in LDS)
 * OV2_OPT_BA_FORCE_LARGE    1: the large-problem BA path (sparse W slots, HBM Cholesky) on a problem of any size
 * OV2_OPT_BA_LIN_DIRECT     1 (with FORCE_LARGE): the lineariser without LDS aggregation of the observer blocks
 * OV2_OPT_BA_SCHUR_CHUNK    columns of the sparse Schur row block kept in LDS per chunk (0 = auto)
 * OV2_OPT_BA_XYZ_LIN_WAVES  wavefronts per work-group of the 3-D-point lineariser: 0 auto, 1, 2
 * OV2_OPT_BA_POSE_ONLY_FUSED 0: ceresPnP through the multi-kernel LM loop instead of the one-kernel form (default 1)
 * OV2_OPT_BA_DETERMINISTIC 1: bit-identical results from run to run (the reference solves with num_threads = 1): every sum that the
 *                           default accumulates with fp64 atomics in arrival order (H, F^T b, W^T C W, the costs) goes through
 *                           per-work-group buffers added up in a fixed order.  Inverse-depth form on the LDS-resident path (up
 *                           to ~70 optimised keyframes); other forms answer OV2_EUNSUPPORTED while it is set.  ~1.7x the solve time.
 * OV2_OPT_FAST_TIE          detectGridFAST: which of several EQUAL best FAST responses of a cell wins.  The reference sorts the cell's corners
 *                           with std::sort (src/feature_extractor.cpp:518, not stable) and takes the first: with more than 16 corners left
 *                           the winner among ties is the standard library's choice.  OV2_FAST_TIE_LIBSTDCXX (default): libstdc++'s
 *                           introsort restated -- the reference as built with g++, cell for cell (tests/test_reference_factors.py runs
 *                           the reference's own source); OV2_FAST_TIE_SCAN_ORDER: the first in scan order (what a stable sort gives)
 * OV2_OPT_LK_ACC            accumulator type of calcOpticalFlowPyrLK's sums (normal matrix, mismatch vector).  OpenCV's LKTrackerInvoker
 *                           is written against `acctype`: int64 on ARM without NEON, FLOAT everywhere else (lkpyramid.cpp).
 *                           OV2_LK_ACC_INT64 (default): exact integer sums -- independent of summation order, the form every kernel
 *                           implements and the oracle's canonical mode.  OV2_LK_ACC_FLOAT_UI4: float accumulators in the order an x86
 *                           OpenCV 4.x build (128-bit universal intrinsics, no FMA) adds them -- what the reference executes at
 *                           src/feature_tracker.cpp:66-69 / :113-116 on a desktop; restated from the public source (oracle:
 *                           ORC_LK_ACC_FLOAT_UI4), bit-exact against that restatement, never checked against an OpenCV binary.
 *                           Applies to ov2_fb_klt* / ov2_lk_track (row-per-lane kernel, any window), ov2_tracker_* / ov2_btracker_* /
 *                           ov2_stereo_match* (both kernels); a tracker captures its graphs at creation: set the option before.
 *                           Measured on EuRoC-like frames: no status flips, positions within 1.5e-4 px of the INT64 mode.
 * OV2_OPT_DEBUG             1: timing laps of ov2_local_ba / detection on stderr (initial value: environment OV2_DEBUG at
 *                           ov2_ctx_create, the only environment variable the library ever reads)                          */
#define OV2_OPT_LK_IMPL            2
#define OV2_LK_IMPL_AUTO           0
#define OV2_LK_IMPL_ROW            1
#define OV2_LK_IMPL_LANE3          2
#define OV2_OPT_TRACK_IMPL         3
#define OV2_TRACK_IMPL_WAVE        0
#define OV2_TRACK_IMPL_ROW         1
#define OV2_OPT_CLAHE_STRIPS       4
#define OV2_OPT_BA_FORCE_LARGE     5
#define OV2_OPT_BA_LIN_DIRECT      6
#define OV2_OPT_BA_SCHUR_CHUNK     7
#define OV2_OPT_BA_XYZ_LIN_WAVES   8
#define OV2_OPT_BA_POSE_ONLY_FUSED 9
#define OV2_OPT_BA_DETERMINISTIC   10
#define OV2_OPT_DEBUG              11
#define OV2_OPT_FAST_TIE           12
#define OV2_FAST_TIE_SCAN_ORDER    0
#define OV2_FAST_TIE_LIBSTDCXX     1
#define OV2_OPT_DETECT_STRIP       15   /* detectSingleScale's response kernel: -1 auto (the strip kernel for batches: the free cells of an image side by
                                           side, ~59 of 64 lanes busy on 35-pixel cells), 0 one wavefront per cell, 1 the strip kernel; same bits */
#define OV2_OPT_LK_ACC             14
#define OV2_LK_ACC_INT64           0
#define OV2_LK_ACC_FLOAT_UI4       1
#define OV2_OPT_BA_TRACE           13   /* 1: ov2_ba_solve / ov2_ba_solve_resident / ov2_xyz_ba_solve / each pass of ov2_local_ba record the
                                           iteration summaries of the solve (ov2_ba_get_trace); the batch entry point does not */
int  ov2_ctx_set_option(ov2_ctx *ctx, int option, int value);
int  ov2_ctx_get_option(ov2_ctx *ctx, int option, int *value);

/* ---- image pyramid -------------------------------------------------
 * Replaces cv::buildOpticalFlowPyramid(img, pyr, Size(win,win), max_level)
 *   src/visual_front_end.cpp:1172, :53 ; src/mapper.cpp:81
 * (withDerivatives=true, pyrBorder=REFLECT_101, derivBorder=CONSTANT).
 * `batch` independent images of identical size are built in one launch set
 * (batch=1 is the drop-in case; batch>1 is the offline batch-of-sequences
 * mode of BASELINE.json config 5).  Image b starts at img + b*img_batch_stride.
 */
int  ov2_pyr_create(ov2_ctx *ctx, int w, int h, int win, int max_level, int batch, ov2_pyr **out);
void ov2_pyr_destroy(ov2_pyr *p);
int  ov2_pyr_levels(const ov2_pyr *p);                 /* levels actually built */
int  ov2_pyr_level_size(const ov2_pyr *p, int level, int *w, int *h);
int  ov2_pyr_batch(const ov2_pyr *p);
/* A batch-1 pyramid that ALIASES batch item `item` of `p` (no copy, no allocation on the device): what the entry points that
 * take batch-1 pyramids (ov2_stereo_match, ov2_fb_klt, ov2_detect_*_d ...) need to work on one sequence of a lock-step batch.
 * The view shares p's `ready` hand-off (a consumer on another context waits for p's last build) and must be destroyed
 * (ov2_pyr_destroy) before p.                                                                                        */
int  ov2_pyr_item_view(const ov2_pyr *p, int item, ov2_pyr **out);
/* (re)build from host images: H2D copy + kernels, asynchronous on ctx's stream.  A batch-1 image is repacked into the context's
 * pinned staging buffer before the call returns (one contiguous DMA whatever the row stride; the caller's buffer is free at
 * once); for batch > 1 the host buffer must stay valid until ov2_ctx_sync / a later blocking call */
int  ov2_pyr_build_h(ov2_ctx *ctx, ov2_pyr *p, const uint8_t *img_h, int stride, size_t img_batch_stride);
/* (re)build from images already resident in HBM */
int  ov2_pyr_build_d(ov2_ctx *ctx, ov2_pyr *p, const uint8_t *img_d, int stride, size_t img_batch_stride);
/* D2H of one level of batch item b: un-padded image (w*h u8) and/or derivative
 * (w*h int16x2); either pointer may be NULL.  Blocking.                        */
int  ov2_pyr_download(ov2_ctx *ctx, const ov2_pyr *p, int b, int level, uint8_t *img_h, int16_t *deriv_h);
/* same but including the `win` border on every side ((w+2win)*(h+2win))        */
int  ov2_pyr_download_padded(ov2_ctx *ctx, const ov2_pyr *p, int b, int level, uint8_t *img_h, int16_t *deriv_h);
/* algorithmic HBM bytes of one build of one image (SURVEY.md 8d "B_pyr")       */
size_t ov2_pyr_algorithmic_bytes(const ov2_pyr *p);

/* ---- CLAHE ----------------------------------------------------------
 * Replaces cv::CLAHE::apply(img_raw, cur_img_) -- src/visual_front_end.cpp:1159 (left image, every
 * frame when use_clahe: 1), src/mapper.cpp:76 (right image) -- for the handle created at
 * src/ov2slam.cpp:85-89: cv::createCLAHE(clip_limit = fclahe_val, tiles = (w/50, h/50)).
 * CV_8UC1 only (what the reference feeds it).  _h: host buffers (drop-in); _d: `batch` images
 * already in HBM, src and dst must not overlap. */
int ov2_clahe_h(ov2_ctx *ctx, const uint8_t *src_h, int w, int h, int stride, double clip_limit, int tiles_x, int tiles_y,
                uint8_t *dst_h, int dst_stride);
int ov2_clahe_d(ov2_ctx *ctx, const uint8_t *src_d, int w, int h, int stride, size_t src_batch_stride, int batch,
                double clip_limit, int tiles_x, int tiles_y, uint8_t *dst_d, int dst_stride, size_t dst_batch_stride);
/* preprocessImage in one call: CLAHE of img_d written straight into the pyramid's level 0, then the
 * coarser levels -- the pair clahe->apply(img_raw, cur_img_) + cv::buildOpticalFlowPyramid(cur_img_, ...)
 * of src/visual_front_end.cpp:1159 + :1172 (mapper.cpp:76 + :81) without the intermediate image.
 * Results are identical to ov2_clahe_d followed by ov2_pyr_build_d; the equalised image stays
 * available as level 0 of the pyramid.                                                            */
int ov2_pyr_build_clahe_d(ov2_ctx *ctx, ov2_pyr *p, const uint8_t *img_d, int stride, size_t img_batch_stride,
                          double clip_limit, int tiles_x, int tiles_y);
/* same from a host image (batch-1 pyramid): one H2D of the raw frame, asynchronous on ctx's stream -- the
 * single-sequence form of VisualFrontEnd::preprocessImage (the image is staged in pinned memory before the call returns) */
int ov2_pyr_build_clahe_h(ov2_ctx *ctx, ov2_pyr *p, const uint8_t *img_h, int stride, double clip_limit, int tiles_x, int tiles_y);
/* the same for `n_items` host images (one pointer each, rows `stride` apart) into items [0, n_items) of a batch pyramid: one repack into
 * pinned memory, ONE H2D, the batched kernels -- the right images of the keyframes a lock-step batch reaches together
 * (src/mapper.cpp:74-81 per keyframe).  clip_limit < 0: no CLAHE (use_clahe: 0).  Asynchronous like ov2_pyr_build_clahe_h.        */
int ov2_pyr_build_clahe_hb(ov2_ctx *ctx, ov2_pyr *p, int n_items, const uint8_t *const *img_h, int stride, double clip_limit, int tiles_x, int tiles_y);

/* ---- Lucas-Kanade --------------------------------------------------
 * ov2_lk_track replaces one cv::calcOpticalFlowPyrLK(prevPyr, nextPyr, prevPts,
 * nextPts, status, err, Size(win,win), max_level, TermCriteria(COUNT+EPS,
 * max_iter, eps), flags, 1e-4)  -- src/feature_tracker.cpp:66-69, :113-116.
 * flags: OV2_LK_USE_INITIAL_FLOW | OV2_LK_GET_MIN_EIGENVALS (the only
 * combination the reference uses); without GET_MIN_EIGENVALS err is left 0.
 * All point buffers are host memory, n points per batch item, batch items
 * contiguous (n_per_item[b] points for item b stored at offset b*n_max).
 */
#define OV2_LK_USE_INITIAL_FLOW   4
#define OV2_LK_GET_MIN_EIGENVALS  8

int ov2_lk_track(ov2_ctx *ctx, const ov2_pyr *prev, const ov2_pyr *next,
                 int win, int max_level, int max_iter, float eps, int flags,
                 const float *prev_xy_h, float *next_xy_inout_h, int n,
                 uint8_t *status_h, float *err_h, int *iters_h /* per point, may be NULL */);

/* ov2_fb_klt replaces FeatureTracker::fbKltTracking (src/feature_tracker.cpp:35-137):
 * forward LK (max_level = nbpyrlvl, clamped to the pyramid), status / err>err_th /
 * 1-px border filter, backward LK at level 0 from the tracked point with the
 * original keypoint as initial guess, reject if |kp - back| > fb_dist.
 * One fused kernel launch.  prior_xy_inout_h: in = initial guess, out = tracked
 * position (entries whose forward level-0 step was skipped keep OpenCV's
 * semantics).  n == 0 returns OV2_OK and touches nothing (:43-46).
 * stats (may be NULL): [0] = total GN iterations, [1] = (point,level) patch builds. */
int ov2_fb_klt(ov2_ctx *ctx, const ov2_pyr *prev, const ov2_pyr *cur,
               int win, int nbpyrlvl, int max_iter, float eps, float err_th, float fb_dist,
               const float *kps_xy_h, float *prior_xy_inout_h, int n,
               uint8_t *status_h, long long stats[2]);

/* Device-resident, batched form used by the offline batch-of-sequences path and
 * by bench.py: kps / priors / status live in HBM, item b uses points
 * [b*n_max, b*n_max + n_per_item_d[b])  (n_per_item_d == NULL -> n_max each).
 * stats_d (may be NULL): 2 x int64 accumulated with atomics (zero it yourself). */
int ov2_fb_klt_d(ov2_ctx *ctx, const ov2_pyr *prev, const ov2_pyr *cur,
                 int win, int nbpyrlvl, int max_iter, float eps, float err_th, float fb_dist,
                 const float *kps_xy_d, float *prior_xy_inout_d, int n_max, const int *n_per_item_d,
                 uint8_t *status_d, long long *stats_d);

/* ---- single-sequence tracker: preprocessImage + kltTracking ----------------------------------
 * One object per camera stream that owns what VisualFrontEnd keeps between frames -- prev_pyr_ / cur_pyr_
 * (src/visual_front_end.hpp) -- plus pinned staging buffers and, optionally, a captured hipGraph of the whole
 * per-frame enqueue.  It replaces, per frame, on the SLAM thread:
 *   VisualFrontEnd::preprocessImage  src/visual_front_end.cpp:1143-1177  (pyramid swap :1169, CLAHE :1159,
 *                                                                         cv::buildOpticalFlowPyramid :1172)
 *   VisualFrontEnd::kltTracking      src/visual_front_end.cpp:132-275    (both fbKltTracking calls :196 / :242, the
 *                                                                         retry of lost prior tracks :213-217 and the
 *                                                                         "motion model is wrong" rule :225-230)
 * with ONE H2D of the frame, ONE H2D of the keypoint block, five small kernels, ONE LK launch, ONE D2H and ONE
 * host synchronisation.  Results are identical to calling ov2_pyr_build_clahe_h + ov2_fb_klt twice.           */
typedef struct ov2_tracker ov2_tracker;
typedef struct {
    int w, h;                    /* image size                                                          */
    int win;                     /* nklt_win_size (9)                                                   */
    int nklt_pyr_lvl;            /* pyramid levels above 0 (3): full-pyramid pass                       */
    int prior_pyr_lvl;           /* nbpyrlvl of the 3-D-prior pass (1, visual_front_end.cpp:188)        */
    int max_iter; float eps;     /* klt_convg_crit_: nmax_iter (30), fmax_px_precision (0.01)           */
    float err_th, fb_dist;       /* nklt_err (30), fmax_fbklt_dist (0.5)                                */
    int use_clahe; double clahe_clip; int tiles_x, tiles_y;   /* use_clahe, fclahe_val, (w/50, h/50)    */
    int n_max;                   /* keypoints per fused launch (>= nbmaxkps; the adapters pass 2 x nbmaxkps).  NOT a
                                    limit on n: a frame can carry more than nbmaxkps keypoints (pruning happens at the
                                    next keyframe, src/map_manager.cpp:74); keypoints beyond n_max run through further
                                    launches of the same kernel, n_max at a time, with identical results              */
    int use_graph;               /* 1: replay a captured hipGraph per frame (falls back to plain enqueue
                                    when the context's stream cannot be captured)                       */
} ov2_tracker_config;

int  ov2_tracker_create(ov2_ctx *ctx, const ov2_tracker_config *cfg, ov2_tracker **out);
void ov2_tracker_destroy(ov2_tracker *t);
/* Pinned staging image the next frame may be written into directly (camera driver / decoder / cv::Mat header
 * over it): a frame passed from there skips the host-side copy.  *stride receives its pitch.               */
uint8_t *ov2_tracker_image_buffer(ov2_tracker *t, int *stride);
/* preprocessImage: swap prev/cur, H2D of the frame, CLAHE (if configured) + pyramid build.  ASYNCHRONOUS: returns
 * after the enqueue so the host can run its motion model while the GPU works; the image is staged through the
 * pinned buffer first, so img_h may be reused immediately.                                                 */
int  ov2_tracker_preprocess(ov2_tracker *t, const uint8_t *img_h, int stride);
/* kltTracking on (prev, cur) of the tracker.  has_prior_h[i] != 0: keypoint i carries a 3-D prior
 * (prior_xy_h[i] = projected map point) and is tracked on prior_pyr_lvl levels first; otherwise pass
 * prior_xy_h[i] = kps_xy_h[i] like the reference (:180-182).  klt_use_prior = 0 ignores has_prior_h.
 * out_xy_h[i]: tracked position (the value the reference hands to updateKeypoint, :204 / :256);
 * status_h[i]: bit 0 = tracked, bit 1 = lost on the prior pass and re-tracked on the full pyramid;
 * *p3p_req (may be NULL): 1 when fewer than a third of the prior tracks were good (bp3preq_, :225-230) -- the lost
 * prior tracks are then re-run from the keypoints themselves in a second launch, exactly like the reference.
 * Blocking (one synchronisation; one more per n_max keypoints beyond the first n_max).  n == 0 returns OV2_OK.     */
int  ov2_tracker_klt(ov2_tracker *t, const float *kps_xy_h, const float *prior_xy_h, const uint8_t *has_prior_h, int n,
                     int klt_use_prior, float *out_xy_h, uint8_t *status_h, int *p3p_req);
/* preprocess + klt in one enqueue (one graph launch when use_graph): the per-frame call of the drop-in.
 * The first frame after creation only builds the pyramid (nothing to track against): status_h is zeroed.   */
int  ov2_tracker_track_frame(ov2_tracker *t, const uint8_t *img_h, int stride, const float *kps_xy_h,
                             const float *prior_xy_h, const uint8_t *has_prior_h, int n, int klt_use_prior,
                             float *out_xy_h, uint8_t *status_h, int *p3p_req);
/* Optional: Frame::computeKeypoint (src/frame.cpp:246-254: undistortImagePoint + bearing vector, what the reference runs for
 * every keypoint it has just tracked, updateKeypoint) for every output position INSIDE the per-frame enqueue -- no second call,
 * no second synchronisation.  Same arguments as ov2_compute_keypoints.  Re-captures the tracker's graphs: call it right after
 * ov2_tracker_create.  ov2_tracker_last_keypoints then returns unpx (2 floats) / bv (3 doubles) per keypoint of the LAST
 * ov2_tracker_klt / _track_frame call (entries of untracked keypoints are computed from their last forward position).   */
int  ov2_tracker_set_calibration(ov2_tracker *t, int model, const double K[4], const double *D, int nD, const double iK[9]);
int  ov2_tracker_last_keypoints(const ov2_tracker *t, int n, float *unpx_xy_h, double *bv_xyz_h);
/* the tracker's pyramids (valid until the next preprocess), e.g. for createKeyframe / stereo matching / detection */
const ov2_pyr *ov2_tracker_cur_pyr(const ov2_tracker *t);
const ov2_pyr *ov2_tracker_prev_pyr(const ov2_tracker *t);
int  ov2_tracker_frames(const ov2_tracker *t);     /* frames preprocessed so far */
int  ov2_tracker_uses_graph(const ov2_tracker *t); /* 1 when the graph path is active */

/* ---- lock-step tracker: `batch` camera streams advance ONE FRAME PER CALL ---------------------------------
 * The offline / batch mode of the reference's benchmark protocol (benchmark_scripts/euroc_bench.sh:3-27 runs whole sequences one
 * after the other; BASELINE.json configs[4] shards them over GPUs): a rank that owns several sequences does not need their frames
 * one stream at a time.  One stream is a chain of ~10 small dependent launches per frame and a rank's streams together saturate
 * the launch rate with the CUs ~5 % busy (profiles/archive/r4_stream_concurrency.txt); here every launch of the per-frame enqueue --
 * frame upload, CLAHE, pyramid, the fused kltTracking kernel (both fbKltTracking calls + retry), Frame::computeKeypoint -- covers
 * all streams at once: same kernels as ov2_tracker_*, grid extended by the batch item, ONE synchronisation per step.
 * Results per item are bit-identical to an ov2_tracker fed the same frames / keypoints (tests/test_gpu_lockstep.py).
 *   items [0, n_active) take part in a call (sequences of different length: order them longest first and shrink n_active as
 *   they end); the state of the other items is not touched.
 *   point arrays hold cfg->n_max slots per item: item b's points are [b*n_max, b*n_max + n_h[b]); n_h[b] <= n_max.
 *   images: img_h[b] = frame of item b, rows `stride` bytes apart.  Three pinned staging sets exist (which = 0 / 1 / 2,
 *   ov2_btracker_image_buffer); frames passed from the slots of one set are not copied on the host.
 *   Look-ahead (optional; an offline host knows the next frames): the step is a three-stage pipeline on three streams --
 *     ov2_btracker_upload(which)    H2D of a filled staging set on the tracker's copy stream
 *     ov2_btracker_prepare(which)   preprocessImage (CLAHE + pyramid) of that set on the tracker's prep stream, into the pyramid set
 *                                   that becomes cur_pyr_ when the frame is tracked
 *     ov2_btracker_track_frame      with exactly those slots as img_h[]: only kltTracking + computeKeypoint remain (its stream waits
 *                                   for the pyramids); with other frames, or without the look-ahead calls, everything runs in order.
 *   A host loop: fill set (f+2)%3 [reader threads] -> upload((f+2)%3) -> prepare((f+1)%3) -> track_frame(frame f = set f%3).
 *   Results do not depend on which form is used.
 * hipGraph replay (cfg->use_graph) is not used here: n_active changes the grids.                                          */
typedef struct ov2_btracker ov2_btracker;
int  ov2_btracker_create(ov2_ctx *ctx, const ov2_tracker_config *cfg, int batch, ov2_btracker **out);
void ov2_btracker_destroy(ov2_btracker *t);
int  ov2_btracker_batch(const ov2_btracker *t);
int  ov2_btracker_frames(const ov2_btracker *t);
/* pinned slot of item `item` in staging set `which` (0 / 1 / 2); *stride receives its pitch */
uint8_t *ov2_btracker_image_buffer(ov2_btracker *t, int which, int item, int *stride);
/* asynchronous H2D of items [0, n_active) of staging set `which` (the caller has filled the slots); the next
 * ov2_btracker_prepare / ov2_btracker_track_frame of exactly those slots consumes the uploaded copy                   */
int  ov2_btracker_upload(ov2_btracker *t, int which, int n_active);
/* asynchronous preprocessImage of items [0, n_active) of staging set `which` for an ov2_btracker_track_frame to come: frames are
 * prepared in order, at most two may wait (the one about to be tracked and the one after it), and once a frame is prepared the
 * track_frame calls must consume exactly the prepared slots.  It overwrites the oldest pyramid set: see ov2_btracker_pyramid_sets   */
int  ov2_btracker_prepare(ov2_btracker *t, int which, int n_active);
/* Frame::computeKeypoint inside the per-step enqueue, as ov2_tracker_set_calibration (one calibration: the sequences of a batch
 * come from one camera rig)                                                                                            */
int  ov2_btracker_set_calibration(ov2_btracker *t, int model, const double K[4], const double *D, int nD, const double iK[9]);
/* preprocessImage + kltTracking of items [0, n_active): the lock-step form of ov2_tracker_track_frame, same per-item semantics
 * (first frame: pyramids only; has_prior_h / klt_use_prior / status bits / p3p_req[b] as there, the "motion model is wrong" retry
 * of visual_front_end.cpp:225-230 included).  Blocking: one synchronisation.                                             */
int  ov2_btracker_track_frame(ov2_btracker *t, int n_active, const uint8_t *const *img_h, int stride, const float *kps_xy_h,
                              const float *prior_xy_h, const uint8_t *has_prior_h, const int *n_h, int klt_use_prior,
                              float *out_xy_h, uint8_t *status_h, int *p3p_req);
/* The same step in two halves: _begin enqueues it (frames, pre-processing or the wait for the prepared pyramids, the tracking
 * kernels) and returns; _end waits for it, returns the results and applies the p3p rule.  Between the two the host is free -- an
 * offline driver issues ov2_btracker_upload / ov2_btracker_prepare of the frames to come there, so that their enqueue cost runs beside
 * the tracking kernels instead of before them.  kps_xy_h / has_prior_h must stay valid until _end (the p3p rule re-reads them).
 * ov2_btracker_track_frame = _begin + _end.                                                                              */
int  ov2_btracker_track_frame_begin(ov2_btracker *t, int n_active, const uint8_t *const *img_h, int stride, const float *kps_xy_h,
                                    const float *prior_xy_h, const uint8_t *has_prior_h, const int *n_h, int klt_use_prior);
int  ov2_btracker_track_frame_end(ov2_btracker *t, float *out_xy_h, uint8_t *status_h, int *p3p_req);
/* unpx (2 floats) / bv (3 doubles) of the first n keypoints of item `item` from the LAST ov2_btracker_track_frame */
int  ov2_btracker_last_keypoints(const ov2_btracker *t, int item, int n, float *unpx_xy_h, double *bv_xyz_h);
/* MapManager::extractKeypoints on the current frame of items [0, n_active) in one call (all sequences of a lock-step batch reach
 * their keyframes together): ov2_detect_singlescale_batch_d / ov2_detect_grid_fast_batch_d on level 0 of the current pyramids
 * with host buffers -- cur_xy_h: n_max slots per item, ncur_h[b] of them valid; out_xy_h: out_cap slots per item
 * (>= 2*(w/cell)*(h/cell), FAST: (w/cell)*(h/cell)); quality_inout / fast_th_inout: one adaptive state per item.          */
int  ov2_btracker_detect_singlescale(ov2_btracker *t, int n_active, int cell, const float *cur_xy_h, const int *ncur_h, const int roi[4],
                                     double *quality_inout, int do_subpix, float *out_xy_h, int out_cap, int *out_n_h);
int  ov2_btracker_detect_grid_fast(ov2_btracker *t, int n_active, int cell, const float *cur_xy_h, const int *ncur_h, int *fast_th_inout,
                                   int mask_mode, int do_subpix, float *out_xy_h, int out_cap, int *out_n_h);
/* the current / previous frame's pyramids: the whole batch, or item `item` as a batch-1 view (owned by the tracker; valid until that
 * pyramid set comes round again, see ov2_btracker_pyramid_sets) -- what the mapper context passes to ov2_stereo_match as `left` */
/* How many pyramid sets the tracker rotates through (8): the pyramids of frame f are overwritten by the pre-processing of frame
 * f + sets -- ov2_btracker_track_frame of that frame, or the ov2_btracker_prepare call for it.  A consumer on another context (the
 * mapper's stereo matching of keyframe f) must be done before the caller issues that call. */
int  ov2_btracker_pyramid_sets(const ov2_btracker *t);
const ov2_pyr *ov2_btracker_cur_pyr(const ov2_btracker *t);
const ov2_pyr *ov2_btracker_prev_pyr(const ov2_btracker *t);
const ov2_pyr *ov2_btracker_cur_item(const ov2_btracker *t, int item);
const ov2_pyr *ov2_btracker_prev_item(const ov2_btracker *t, int item);

/* ---- keypoint detection ---------------------------------------------
 * mask_mode for the FAST grid detector: the reference passes a
 * CV_32F mask to cv::FastFeatureDetector::detect, which reads it as bytes.     */
#define OV2_MASK_AS_EXECUTED 0
#define OV2_MASK_INTENDED    1

/* FeatureExtractor::detectGridFAST (src/feature_extractor.cpp:443-570).
 * fast_th_inout mirrors the member nfast_th_ (adapted at :546-552).
 * out_xy_h capacity (w/cell)*(h/cell) points.  do_subpix=0 skips cv::cornerSubPix. */
int ov2_detect_grid_fast(ov2_ctx *ctx, const uint8_t *img_h, int w, int h, int stride, int cell,
                         const float *cur_xy_h, int ncur, int *fast_th_inout, int mask_mode,
                         int do_subpix, float *out_xy_h, int *out_n);

/* FeatureExtractor::detectSingleScale (src/feature_extractor.cpp:288-440).
 * roi = {x, y, width, height} (the 5-px border rect of camera_calibration.cpp:72-73).
 * quality_inout mirrors dmaxquality_ (:418-423).  out_xy_h capacity 2*(w/cell)*(h/cell). */
int ov2_detect_singlescale(ov2_ctx *ctx, const uint8_t *img_h, int w, int h, int stride, int cell,
                           const float *cur_xy_h, int ncur, const int roi[4], double *quality_inout,
                           int do_subpix, float *out_xy_h, int *out_n);

/* Device-resident forms of the two detectors: the image is level 0 of batch item `item` of a pyramid that already lives in
 * HBM.  At a keyframe the reference passes cur_img_ -- the CLAHE'd frame preprocessImage also built cur_pyr_ from -- to
 * MapManager::extractKeypoints (src/map_manager.cpp:286-341, detector choice :312-320): that image IS level 0 of the
 * tracker's current pyramid (ov2_tracker_cur_pyr), so a keyframe costs no upload.  Results are identical to the host-image
 * forms on the same pixels.  One host synchronisation per call.                                                          */
int ov2_detect_grid_fast_d(ov2_ctx *ctx, const ov2_pyr *pyr, int item, int cell, const float *cur_xy_h, int ncur,
                           int *fast_th_inout, int mask_mode, int do_subpix, float *out_xy_h, int *out_n);
int ov2_detect_singlescale_d(ov2_ctx *ctx, const ov2_pyr *pyr, int item, int cell, const float *cur_xy_h, int ncur,
                             const int roi[4], double *quality_inout, int do_subpix, float *out_xy_h, int *out_n);

/* The same detectors on EVERY batch item of the pyramid in one call (the offline batch-of-sequences mode: all sequences reach
 * a keyframe together).  Everything but the adaptive per-sequence state stays on the device:
 *   cur_xy_d   device, batch x cur_cap points (x, y): the items' current keypoints; ncur_d device, batch counts (NULL: none)
 *   out_xy_d   device, batch x out_cap points; out_cap >= (w/cell)*(h/cell) for FAST, twice that for single scale
 *   quality_inout / fast_th_inout   host, one entry per item (dmaxquality_ / nfast_th_ of that sequence), updated like the
 *              single-image forms; out_n_h host, points written per item.
 * Results per item are identical to the single-image forms.  One host synchronisation per call.                        */
int ov2_detect_singlescale_batch_d(ov2_ctx *ctx, const ov2_pyr *pyr, int cell, const float *cur_xy_d, int cur_cap, const int *ncur_d,
                                   const int roi[4], double *quality_inout, int do_subpix, float *out_xy_d, int out_cap, int *out_n_h);
int ov2_detect_grid_fast_batch_d(ov2_ctx *ctx, const ov2_pyr *pyr, int cell, const float *cur_xy_d, int cur_cap, const int *ncur_d,
                                 int *fast_th_inout, int mask_mode, int do_subpix, float *out_xy_d, int out_cap, int *out_n_h);

/* cv::cornerSubPix(im, pts, Size(hw,hw), Size(-1,-1), TermCriteria(EPS+MAX_ITER, max_iter, eps))
 * src/feature_extractor.cpp:434, :564 (hw = 3, 30, 0.01).  In place.             */
int ov2_corner_subpix(ov2_ctx *ctx, const uint8_t *img_h, int w, int h, int stride,
                      float *xy_inout_h, int n, int half_win, int max_iter, double eps);

/* ---- local bundle adjustment ----------------------------------------
 * Replaces the two ceres::Solve calls of Optimizer::localBA
 * (src/optimizer.cpp:479 and :618) on the anchored-inverse-depth problem built at
 * :128-407.  The adapter walks the map on the CPU (as the reference does) into
 * the flat arrays below.  Residual types follow src/ceres_parametrization.cpp.
 */
enum {
    OV2_RES_LEFT       = 0, /* DirectLeftSE3::ReprojectionErrorKSE3AnchInvDepth         :361-473 */
    OV2_RES_RIGHT      = 1, /* DirectLeftSE3::ReprojectionErrorRightCamKSE3AnchInvDepth :579-712 */
    OV2_RES_RIGHT_ANCH = 2, /* DirectLeftSE3::ReprojectionErrorRightAnchCamKSE3AnchInvDepth :476-577 */
    OV2_RES_PNP        = 3  /* DirectLeftSE3::ReprojectionErrorSE3 (fixed world point) :301-358 -- the factor of
                               MultiViewGeometry::ceresPnP, src/multi_view_geometry.cpp:492-586; uses res_kf, res_xyz,
                               res_uv, res_sigma and calib_l; res_lm is ignored */
};

typedef struct {
    int n_kf;                    /* keyframe poses Twc, [tx ty tz qx qy qz qw] each (se3_param_block.hpp:40-46) */
    const double *poses;         /* 7*n_kf                                        */
    const uint8_t *kf_const;     /* n_kf; 1 = SetParameterBlockConstant (optimizer.cpp:176-185, :229-246, :397-407) */
    int n_lm;                    /* anchored inverse-depth landmarks              */
    const double *invdepth;      /* n_lm                                          */
    const int *lm_anchor_kf;     /* n_lm; index into poses                        */
    const double *lm_anchor_uv;  /* 2*n_lm; undistorted anchor pixel              */
    int n_res;                   /* 2-row residual blocks                         */
    const uint8_t *res_type;     /* n_res; OV2_RES_*                              */
    const int *res_kf;           /* n_res; observing keyframe (unused for RIGHT_ANCH) */
    const int *res_lm;           /* n_res                                         */
    const double *res_uv;        /* 2*n_res; observed pixel                       */
    const double *res_sigma;     /* n_res; 2^scale (always 1 in the reference)    */
    const uint8_t *res_active;   /* n_res or NULL; 0 = residual block removed (optimizer.cpp:500-592) */
    const double *res_xyz;       /* 3*n_res; world point of OV2_RES_PNP blocks (ignored for the others); NULL if none */
    double calib_l[4];           /* fx fy cx cy, constant block                   */
    double calib_r[4];
    double T_rl[7];              /* right <- left extrinsic, [t q], constant block */
} ov2_ba_problem;

typedef struct {
    int max_iter;                /* 5 (robust pass) / 10 (L2 pass), optimizer.cpp:461, :611 */
    double function_tolerance;   /* 1e-3, :462                                    */
    double gradient_tolerance;   /* Ceres default 1e-10                           */
    double parameter_tolerance;  /* Ceres default 1e-8                            */
    double huber_delta;          /* sqrt(5.9915), :49; <= 0: no loss function     */
    double initial_radius;       /* 1e4  (Ceres initial_trust_region_radius)      */
    double max_radius;           /* 1e16                                          */
    double min_radius;           /* 1e-32                                         */
    double min_lm_diagonal;      /* 1e-6                                          */
    double max_lm_diagonal;      /* 1e32                                          */
    double min_relative_decrease;/* 1e-3                                          */
    int jacobi_scaling;          /* 1                                             */
    int max_consecutive_invalid_steps; /* 5                                       */
    double max_solver_time_s;    /* Ceres max_solver_time_in_seconds: 0.2 / 0.1 s in localBA when force_realtime (optimizer.cpp:464-468,
                                    :612), 5 ms in ceresPnP (multi_view_geometry.cpp:546), 10-20 ms in structureOnlyBA; <= 0 = no limit
                                    (the default: results are then independent of machine load).  The limit is checked by the host
                                    between chunks of 2 LM iterations; when it fires no further iteration is started and the solve
                                    returns the last accepted state with OV2_TERM_NO_CONVERGENCE, like Ceres' "maximum solver time" exit */
} ov2_ba_options;

enum {
    OV2_TERM_NO_CONVERGENCE = 0, OV2_TERM_FUNCTION_TOL = 1, OV2_TERM_PARAMETER_TOL = 2,
    OV2_TERM_GRADIENT_TOL = 3, OV2_TERM_MIN_RADIUS = 4, OV2_TERM_INVALID_STEPS = 5,
    OV2_TERM_FAILURE = 6
};

typedef struct {
    double *poses_out;           /* 7*n_kf                                        */
    double *invdepth_out;        /* n_lm                                          */
    double *chi2_last_eval;      /* n_res: chi2err_ cached by the last Evaluate */
    uint8_t *depthpos_last_eval; /* n_res: isdepthpositive_ likewise              */
    int iterations;              /* LM iterations executed (Ceres summary.iterations.size()-1) */
    int num_successful_steps;
    double initial_cost, final_cost;
    int termination;             /* OV2_TERM_*                                    */
    double solve_ms;             /* device time of the solve (HIP events)         */
} ov2_ba_result;

void ov2_ba_default_options(ov2_ba_options *o);
/* One ceres::Solve: H2D of the problem, the whole LM loop on the device (a fixed kernel sequence,
 * one host synchronisation), D2H of the result.  When p->res_active is given, the entries of
 * r->chi2_last_eval / r->depthpos_last_eval that belong to inactive residual blocks are IN/OUT:
 * they keep the caller's values, like the cached chi2err_ of a removed residual block (N4).
 * Size: up to ~90 optimised keyframes the reduced system is solved in one work-group's LDS; beyond that (a loop-closure
 * fullBA) a sparse-W / HBM-Cholesky path takes over by itself, same results, up to 1024 optimised keyframes (the dense reduced
 * system: 3 x 302 MB at the cap), with or without OV2_RES_PNP blocks; past that OV2_EUNSUPPORTED with a message, nothing enqueued. */
int  ov2_ba_solve(ov2_ctx *ctx, const ov2_ba_problem *p, const ov2_ba_options *o, ov2_ba_result *r);

/* Iteration trace of the LAST one-problem solve of this context (OV2_OPT_BA_TRACE = 1): one entry per iteration that Ceres'
 * TrustRegionMinimizer pushes into Solver::Summary::iterations (Thirdparty/ceres-solver/internal/ceres/trust_region_minimizer.cc:313-337;
 * include/ceres/iteration_callback.h:45-150) -- entry 0 is the starting point; an iteration that ends the solve inside the loop
 * (parameter / function tolerance, :706-748) is not recorded, as in Ceres.  gradient_norm is NaN (the device forms the max norm only).
 * *n = entries recorded by the solve (the library keeps the first 64), buf receives min(*n, 64, cap).  tests/test_gpu_ba.py compares it
 * with the oracle's trace; tests/test_reference_trlm.py compares both with Ceres' own loop compiled in place.                       */
typedef struct {
    int iteration, step_is_valid, step_is_successful, reserved_;
    double cost, cost_change, gradient_max_norm, gradient_norm, step_norm, relative_decrease, trust_region_radius;
} ov2_ba_iter;
int  ov2_ba_get_trace(ov2_ctx *ctx, ov2_ba_iter *buf, int cap, int *n);

/* Same solve on a problem that is already resident in HBM (upload once, solve many times from the
 * same initial parameters); used by bench.py so that the timed region starts with inputs in HBM. */
typedef struct ov2_ba_dev ov2_ba_dev;
int  ov2_ba_create(ov2_ctx *ctx, const ov2_ba_problem *p, ov2_ba_dev **out);
int  ov2_ba_solve_resident(ov2_ctx *ctx, ov2_ba_dev *dev, const ov2_ba_options *o, ov2_ba_result *r);
void ov2_ba_destroy(ov2_ba_dev *dev);

/* Optimizer::localBA's whole solve stage (src/optimizer.cpp:436-735) in ONE call with the problem resident in HBM between the
 * two ceres::Solve calls: one sort + upload of the residual blocks, pass 1 (Huber unless !use_robust_cost), the outlier test
 * on the values cached by the last Evaluate (chi2err_ > robust_mono_th or depth <= 0, :492-594) and the removal
 * of the outlier blocks ON THE DEVICE, pass 2 (only if apply_l2_after_robust && use_robust_cost && !stop_requested && outliers
 * were found, :603-604; loss reset to L2 only when a left AND a right-camera block remain, :606-608 -- mono runs keep Huber),
 * the second outlier test on the blocks still in the problem (:637-735), one download.  Two ov2_ba_solve calls return the same
 * (tests/test_gpu_ba.py) but sort, upload and download the residual blocks twice: 11.5 -> see bench `localba_two_pass_stereo`.
 * pass1 / pass2 carry the iteration caps and tolerances (their huber_delta is ignored: the protocol sets it).  Inverse-depth
 * problems without OV2_RES_PNP blocks; same size limits as ov2_ba_solve.                                                */
typedef struct {
    double robust_mono_th;       /* 5.9915 (slam_params.hpp robust_mono_th_)                              */
    int use_robust_cost;         /* localBA's buse_robust_cost argument                                   */
    int apply_l2_after_robust;   /* apply_l2_after_robust_                                                */
    int stop_requested;          /* a stop that is already known at entry (ORed with *stop_flag)          */
    const volatile int *stop_flag;/* or NULL.  The LIVE Optimizer::bstop_localba_ (include/optimizer.hpp:48-49): the reference tests
                                    !stopLocalBA() AFTER its first ceres::Solve (:603-604) and Estimator::addNewKf raises the flag from
                                    another thread while pass 1 runs, so the library reads *stop_flag once, right before it decides
                                    on pass 2 (after pass 1 and the first outlier test), never at entry                              */
    ov2_ba_options pass1, pass2; /* max_iter 5 / 10, function_tolerance 1e-3 (:461-462, :611).  max_solver_time_s: the reference runs
                                    pass 1 with 0.2 s (0.4 s unless force_realtime, :463-467) and pass 2 with HALF of that (:612); the
                                    defaults here are 0 = no limit (results independent of machine load): an adapter that wants the
                                    reference's limits sets pass1.max_solver_time_s = t and pass2.max_solver_time_s = t / 2           */
} ov2_local_ba_options;
typedef struct {
    double *poses_out;           /* 7*n_kf                                                                */
    double *invdepth_out;        /* n_lm                                                                  */
    uint8_t *bad_obs;            /* n_res: 1 = outlier after the whole protocol (remove the observation)  */
    uint8_t *bad_after_pass1;    /* n_res or NULL: verdicts of the first test only                        */
    double *chi2_last_eval;      /* n_res or NULL (not downloaded)                                        */
    uint8_t *depthpos_last_eval; /* n_res or NULL                                                         */
    int l2_done;                 /* pass 2 ran (and succeeded)                                            */
    int pass2_error;             /* OV2_OK, or why pass 2 could not run: the call then still returns OV2_OK with the valid result of
                                    pass 1 + first outlier test in every output (what the reference keeps when its second Solve
                                    gives up); the message is in ov2_last_error()                                                    */
    int n_bad_pass1, n_bad_total;
    int iterations[2], num_successful_steps[2], termination[2];
    double initial_cost[2], final_cost[2];
    double solve_ms[2];          /* device time of each pass                                              */
    int status;                  /* OV2_OK, or this problem's error code (ABI 600).  ov2_local_ba: the value it returns.  ov2_local_ba_batch:
                                    every problem is attempted; r[i].status tells which results are valid and the call returns the
                                    first non-OK status (OV2_OK when all are)                                                        */
} ov2_local_ba_result;
void ov2_local_ba_default_options(ov2_local_ba_options *o);
int  ov2_local_ba(ov2_ctx *ctx, const ov2_ba_problem *p, const ov2_local_ba_options *o, ov2_local_ba_result *r);
/* The same protocol for n problems at once -- the estimator side of the lock-step batch of sequences (BASELINE configs[4]; the
 * reference runs one Optimizer::localBA per sequence on that sequence's estimator thread, src/estimator.cpp:71-93).  Every kernel
 * of the solver is launched ONCE for the batch (grid.z = problem; a problem that has converged, or takes no second pass, drops
 * out inside the kernels), so n solves cost the launches of one and the one-work-group kernels (factorisation, trust-region
 * bookkeeping) run side by side.  p, o, r: n entries each; the entries of o share robust_mono_th, use_robust_cost,
 * apply_l2_after_robust, pass1 and pass2 (OV2_EINVAL otherwise) -- stop_requested / stop_flag are per problem and read after the
 * first pass of the batch.  Per problem the result is what ov2_local_ba returns for it (same device code; floating-point sums
 * over work-groups are grouped by the batch's grid: parity 1e-7, tests/test_gpu_ba_batch.py); solve_ms is the device time of
 * the batch's pass.  Problems the shared launches do not cover (more optimised keyframes than the LDS-resident path holds,
 * OV2_RES_PNP blocks, no landmarks, OV2_OPT_BA_DETERMINISTIC) are solved one after the other through ov2_local_ba in the same
 * call; *n_batched (or NULL) = how many shared the launches; a failure of one of them (r[i].status) does not stop the others.
 * max_solver_time_s is ONE host-clock budget for the batch's shared pass (the batch advances iteration by iteration, so a slow
 * window ends the pass for all): a caller that wants the reference's per-problem limit leaves it at 0 here, or calls ov2_local_ba
 * per problem.                                                                                                                 */
int  ov2_local_ba_batch(ov2_ctx *ctx, int n, const ov2_ba_problem *p, const ov2_local_ba_options *o, ov2_local_ba_result *r, int *n_batched);

/* ------------------------------------------------------------------ */
/* Optimizer::structureOnlyBA                                           */
/* ------------------------------------------------------------------ */
/* One ceres::Solve of Optimizer::structureOnlyBA (src/optimizer.cpp:2594-2781, called at
 * src/loop_closer.cpp:353 on the map points merged by a loop closure): 3-D world points
 * (PointXYZParametersBlock) are the only variables; every keyframe pose, both calibrations and the stereo
 * extrinsic are constant blocks.  Residual blocks:
 *   OV2_XYZ_LEFT   DirectLeftSE3::ReprojectionErrorKSE3XYZ          (left camera,  :2692-2702, :2719-2727)
 *   OV2_XYZ_RIGHT  DirectLeftSE3::ReprojectionErrorRightCamKSE3XYZ  (right camera through T_rl, :2704-2715)
 * Options: the reference uses DENSE_SCHUR / LM, max_num_iterations 10, function_tolerance 1e-3, Huber
 * sqrt(robust_mono_th) (:2599-2601, :2742-2758) -- fill an ov2_ba_options accordingly; its 10-20 ms
 * max_solver_time_in_seconds has no counterpart.  The function never changes poses.                    */
enum { OV2_XYZ_LEFT = 0, OV2_XYZ_RIGHT = 1 };
typedef struct {
    int n_kf;
    const double *poses;         /* 7*n_kf  [tx ty tz qx qy qz qw] of Twc (constant)          */
    int n_pts;
    const double *xyz;           /* 3*n_pts world points, initial values                      */
    int n_res;
    const uint8_t *res_type;     /* n_res   OV2_XYZ_*                                         */
    const int *res_kf;           /* n_res   observing keyframe                                */
    const int *res_pt;           /* n_res   observed point                                    */
    const double *res_uv;        /* 2*n_res undistorted pixel (unpx_ / runpx_)                */
    const double *res_sigma;     /* n_res   2^scale                                           */
    const uint8_t *res_active;   /* n_res or NULL                                             */
    double calib_l[4], calib_r[4], T_rl[7];
} ov2_sba_problem;
typedef struct {
    double *xyz_out;             /* 3*n_pts                                                   */
    double *chi2_last_eval;      /* n_res, in/out like ov2_ba_result (may be NULL)            */
    uint8_t *depthpos_last_eval; /* n_res, in/out (may be NULL)                               */
    int iterations, num_successful_steps;
    double initial_cost, final_cost;
    int termination;             /* OV2_TERM_*                                                */
    double solve_ms;
} ov2_sba_result;
int ov2_structure_ba(ov2_ctx *ctx, const ov2_sba_problem *p, const ov2_ba_options *o, ov2_sba_result *r);

/* ------------------------------------------------------------------ */
/* Bundle adjustment over 3-D points with variable poses (buse_inv_depth: 0) */
/* ------------------------------------------------------------------ */
/* One ceres::Solve of Optimizer::localBA / looseBA / fullBA when `buse_inv_depth: 0`: map points enter as
 * PointXYZParametersBlock (3 doubles, elimination group 0, src/optimizer.cpp:207-209) and every observation is a
 *   OV2_XYZ_LEFT   DirectLeftSE3::ReprojectionErrorKSE3XYZ          {calib, pose, X}            (:333-384, :366-375)
 *   OV2_XYZ_RIGHT  DirectLeftSE3::ReprojectionErrorRightCamKSE3XYZ  {calib_r, pose, T_rl, X}    (:347-357)
 * residual block with a VARIABLE keyframe pose (factors src/ceres_parametrization.cpp:107-298); kf_const marks the
 * SetParameterBlockConstant keyframes (:397-407).  Same options, termination codes and N4 outputs as ov2_ba_solve; the
 * Schur complement eliminates 3x3 point blocks.  No shipped parameter file selects this branch; the inverse-depth form
 * (ov2_ba_solve) is what every preset runs.  Limit: ~450 optimised keyframes (OV2_EUNSUPPORTED beyond): W stays dense in
 * this form (3 rows per wavefront in LDS); beyond ~90 keyframes its reduced system is factored by the same multi-kernel
 * Cholesky on HBM as ov2_ba_solve's large-problem path (which reaches 1024 keyframes with a sparse W).              */
typedef struct {
    int n_kf;
    const double *poses;         /* 7*n_kf  [tx ty tz qx qy qz qw] of Twc, initial values      */
    const uint8_t *kf_const;     /* n_kf; 1 = constant block (NULL: every pose variable)       */
    int n_pts;
    const double *xyz;           /* 3*n_pts world points, initial values                       */
    int n_res;
    const uint8_t *res_type;     /* n_res   OV2_XYZ_*                                          */
    const int *res_kf;           /* n_res   observing keyframe                                 */
    const int *res_pt;           /* n_res   observed point                                     */
    const double *res_uv;        /* 2*n_res undistorted pixel (unpx_ / runpx_)                 */
    const double *res_sigma;     /* n_res   2^scale                                            */
    const uint8_t *res_active;   /* n_res or NULL                                              */
    double calib_l[4], calib_r[4], T_rl[7];
} ov2_xyzba_problem;
typedef struct {
    double *poses_out;           /* 7*n_kf                                                     */
    double *xyz_out;             /* 3*n_pts                                                    */
    double *chi2_last_eval;      /* n_res, in/out like ov2_ba_result                           */
    uint8_t *depthpos_last_eval; /* n_res, in/out                                              */
    int iterations, num_successful_steps;
    double initial_cost, final_cost;
    int termination;             /* OV2_TERM_*                                                 */
    double solve_ms;
} ov2_xyzba_result;
int ov2_xyz_ba_solve(ov2_ctx *ctx, const ov2_xyzba_problem *p, const ov2_ba_options *o, ov2_xyzba_result *r);

/* ------------------------------------------------------------------ */
/* Per-keypoint undistortion + bearing vector                           */
/* ------------------------------------------------------------------ */
/* Frame::computeKeypoint (src/frame.cpp:246-254) for n keypoints in one launch:
 *   unpx = CameraCalibration::undistortImagePoint(px)   (src/camera_calibration.cpp:313-333:
 *          cv::undistortPoints(.., K, D, noArray(), K) for model pinhole, 5 iterations;
 *          cv::fisheye::undistortPoints(.., K, D, Mat(), K) for model fisheye; `return pt` when D is empty)
 *   bv   = normalize(iK * (unpx.x, unpx.y, 1))          (iK = the reference's K_.inverse(), row-major)
 * K = (fx, fy, cx, cy); D / nD = distortion coefficients (pinhole: 4, 5, 8 or 12; fisheye: 4; 0 = none).
 * px / unpx: n x (x,y) float; bv: n x 3 double, may be NULL.  The reference calls this per keypoint from
 * Frame::addKeypoint / updateKeypoint (src/frame.cpp:257-354) and for right-image points (:408).      */
#define OV2_CAM_PINHOLE 0
#define OV2_CAM_FISHEYE 1
int ov2_compute_keypoints(ov2_ctx *ctx, int model, const double K[4], const double *D, int nD, const double iK[9],
                          const float *px_xy_h, int n, float *unpx_xy_h, double *bv_xyz_h);
/* same on device-resident buffers (asynchronous on ctx's stream), e.g. straight on the output of ov2_fb_klt_d */
int ov2_compute_keypoints_d(ov2_ctx *ctx, int model, const double K[4], const double *D, int nD, const double iK[9],
                            const float *px_xy_d, int n, float *unpx_xy_d, double *bv_xyz_d);

/* ------------------------------------------------------------------ */
/* Stereo matching front half (MapManager::stereoMatching,              */
/* src/map_manager.cpp:367-611)                                         */
/* ------------------------------------------------------------------ */
/* FeatureTracker::getLineMinSAD (src/feature_tracker.cpp:138-206) for n keypoints in one launch, as called at
 * src/map_manager.cpp:431 on the coarsest pyramid level of a rectified pair: pts are ALREADY scaled to
 * `level` (kp.px_ * downpyrcoef), nwinsize odd (the reference uses 7), go_left = bgoleft.
 * xprior[i] = best column at that level or -1 (multiply by uppyrcoef like :433); l1err[i] = the minimal mean
 * absolute difference, 255 when nothing qualified (the reference leaves it unset on its early returns).
 * left/right: batch-1 pyramids of the two images (the image of `level` is read, REPLICATE border as
 * cv::getRectSubPix does); keypoints must lie inside that image.                                        */
int ov2_line_min_sad(ov2_ctx *ctx, const ov2_pyr *left, const ov2_pyr *right, int level, int nwinsize, int go_left,
                     const float *pts_xy_h, int n, float *xprior_h, float *l1err_h);
/* Epipolar gate of src/map_manager.cpp:568-590 for n (left keypoint, tracked right keypoint) pairs:
 *   runpx = pcalib_rightcam_->undistortImagePoint(rkps[i])        (model / K / D / nD as in ov2_compute_keypoints)
 *   rect != 0: epi_err = |lunpx.y - runpx.y| and rkps[i].y = lunpx[i].y (written back, :578)
 *   rect == 0: epi_err = MultiViewGeometry::computeSampsonDistance(Frl, lunpx, runpx)  (src/multi_view_geometry.cpp:797-822)
 *   ok[i] = epi_err <= 2.   runpx_xy_h / epi_err_h may be NULL.                                          */
int ov2_stereo_epipolar_check(ov2_ctx *ctx, int rect, const double Frl[9], int model, const double K[4], const double *D, int nD,
                              const float *lunpx_xy_h, float *rkps_xy_inout_h, int n, float *runpx_xy_h, float *epi_err_h, uint8_t *ok_h);

/* MapManager::stereoMatching's data path (src/map_manager.cpp:367-611) for the n keypoints of a keyframe in ONE enqueue and ONE
 * synchronisation (the three calls above need one each, plus a second fbKltTracking):
 *   rect != 0   getLineMinSAD on pyramid level nklt_pyr_lvl (window 7, searching left) gives the x prior of every keypoint
 *               without a 3-D prior when it lies in [0, kp.x] (:421-439)
 *   tracking    keypoints with has_prior3d_h[i] != 0 are tracked from priors3d_h[i] on 1 level first; the ones that fail join
 *               the second call with the first call's forward result as prior (:533-538: v3dpriors was updated in place); everything else runs on nklt_pyr_lvl levels (:544-565)
 *   gate        ov2_stereo_epipolar_check on the tracked right keypoints (model / K / D of the RIGHT camera, kps_unpx_h = the
 *               left keypoints' undistorted pixels)
 * Outputs: stereo_ok_h[i] (what decides updateKeypointStereo, :584) and right_px_h[i] (rect: y replaced by the left keypoint's,
 * :578; (0, 0) where the tracking itself failed -- a tracked point that the gate rejects keeps its position, stereo_ok 0).  has_prior3d_h / priors3d_h may be NULL (no map-point priors).  The right pyramid may
 * still be building on this context's stream (ov2_pyr_build_clahe_h is asynchronous).                                      */
int ov2_stereo_match(ov2_ctx *ctx, const ov2_pyr *left, const ov2_pyr *right, int nklt_win_size, int nklt_pyr_lvl, int max_iter,
                     float eps, float nklt_err, float fmax_fbklt_dist, int rect, const double Frl[9], int model, const double K[4],
                     const double *D, int nD, const float *kps_px_h, const float *kps_unpx_h, const float *priors3d_h,
                     const uint8_t *has_prior3d_h, int n, float *right_px_h, uint8_t *stereo_ok_h);

/* ov2_stereo_match for the keyframes of a lock-step batch (all sequences of a rank reach their keyframes together): items [0, n_items) of
 * two batch pyramids (left: e.g. ov2_btracker_cur_pyr at the keyframe; right: ov2_pyr_build_clahe_hb), n_max point slots per item
 * (item b's points are [b*n_max, b*n_max + n_h[b])), ONE enqueue and ONE synchronisation for all items -- the same kernels with the grid
 * extended by the item.  Per item the outputs equal ov2_stereo_match on that item.                                                */
int ov2_stereo_match_batch(ov2_ctx *ctx, const ov2_pyr *left, const ov2_pyr *right, int n_items, int nklt_win_size, int nklt_pyr_lvl, int max_iter,
                           float eps, float nklt_err, float fmax_fbklt_dist, int rect, const double Frl[9], int model, const double K[4],
                           const double *D, int nD, int n_max, const float *kps_px_h, const float *kps_unpx_h, const float *priors3d_h,
                           const uint8_t *has_prior3d_h, const int *n_h, float *right_px_h, uint8_t *stereo_ok_h);

#ifdef __cplusplus
}
#endif
#endif /* OV2SLAM_HIP_H */
