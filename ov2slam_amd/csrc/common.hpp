// common.hpp -- shared declarations of libov2slam_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/ov2slam_hip.h"

#define OV2_MAX_LEVELS 8

// ---- error plumbing (never throws across the C ABI) ------------------------
void ov2_set_error(const char *fmt, ...);

#define OV2_HIP_CHECK(expr)                                                        \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            ov2_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,            \
                          hipGetErrorString(_e));                                  \
            return OV2_EHIP;                                                       \
        }                                                                          \
    } while (0)

#define OV2_REQUIRE(cond, code, msg)                                               \
    do {                                                                           \
        if (!(cond)) {                                                             \
            ov2_set_error("%s:%d: %s", __FILE__, __LINE__, msg);                   \
            return (code);                                                         \
        }                                                                          \
    } while (0)

// ---- context ---------------------------------------------------------------
struct ov2_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    // grow-only scratch (device + pinned host), reused across calls of this ctx
    void *d_scratch = nullptr;  size_t d_scratch_bytes = 0;
    void *h_scratch = nullptr;  size_t h_scratch_bytes = 0;
    // LK statistics: work-groups add their (iterations, patch builds) into one of LK_STAT_SLOTS cache lines
    // (same-address device atomics retire at ~6 ns each: 2 per work-group on one address would put a
    // 200 us floor under a 16k-work-group launch); a one-block kernel folds the lines into the caller's pair.
    unsigned long long *stat_slots = nullptr;
    int sobel_dy_order = OV2_SOBEL_DY_OPENCV_ROWFILTER;   // ov2_ctx_set_option(OV2_OPT_SOBEL_DY_ORDER)
    // Path selection / test forcing (ov2_ctx_set_option, OV2_OPT_*).  The library reads NO environment variable after
    // ov2_ctx_create: the entry points are called from several threads of a host that may setenv() concurrently.
    int lk_impl = OV2_LK_IMPL_AUTO;            // OV2_OPT_LK_IMPL
    int track_impl = OV2_TRACK_IMPL_WAVE;      // OV2_OPT_TRACK_IMPL
    int lk_acc = OV2_LK_ACC_INT64;             // OV2_OPT_LK_ACC
    int clahe_strips = -1;                     // OV2_OPT_CLAHE_STRIPS: -1 auto, 0 never, 1 whenever the geometry allows
    int ba_force_large = 0;                    // OV2_OPT_BA_FORCE_LARGE
    int ba_lin_direct = 0;                     // OV2_OPT_BA_LIN_DIRECT
    int ba_schur_chunk = 0;                    // OV2_OPT_BA_SCHUR_CHUNK (columns; 0 = auto)
    int ba_xyz_lin_waves = 0;                  // OV2_OPT_BA_XYZ_LIN_WAVES (0 = auto, 1, 2)
    int ba_pose_only_fused = 1;                // OV2_OPT_BA_POSE_ONLY_FUSED
    int ba_deterministic = 0;                  // OV2_OPT_BA_DETERMINISTIC
    int ba_trace = 0;                          // OV2_OPT_BA_TRACE: the one-problem solves record their iteration summaries (ov2_ba_get_trace)
    void *ba_trace_d = nullptr, *ba_trace_h = nullptr; int ba_trace_n = 0;
    int det_fast_tie = 1;                      // OV2_OPT_FAST_TIE: OV2_FAST_TIE_LIBSTDCXX (the reference as built with g++)
    hipStream_t det_aux_stream = nullptr; hipEvent_t det_ev[2] = {nullptr, nullptr};   // batched detectors: the passes alternate between the context's stream and this one
    int det_strip = -1;                        // OV2_OPT_DETECT_STRIP: -1 auto (batches), 0 one wavefront per cell, 1 the strip kernel
    void *ba_det_pool = nullptr; size_t ba_det_bytes = 0;   // OV2_OPT_BA_DETERMINISTIC: per-work-group copies of H / F^T b / G (grow-only)
    // ov2_local_ba_batch: persistent host threads that prepare the problems of a batch (created with the first batch; ba.hip owns the type)
    void *ba_host_pool = nullptr; void (*ba_host_pool_free)(void *) = nullptr;
    hipEvent_t ba_ev[2] = {nullptr, nullptr};              // the two timing events of a solve (created with the first one: a pair per pass was 25 us)
    int debug = 0;                             // OV2_OPT_DEBUG; initial value: environment OV2_DEBUG, read once by ov2_ctx_create
    // pinned staging of host images on their way to the device: its own buffer (h_scratch is rewritten by the next call's small
    // arrays while an asynchronous image upload may still be in flight) and an event that says when it may be refilled
    void *h_img = nullptr;  size_t h_img_bytes = 0;
    hipEvent_t img_ev = nullptr;  bool img_pending = false;
    int reserve_device(size_t bytes);
    int reserve_host(size_t bytes);
    int reserve_stat_slots();
    // host image (any row stride, pageable or pinned) -> device buffer with pitch dst_pitch, asynchronous on the stream; the
    // caller's buffer is free again when this returns
    int upload_image(void *dst_d, size_t dst_pitch, const uint8_t *src_h, size_t src_stride, size_t w, size_t h);
    // n images (one pointer each) into device slots `item_bytes` apart: one repack into the pinned staging buffer, ONE DMA
    int upload_images(void *dst_d, size_t dst_pitch, size_t item_bytes, const uint8_t *const *src_h, int n, size_t src_stride, size_t w, size_t h);
    // the reverse, synchronous: h rows of w bytes at device pitch src_pitch -> host rows at dst_stride (one contiguous DMA into the
    // pinned staging buffer, then a host-side repack)
    int download_image(uint8_t *dst_h, size_t dst_stride, const void *src_d, size_t src_pitch, size_t w, size_t h);
};
#define LK_STAT_SLOTS 256
#define LK_STAT_STRIDE 16      // unsigned long long per slot (128 B)

// ---- pyramid layout in HBM ---------------------------------------------------
// One allocation per ov2_pyr holding `batch` items of identical layout.  Per
// level: a padded u8 image (REFLECT_101 border, ROI origin 16-byte aligned) and a
// padded int16x2 Scharr derivative (zero border, ROI origin 64-byte aligned).
struct PyrLevelDesc {
    int w, h;
    int img_pitch;        // bytes per padded image row (multiple of 64)
    int der_pitch;        // int16x2 elements (4 B) per padded derivative row (multiple of 16)
    long long img_roi;    // byte offset (from the item base) of image ROI pixel (0,0)
    long long der_roi;    // byte offset (from the item base) of derivative ROI element (0,0)
    int img_padx, der_padx, pady;
};

struct PyrDesc {
    uint8_t *base;            // device pointer of batch item 0
    long long item_stride;    // bytes between batch items
    int n_levels;
    int win;
    int batch;
    PyrLevelDesc lv[OV2_MAX_LEVELS];
};

struct ov2_pyr {
    PyrDesc d;
    int w = 0, h = 0, max_level = 0;
    size_t bytes = 0;
    int device = 0;
    // Cross-context hand-off (the front-end's pyramid is read by the mapper thread's context):
    // every (re)build records `ready` on the producing stream; a consumer entry point running on another stream
    // waits on it before its first kernel (ov2_pyr_wait_ready).  Same-stream use costs nothing.
    hipEvent_t ready = nullptr;
    hipStream_t producer = nullptr;
    bool built = false;
    // ov2_pyr_item_view: a batch-1 alias of one item of `parent` (owns neither the memory nor the event; the hand-off state is the parent's)
    const ov2_pyr *parent = nullptr;
};
int ov2_pyr_mark_ready(ov2_ctx *ctx, ov2_pyr *p);              // after the last kernel of a build was enqueued
int ov2_pyr_wait_ready(ov2_ctx *ctx, const ov2_pyr *p);        // before the first kernel of a consumer

// kernels' host launchers (defined in the .hip files)
// from_level = 1: levels 0 and 1 (borders included) are already in place (k_clahe_apply_pyr)
int ov2_launch_pyr_build(ov2_ctx *ctx, ov2_pyr *p, const uint8_t *img_d, int stride, size_t img_batch_stride, int from_level = 0);
// cv::CLAHE::apply on `batch` device images; border > 0: dst is a padded pyramid level, its REFLECT_101 border is written too
int ov2_launch_clahe(ov2_ctx *ctx, const uint8_t *src_d, int w, int h, int stride, size_t src_batch_stride, int batch,
                     double clip_limit, int tiles_x, int tiles_y, uint8_t *dst_d, int dst_stride, size_t dst_batch_stride,
                     uint8_t *lut_d, int border, const struct PyrDesc *pyr = nullptr, int *level1_done = nullptr);
// fused VisualFrontEnd::kltTracking launch (lk.hip), device pointers only
int ov2_launch_track_klt(hipStream_t s, const ov2_pyr *prev, const ov2_pyr *cur, int win, int lvl_prior, int lvl_full,
                         int max_iter, float eps, float err_th, float fb_dist, int n_max, const int *n_dev,
                         const float *kps, const float *priors, const uint8_t *flags, float *out_xy, uint8_t *status, int *iters,
                         const float *sad_x = nullptr, float sad_up = 0.f,       // sad_x: stereo mode (lk.hip: k_track_klt)
                         int track_impl = OV2_TRACK_IMPL_WAVE,
                         int items = 1,    // items > 1 (trackb.hip): batch items [0, items) of both pyramids, n_max point slots and one n_dev entry per item
                         int lk_acc = OV2_LK_ACC_INT64);   // OV2_OPT_LK_ACC   // items > 1 (trackb.hip): batch items [0, items) of both pyramids, n_max point slots and one n_dev entry per item
