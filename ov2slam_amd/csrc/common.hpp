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
    // Tiled copy for the batch LK kernel (k_fb_klt3): 16 x 8-pixel tiles of 128 bytes = one cache line each, covering the
    // image plus 16 columns / rows of padding on the left / top (REFLECT_101 border inside it) and the slack LK's block
    // fetches can touch on the right / bottom.  A 16-row block of 16-byte row segments then spans 4-6 lines instead of
    // 12-18 (the kernel is bound by line fills).  til_base < 0: this pyramid has no tiled copy.
    long long til_base;   // byte offset (from the item base) of tile (0, 0)
    int til_ntx, til_nty; // tiles per row / tile rows
};
#define OV2_TIL_PAD 16        // pixel (x, y) lives at tile column (x + 16) >> 4, tile row (y + 16) >> 3
__host__ __device__ __forceinline__ long long ov2_til_offset(const PyrLevelDesc &L, int x, int y)
{
    const int X = x + OV2_TIL_PAD, Y = y + OV2_TIL_PAD;
    return L.til_base + ((long long)((Y >> 3) * L.til_ntx + (X >> 4)) << 7) + ((Y & 7) << 4) + (X & 15);
}

struct PyrDesc {
    uint8_t *base;            // device pointer of batch item 0
    long long item_stride;    // bytes between batch items
    int n_levels;
    int win;
    int batch;
    int tiled;                // 1: every level also has its tiled copy (batch pyramids)
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
};
int ov2_pyr_mark_ready(ov2_ctx *ctx, ov2_pyr *p);              // after the last kernel of a build was enqueued
int ov2_pyr_wait_ready(ov2_ctx *ctx, const ov2_pyr *p);        // before the first kernel of a consumer

// kernels' host launchers (defined in the .hip files)
// from_level = 1: levels 0 and 1 (borders included) are already in place (k_clahe_apply_pyr)
int ov2_launch_pyr_build(ov2_ctx *ctx, ov2_pyr *p, const uint8_t *img_d, int stride, size_t img_batch_stride, int from_level = 0);
// cv::CLAHE::apply on `batch` device images; border > 0: dst is a padded pyramid level, its REFLECT_101 border is written too
int ov2_launch_clahe(ov2_ctx *ctx, const uint8_t *src_d, int w, int h, int stride, size_t src_batch_stride, int batch,
                     double clip_limit, int tiles_x, int tiles_y, uint8_t *dst_d, int dst_stride, size_t dst_batch_stride,
                     uint8_t *lut_d, int border, long long til_delta = 0, int til_ntx = 0, const struct PyrDesc *pyr = nullptr,
                     int *level1_done = nullptr);
// fused VisualFrontEnd::kltTracking launch (lk.hip), device pointers only
int ov2_launch_track_klt(hipStream_t s, const ov2_pyr *prev, const ov2_pyr *cur, int win, int lvl_prior, int lvl_full,
                         int max_iter, float eps, float err_th, float fb_dist, int n_max, const int *n_dev,
                         const float *kps, const float *priors, const uint8_t *flags, float *out_xy, uint8_t *status, int *iters,
                         const float *sad_x = nullptr, float sad_up = 0.f);      // sad_x: stereo mode (lk.hip: k_track_klt)
