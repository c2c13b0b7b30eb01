// ctx.hip -- context / error handling of libov2slam_hip.so
#include "common.hpp"
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void ov2_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ov2_ctx::reserve_device(size_t bytes)
{
    if (bytes <= d_scratch_bytes) return OV2_OK;
    if (d_scratch) { OV2_HIP_CHECK(hipStreamSynchronize(stream)); OV2_HIP_CHECK(hipFree(d_scratch)); d_scratch = nullptr; d_scratch_bytes = 0; }
    size_t cap = bytes + bytes / 2 + 4096;
    OV2_HIP_CHECK(hipMalloc(&d_scratch, cap));
    d_scratch_bytes = cap;
    return OV2_OK;
}

int ov2_ctx::reserve_host(size_t bytes)
{
    if (bytes <= h_scratch_bytes) return OV2_OK;
    if (h_scratch) { OV2_HIP_CHECK(hipStreamSynchronize(stream)); OV2_HIP_CHECK(hipHostFree(h_scratch)); h_scratch = nullptr; h_scratch_bytes = 0; }
    size_t cap = bytes + bytes / 2 + 4096;
    OV2_HIP_CHECK(hipHostMalloc(&h_scratch, cap, hipHostMallocCoherent));   // (ba_run polls a flag word the device writes: coherence stated, not left to HIP_HOST_COHERENT)
    h_scratch_bytes = cap;
    return OV2_OK;
}

// Row-pitched upload through pinned memory: hipMemcpy2DAsync from a pageable image whose width is not a multiple of the pitches
// (KITTI: 1241-byte rows into a 1248-byte pitch) took 2.8 ms per frame -- 376 row transfers; staged, it is one host-side repack
// (rows of w bytes, ~25 us) and ONE contiguous DMA.  The staging buffer is refilled only after the previous upload's event.
int ov2_ctx::upload_image(void *dst_d, size_t dst_pitch, const uint8_t *src_h, size_t src_stride, size_t w, size_t h)
{
    const size_t bytes = dst_pitch * h;
    if (img_pending) { OV2_HIP_CHECK(hipEventSynchronize(img_ev)); img_pending = false; }
    if (!img_ev) OV2_HIP_CHECK(hipEventCreateWithFlags(&img_ev, hipEventDisableTiming));
    if (bytes > h_img_bytes) {
        if (h_img) { OV2_HIP_CHECK(hipHostFree(h_img)); h_img = nullptr; h_img_bytes = 0; }
        OV2_HIP_CHECK(hipHostMalloc(&h_img, bytes + bytes / 4 + 4096, hipHostMallocDefault));
        h_img_bytes = bytes + bytes / 4 + 4096;
    }
    uint8_t *st = (uint8_t *)h_img;
    if (src_stride == dst_pitch) memcpy(st, src_h, dst_pitch * (h - 1) + w);
    else for (size_t y = 0; y < h; y++) memcpy(st + y * dst_pitch, src_h + y * src_stride, w);
    OV2_HIP_CHECK(hipMemcpyAsync(dst_d, st, bytes, hipMemcpyHostToDevice, stream));
    OV2_HIP_CHECK(hipEventRecord(img_ev, stream));
    img_pending = true;
    return OV2_OK;
}

int ov2_ctx::upload_images(void *dst_d, size_t dst_pitch, size_t item_bytes, const uint8_t *const *src_h, int n, size_t src_stride, size_t w, size_t h)
{
    const size_t bytes = item_bytes * (size_t)n;
    if (img_pending) { OV2_HIP_CHECK(hipEventSynchronize(img_ev)); img_pending = false; }
    if (!img_ev) OV2_HIP_CHECK(hipEventCreateWithFlags(&img_ev, hipEventDisableTiming));
    if (bytes > h_img_bytes) {
        if (h_img) { OV2_HIP_CHECK(hipHostFree(h_img)); h_img = nullptr; h_img_bytes = 0; }
        OV2_HIP_CHECK(hipHostMalloc(&h_img, bytes + bytes / 4 + 4096, hipHostMallocDefault));
        h_img_bytes = bytes + bytes / 4 + 4096;
    }
    for (int b = 0; b < n; b++) {
        uint8_t *st = (uint8_t *)h_img + item_bytes * (size_t)b;
        if (src_stride == dst_pitch) memcpy(st, src_h[b], dst_pitch * (h - 1) + w);
        else for (size_t y = 0; y < h; y++) memcpy(st + y * dst_pitch, src_h[b] + y * src_stride, w);
    }
    OV2_HIP_CHECK(hipMemcpyAsync(dst_d, h_img, bytes, hipMemcpyHostToDevice, stream));
    OV2_HIP_CHECK(hipEventRecord(img_ev, stream));
    img_pending = true;
    return OV2_OK;
}

int ov2_ctx::download_image(uint8_t *dst_h, size_t dst_stride, const void *src_d, size_t src_pitch, size_t w, size_t h)
{
    if (w == 0 || h == 0) return OV2_OK;
    const size_t bytes = src_pitch * (h - 1) + w;
    if (img_pending) { OV2_HIP_CHECK(hipEventSynchronize(img_ev)); img_pending = false; }
    if (bytes > h_img_bytes) {
        if (h_img) { OV2_HIP_CHECK(hipHostFree(h_img)); h_img = nullptr; h_img_bytes = 0; }
        OV2_HIP_CHECK(hipHostMalloc(&h_img, bytes + bytes / 4 + 4096, hipHostMallocDefault));
        h_img_bytes = bytes + bytes / 4 + 4096;
    }
    OV2_HIP_CHECK(hipMemcpyAsync(h_img, src_d, bytes, hipMemcpyDeviceToHost, stream));
    OV2_HIP_CHECK(hipStreamSynchronize(stream));
    const uint8_t *st = (const uint8_t *)h_img;
    if (src_pitch == dst_stride) memcpy(dst_h, st, bytes);
    else for (size_t y = 0; y < h; y++) memcpy(dst_h + y * dst_stride, st + y * src_pitch, w);
    return OV2_OK;
}

extern "C" {

int ov2_version(void) { return OV2_ABI_VERSION; }

const char *ov2_last_error(void) { return g_err; }

static int ctx_create_common(int device, hipStream_t stream, bool own, ov2_ctx **out, int priority = 0)
{
    OV2_REQUIRE(out != nullptr, OV2_EINVAL, "out == NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        ov2_set_error("no HIP device visible (%s)", e == hipSuccess ? "count == 0" : hipGetErrorString(e));
        return OV2_ENODEVICE;
    }
    OV2_REQUIRE(device >= 0 && device < ndev, OV2_EINVAL, "device index out of range");
    OV2_HIP_CHECK(hipSetDevice(device));
    ov2_ctx *c = new (std::nothrow) ov2_ctx();
    OV2_REQUIRE(c != nullptr, OV2_ENOMEM, "out of host memory");
    c->device = device;
    if (const char *e = getenv("OV2_DEBUG")) c->debug = e[0] == '1';     // the one environment read of the library
    if (own) {
        hipError_t se;
        if (priority == 0) se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        else {
            // HIP numbers priorities downwards: `greatest` (the numerically smallest value) is served first
            int least = 0, greatest = 0;
            se = hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (se == hipSuccess) se = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, priority > 0 ? greatest : least);
        }
        if (se != hipSuccess) { delete c; ov2_set_error("hipStreamCreate: %s", hipGetErrorString(se)); return OV2_EHIP; }
        c->owns_stream = true;
    } else {
        c->stream = stream;
        c->owns_stream = false;
    }
    *out = c;
    return OV2_OK;
}

int ov2_ctx::reserve_stat_slots()
{
    if (stat_slots) return OV2_OK;
    const size_t bytes = (size_t)LK_STAT_SLOTS * LK_STAT_STRIDE * sizeof(unsigned long long);
    OV2_HIP_CHECK(hipMalloc((void **)&stat_slots, bytes));
    OV2_HIP_CHECK(hipMemsetAsync(stat_slots, 0, bytes, stream));
    return OV2_OK;
}

int ov2_ctx_create(int device, ov2_ctx **out) { return ctx_create_common(device, nullptr, true, out); }

int ov2_ctx_create_with_priority(int device, int priority, ov2_ctx **out) { return ctx_create_common(device, nullptr, true, out, priority); }

int ov2_ctx_create_on_stream(int device, void *hip_stream, ov2_ctx **out)
{
    return ctx_create_common(device, (hipStream_t)hip_stream, false, out);
}

void ov2_ctx_destroy(ov2_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->stat_slots) (void)hipFree(ctx->stat_slots);
    if (ctx->ba_det_pool) (void)hipFree(ctx->ba_det_pool);
    if (ctx->ba_trace_d) (void)hipFree(ctx->ba_trace_d);
    free(ctx->ba_trace_h);
    if (ctx->ba_host_pool && ctx->ba_host_pool_free) ctx->ba_host_pool_free(ctx->ba_host_pool);
    for (int i = 0; i < 2; i++) if (ctx->ba_ev[i]) (void)hipEventDestroy(ctx->ba_ev[i]);
    if (ctx->h_scratch) (void)hipHostFree(ctx->h_scratch);
    if (ctx->h_img) (void)hipHostFree(ctx->h_img);
    if (ctx->img_ev) (void)hipEventDestroy(ctx->img_ev);
    if (ctx->det_aux_stream) { (void)hipStreamSynchronize(ctx->det_aux_stream); (void)hipStreamDestroy(ctx->det_aux_stream); }
    for (int i = 0; i < 2; i++) if (ctx->det_ev[i]) (void)hipEventDestroy(ctx->det_ev[i]);
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int ov2_ctx_sync(ov2_ctx *ctx)
{
    OV2_REQUIRE(ctx != nullptr, OV2_EINVAL, "ctx == NULL");
    OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return OV2_OK;
}

int ov2_ctx_set_option(ov2_ctx *ctx, int option, int value)
{
    OV2_REQUIRE(ctx != nullptr, OV2_EINVAL, "ctx == NULL");
    switch (option) {
    case OV2_OPT_SOBEL_DY_ORDER:
        OV2_REQUIRE(value == OV2_SOBEL_DY_OPENCV_ROWFILTER || value == OV2_SOBEL_DY_EXACT_SUM, OV2_EINVAL, "unknown Sobel dy order");
        ctx->sobel_dy_order = value;
        return OV2_OK;
    case OV2_OPT_LK_IMPL:
        OV2_REQUIRE(value >= OV2_LK_IMPL_AUTO && value <= OV2_LK_IMPL_LANE3, OV2_EINVAL, "unknown LK kernel choice");
        ctx->lk_impl = value; return OV2_OK;
    case OV2_OPT_TRACK_IMPL:
        OV2_REQUIRE(value == OV2_TRACK_IMPL_WAVE || value == OV2_TRACK_IMPL_ROW, OV2_EINVAL, "unknown tracker kernel choice");
        ctx->track_impl = value; return OV2_OK;
    case OV2_OPT_CLAHE_STRIPS:
        OV2_REQUIRE(value >= -1 && value <= 2, OV2_EINVAL, "OV2_OPT_CLAHE_STRIPS takes -1, 0, 1 or 2");
        ctx->clahe_strips = value; return OV2_OK;
    case OV2_OPT_BA_FORCE_LARGE:     ctx->ba_force_large = value != 0; return OV2_OK;
    case OV2_OPT_BA_LIN_DIRECT:      ctx->ba_lin_direct = value != 0; return OV2_OK;
    case OV2_OPT_BA_SCHUR_CHUNK:
        OV2_REQUIRE(value == 0 || value >= 6, OV2_EINVAL, "OV2_OPT_BA_SCHUR_CHUNK takes 0 or >= 6 columns");
        ctx->ba_schur_chunk = value; return OV2_OK;
    case OV2_OPT_BA_XYZ_LIN_WAVES:
        OV2_REQUIRE(value >= 0 && value <= 2, OV2_EINVAL, "OV2_OPT_BA_XYZ_LIN_WAVES takes 0, 1 or 2");
        ctx->ba_xyz_lin_waves = value; return OV2_OK;
    case OV2_OPT_BA_POSE_ONLY_FUSED: ctx->ba_pose_only_fused = value != 0; return OV2_OK;
    case OV2_OPT_BA_DETERMINISTIC:   ctx->ba_deterministic = value != 0; return OV2_OK;
    case OV2_OPT_BA_TRACE:           ctx->ba_trace = value != 0; return OV2_OK;
    case OV2_OPT_DETECT_STRIP:
        OV2_REQUIRE(value >= -1 && value <= 1, OV2_EINVAL, "OV2_OPT_DETECT_STRIP takes -1, 0 or 1");
        ctx->det_strip = value; return OV2_OK;
    case OV2_OPT_LK_ACC:
        OV2_REQUIRE(value == OV2_LK_ACC_INT64 || value == OV2_LK_ACC_FLOAT_UI4, OV2_EINVAL, "OV2_OPT_LK_ACC takes OV2_LK_ACC_INT64 or OV2_LK_ACC_FLOAT_UI4");
        ctx->lk_acc = value; return OV2_OK;
    case OV2_OPT_DEBUG:              ctx->debug = value != 0; return OV2_OK;
    case OV2_OPT_FAST_TIE:
        OV2_REQUIRE(value == OV2_FAST_TIE_SCAN_ORDER || value == OV2_FAST_TIE_LIBSTDCXX, OV2_EINVAL, "OV2_OPT_FAST_TIE takes 0 or 1");
        ctx->det_fast_tie = value; return OV2_OK;
    default:
        ov2_set_error("unknown context option %d", option);
        return OV2_EINVAL;
    }
}

int ov2_ctx_get_option(ov2_ctx *ctx, int option, int *value)
{
    OV2_REQUIRE(ctx != nullptr && value != nullptr, OV2_EINVAL, "NULL argument");
    switch (option) {
    case OV2_OPT_SOBEL_DY_ORDER:     *value = ctx->sobel_dy_order; return OV2_OK;
    case OV2_OPT_LK_IMPL:            *value = ctx->lk_impl; return OV2_OK;
    case OV2_OPT_TRACK_IMPL:         *value = ctx->track_impl; return OV2_OK;
    case OV2_OPT_CLAHE_STRIPS:       *value = ctx->clahe_strips; return OV2_OK;
    case OV2_OPT_BA_FORCE_LARGE:     *value = ctx->ba_force_large; return OV2_OK;
    case OV2_OPT_BA_LIN_DIRECT:      *value = ctx->ba_lin_direct; return OV2_OK;
    case OV2_OPT_BA_SCHUR_CHUNK:     *value = ctx->ba_schur_chunk; return OV2_OK;
    case OV2_OPT_BA_XYZ_LIN_WAVES:   *value = ctx->ba_xyz_lin_waves; return OV2_OK;
    case OV2_OPT_BA_POSE_ONLY_FUSED: *value = ctx->ba_pose_only_fused; return OV2_OK;
    case OV2_OPT_BA_DETERMINISTIC:   *value = ctx->ba_deterministic; return OV2_OK;
    case OV2_OPT_BA_TRACE:           *value = ctx->ba_trace; return OV2_OK;
    case OV2_OPT_DETECT_STRIP:       *value = ctx->det_strip; return OV2_OK;
    case OV2_OPT_LK_ACC:             *value = ctx->lk_acc; return OV2_OK;
    case OV2_OPT_DEBUG:              *value = ctx->debug; return OV2_OK;
    case OV2_OPT_FAST_TIE:           *value = ctx->det_fast_tie; return OV2_OK;
    default:
        ov2_set_error("unknown context option %d", option);
        return OV2_EINVAL;
    }
}

void *ov2_ctx_stream(ov2_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

} // extern "C"
