// pyramid.hip -- device-resident optical-flow pyramid for gfx950.
//
// Replaces cv::buildOpticalFlowPyramid(img, pyr, Size(win,win), maxLevel,
// withDerivatives=true, BORDER_REFLECT_101, BORDER_CONSTANT) as called at
// /root/reference/src/visual_front_end.cpp:1172, :53 and src/mapper.cpp:81.
// Arithmetic (all integer, bit-exact vs the oracle):
//   level l>0 : 5x5 [1 4 6 4 1]^2 pyrDown, (sum + 128) >> 8, REFLECT_101
//   derivative: Scharr 3/10/3, int16 (dx,dy) interleaved, REFLECT_101 on the image
// Layout: see PyrDesc in common.hpp.  Every level image carries a REFLECT_101
// border of >= win pixels so that (a) LK windows that hang over the image edge
// read legal memory exactly like OpenCV's padded Mats and (b) the 5x5 / 3x3
// stencils of the next kernel need no border logic when reading.
#include "common.hpp"

#pragma clang fp contract(off)

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// ---- level 0: copy + REFLECT_101 border ---------------------------------------
__global__ __launch_bounds__(256) void k_pyr_level0(PyrDesc P, const uint8_t *__restrict__ src, int stride, long long src_item_stride)
{
    const PyrLevelDesc L = P.lv[0];
    const int b = blockIdx.z;
    const int pw = L.w + 2 * P.win, ph = L.h + 2 * P.win;
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    const int py = blockIdx.y;
    if (px >= pw || py >= ph) return;
    const int x = px - P.win, y = py - P.win;
    const int sx = reflect101(x, L.w), sy = reflect101(y, L.h);
    uint8_t *dst = P.base + (long long)b * P.item_stride + L.img_roi;
    dst[(long long)y * L.img_pitch + x] = src[(long long)b * src_item_stride + (long long)sy * stride + sx];
}

// ---- level l>0: pyrDown of level l-1 (+ border) --------------------------------
// One thread per padded output pixel; the border pixels recompute the stencil at
// the reflected coordinate (no dependency on other threads).  The source level's
// own REFLECT_101 border supplies the 2-pixel stencil overhang.
__global__ __launch_bounds__(256) void k_pyr_down(PyrDesc P, int level)
{
    const PyrLevelDesc S = P.lv[level - 1];
    const PyrLevelDesc L = P.lv[level];
    const int b = blockIdx.z;
    const int pw = L.w + 2 * P.win, ph = L.h + 2 * P.win;
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    const int py = blockIdx.y;
    if (px >= pw || py >= ph) return;
    const int x = px - P.win, y = py - P.win;
    const int ox = reflect101(x, L.w), oy = reflect101(y, L.h);
    const uint8_t *s = P.base + (long long)b * P.item_stride + S.img_roi;
    int acc = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int wy = (k == 0 || k == 4) ? 1 : (k == 2 ? 6 : 4);
        const uint8_t *r = s + (long long)(2 * oy - 2 + k) * S.img_pitch + (2 * ox - 2);
        const int row = r[2] * 6 + (r[1] + r[3]) * 4 + r[0] + r[4];
        acc += wy * row;
    }
    uint8_t *dst = P.base + (long long)b * P.item_stride + L.img_roi;
    dst[(long long)y * L.img_pitch + x] = (uint8_t)((acc + 128) >> 8);
}

// ---- Scharr derivative of one level ---------------------------------------------
__global__ __launch_bounds__(256) void k_scharr(PyrDesc P, int level)
{
    const PyrLevelDesc L = P.lv[level];
    const int b = blockIdx.z;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= L.w || y >= L.h) return;
    const uint8_t *s = P.base + (long long)b * P.item_stride + L.img_roi + (long long)y * L.img_pitch + x;
    const int p = L.img_pitch;
    const int a00 = s[-p - 1], a01 = s[-p], a02 = s[-p + 1];
    const int a10 = s[-1],                 a12 = s[1];
    const int a20 = s[p - 1],  a21 = s[p],  a22 = s[p + 1];
    // t0 = 3*(up+down) + 10*mid per column ; t1 = down - up per column
    const int t0l = (a00 + a20) * 3 + a10 * 10, t0r = (a02 + a22) * 3 + a12 * 10;
    const int t1l = a20 - a00, t1c = a21 - a01, t1r = a22 - a02;
    const int dx = t0r - t0l;
    const int dy = (t1l + t1r) * 3 + t1c * 10;
    uint32_t *d = (uint32_t *)(P.base + (long long)b * P.item_stride + L.der_roi) + (long long)y * L.der_pitch + x;
    *d = ((uint32_t)(uint16_t)(int16_t)dx) | (((uint32_t)(uint16_t)(int16_t)dy) << 16);
}

int ov2_launch_pyr_build(ov2_ctx *ctx, ov2_pyr *p, const uint8_t *img_d, int stride, size_t img_batch_stride)
{
    const PyrDesc &P = p->d;
    const int win = P.win;
    {
        const PyrLevelDesc &L = P.lv[0];
        dim3 grid((L.w + 2 * win + 255) / 256, L.h + 2 * win, P.batch);
        hipLaunchKernelGGL(k_pyr_level0, grid, dim3(256), 0, ctx->stream, P, img_d, stride, (long long)img_batch_stride);
    }
    for (int l = 0; l < P.n_levels; l++) {
        const PyrLevelDesc &L = P.lv[l];
        if (l > 0) {
            dim3 grid((L.w + 2 * win + 255) / 256, L.h + 2 * win, P.batch);
            hipLaunchKernelGGL(k_pyr_down, grid, dim3(256), 0, ctx->stream, P, l);
        }
        dim3 grid((L.w + 255) / 256, L.h, P.batch);
        hipLaunchKernelGGL(k_scharr, grid, dim3(256), 0, ctx->stream, P, l);
    }
    OV2_HIP_CHECK(hipGetLastError());
    return OV2_OK;
}

// ---- C ABI ----------------------------------------------------------------------
static inline long long round_up(long long v, long long a) { return (v + a - 1) / a * a; }

extern "C" {

int ov2_pyr_create(ov2_ctx *ctx, int w, int h, int win, int max_level, int batch, ov2_pyr **out)
{
    OV2_REQUIRE(ctx && out, OV2_EINVAL, "ctx/out == NULL");
    *out = nullptr;
    OV2_REQUIRE(w > 0 && h > 0 && win > 2 && win <= 31 && max_level >= 0 && max_level < OV2_MAX_LEVELS && batch >= 1,
                OV2_EINVAL, "bad pyramid geometry");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    ov2_pyr *p = new (std::nothrow) ov2_pyr();
    OV2_REQUIRE(p != nullptr, OV2_ENOMEM, "out of host memory");
    p->w = w; p->h = h; p->max_level = max_level; p->device = ctx->device;
    PyrDesc &D = p->d;
    memset(&D, 0, sizeof(D));
    D.win = win; D.batch = batch;
    long long off = 0;
    int lw = w, lh = h;
    for (int l = 0; l <= max_level; l++) {
        PyrLevelDesc &L = D.lv[l];
        L.w = lw; L.h = lh;
        L.pady = win;
        L.img_padx = (int)round_up(win + 3, 16);           // aligned-dword row reads may start 3 B early
        L.img_pitch = (int)round_up(L.img_padx + lw + win + 8, 64);
        L.der_padx = 16;
        if (L.der_padx < win) L.der_padx = (int)round_up(win, 16);
        L.der_pitch = (int)round_up(L.der_padx + lw + win, 16);
        const long long img_bytes = (long long)(lh + 2 * win) * L.img_pitch;
        const long long der_bytes = (long long)(lh + 2 * win) * L.der_pitch * 4;
        off = round_up(off, 256);
        L.img_roi = off + (long long)L.pady * L.img_pitch + L.img_padx;
        off += img_bytes;
        off = round_up(off, 256);
        L.der_roi = off + ((long long)L.pady * L.der_pitch + L.der_padx) * 4;
        off += der_bytes;
        D.n_levels = l + 1;
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
        if (lw <= win || lh <= win) break;     // buildOpticalFlowPyramid stops early
    }
    D.item_stride = round_up(off, 4096);
    p->bytes = (size_t)D.item_stride * (size_t)batch;
    hipError_t e = hipMalloc((void **)&D.base, p->bytes);
    if (e != hipSuccess) { delete p; ov2_set_error("hipMalloc(%zu): %s", p->bytes, hipGetErrorString(e)); return OV2_ENOMEM; }
    // derivative borders are BORDER_CONSTANT(0) and never written again
    e = hipMemsetAsync(D.base, 0, p->bytes, ctx->stream);
    if (e != hipSuccess) { (void)hipFree(D.base); delete p; ov2_set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return OV2_EHIP; }
    *out = p;
    return OV2_OK;
}

void ov2_pyr_destroy(ov2_pyr *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->d.base) (void)hipFree(p->d.base);
    delete p;
}

int ov2_pyr_levels(const ov2_pyr *p) { return p ? p->d.n_levels : 0; }
int ov2_pyr_batch(const ov2_pyr *p) { return p ? p->d.batch : 0; }

int ov2_pyr_level_size(const ov2_pyr *p, int level, int *w, int *h)
{
    OV2_REQUIRE(p && w && h && level >= 0 && level < p->d.n_levels, OV2_EINVAL, "bad level");
    *w = p->d.lv[level].w; *h = p->d.lv[level].h;
    return OV2_OK;
}

int ov2_pyr_build_d(ov2_ctx *ctx, ov2_pyr *p, const uint8_t *img_d, int stride, size_t img_batch_stride)
{
    OV2_REQUIRE(ctx && p && img_d, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(stride >= p->w, OV2_EINVAL, "stride < width");
    OV2_REQUIRE(p->d.batch == 1 || img_batch_stride >= (size_t)stride * (size_t)p->h, OV2_EINVAL, "img_batch_stride too small");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    return ov2_launch_pyr_build(ctx, p, img_d, stride, img_batch_stride);
}

int ov2_pyr_build_h(ov2_ctx *ctx, ov2_pyr *p, const uint8_t *img_h, int stride, size_t img_batch_stride)
{
    OV2_REQUIRE(ctx && p && img_h, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(stride >= p->w, OV2_EINVAL, "stride < width");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t item = (size_t)p->w * (size_t)p->h;
    const int rc = ctx->reserve_device(item * (size_t)p->d.batch);
    if (rc != OV2_OK) return rc;
    if (p->d.batch > 1) OV2_REQUIRE(img_batch_stride >= (size_t)stride * (size_t)p->h, OV2_EINVAL, "img_batch_stride too small");
    for (int b = 0; b < p->d.batch; b++) {
        OV2_HIP_CHECK(hipMemcpy2DAsync((uint8_t *)ctx->d_scratch + item * b, (size_t)p->w,
                                       img_h + img_batch_stride * b, (size_t)stride, (size_t)p->w, (size_t)p->h,
                                       hipMemcpyHostToDevice, ctx->stream));
    }
    return ov2_launch_pyr_build(ctx, p, (const uint8_t *)ctx->d_scratch, p->w, item);
}

static int pyr_download_impl(ov2_ctx *ctx, const ov2_pyr *p, int b, int level, uint8_t *img_h, int16_t *deriv_h, int padded)
{
    OV2_REQUIRE(ctx && p, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(level >= 0 && level < p->d.n_levels && b >= 0 && b < p->d.batch, OV2_EINVAL, "bad level/batch index");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    const PyrLevelDesc &L = p->d.lv[level];
    const int pad = padded ? p->d.win : 0;
    const int ow = L.w + 2 * pad, oh = L.h + 2 * pad;
    const uint8_t *item = p->d.base + (long long)b * p->d.item_stride;
    if (img_h) {
        const uint8_t *src = item + L.img_roi - (long long)pad * L.img_pitch - pad;
        OV2_HIP_CHECK(hipMemcpy2DAsync(img_h, (size_t)ow, src, (size_t)L.img_pitch, (size_t)ow, (size_t)oh,
                                       hipMemcpyDeviceToHost, ctx->stream));
    }
    if (deriv_h) {
        const uint8_t *src = item + L.der_roi - ((long long)pad * L.der_pitch + pad) * 4;
        OV2_HIP_CHECK(hipMemcpy2DAsync(deriv_h, (size_t)ow * 4, src, (size_t)L.der_pitch * 4, (size_t)ow * 4, (size_t)oh,
                                       hipMemcpyDeviceToHost, ctx->stream));
    }
    OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return OV2_OK;
}

int ov2_pyr_download(ov2_ctx *ctx, const ov2_pyr *p, int b, int level, uint8_t *img_h, int16_t *deriv_h)
{
    return pyr_download_impl(ctx, p, b, level, img_h, deriv_h, 0);
}

int ov2_pyr_download_padded(ov2_ctx *ctx, const ov2_pyr *p, int b, int level, uint8_t *img_h, int16_t *deriv_h)
{
    return pyr_download_impl(ctx, p, b, level, img_h, deriv_h, 1);
}

size_t ov2_pyr_algorithmic_bytes(const ov2_pyr *p)
{
    if (!p) return 0;
    // read level 0 once, write levels >= 1 (u8), write every derivative (int16x2)
    size_t bytes = (size_t)p->d.lv[0].w * p->d.lv[0].h;
    for (int l = 0; l < p->d.n_levels; l++) {
        const size_t px = (size_t)p->d.lv[l].w * p->d.lv[l].h;
        if (l > 0) bytes += px;
        bytes += 4 * px;
    }
    return bytes;
}

} // extern "C"
