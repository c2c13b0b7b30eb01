// clahe.hip -- cv::CLAHE::apply (CV_8UC1) for gfx950.
//
// Replaces ptracker_->pclahe_->apply(img_raw, cur_img_) at /root/reference/src/visual_front_end.cpp:1159
// and src/mapper.cpp:76 (handle created at src/ov2slam.cpp:85-89: clip = fclahe_val, tiles = (w/50, h/50)).
// Two kernels, integer histogram work + fp32 interpolation without FMA contraction (bit-exact vs oracle):
//   k_clahe_lut   : one workgroup per (tile, image): 256-bin histogram in LDS (ds_add), clip,
//                   redistribution, inclusive scan, LUT = saturate(cvRound(cdf * 255 / tile_area))
//   k_clahe_apply : 4 pixels per thread (dword load/store), bilinear blend of the four tile LUTs
#include "common.hpp"
#include <math.h>

#pragma clang fp contract(off)

__device__ __forceinline__ int c_reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

struct ClaheParams {
    int w, h, stride, tiles_x, tiles_y, tw, th, clip;
    float lut_scale, inv_tw, inv_th;
    long long src_item_stride, dst_item_stride;
    int dst_stride;
};

__global__ __launch_bounds__(256) void k_clahe_lut(ClaheParams P, const uint8_t *__restrict__ src, uint8_t *__restrict__ lut)
{
    __shared__ int hist[256];
    __shared__ int scan[256];
    __shared__ int s_clipped;
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int ty = t / P.tiles_x, tx = t - ty * P.tiles_x;
    const uint8_t *img = src + (long long)b * P.src_item_stride;
    hist[tid] = 0;
    if (tid == 0) s_clipped = 0;
    __syncthreads();
    const int npx = P.tw * P.th;
    for (int e = tid; e < npx; e += 256) {
        const int ly = e / P.tw, lx = e - ly * P.tw;
        const int y = c_reflect101(ty * P.th + ly, P.h), x = c_reflect101(tx * P.tw + lx, P.w);   // right/bottom REFLECT_101 padding
        atomicAdd(&hist[img[(long long)y * P.stride + x]], 1);
    }
    __syncthreads();
    int hv = hist[tid];
    if (P.clip > 0) {
        if (hv > P.clip) { atomicAdd(&s_clipped, hv - P.clip); hv = P.clip; }
        __syncthreads();
        const int clipped = s_clipped;
        const int batch = clipped / 256, residual = clipped - batch * 256;
        hv += batch;
        if (residual != 0) {
            int step = 256 / residual; if (step < 1) step = 1;
            // serial loop `for (i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++`
            if (tid % step == 0 && tid / step < residual) hv++;
        }
    }
    // inclusive scan over the 256 bins (Hillis-Steele in LDS)
    scan[tid] = hv;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = tid >= off ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    int r = __float2int_rn((float)scan[tid] * P.lut_scale);            // saturate_cast<uchar>(float)
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    lut[((long long)b * P.tiles_x * P.tiles_y + t) * 256 + tid] = (uint8_t)r;
}

__global__ __launch_bounds__(256) void k_clahe_apply(ClaheParams P, const uint8_t *__restrict__ src, const uint8_t *__restrict__ lut,
                                                     uint8_t *__restrict__ dst)
{
    const int b = blockIdx.z;
    const int y = blockIdx.y;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= P.w) return;
    const uint8_t *srow = src + (long long)b * P.src_item_stride + (long long)y * P.stride;
    uint8_t *drow = dst + (long long)b * P.dst_item_stride + (long long)y * P.dst_stride;
    const uint8_t *L = lut + (long long)b * P.tiles_x * P.tiles_y * 256;
    const float tyf = (float)y * P.inv_th - 0.5f;
    int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
    const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
    ty1 = ty1 < 0 ? 0 : ty1; ty2 = ty2 > P.tiles_y - 1 ? P.tiles_y - 1 : ty2;
    const uint8_t *L1 = L + (long long)ty1 * P.tiles_x * 256, *L2 = L + (long long)ty2 * P.tiles_x * 256;
    const bool vec = x0 + 3 < P.w && ((((size_t)srow | (size_t)drow) + x0) & 3) == 0 && (((size_t)srow | (size_t)drow) & 3) == 0;
    uint32_t in = 0, out = 0;
    if (vec) in = *(const uint32_t *)(srow + x0);
    for (int j = 0; j < 4; j++) {
        const int x = x0 + j;
        if (x >= P.w) break;
        const int v = vec ? (int)((in >> (8 * j)) & 0xFF) : (int)srow[x];
        const float txf = (float)x * P.inv_tw - 0.5f;
        int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
        const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
        tx1 = tx1 < 0 ? 0 : tx1; tx2 = tx2 > P.tiles_x - 1 ? P.tiles_x - 1 : tx2;
        const float l11 = (float)L1[tx1 * 256 + v], l12 = (float)L1[tx2 * 256 + v];
        const float l21 = (float)L2[tx1 * 256 + v], l22 = (float)L2[tx2 * 256 + v];
        const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
        int r = __float2int_rn(res);
        r = r < 0 ? 0 : (r > 255 ? 255 : r);
        if (vec) out |= (uint32_t)r << (8 * j); else drow[x] = (uint8_t)r;
    }
    if (vec) *(uint32_t *)(drow + x0) = out;
}

static int clahe_launch(ov2_ctx *ctx, const uint8_t *src_d, int w, int h, int stride, size_t src_batch_stride, int batch,
                        double clip_limit, int tiles_x, int tiles_y, uint8_t *dst_d, int dst_stride, size_t dst_batch_stride,
                        uint8_t *lut_d)
{
    ClaheParams P;
    int ew = w, eh = h;
    if (!(w % tiles_x == 0 && h % tiles_y == 0)) { ew = w + (tiles_x - w % tiles_x); eh = h + (tiles_y - h % tiles_y); }
    P.w = w; P.h = h; P.stride = stride; P.tiles_x = tiles_x; P.tiles_y = tiles_y;
    P.tw = ew / tiles_x; P.th = eh / tiles_y;
    const int total = P.tw * P.th;
    P.lut_scale = (float)(256 - 1) / (float)total;
    P.clip = 0;
    if (clip_limit > 0.0) { P.clip = (int)(clip_limit * total / 256); if (P.clip < 1) P.clip = 1; }
    P.inv_tw = 1.0f / (float)P.tw; P.inv_th = 1.0f / (float)P.th;
    P.src_item_stride = (long long)src_batch_stride; P.dst_item_stride = (long long)dst_batch_stride; P.dst_stride = dst_stride;
    hipLaunchKernelGGL(k_clahe_lut, dim3(tiles_x * tiles_y, batch), dim3(256), 0, ctx->stream, P, src_d, lut_d);
    hipLaunchKernelGGL(k_clahe_apply, dim3((w + 1023) / 1024, h, batch), dim3(256), 0, ctx->stream, P, src_d, lut_d, dst_d);
    OV2_HIP_CHECK(hipGetLastError());
    return OV2_OK;
}

extern "C" {

int ov2_clahe_d(ov2_ctx *ctx, const uint8_t *src_d, int w, int h, int stride, size_t src_batch_stride, int batch,
                double clip_limit, int tiles_x, int tiles_y, uint8_t *dst_d, int dst_stride, size_t dst_batch_stride)
{
    OV2_REQUIRE(ctx && src_d && dst_d, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(w > 0 && h > 0 && stride >= w && dst_stride >= w && batch >= 1 && tiles_x > 0 && tiles_y > 0, OV2_EINVAL, "bad geometry");
    OV2_REQUIRE(tiles_x <= w && tiles_y <= h, OV2_EINVAL, "more tiles than pixels");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t lut_bytes = (size_t)batch * tiles_x * tiles_y * 256;
    const int rc = ctx->reserve_device(lut_bytes);
    if (rc != OV2_OK) return rc;
    return clahe_launch(ctx, src_d, w, h, stride, src_batch_stride, batch, clip_limit, tiles_x, tiles_y, dst_d, dst_stride,
                        dst_batch_stride, (uint8_t *)ctx->d_scratch);
}

int ov2_clahe_h(ov2_ctx *ctx, const uint8_t *src_h, int w, int h, int stride, double clip_limit, int tiles_x, int tiles_y,
                uint8_t *dst_h, int dst_stride)
{
    OV2_REQUIRE(ctx && src_h && dst_h, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(w > 0 && h > 0 && stride >= w && dst_stride >= w && tiles_x > 0 && tiles_y > 0, OV2_EINVAL, "bad geometry");
    OV2_REQUIRE(tiles_x <= w && tiles_y <= h, OV2_EINVAL, "more tiles than pixels");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t img = ((size_t)w * h + 255) & ~(size_t)255, lut_bytes = (size_t)tiles_x * tiles_y * 256;
    const int rc = ctx->reserve_device(2 * img + lut_bytes);
    if (rc != OV2_OK) return rc;
    uint8_t *ds = (uint8_t *)ctx->d_scratch;
    OV2_HIP_CHECK(hipMemcpy2DAsync(ds, (size_t)w, src_h, (size_t)stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice, ctx->stream));
    const int rc2 = clahe_launch(ctx, ds, w, h, w, 0, 1, clip_limit, tiles_x, tiles_y, ds + img, w, 0, ds + 2 * img);
    if (rc2 != OV2_OK) return rc2;
    OV2_HIP_CHECK(hipMemcpy2DAsync(dst_h, (size_t)dst_stride, ds + img, (size_t)w, (size_t)w, (size_t)h, hipMemcpyDeviceToHost, ctx->stream));
    OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return OV2_OK;
}

} // extern "C"
