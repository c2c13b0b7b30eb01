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

// One WAVEFRONT per (tile, image), four tiles per workgroup, no workgroup barriers: 16 lanes cover one
// tile row as aligned dwords (<= 64 bytes), so a wavefront histograms 4 rows per trip into its own
// 256-bin LDS histogram; clip / redistribute / scan run on 4 bins per lane with wave shuffles.
__device__ __forceinline__ void clahe_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void k_clahe_lut(ClaheParams P, const uint8_t *__restrict__ src, uint8_t *__restrict__ lut)
{
    __shared__ int hist_all[4][4][256];                              // [wave][lane & 3][bin]: fewer same-address ds_add collisions
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ntiles = P.tiles_x * P.tiles_y;
    const int t = blockIdx.x * 4 + wave, b = blockIdx.y;
    if (t >= ntiles) return;                                        // whole wavefront exits
    int *hist = hist_all[wave][lane & 3];
    int *hist0 = hist_all[wave][0];
    const int ty = t / P.tiles_x, tx = t - ty * P.tiles_x;
    const uint8_t *img = src + (long long)b * P.src_item_stride;
#pragma unroll
    for (int k = 0; k < 16; k++) hist0[lane + 64 * k] = 0;
    clahe_wave_sync();
    const int sub = lane >> 4, l16 = lane & 15;
    const int x_begin = tx * P.tw, x_end = x_begin + P.tw;          // tile columns in padded coordinates
    const bool fast = x_end <= P.w && P.tw <= 61 && ((P.stride | (int)(size_t)img) & 3) == 0;
    const int xa = (x_begin & ~3) + 4 * l16;                        // aligned dword l16 of a row segment
    for (int ly = sub; ly < P.th; ly += 4) {
        const int y = c_reflect101(ty * P.th + ly, P.h);            // bottom REFLECT_101 padding
        const uint8_t *row = img + y * P.stride;
        if (fast) {
            if (xa < x_end) {
                const uint32_t v = *(const uint32_t *)(row + xa);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int x = xa + k;
                    if (x >= x_begin && x < x_end) atomicAdd(&hist[(v >> (8 * k)) & 0xFF], 1);
                }
            }
        } else {
            for (int lx = l16; lx < P.tw; lx += 16) atomicAdd(&hist[row[c_reflect101(x_begin + lx, P.w)]], 1);
        }
    }
    clahe_wave_sync();
    // lane owns bins 4*lane .. 4*lane+3
    int hv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) hv[k] = hist0[4 * lane + k] + hist0[256 + 4 * lane + k] + hist0[512 + 4 * lane + k] + hist0[768 + 4 * lane + k];
    if (P.clip > 0) {
        int over = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) if (hv[k] > P.clip) { over += hv[k] - P.clip; hv[k] = P.clip; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) over += __shfl_xor(over, off, 64);
        const int clipped = over;
        const int batch = clipped / 256, residual = clipped - batch * 256;
        int step = residual ? 256 / residual : 1; if (step < 1) step = 1;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int bin = 4 * lane + k;
            hv[k] += batch;
            // serial loop `for (i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++`
            if (residual != 0 && bin % step == 0 && bin / step < residual) hv[k]++;
        }
    }
    // inclusive scan over the 256 bins: in-lane prefix + wave scan of the lane totals
    hv[1] += hv[0]; hv[2] += hv[1]; hv[3] += hv[2];
    int tot = hv[3], v = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(v, off, 64);
        if (lane >= off) v += u;
    }
    const int base = v - tot;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int r = __float2int_rn((float)(hv[k] + base) * P.lut_scale);   // saturate_cast<uchar>(float)
        r = r < 0 ? 0 : (r > 255 ? 255 : r);
        packed |= (uint32_t)r << (8 * k);
    }
    *(uint32_t *)(lut + ((long long)b * ntiles + t) * 256 + 4 * lane) = packed;
}

// One workgroup per (row of interpolation cells, image): pixels of rows whose two surrounding tile rows are
// (cy-1, cy).  LDS holds, for every cell column, the four surrounding tile LUTs packed as one dword per gray
// value, plus the per-column cell index and horizontal weight -- a pixel costs two LDS reads.
#define CLAHE_MAX_CELLS 48
__global__ __launch_bounds__(256) void k_clahe_apply(ClaheParams P, const uint8_t *__restrict__ src, const uint8_t *__restrict__ lut,
                                                     uint8_t *__restrict__ dst)
{
    extern __shared__ __align__(16) unsigned char clahe_smem[];
    const int ncx = P.tiles_x + 1;
    uint32_t *lut4 = (uint32_t *)clahe_smem;                            // ncx * 256
    float *xaT = (float *)(lut4 + ncx * 256);                           // w
    uint8_t *cxT = (uint8_t *)(xaT + P.w);                              // w
    const int cy = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int ty1 = max(cy - 1, 0), ty2 = min(cy, P.tiles_y - 1);
    const uint8_t *L = lut + (long long)b * P.tiles_x * P.tiles_y * 256;
    for (int e = tid; e < ncx * 256; e += 256) {
        const int cx = e >> 8, v = e & 255;
        const int tx1 = max(cx - 1, 0), tx2 = min(cx, P.tiles_x - 1);
        lut4[e] = (uint32_t)L[(ty1 * P.tiles_x + tx1) * 256 + v] | ((uint32_t)L[(ty1 * P.tiles_x + tx2) * 256 + v] << 8) |
                  ((uint32_t)L[(ty2 * P.tiles_x + tx1) * 256 + v] << 16) | ((uint32_t)L[(ty2 * P.tiles_x + tx2) * 256 + v] << 24);
    }
    for (int x = tid; x < P.w; x += 256) {
        const float txf = (float)x * P.inv_tw - 0.5f;
        const int fx = (int)floorf(txf);
        xaT[x] = txf - (float)fx;
        cxT[x] = (uint8_t)(fx + 1);
    }
    __syncthreads();
    // candidate rows of this cell row (membership decided by the float formula, like the reference)
    const int ys = max(0, (int)floorf(((float)cy - 0.5f) * (float)P.th) - 1), ye = min(P.h, (int)ceilf(((float)cy + 0.5f) * (float)P.th) + 2);
    const uint8_t *simg = src + (long long)b * P.src_item_stride;
    uint8_t *dimg = dst + (long long)b * P.dst_item_stride;
    const bool aligned = ((P.stride | P.dst_stride | (int)(size_t)simg | (int)(size_t)dimg) & 3) == 0;
    const int ndw = (P.w + 3) >> 2;
    const int total = (ye - ys) * ndw;
    for (int e = tid; e < total; e += 256) {
        const int ry = e / ndw, dwi = e - ry * ndw;
        const int y = ys + ry, xb = 4 * dwi;
        const float tyf = (float)y * P.inv_th - 0.5f;
        const int fy = (int)floorf(tyf);
        if (fy + 1 != cy) continue;                                    // row belongs to another cell row
        const float ya = tyf - (float)fy, ya1 = 1.0f - ya;
        const uint8_t *srow = simg + y * P.stride;
        uint8_t *drow = dimg + y * P.dst_stride;
        const bool full = aligned && xb + 3 < P.w;
        uint32_t in = 0;
        if (full) in = *(const uint32_t *)(srow + xb);
        else for (int k = 0; k < 4; k++) if (xb + k < P.w) in |= (uint32_t)srow[xb + k] << (8 * k);
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = min(xb + k, P.w - 1);
            const float xa = xaT[x], xa1 = 1.0f - xa;
            const uint32_t q = lut4[((int)cxT[x] << 8) + ((in >> (8 * k)) & 0xFF)];
            const float l11 = (float)(q & 0xFF), l12 = (float)((q >> 8) & 0xFF), l21 = (float)((q >> 16) & 0xFF), l22 = (float)(q >> 24);
            const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
            int r = __float2int_rn(res);
            r = r < 0 ? 0 : (r > 255 ? 255 : r);
            out |= (uint32_t)r << (8 * k);
        }
        if (full) *(uint32_t *)(drow + xb) = out;
        else for (int k = 0; k < 4; k++) if (xb + k < P.w) drow[xb + k] = (uint8_t)(out >> (8 * k));
    }
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
    hipLaunchKernelGGL(k_clahe_lut, dim3((tiles_x * tiles_y + 3) / 4, batch), dim3(256), 0, ctx->stream, P, src_d, lut_d);
    const size_t apply_lds = (size_t)(tiles_x + 1) * 1024 + (size_t)w * 5 + 16;
    OV2_REQUIRE(tiles_x + 1 <= CLAHE_MAX_CELLS && apply_lds <= 160 * 1024, OV2_EUNSUPPORTED, "CLAHE: too many tile columns / too wide an image for the LDS tables");
    OV2_HIP_CHECK(hipFuncSetAttribute((const void *)k_clahe_apply, hipFuncAttributeMaxDynamicSharedMemorySize, (int)apply_lds));
    hipLaunchKernelGGL(k_clahe_apply, dim3(tiles_y + 1, batch), dim3(256), apply_lds, ctx->stream, P, src_d, lut_d, dst_d);
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
