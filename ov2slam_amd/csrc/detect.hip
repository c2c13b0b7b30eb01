// detect.hip -- grid keypoint detection for gfx950.
//
// Replaces FeatureExtractor::detectGridFAST  (/root/reference/src/feature_extractor.cpp:443-570)
//      and FeatureExtractor::detectSingleScale (/root/reference/src/feature_extractor.cpp:288-440)
// including the OpenCV calls inside them (FAST-9/16 + NMS, GaussianBlur 3x3,
// cornerMinEigenVal(3,3), minMaxLoc, circle(FILLED), cornerSubPix).
//
// Structure (per image):
//   1. k_fast_cells / k_mineig_cells : one workgroup (FAST) / one wavefront (min-eigenvalue) per grid cell,
//      produces the cell's response map (NMS'ed FAST score bytes / min-eigenvalue floats).
//      Each also leaves the cell's selection CANDIDATES (CellCand): what the selection would take from an untouched mask.
//   2. k_grid_select : ONE workgroup.  The exclusion mask (the reference's CV_32F ones image with
//      zeroed discs) lives in LDS as a bitmask.  The reference walks the cells serially and every
//      accepted point zeroes a disc of radius cell/4 that can only reach the 8 neighbouring cells,
//      so cell (r,c) depends on (r,c-1), (r-1,c-1), (r-1,c), (r-1,c+1) only.  Wavefront w walks along cell row w
//      behind a per-row progress counter (no work-group barrier); a cell whose candidates are still unmasked costs two
//      LDS bit tests, any other cell the full masked scan (a lane owns one row of the cell: its mask slice is one 64-bit
//      string, its responses sit in registers, the maps are stored column-major so that column loads coalesce).
//      Bit-identical results to the serial raster order.
//   3. k_corner_subpix : cv::cornerSubPix, one wavefront per point (parallel patch, ordered accumulation).
#include "common.hpp"
#include <limits.h>
#include <float.h>
#include <math.h>
#include <algorithm>
#include <mutex>

#pragma clang fp contract(off)

#define DET_MAX_CELL 64
#ifndef DET_TWO_STREAMS
#define DET_TWO_STREAMS 1
#endif
#ifndef DET_CHUNK
#define DET_CHUNK 1024     // images per pass of the batch entry points (the response maps of a pass live in the scratch: 1.3 MB per EuRoC image)
#endif

__device__ __forceinline__ int d_reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// ---------------------------------------------------------------------------------
// per-cell selection candidates (computed by the cell kernels while the response map is still in LDS)
// ---------------------------------------------------------------------------------
// The selection sweep (k_grid_select) is a dependent chain over the cells; what a cell would select if NO earlier disc
// reached into it does not depend on the chain and is computed here, in parallel over all cells:
//   p1 / v1 : first maximum of the response in raster order (minMaxLoc; FAST: among the corners the mask mode can select at all)
//   p2 / v2 : [single scale] first maximum outside the disc cv::circle would draw around p1
// An exclusion mask only ever removes candidates (masked responses become 0), so when p1 (p2) is positive and its mask bit is
// still set when the sweep reaches the cell, it IS the masked arg-max -- the sweep then costs two LDS bit tests per cell instead
// of a load batch and two arg-max passes, and falls back to the full scan otherwise.
struct CellCand { int p1; float v1; int p2; float v2; };      // p = lx | ly << 16 (cell-local pixel), -1: none

// Batched launches (one image per batch item of a pyramid, ov2_detect_*_batch_d): per-item strides and parameters; everything
// zero / NULL for a single image.  The cell kernels run ncells work-groups per item, the selection one work-group per item,
// the sub-pixel refinement a wavefront per (output slot, item).
struct DetBatch {
    long long img_stride;      // bytes between the items' images
    int ncells;                // cells per image
    int cur_stride;            // float2 entries between the items' current-keypoint lists
    int out_stride;            // float2 entries between the items' output lists
    const int *ncur;           // per-item number of current keypoints   (NULL: ncur_all)
    const float2 *cur;         // item 0's current keypoints (the cell kernels skip occupied cells like the reference's loop does)
    int ncur_all;              // number of current keypoints when ncur == NULL
    const int *fast_th;        // per-item FAST threshold                (NULL: the scalar argument)
    const double *quality;     // per-item quality level                 (NULL: SelectParams::quality)
};

// wave-wide arg-max of (value, smaller index wins ties); all 64 lanes participate
__device__ __forceinline__ void wave_argmax_f(float &v, int &idx)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

// midpoint circle half-widths (drawing.cpp Circle()): rows +-dy get dx, rows +-dx get dy; hw[0..63], -1 = row not touched
__device__ __forceinline__ void d_circle_halfwidths(int *hw, int radius)
{
    for (int k = 0; k < 64; k++) hw[k] = -1;
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        if (dx > hw[dy]) hw[dy] = dx;
        if (dy > hw[dx]) hw[dx] = dy;
        dy++; err += plus; plus += 2;
        const int m = (err <= 0) - 1;
        err -= minus & m; dx += m; minus -= m & 2;
    }
}

__device__ __forceinline__ void block_argmax_f(float &v, int &idx, float *s_v, int *s_i)
{
    wave_argmax_f(v, idx);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) { s_v[wave] = v; s_i[wave] = idx; }
    __syncthreads();
    v = s_v[0]; idx = s_i[0];
    for (int k = 1; k < nw; k++) {
        const float ov = s_v[k]; const int oi = s_i[k];
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

// true when a current keypoint lies in cell (r, c): the reference's loops `continue` on such cells before computing anything
// (voccupcells, :296-319 / :451-474: r = (int)(y / cs), c = (int)(x / cs)); all lanes of the calling wavefront get the answer
__device__ __forceinline__ bool d_cell_occupied(const DetBatch &B, int item, int cs, int r, int c)
{
    const int n = B.ncur ? B.ncur[item] : B.ncur_all;
    if (n <= 0 || B.cur == nullptr) return false;
    const float2 *cur = B.cur + (long long)item * B.cur_stride;
    bool hit = false;
    for (int i = threadIdx.x & 63; i < n; i += 64) {
        const float2 p = cur[i];
        if ((int)(p.y / (float)cs) == r && (int)(p.x / (float)cs) == c) hit = true;
    }
    return __builtin_amdgcn_ballot_w64(hit) != 0;
}

// ---------------------------------------------------------------------------------
// FAST-9/16 + score + 3x3 NMS on the cs x cs sub-image of every cell (cv::FAST semantics)
// out: per cell cs*cs bytes, NMS-surviving corners hold their score (>0), everything else 0
// ---------------------------------------------------------------------------------
__constant__ int c_fast_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
__constant__ int c_fast_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

__global__ __launch_bounds__(256) void k_fast_cells(const uint8_t *__restrict__ img, int w, int h, int stride,
                                                    int cs, int nwcells, int threshold, uint8_t *__restrict__ nms_out,
                                                    int mask_mode, CellCand *__restrict__ cand_out, DetBatch B)
{
    __shared__ uint8_t tile[DET_MAX_CELL * DET_MAX_CELL];
    __shared__ uint8_t score[DET_MAX_CELL * DET_MAX_CELL];
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    const int item = blockIdx.x / B.ncells, cell = blockIdx.x - item * B.ncells;
    const int x0 = (cell % nwcells) * cs, y0 = (cell / nwcells) * cs;
    const int npx = cs * cs;
    img += (long long)item * B.img_stride;
    nms_out += (long long)item * B.ncells * npx;
    cand_out += (long long)item * B.ncells;
    if (B.fast_th) { const int t = B.fast_th[item]; threshold = t < 0 ? 0 : (t > 255 ? 255 : t); }
    if (d_cell_occupied(B, item, cs, cell / nwcells, cell % nwcells)) return;      // (work-group-uniform: every wavefront scans the whole list)
    for (int p = threadIdx.x; p < npx; p += blockDim.x) {
        const int ly = p / cs, lx = p - ly * cs;
        tile[p] = img[(long long)(y0 + ly) * stride + x0 + lx];
        score[p] = 0;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < npx; p += blockDim.x) {
        const int ly = p / cs, lx = p - ly * cs;
        if (ly < 3 || ly >= cs - 3 || lx < 3 || lx >= cs - 3) continue;
        const int v = tile[p];
        int ring[16];
        unsigned dark = 0, bright = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            ring[k] = tile[p + c_fast_dy[k] * cs + c_fast_dx[k]];
            dark |= (unsigned)(ring[k] < v - threshold) << k;
            bright |= (unsigned)(ring[k] > v + threshold) << k;
        }
        unsigned md = dark | (dark << 16), mb = bright | (bright << 16);
        unsigned rd = md, rb = mb;
#pragma unroll
        for (int i = 1; i <= 8; i++) { rd &= md >> i; rb &= mb >> i; }
        if (((rd | rb) & 0xFFFFu) == 0) continue;
        // cornerScore<16>
        int d[25];
#pragma unroll
        for (int k = 0; k < 25; k++) d[k] = v - ring[k & 15];
        int a0 = threshold;
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            int a = min(d[k + 1], min(d[k + 2], d[k + 3]));
            a = min(a, min(d[k + 4], min(d[k + 5], min(d[k + 6], min(d[k + 7], d[k + 8])))));
            a0 = max(a0, min(a, d[k]));
            a0 = max(a0, min(a, d[k + 9]));
        }
        int b0 = -a0;
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            int b = max(d[k + 1], max(d[k + 2], d[k + 3]));
            b = max(b, max(d[k + 4], max(d[k + 5], max(d[k + 6], max(d[k + 7], d[k + 8])))));
            b0 = min(b0, max(b, d[k]));
            b0 = min(b0, max(b, d[k + 9]));
        }
        score[p] = (uint8_t)(-b0 - 1);
    }
    __syncthreads();
    uint8_t *out = nms_out + (long long)cell * npx;
    for (int p = threadIdx.x; p < npx; p += blockDim.x) {
        const int ly = p / cs, lx = p - ly * cs;
        int keep = 0;
        const int s = score[p];
        if (s > 0 && ly >= 3 && ly < cs - 3 && lx >= 3 && lx < cs - 3) {
            keep = s > score[p - 1] && s > score[p + 1] && s > score[p - cs - 1] && s > score[p - cs] &&
                   s > score[p - cs + 1] && s > score[p + cs - 1] && s > score[p + cs] && s > score[p + cs + 1];
        }
        out[lx * cs + ly] = keep ? (uint8_t)s : 0;            // column-major: k_grid_select reads a row per lane, coalesced
        tile[p] = keep ? (uint8_t)s : 0;                      // (the cell's pixels are no longer needed)
    }
    __syncthreads();
    // candidate: the best corner the selection could take from an untouched mask.  AS_EXECUTED reads the CV_32F ones-mask
    // as bytes (N3): only pixels with (lx & 3) >= 2 ever see a non-zero byte
    float bv = 0.f; int bi = 0x7FFFFFFF;
    for (int p = threadIdx.x; p < npx; p += blockDim.x) {
        const int ly = p / cs, lx = p - ly * cs;
        const float v = (float)tile[p];
        if ((mask_mode != OV2_MASK_AS_EXECUTED || (lx & 3) >= 2) && v > bv) { bv = v; bi = p; }
    }
    block_argmax_f(bv, bi, s_v, s_i);
    // how many corners FAST's mask filter lets through on an untouched mask, and how many of them share the best response: the reference
    // sorts them with std::sort (src/feature_extractor.cpp:518), whose choice among equal responses depends on the count (k_grid_select)
    __shared__ int s_cnt[2];
    if (threadIdx.x == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
    __syncthreads();
    int n_l = 0, t_l = 0;
    for (int p = threadIdx.x; p < npx; p += blockDim.x) {
        const int ly = p / cs, lx = p - ly * cs;
        const float v = (float)tile[p];
        if ((mask_mode != OV2_MASK_AS_EXECUTED || (lx & 3) >= 2) && v > 0.f) { n_l++; t_l += v == bv; }
    }
    if (n_l) atomicAdd(&s_cnt[0], n_l);
    if (t_l) atomicAdd(&s_cnt[1], t_l);
    __syncthreads();
    if (threadIdx.x == 0) {
        CellCand cd;
        const int by = bi / cs, bx = bi - by * cs;
        cd.p1 = bv > 0.f ? (bx | (by << 16)) : -1; cd.v1 = bv;
        cd.p2 = min(s_cnt[0], 0xFFFF) | (min(s_cnt[1], 0x7FFF) << 16); cd.v2 = 0.f;       // (FAST cells have no second candidate: the slot carries the counts)
        cand_out[cell] = cd;
    }
}

// ---------------------------------------------------------------------------------
// min-eigenvalue response of every cell: GaussianBlur 3x3 (parent pixels, fixed point) ->
// Sobel/3060 -> (dx^2, dxdy, dy^2) -> 3x3 box (double sums, sliding column) -> lambda_min
// ---------------------------------------------------------------------------------
// ONE WAVEFRONT per cell, lane = cell column, marching down the rows: the blurred cell goes to LDS (bytes), everything after
// it stays in registers -- Sobel of a row from nine LDS bytes, the neighbouring columns' products through ds_bpermute, the
// three double row sums of rows y-1, y, y+1 as a sliding window, ColumnSum's recurrence (SUM + row[y+1], then - row[y-1],
// REFLECT_101 at the cell's edges) exactly as the serial code runs it.  lambda_min is staged row-major in LDS for the
// candidates and the coalesced column-major store.  5 bytes of LDS per pixel and no work-group barrier (round 1 / early round 2:
// 256 threads per cell, five barrier-separated phases over 45 bytes of LDS per pixel -- 2 work-groups per CU, 13 us per image
// in a batch, 24 us for one image).
__global__ __launch_bounds__(64) void k_mineig_cells(const uint8_t *__restrict__ img, int w, int h, int stride,
                                                     int cs, int nwcells, float *__restrict__ hmap_out, int dy_order,
                                                     int radius, CellCand *__restrict__ cand_out, DetBatch B)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int s_hw[64];
    const int npx = cs * cs;
    float *lam = (float *)smem;                               // npx, row-major
    uint8_t *blur = (uint8_t *)(lam + npx);                   // npx
    const int item = blockIdx.x / B.ncells, cell = blockIdx.x - item * B.ncells;
    const int x0 = (cell % nwcells) * cs, y0 = (cell / nwcells) * cs;
    img += (long long)item * B.img_stride;
    hmap_out += (long long)item * B.ncells * npx;
    cand_out += (long long)item * B.ncells;
    const int lane = threadIdx.x;
    if (d_cell_occupied(B, item, cs, cell / nwcells, cell % nwcells)) return;
    const bool act = lane < cs;
    const int x = act ? lane : cs - 1;                        // idle lanes shadow the last column (their results are never stored)
    if (lane == 0) d_circle_halfwidths(s_hw, radius);

    // ---- GaussianBlur 3x3 on the parent image (REFLECT_101 at the IMAGE border), (1 2 1) x (1 2 1), (s + 8) >> 4 ----
    {
        const int gx0 = d_reflect101(x0 + x - 1, w), gx1 = d_reflect101(x0 + x, w), gx2 = d_reflect101(x0 + x + 1, w);
        auto hrow = [&](int gy) {
            const uint8_t *row = img + (long long)d_reflect101(gy, h) * stride;
            return (int)row[gx0] + 2 * (int)row[gx1] + (int)row[gx2];
        };
        // eight rows per trip, their ten source rows (30 byte loads per lane) in flight together: one row per trip made the
        // stage a chain of cs + 2 dependent L2 round trips -- most of the kernel's latency for a single image
        for (int j0 = 0; j0 < cs; j0 += 8) {
            int hv[10];
#pragma unroll
            for (int u = 0; u < 10; u++) hv[u] = hrow(y0 + j0 - 1 + u);          // (rows past the cell: valid image rows, never stored)
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (act && j0 + u < cs) blur[(j0 + u) * cs + x] = (uint8_t)((hv[u] + 2 * hv[u + 1] + hv[u + 2] + 8) >> 4);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // ---- Sobel (REFLECT_101 at the CELL's edges: the blurred cell is a Mat of its own), products, row sums ----
    const float f1 = (float)(1.0 / (4.0 * 3.0 * 255.0)), f0 = (float)(2.0 * (1.0 / (4.0 * 3.0 * 255.0)));
    const int xl = x == 0 ? 1 : x - 1, xr = x == cs - 1 ? cs - 2 : x + 1;
    auto rowsum = [&](int yy, double (&sum)[3]) {
        const int ym = yy == 0 ? 1 : yy - 1, yp = yy == cs - 1 ? cs - 2 : yy + 1;
        const uint8_t *bm = blur + ym * cs, *bc = blur + yy * cs, *bp = blur + yp * cs;
        const int a00 = bm[xl], a01 = bm[x], a02 = bm[xr], a10 = bc[xl], a12 = bc[xr], a20 = bp[xl], a21 = bp[x], a22 = bp[xr];
        const float r0 = (float)(a02 - a00), r1 = (float)(a12 - a10), r2 = (float)(a22 - a20);
        const float dx = (r0 + r2) * f1 + r1 * f0;
        float dy;
        if (dy_order == OV2_SOBEL_DY_EXACT_SUM) {          // round 1's order: scale applied to the exact integer difference
            const float s0 = (float)(a00 + 2 * a01 + a02), s2 = (float)(a20 + 2 * a21 + a22);
            dy = (s2 - s0) * f1;
        } else {
            // cv::Sobel(dx = 0, dy = 1, scale): the scale goes into the smoothing kernel, the row pass is the generic
            // RowFilter<uchar, float> ((p[x-1] k0 + p[x] k1) + p[x+1] k2, every operation rounded), the column pass the exact
            // difference of two rounded rows (include/ov2slam_hip.h: OV2_OPT_SOBEL_DY_ORDER)
            const float s0 = ((float)a00 * f1 + (float)a01 * f0) + (float)a02 * f1;
            const float s2 = ((float)a20 * f1 + (float)a21 * f0) + (float)a22 * f1;
            dy = s2 - s0;
        }
        const float v0 = dx * dx, v1 = dx * dy, v2 = dy * dy;
        // RowSum<float, double> over columns x-1, x, x+1 (REFLECT_101): the neighbours' products come from their lanes
        const float l0 = __shfl(v0, xl, 64), l1 = __shfl(v1, xl, 64), l2 = __shfl(v2, xl, 64);
        const float q0 = __shfl(v0, xr, 64), q1 = __shfl(v1, xr, 64), q2 = __shfl(v2, xr, 64);
        sum[0] = (double)l0; sum[0] = sum[0] + (double)v0; sum[0] = sum[0] + (double)q0;
        sum[1] = (double)l1; sum[1] = sum[1] + (double)v1; sum[1] = sum[1] + (double)q1;
        sum[2] = (double)l2; sum[2] = sum[2] + (double)v2; sum[2] = sum[2] + (double)q2;
    };
    double rp[3] = {0, 0, 0}, rc[3], rn[3], SUM[3];
    rowsum(0, rc); rowsum(1, rn);
#pragma unroll
    for (int ch = 0; ch < 3; ch++) { SUM[ch] = 0; SUM[ch] += rn[ch]; SUM[ch] += rc[ch]; }      // rows[-1 -> 1], rows[0]
    float bv = -INFINITY; int bi = 0x7FFFFFFF;
    for (int y = 0; y < cs; y++) {
        float cov[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const double nxt = (y + 1 <= cs - 1) ? rn[ch] : rp[ch];     // rows[reflect(y + 1)]
            const double prv = (y == 0) ? rn[ch] : rp[ch];             // rows[reflect(y - 1)]
            const double s0 = SUM[ch] + nxt;
            cov[ch] = (float)s0;
            SUM[ch] = s0 - prv;
        }
        const float a = cov[0] * 0.5f, b = cov[1], c = cov[2] * 0.5f;
        const float lm = (a + c) - sqrtf((a - c) * (a - c) + b * b);
        if (act) {
            lam[y * cs + x] = lm;
            if (lm > bv) { bv = lm; bi = y * cs + x; }                  // first maximum of this column; raster order across lanes below
        }
#pragma unroll
        for (int ch = 0; ch < 3; ch++) { rp[ch] = rc[ch]; rc[ch] = rn[ch]; }
        if (y + 2 <= cs - 1) rowsum(y + 2, rn);                          // (wave-uniform)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // column-major map (k_grid_select's full scan reads a row per lane, coalesced), written with coalesced stores
    float *out = hmap_out + (long long)cell * npx;
    for (int e = lane; e < npx; e += 64) { const int lx = e / cs, ly = e - lx * cs; out[e] = lam[ly * cs + lx]; }
    // candidates: first maximum (raster order), and first maximum outside the disc around it
    wave_argmax_f(bv, bi);
    const int p1y = bi / cs, p1x = bi - p1y * cs;
    float bv2 = -INFINITY; int bi2 = 0x7FFFFFFF;
    if (act) {
        const int adx = x > p1x ? x - p1x : p1x - x;
        for (int y = 0; y < cs; y++) {
            const int ady = y > p1y ? y - p1y : p1y - y;
            const bool in_disc = ady <= radius && s_hw[ady < 64 ? ady : 63] >= 0 && adx <= s_hw[ady < 64 ? ady : 63];
            const float v = lam[y * cs + x];
            if (!in_disc && v > bv2) { bv2 = v; bi2 = y * cs + x; }
        }
    }
    wave_argmax_f(bv2, bi2);
    if (lane == 0) {
        CellCand cd;
        cd.p1 = p1x | (p1y << 16); cd.v1 = bv;
        const int p2y = bi2 / cs, p2x = bi2 - p2y * cs;
        cd.p2 = bi2 == 0x7FFFFFFF ? -1 : (p2x | (p2y << 16)); cd.v2 = bv2;
        cand_out[cell] = cd;
    }
}

// ---------------------------------------------------------------------------------
// The same responses for BATCHES of images: k_mineig_strip.  One wavefront per cell keeps 35 of 64 lanes busy on the reference's
// 35-pixel cells and marches occupied cells' wavefronts to an early exit; the kernel is VALU-issue bound (profiles/r6_detect_*), so
// lanes are what there is to win.  Here the FREE cells of an image are laid side by side as one strip of columns (cells are Mats of
// their own: REFLECT_101 at the cell's edges, so neighbours in the strip need not be neighbours in the image) and a work-group of K
// wavefronts takes NCELL of them: lane = strip column, wavefront k owns columns [k O, (k + 1) O) and carries two more on either
// side -- a column's Sobel needs its neighbours' blurred pixels, its box sum their products -- so that no value ever crosses a
// wavefront (59 of 64 lanes own a column for cs = 35: NCELL = 5, K = 3).  Every lane runs, for its column, exactly the sequence of
// operations the one-wavefront-per-cell kernel runs (same rounding, same order: the maps and candidates are bit-identical); the
// neighbours come through DPP wave shifts instead of ds_bpermute, a wavefront's lambda_min columns are staged in LDS (row pitch 65:
// row writes and column reads both conflict-free) for the second candidate and for the map store -- the owned columns of one cell
// are ONE contiguous run of the column-major map, written with coalesced stores (a lane storing its own column row by row touches
// 64 cache lines per instruction: measured 2.2x the whole kernel) --, the per-cell arg-max reductions are 64-bit LDS atomics on
// (ordered value, ~index).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float d_wave_from_left(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, true)); }   // wave_shr:1: lane i <- lane i - 1
__device__ __forceinline__ float d_wave_from_right(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true)); }  // wave_shl:1: lane i <- lane i + 1
__device__ __forceinline__ int d_wave_from_left(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true); }
__device__ __forceinline__ int d_wave_from_right(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true); }
__device__ __forceinline__ unsigned long long d_argmax_key(float v, int idx)
{
    unsigned u = __float_as_uint(v);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;                // monotone map float -> unsigned
    return ((unsigned long long)u << 32) | (unsigned)(0x7FFFFFFF - idx);      // larger value first, then the smaller index
}
__device__ __forceinline__ float d_argmax_key_value(unsigned long long k)
{
    unsigned u = (unsigned)(k >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(u);
}

#define STRIP_MAX_CELLS 8
// the free cells of every image in raster order (the reference's loops `continue` on cells that hold a current keypoint, :296-319):
// free_list[item * (ncells + 1)] = count, then the cell indices.  One wavefront per image.
__global__ __launch_bounds__(64) void k_free_cells(int cs, int nwcells, DetBatch B, int *__restrict__ free_list)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned *occ = (unsigned *)smem;
    const int item = blockIdx.x, lane = threadIdx.x, ncells = B.ncells, nhcells = ncells / nwcells;
    const int occ_words = (ncells + 31) / 32;
    for (int i = lane; i < occ_words; i += 64) occ[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int n = B.ncur ? B.ncur[item] : B.ncur_all;
    if (n > 0 && B.cur != nullptr) {
        const float2 *cur = B.cur + (long long)item * B.cur_stride;
        for (int i = lane; i < n; i += 64) {
            const float2 p = cur[i];
            const int r = (int)(p.y / (float)cs), c = (int)(p.x / (float)cs);
            if (r >= 0 && r < nhcells && c >= 0 && c < nwcells) atomicOr(&occ[(r * nwcells + c) >> 5], 1u << ((r * nwcells + c) & 31));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int *out = free_list + (long long)item * (ncells + 1);
    int seen = 0;
    for (int base = 0; base < ncells; base += 64) {
        const int c = base + lane;
        const bool fr = c < ncells && !((occ[c >> 5] >> (c & 31)) & 1u);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(fr);
        if (fr) out[1 + seen + __popcll(m & ((1ull << lane) - 1ull))] = c;
        seen += __popcll(m);
    }
    if (lane == 0) out[0] = seen;
}
__global__ __launch_bounds__(512) void k_mineig_strip(const uint8_t *__restrict__ img, int w, int h, int stride,
                                                      int cs, int nwcells, float *__restrict__ hmap_out, int dy_order,
                                                      int radius, CellCand *__restrict__ cand_out, DetBatch B, int ncell_wg, int owned,
                                                      const int *__restrict__ free_list)
{
    extern __shared__ __align__(16) unsigned char smem[];       // the blurred columns: cs rows x blockDim bytes, then the wavefronts' lambda_min
    __shared__ int s_hw[64];
    __shared__ int s_cell[STRIP_MAX_CELLS];
    __shared__ int s_nmine;
    __shared__ unsigned long long s_key1[STRIP_MAX_CELLS], s_key2[STRIP_MAX_CELLS];
    const int item = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nt = blockDim.x;
    const int npx = cs * cs, ncells = B.ncells;
    img += (long long)item * B.img_stride;
    hmap_out += (long long)item * ncells * npx;
    cand_out += (long long)item * ncells;
    // ---- this work-group's share of the image's free cells (k_free_cells) ----
    if (tid < STRIP_MAX_CELLS) { s_key1[tid] = 0; s_key2[tid] = 0; s_cell[tid] = -1; }
    if (tid == 0) d_circle_halfwidths(s_hw, radius);
    {
        const int *fl = free_list + (long long)item * (ncells + 1);
        const int first = blockIdx.x * ncell_wg, left = fl[0] - first;
        if (left <= 0) return;                                     // (work-group uniform)
        if (tid == 0) s_nmine = left > ncell_wg ? ncell_wg : left;
        __syncthreads();
        if (tid < ncell_wg && tid < left) s_cell[tid] = fl[1 + first + tid];
    }
    __syncthreads();
    const int nmine = s_nmine;
    if (nmine == 0) return;
    // ---- this lane's column ----
    const int g = wave * owned - 2 + lane;                     // strip column; two carried columns on either side of the owned range
    const int ncols = nmine * cs;
    const bool valid = g >= 0 && g < ncols;
    const bool owns = valid && lane >= 2 && lane < 2 + owned;
    const int gc = valid ? g : 0;
    const int slot = gc / cs, x = gc - slot * cs;
    const int cell = s_cell[slot];
    const int x0 = (cell % nwcells) * cs, y0 = (cell / nwcells) * cs;
    uint8_t *blur = smem + tid;                                 // row y of this column: blur[y * nt]
    float *lamw = (float *)(smem + ((cs * nt + 15) & ~15)) + wave * cs * 65;     // this wavefront's lambda_min: lamw[y * 65 + lane]
    // ---- GaussianBlur 3x3 on the parent image (REFLECT_101 at the IMAGE border), (1 2 1) x (1 2 1), (s + 8) >> 4 ----
    {
        const int gx0 = d_reflect101(x0 + x - 1, w), gx1 = d_reflect101(x0 + x, w), gx2 = d_reflect101(x0 + x + 1, w);
        auto hrow = [&](int gy) {
            const uint8_t *row = img + (long long)d_reflect101(gy, h) * stride;
            return (int)row[gx0] + 2 * (int)row[gx1] + (int)row[gx2];
        };
        for (int j0 = 0; j0 < cs; j0 += 8) {
            int hv[10];
#pragma unroll
            for (int u = 0; u < 10; u++) hv[u] = hrow(y0 + j0 - 1 + u);
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (j0 + u < cs) blur[(j0 + u) * nt] = (uint8_t)((hv[u] + 2 * hv[u + 1] + hv[u + 2] + 8) >> 4);
        }
    }
    // (a lane reads back its OWN bytes only: no barrier)
    const float f1 = (float)(1.0 / (4.0 * 3.0 * 255.0)), f0 = (float)(2.0 * (1.0 / (4.0 * 3.0 * 255.0)));
    const bool at_l = x == 0, at_r = x == cs - 1;
    // per image row of the cell: this column's blurred pixel and its REFLECT_101 neighbours, as the row filter sees them
    auto blur_row = [&](int yy, int &bl, int &bc, int &br) {
        bc = blur[yy * nt];
        const int L = d_wave_from_left(bc), R = d_wave_from_right(bc);
        bl = at_l ? R : L; br = at_r ? L : R;
    };
    // Sobel pieces of one blurred row: d = right - left (dx's row term), s = the smoothed row (dy's row term, in the selected order)
    struct RowT { float d, s; int sx; };
    auto row_terms = [&](int yy) {
        int bl, bc, br;
        blur_row(yy, bl, bc, br);
        RowT t;
        t.d = (float)(br - bl);
        t.sx = bl + 2 * bc + br;
        t.s = ((float)bl * f1 + (float)bc * f0) + (float)br * f1;
        return t;
    };
    auto rowsum = [&](const RowT &tm, const RowT &tc, const RowT &tp, double (&sum)[3]) {
        const float dx = (tm.d + tp.d) * f1 + tc.d * f0;
        float dy;
        if (dy_order == OV2_SOBEL_DY_EXACT_SUM) dy = ((float)tp.sx - (float)tm.sx) * f1;
        else dy = tp.s - tm.s;
        const float v0 = dx * dx, v1 = dx * dy, v2 = dy * dy;
        const float L0 = d_wave_from_left(v0), L1 = d_wave_from_left(v1), L2 = d_wave_from_left(v2);
        const float R0 = d_wave_from_right(v0), R1 = d_wave_from_right(v1), R2 = d_wave_from_right(v2);
        const float l0 = at_l ? R0 : L0, l1 = at_l ? R1 : L1, l2 = at_l ? R2 : L2;
        const float q0 = at_r ? L0 : R0, q1 = at_r ? L1 : R1, q2 = at_r ? L2 : R2;
        sum[0] = (double)l0; sum[0] = sum[0] + (double)v0; sum[0] = sum[0] + (double)q0;
        sum[1] = (double)l1; sum[1] = sum[1] + (double)v1; sum[1] = sum[1] + (double)q1;
        sum[2] = (double)l2; sum[2] = sum[2] + (double)v2; sum[2] = sum[2] + (double)q2;
    };
    // blurred rows y - 1, y, y + 1 of the Sobel stencil (REFLECT_101 at the cell's top / bottom: row -1 = row 1, row cs = row cs - 2)
    RowT t0 = row_terms(0), t1 = row_terms(1), t2 = row_terms(cs > 2 ? 2 : cs - 1);
    double rp[3] = {0, 0, 0}, rc[3], rn[3], SUM[3];
    rowsum(t1, t0, t1, rc);                                     // row 0: rows (1, 0, 1)
    rowsum(t0, t1, t2, rn);                                     // row 1: rows (0, 1, 2)
#pragma unroll
    for (int ch = 0; ch < 3; ch++) { SUM[ch] = 0; SUM[ch] += rn[ch]; SUM[ch] += rc[ch]; }
    float bv = -INFINITY; int bi = 0x7FFFFFFF;
    // ta, tb: blurred rows y + 1 and y + 2 when rn is about to be computed for row y + 2  (rows (y + 1, y + 2, y + 3))
    RowT ta = t1, tb = t2;
    for (int y = 0; y < cs; y++) {
        float cov[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const double nxt = (y + 1 <= cs - 1) ? rn[ch] : rp[ch];
            const double prv = (y == 0) ? rn[ch] : rp[ch];
            const double s0 = SUM[ch] + nxt;
            cov[ch] = (float)s0;
            SUM[ch] = s0 - prv;
        }
        const float a = cov[0] * 0.5f, b = cov[1], c = cov[2] * 0.5f;
        const float lm = (a + c) - sqrtf((a - c) * (a - c) + b * b);
        lamw[y * 65 + lane] = lm;
        if (owns && lm > bv) { bv = lm; bi = y * cs + x; }
#pragma unroll
        for (int ch = 0; ch < 3; ch++) { rp[ch] = rc[ch]; rc[ch] = rn[ch]; }
        if (y + 2 <= cs - 1) {                                  // (wave-uniform) row y + 2: blurred rows y + 1, y + 2, y + 3 (reflected at the bottom)
            const int y3 = y + 3 <= cs - 1 ? y + 3 : cs - 2;
            const RowT tn = (y + 3 <= cs - 1) ? row_terms(y3) : ta;            // row cs reflects to row cs - 2 = the stencil's top row
            rowsum(ta, tb, tn, rn);
            ta = tb; tb = tn;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- the map: this wavefront's owned columns, cell by cell one contiguous run of the column-major layout ----
    {
        const int c_first = wave * owned, c_end = min(ncols, c_first + owned);          // owned strip columns [c_first, c_end)
        const int total = (c_end - c_first) * cs;
        // element e of the run: column c_first + e / cs, row e % cs (advanced incrementally: 64 elements per trip)
        int col = c_first + lane / cs, yy = lane - (lane / cs) * cs;
        const int dcol = 64 / cs, dy = 64 - dcol * cs;
        for (int e = lane; e < total; e += 64) {
            const int sl = col / cs, xx = col - sl * cs;
            hmap_out[(long long)s_cell[sl] * npx + xx * cs + yy] = lamw[yy * 65 + (col - c_first + 2)];
            col += dcol; yy += dy;
            if (yy >= cs) { yy -= cs; col++; }
        }
    }
    // ---- candidates: first maximum of the cell in raster order, then the first maximum outside the disc around it ----
    if (owns && bi != 0x7FFFFFFF) atomicMax(&s_key1[slot], d_argmax_key(bv, bi));
    __syncthreads();
    const unsigned long long k1 = s_key1[slot];
    const int bi1 = 0x7FFFFFFF - (int)(unsigned)(k1 & 0xFFFFFFFFull);
    const int p1y = bi1 / cs, p1x = bi1 - p1y * cs;
    float bv2 = -INFINITY; int bi2 = 0x7FFFFFFF;
    if (owns) {
        const int adx = x > p1x ? x - p1x : p1x - x;
        for (int y = 0; y < cs; y++) {
            const int ady = y > p1y ? y - p1y : p1y - y;
            const bool in_disc = ady <= radius && s_hw[ady < 64 ? ady : 63] >= 0 && adx <= s_hw[ady < 64 ? ady : 63];
            const float v = lamw[y * 65 + lane];
            if (!in_disc && v > bv2) { bv2 = v; bi2 = y * cs + x; }
        }
        if (bi2 != 0x7FFFFFFF) atomicMax(&s_key2[slot], d_argmax_key(bv2, bi2));
    }
    __syncthreads();
    if (tid < nmine) {
        const unsigned long long a1 = s_key1[tid], a2 = s_key2[tid];
        CellCand cd;
        const int i1 = 0x7FFFFFFF - (int)(unsigned)(a1 & 0xFFFFFFFFull);
        cd.p1 = (i1 % cs) | ((i1 / cs) << 16); cd.v1 = d_argmax_key_value(a1);
        if (a2 == 0) { cd.p2 = -1; cd.v2 = -INFINITY; }
        else { const int i2 = 0x7FFFFFFF - (int)(unsigned)(a2 & 0xFFFFFFFFull); cd.p2 = (i2 % cs) | ((i2 / cs) << 16); cd.v2 = d_argmax_key_value(a2); }
        cand_out[s_cell[tid]] = cd;
    }
}

// ---------------------------------------------------------------------------------
// selection sweep
// ---------------------------------------------------------------------------------
struct SelectParams {
    int w, h, cs, nwcells, nhcells, radius, mask_words_per_row;
    int mode;            // 0 = FAST, 1 = single scale
    int mask_mode;       // OV2_MASK_AS_EXECUTED / OV2_MASK_INTENDED (FAST only)
    int fast_tie;        // OV2_FAST_TIE_SCAN_ORDER / OV2_FAST_TIE_LIBSTDCXX (FAST only): which of several equal best responses wins
    int sort_slots;      // LIBSTDCXX tie order: sort scratch areas at the end of the dynamic LDS (1..16; 16 = one per wavefront, never contended)
    int ncur;
    int roi_x, roi_y, roi_w, roi_h;
    double quality;
};

struct SelectOut {
    int n;               // points written to out_xy
    int nboccup, nbempty, nbkps;
    int nslow, pad_;             // cells of the sweep that needed the full masked scan (a candidate was hit by a neighbour's disc)
    unsigned long long dbg[4];   // wall_clock64 ticks (100 MHz): init, prologue, sweep, compaction
};

__device__ __forceinline__ void mask_clear_span(unsigned *mask, int wpr, int y, int xa, int xb)
{
    // clear bits [xa, xb] of row y (already clipped, xa <= xb)
    unsigned *row = mask + y * wpr;
    const int wa = xa >> 5, wb = xb >> 5;
    for (int wd = wa; wd <= wb; wd++) {
        const int lo = wd == wa ? (xa & 31) : 0, hi = wd == wb ? (xb & 31) : 31;
        const unsigned bits = (hi == 31 ? 0xFFFFFFFFu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
        atomicAnd(&row[wd], ~bits);
    }
}

// cv::circle(mask, (cx,cy), R, 0, FILLED) using the precomputed midpoint half-widths hw[0..R];
// `lane`/`nlanes` cooperate over the 2R+1 scan lines.
__device__ __forceinline__ void mask_draw_circle(unsigned *mask, const SelectParams &P, const int *hw,
                                                 int cx, int cy, int lane, int nlanes)
{
    for (int k = lane - P.radius; k <= P.radius; k += nlanes) {
        const int y = cy + k;
        if (y < 0 || y >= P.h) continue;
        const int half = hw[k < 0 ? -k : k];
        if (half < 0) continue;
        int xa = cx - half, xb = cx + half;
        if (xa >= P.w || xb < 0) continue;
        xa = xa < 0 ? 0 : xa; xb = xb > P.w - 1 ? P.w - 1 : xb;
        mask_clear_span(mask, P.mask_words_per_row, y, xa, xb);
    }
}

__device__ __forceinline__ int mask_test(const unsigned *mask, int wpr, int x, int y)
{
    return (mask[y * wpr + (x >> 5)] >> (x & 31)) & 1u;
}

// std::sort(vkps.begin(), vkps.end(), compare_response) as libstdc++ runs it (bits/stl_algo.h: introsort with the median of three moved to the
// front, unguarded partition, final insertion sort with its 16-element threshold), on packed records score << 16 | ly << 8 | lx in LDS,
// comparator a.score > b.score; one lane.  Only the order of EQUAL scores depends on these details: the reference takes element 0
// (src/feature_extractor.cpp:518-522), so among equal best responses of a cell with more than 16 corners it takes libstdc++'s pick.
// (oracle/detect.c: fk_std_sort is the same code on the CPU; tests/test_reference_factors.py runs the reference's own source against it.)
#define FK_LESS(a, b) (((a) >> 16) > ((b) >> 16))
__device__ __forceinline__ void d_fk_unguarded_linear_insert(int *last)
{
    const int val = *last; int *next = last - 1;
    while (FK_LESS(val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}
__device__ __forceinline__ void d_fk_insertion_sort(int *first, int *last)
{
    if (first == last) return;
    for (int *i = first + 1; i != last; ++i) {
        if (FK_LESS(*i, *first)) { const int val = *i; for (int *q = i; q != first; --q) *q = *(q - 1); *first = val; }
        else d_fk_unguarded_linear_insert(i);
    }
}
__device__ __forceinline__ void d_fk_swap(int *a, int *b) { const int t = *a; *a = *b; *b = t; }
__device__ void d_fk_std_sort(int *first, int n, int *stk)
{
    if (n <= 0) return;
    int lg = 0; for (int m = n; m > 1; m >>= 1) lg++;
    // __introsort_loop, its recursion on the right part unrolled onto a small stack of (first, last, depth) offsets (in LDS: `stk`, 3 x 24 ints)
    int sp = 0;
    stk[0] = 0; stk[1] = n; stk[2] = 2 * lg; sp = 1;
    while (sp > 0) {
        --sp;
        int *f = first + stk[3 * sp], *l = first + stk[3 * sp + 1]; int depth = stk[3 * sp + 2];
        while (l - f > 16) {
            if (depth == 0) { d_fk_insertion_sort(f, l); break; }                 // (libstdc++: heap sort; never reached on cells -- the oracle counts it)
            --depth;
            int *mid = f + (l - f) / 2, *a = f + 1, *b = mid, *c = l - 1;
            if (FK_LESS(*a, *b)) { if (FK_LESS(*b, *c)) d_fk_swap(f, b); else if (FK_LESS(*a, *c)) d_fk_swap(f, c); else d_fk_swap(f, a); }
            else if (FK_LESS(*a, *c)) d_fk_swap(f, a);
            else if (FK_LESS(*b, *c)) d_fk_swap(f, c);
            else d_fk_swap(f, b);
            int *pf = f + 1, *pl = l;
            for (;;) {
                while (FK_LESS(*pf, *f)) ++pf;
                --pl;
                while (FK_LESS(*f, *pl)) --pl;
                if (!(pf < pl)) break;
                d_fk_swap(pf, pl);
                ++pf;
            }
            // __introsort_loop(cut, last, depth) first, then the loop goes on with [first, cut): the order of the two does not change the
            // result (disjoint ranges), so the right part waits on the stack
            if (sp < 24) { stk[3 * sp] = (int)(pf - first); stk[3 * sp + 1] = (int)(l - first); stk[3 * sp + 2] = depth; sp++; }
            l = pf;
        }
    }
    if (n > 16) { d_fk_insertion_sort(first, first + 16); for (int *i = first + 16; i != first + n; ++i) d_fk_unguarded_linear_insert(i); }
    else d_fk_insertion_sort(first, first + n);
}
#undef FK_LESS
#ifndef DET_SEL_THREADS_BATCH
#define DET_SEL_THREADS_BATCH 512
#endif
#ifndef DET_SEL_MIN_WAVES
#define DET_SEL_MIN_WAVES 4          // wavefronts per SIMD the batch instance is compiled for (register cap 512 / this)
#endif
#define DET_SORT_CAP 512        // corners of a cell that can enter the emulated sort (non-adjacent after NMS: <= 32 x 32 / 2 for cs = 64, halved again by N3)

// MODE 0 = FAST scores (bytes), 1 = min-eigenvalue (floats); a cell row (<= MAXROW * CHUNKS columns) is held in registers
// MAXROW columns at a time: (36,1) and (52,1) cover the reference cell sizes with compile-time column indices, (32,2) the rest
// NT: threads of the launch -- 1024 for one image (a wavefront per row of cells: latency), DET_SEL_THREADS_BATCH for batches (throughput:
// fewer wavefronts and registers per work-group, more work-groups per CU)
template <int MODE, int MAXROW, int CHUNKS, int NT>
__global__ __launch_bounds__(NT, NT == 1024 ? 1 : DET_SEL_MIN_WAVES) void k_grid_select(SelectParams P, const float2 *__restrict__ cur_xy,
                                                      const uint8_t *__restrict__ nms_maps,
                                                      const float *__restrict__ hmaps, const CellCand *__restrict__ cand,
                                                      float2 *__restrict__ out_xy, SelectOut *__restrict__ out, DetBatch B)
{
    extern __shared__ __align__(16) unsigned char smem[];
    {   // batch item of this work-group
        const int item = blockIdx.x;
        const long long map_items = (long long)item * B.ncells * P.cs * P.cs;
        cur_xy += (long long)item * B.cur_stride; nms_maps += map_items; hmaps += map_items; cand += (long long)item * B.ncells;
        out_xy += (long long)item * B.out_stride; out += item;
        if (B.ncur) P.ncur = B.ncur[item];
        if (B.quality) P.quality = B.quality[item];
    }
    unsigned *mask = (unsigned *)smem;                                   // h * wpr words
    const int mask_words = P.h * P.mask_words_per_row;
    const int ncells = P.nhcells * P.nwcells;
    int *hw = (int *)(mask + mask_words);                                // radius+1 (padded to 64)
    int *s_cand = hw + 64;                                               // 4 * ncells: the cells' candidates (CellCand as 4 dwords)
    int *progress = s_cand + 4 * ncells;                                 // nhcells: cells completed in each row of cells
    uint8_t *occ = (uint8_t *)(progress + P.nhcells);                    // (nh+1)*(nw+1)
    const int nocc = (P.nhcells + 1) * (P.nwcells + 1);
    // per-cell results (cell order): prim/sec as packed (x | y << 16), -1 = none
    int *prim = (int *)(occ + ((nocc + 3) & ~3));
    int *sec = prim + ncells;

    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;

    unsigned long long tk0 = wall_clock64();
    for (int i = tid; i < 4 * ncells; i += nthreads) s_cand[i] = ((const int *)cand)[i];
    for (int i = tid; i < P.nhcells; i += nthreads) progress[i] = 0;
    __shared__ int s_nslow;
    // FAST tie-break (fast_tie): a wavefront's corners of the current cell + the sort's stack.  Carved out of the DYNAMIC allocation
    // (sort_slots areas, 0 unless the libstdc++ order is asked for) so that the host's LDS guard sees it; an area is taken under
    // its lock for the duration of one sort (with 16 areas wavefront w owns area w and the lock is never contended)
    int *s_sort = sec + ncells;
    __shared__ int s_sort_lock[16];
    if (tid < 16) s_sort_lock[tid] = 0;
    if (tid == 0) s_nslow = 0;
    for (int i = tid; i < mask_words; i += nthreads) mask[i] = 0xFFFFFFFFu;
    for (int i = tid; i < nocc; i += nthreads) occ[i] = 0;
    for (int i = tid; i < ncells; i += nthreads) { prim[i] = -1; sec[i] = -1; }
    if (tid == 0) d_circle_halfwidths(hw, P.radius);
    __syncthreads();
    const unsigned long long tk1 = wall_clock64();
    // prologue (:296-319 / :451-474): occupancy + exclusion discs of the current keypoints
    for (int i = tid; i < P.ncur; i += nthreads) {
        const float2 p = cur_xy[i];
        const int r = (int)(p.y / (float)P.cs), c = (int)(p.x / (float)P.cs);
        if (r >= 0 && r <= P.nhcells && c >= 0 && c <= P.nwcells) occ[r * (P.nwcells + 1) + c] = 1;
    }
    for (int i = wave; i < P.ncur; i += nwaves) {
        const float2 p = cur_xy[i];
        mask_draw_circle(mask, P, hw, __float2int_rn(p.x), __float2int_rn(p.y), lane, 64);
    }
    __syncthreads();

    const unsigned long long tk2 = wall_clock64();
    const int npx = P.cs * P.cs;
    // cv::circle of an accepted point on the sweep's fast path: 2 radius + 1 <= 33 scan lines, one per lane, the half-width of a
    // lane's line in a register (the generic mask_draw_circle reads it from LDS: one more round trip on the chain)
    const int my_k = lane - P.radius;
    const int my_hw = lane <= 2 * P.radius ? hw[my_k < 0 ? -my_k : my_k] : -1;
    auto fast_draw = [&](int cx, int cy) {
        const int y = cy + my_k;
        if (my_hw < 0 || y < 0 || y >= P.h) return;
        int xa = cx - my_hw, xb = cx + my_hw;
        if (xa >= P.w || xb < 0) return;
        xa = xa < 0 ? 0 : xa; xb = xb > P.w - 1 ? P.w - 1 : xb;
        mask_clear_span(mask, P.mask_words_per_row, y, xa, xb);
    };
    // The sweep.  Wavefront w walks along cell rows w, w + 16, ...; cell (r, c) may start when cells (r - 1, <= c + 1) are done
    // (the discs of an accepted point reach the 8 neighbouring cells only) -- a per-row progress counter in LDS, no
    // work-group barrier: a cell that needs the full scan delays its dependants only.  Bit-identical to raster order.
    for (int r = wave; r < P.nhcells; r += nwaves) {
        for (int c = 0; c < P.nwcells; c++) {
            // what does not depend on the neighbours is read BEFORE the wait: the cell's candidates (wave-uniform) and its occupancy
            const int cell = r * P.nwcells + c;
            const int q_p1 = s_cand[4 * cell], q_v1 = s_cand[4 * cell + 1], q_p2 = s_cand[4 * cell + 2], q_v2 = s_cand[4 * cell + 3];
            const int occd = occ[r * (P.nwcells + 1) + c];
            if (r > 0) {
                const int need = min(c + 2, P.nwcells);
                while (__hip_atomic_load(&progress[r - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            do {
            if (occd) break;
            const int x0 = c * P.cs, y0 = r * P.cs;
            if (!(x0 + P.cs < P.w - 1 && y0 + P.cs < P.h - 1)) break;    // :350 / :510
            const int c_p1 = __builtin_amdgcn_readfirstlane(q_p1), c_p2 = __builtin_amdgcn_readfirstlane(q_p2);
            const float c_v1 = __int_as_float(__builtin_amdgcn_readfirstlane(q_v1));
            const float c_v2 = __int_as_float(__builtin_amdgcn_readfirstlane(q_v2));
            const int p1x = c_p1 >= 0 ? (c_p1 & 0xFFFF) : 0, p1y = c_p1 >= 0 ? (c_p1 >> 16) : 0;
            const int p2x = c_p2 >= 0 ? (c_p2 & 0xFFFF) : 0, p2y = c_p2 >= 0 ? (c_p2 >> 16) : 0;
            // Full scan (a candidate was masked): lane ly owns row ly of the cell (cs <= 64): its slice of the exclusion
            // mask is ONE 64-bit string (bit j <-> pixel x0 + j), the response map is column-major so that the loads of a
            // column are coalesced.  Raster-order tie-breaking: strict `>` along the row, then the smaller pixel index in
            // the wave-wide arg-max.
            const int ly = lane;
            const bool row_ok = ly < P.cs;
            const int rowbase = ly * P.cs;
            auto row_mask = [&](int xfirst) -> unsigned long long {      // 64 mask bits starting at column xfirst, row y0 + ly
                if (!row_ok) return 0ull;
                const unsigned *mr = mask + (y0 + ly) * P.mask_words_per_row;
                const int w0 = xfirst >> 5, sh = xfirst & 31, last = P.mask_words_per_row - 1;
                const unsigned long long lo = (unsigned long long)mr[w0] | ((unsigned long long)mr[min(w0 + 1, last)] << 32);
                const unsigned long long hi = (unsigned long long)mr[min(w0 + 2, last)];
                return sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
            };
            // The row of responses goes to registers in ONE batch of loads per chunk of MAXROW columns.  Cells up to MAXROW
            // pixels wide -- every reference configuration -- need one chunk, loaded once for both arg-max passes; wider
            // cells (<= 64) take two chunks per pass.  Columns >= cs (and lanes >= cs) hold NaN: every comparison with
            // them is false, so the inner loops carry no per-pixel bounds predicates.
            float curv[MAXROW];
            constexpr bool single = CHUNKS == 1;
            bool have = false;
            const long long cbase = (long long)cell * npx + (row_ok ? ly : 0);
            const float qnan = __builtin_nanf("");
#define SEL_LOAD_CHUNK(JB)                                                                                       \
            _Pragma("unroll") for (int q = 0; q < MAXROW; q++) {                                                 \
                const int col = (JB) + q, qc = col < P.cs ? col : 0;                                             \
                const float v = MODE == 0 ? (float)nms_maps[cbase + qc * P.cs] : hmaps[cbase + qc * P.cs];       \
                curv[q] = (row_ok && col < P.cs) ? v : qnan;                                                     \
            }
            if (MODE == 0) {
                if (c_p1 < 0) break;                                      // no selectable corner whatever the mask says
                float bv = 0.f; int bi = 0x7FFFFFFF;
                // AS_EXECUTED: the CV_32F ones-mask read as bytes -- byte (lx & 3) of float lx >> 2 (N3)
                const int mcol = P.mask_mode == OV2_MASK_AS_EXECUTED ? (p1x >> 2) : p1x;
                int mx_ = p1x, my_ = p1y;
                // fast_tie: the candidate (first of the best responses in scan order) is what std::sort puts in front unless the cell
                // keeps more than 16 corners AND several share the best response -- the counts of the untouched mask bound both
                const bool tie_mode = P.fast_tie == OV2_FAST_TIE_LIBSTDCXX;
                const bool hard = tie_mode && c_p2 >= 0 && (c_p2 & 0xFFFF) > 16 && (c_p2 >> 16) >= 2;
                if (!hard && mask_test(mask, P.mask_words_per_row, x0 + mcol, y0 + p1y)) bv = c_v1;
                else {
                    // best response among the mask-surviving FAST corners, raster order on ties
                    if (lane == 0) atomicAdd(&s_nslow, 1);
                    const unsigned long long mb = row_mask(x0);
                    if (single) { SEL_LOAD_CHUNK(0) }
#pragma unroll
                    for (int ch = 0; ch < CHUNKS; ch++) {
                        const int jb = ch * MAXROW;
                        if (!single) { SEL_LOAD_CHUNK(jb) }
#pragma unroll
                        for (int q = 0; q < MAXROW; q++) {
                            const int lx = jb + q;
                            const unsigned long long bit = P.mask_mode == OV2_MASK_AS_EXECUTED ? (((lx & 3) >= 2) ? (mb >> (lx >> 2)) : 0ull) : (mb >> (lx & 63));
                            if ((bit & 1ull) && curv[q] > bv) { bv = curv[q]; bi = rowbase + lx; }   // NaN (padding) compares false
                        }
                    }
                    wave_argmax_f(bv, bi);
                    my_ = bi / P.cs; mx_ = bi - my_ * P.cs;
                    if (tie_mode && bv > 0.f) {
                        // the cell's corners that pass the mask, and how many of them share the best response
                        int ns = 0, nt = 0;
#pragma unroll
                        for (int ch = 0; ch < CHUNKS; ch++) {
                            const int jb = ch * MAXROW;
                            if (!single) { SEL_LOAD_CHUNK(jb) }
#pragma unroll
                            for (int q = 0; q < MAXROW; q++) {
                                const int lx = jb + q;
                                const unsigned long long bit = P.mask_mode == OV2_MASK_AS_EXECUTED ? (((lx & 3) >= 2) ? (mb >> (lx >> 2)) : 0ull) : (mb >> (lx & 63));
                                if ((bit & 1ull) && curv[q] > 0.f) { ns++; nt += curv[q] == bv; }
                            }
                        }
                        int off = ns;                                         // inclusive prefix over the rows (lane = row)
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(off, d, 64); if (lane >= d) off += t; }
                        const int total = __shfl(off, 63, 64);
                        int ntie = nt;
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1) ntie += __shfl_xor(ntie, d, 64);
                        if (total > 16 && ntie >= 2 && total <= DET_SORT_CAP) {
                            const int slot = wave % P.sort_slots;
                            int *srt = s_sort + slot * (DET_SORT_CAP + 72);
                            if (P.sort_slots < 16) {
                                if (lane == 0) while (atomicCAS(&s_sort_lock[slot], 0, 1) != 0) __builtin_amdgcn_s_sleep(1);
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); __builtin_amdgcn_wave_barrier();
                            }
                            int w_ = off - ns;                                // scan order: rows ascending, columns ascending inside a row
#pragma unroll
                            for (int ch = 0; ch < CHUNKS; ch++) {
                                const int jb = ch * MAXROW;
                                if (!single) { SEL_LOAD_CHUNK(jb) }
#pragma unroll
                                for (int q = 0; q < MAXROW; q++) {
                                    const int lx = jb + q;
                                    const unsigned long long bit = P.mask_mode == OV2_MASK_AS_EXECUTED ? (((lx & 3) >= 2) ? (mb >> (lx >> 2)) : 0ull) : (mb >> (lx & 63));
                                    if ((bit & 1ull) && curv[q] > 0.f) srt[w_++] = ((int)curv[q] << 16) | (ly << 8) | lx;
                                }
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            if (lane == 0) d_fk_std_sort(srt, total, srt + DET_SORT_CAP);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            const int win = srt[0];
                            mx_ = win & 0xFF; my_ = (win >> 8) & 0xFF;
                            if (P.sort_slots < 16) {
                                __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                if (lane == 0) __hip_atomic_store(&s_sort_lock[slot], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                }
                if (bv >= 20.f) {                                         // :521
                    const int px = x0 + mx_, py = y0 + my_;
                    if (lane == 0) prim[cell] = px | (py << 16);
                    fast_draw(px, py);                                    // :527
                }
            } else {
                // minMaxLoc of response * mask: first maximum, row-major.  Pixel 0 of the row initialises (lanes >= cs: no index)
                auto scan = [&](float &bv, int &bi) {
                    if (lane == 0 && !have) atomicAdd(&s_nslow, 1);
                    const unsigned long long mb = row_mask(x0);
                    bv = -FLT_MAX; bi = 0x7FFFFFFF;
                    if (single && !have) { SEL_LOAD_CHUNK(0) have = true; }
#pragma unroll
                    for (int ch = 0; ch < CHUNKS; ch++) {
                        const int jb = ch * MAXROW;
                        if (!single) { SEL_LOAD_CHUNK(jb) }
#pragma unroll
                        for (int q = 0; q < MAXROW; q++) {
                            const int lx = jb + q;
                            const float v = curv[q] * (((mb >> (lx & 63)) & 1ull) ? 1.f : 0.f);     // NaN (padding) compares false
                            const bool take = lx == 0 ? row_ok : (v > bv);
                            if (take) { bv = v; bi = rowbase + lx; }
                        }
                    }
                    wave_argmax_f(bv, bi);
                };
                float bv; int bi;
                // both candidates are tested at once (p2 lies outside the disc of p1 by construction, so its test does not wait for
                // that disc): in the common case the cell costs two LDS bit tests and two disc draws, no intermediate fence
                const bool t1 = mask_test(mask, P.mask_words_per_row, x0 + p1x, y0 + p1y) != 0;
                const bool t2 = mask_test(mask, P.mask_words_per_row, x0 + p2x, y0 + p2y) != 0;
                // pass 0
                const bool f0 = c_v1 > 0.f && t1;
                int mx_ = p1x, my_ = p1y;
                if (f0) bv = c_v1; else { scan(bv, bi); my_ = bi / P.cs; mx_ = bi - my_ * P.cs; }
                int mx = x0 + mx_, my = y0 + my_;
                if (mx < P.roi_x || my < P.roi_y || mx >= P.roi_x + P.roi_w || my >= P.roi_y + P.roi_h) break;   // `continue` at :363-368
                if (!((double)bv >= P.quality)) break;                    // nothing accepted: the second minMaxLoc sees the same mask and value
                if (lane == 0) prim[cell] = mx | (my << 16);
                fast_draw(mx, my);
                // pass 1: the candidate outside the disc just drawn, if the first pass took its candidate
                const bool f1 = f0 && c_p2 >= 0 && c_v2 > 0.f && t2;
                if (f1) { bv = c_v2; mx_ = p2x; my_ = p2y; }
                else {
                    // make the disc visible to this wave's full scan
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    scan(bv, bi); my_ = bi / P.cs; mx_ = bi - my_ * P.cs;
                }
                mx = x0 + mx_; my = y0 + my_;
                if (mx < P.roi_x || my < P.roi_y || mx >= P.roi_x + P.roi_w || my >= P.roi_y + P.roi_h) break;   // :379-384
                if ((double)bv >= P.quality) {
                    if (lane == 0) sec[cell] = mx | (my << 16);
                    fast_draw(mx, my);
                }
            }
#undef SEL_LOAD_CHUNK
            } while (0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(&progress[r], c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();

    const unsigned long long tk3 = wall_clock64();
    // compaction in cell order by wavefront 0: 64 cells per trip, positions from ballot prefix counts
    if (wave == 0) {
        int n = 0, nboccup = 0;
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int base = 0; base < ncells; base += 64) {
            const int i = base + lane;
            const bool in = i < ncells;
            const int r = in ? i / P.nwcells : 0, c = in ? i - r * P.nwcells : 0;
            const bool oc = in && occ[r * (P.nwcells + 1) + c];
            nboccup += __popcll(__builtin_amdgcn_ballot_w64(oc));
            const int pv = in ? prim[i] : -1;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(pv >= 0);
            if (pv >= 0) out_xy[n + __popcll(m & lt)] = make_float2((float)(pv & 0xFFFF), (float)(pv >> 16));
            n += __popcll(m);
        }
        const int nbprim = n, nbempty = ncells - nboccup;
        if (MODE == 1 && nbprim + nboccup < ncells) {                    // :400-414: the first nbsec secondaries in cell order
            const int nbsec = ncells - (nbprim + nboccup);
            int k = 0;
            for (int base = 0; base < ncells && k < nbsec; base += 64) {
                const int i = base + lane;
                const int sv = i < ncells ? sec[i] : -1;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(sv >= 0);
                const int pos = k + __popcll(m & lt);
                if (sv >= 0 && pos < nbsec) out_xy[nbprim + pos] = make_float2((float)(sv & 0xFFFF), (float)(sv >> 16));
                k += __popcll(m);
            }
            n = nbprim + (k < nbsec ? k : nbsec);
        }
      if (lane == 0) {
        out->n = n; out->nboccup = nboccup; out->nbempty = nbempty; out->nbkps = nbprim; out->nslow = s_nslow; out->pad_ = 0;
        out->dbg[0] = tk1 - tk0; out->dbg[1] = tk2 - tk1; out->dbg[2] = tk3 - tk2; out->dbg[3] = wall_clock64() - tk3;
      }
    }
}

// ---------------------------------------------------------------------------------
// cv::cornerSubPix (imgproc/src/cornersubpix.cpp), one lane per point
// ---------------------------------------------------------------------------------
#define SP_MAX_HALF 5
struct SubpixParams {
    long long img_item_stride;          // batched launches (blockIdx.y = item): bytes between images,
    int xy_item_stride, ndev_item_stride;   // float2 entries between point lists, ints between the device-side counts
    int w, h, stride, n, half_win, max_iters;
    double eps2;
    float e[2 * SP_MAX_HALF + 1];       // exp(-((i-hw)/hw)^2) computed by the host libm (like the reference)
};

// getRectSubPix(u8 -> f32), patch pw x ph around (cx, cy)  (imgproc/src/samplers.cpp)
template <int PW>
__device__ void d_get_rect_subpix(const uint8_t *__restrict__ src, int src_step, int sw, int sh,
                                  float *dst, float cx_f, float cy_f)
{
    const int pw = PW, ph = PW;
    const double cxd = (double)cx_f - (pw - 1) * 0.5, cyd = (double)cy_f - (ph - 1) * 0.5;
    int ipx = (int)floor(cxd), ipy = (int)floor(cyd);
    if (0 <= ipx && ipx + pw < sw && 0 <= ipy && ipy + ph < sh) {
        float a = (float)(cxd - ipx), b = (float)(cyd - ipy);
        a = a > 0.0001f ? a : 0.0001f;
        const float a12 = a * (1.f - b), a22 = a * b, b1 = 1.f - b, b2 = b;
        const double s = (1. - (double)a) / (double)a;
        const uint8_t *p = src + (long long)ipy * src_step + ipx;
        for (int i = 0; i < ph; i++, p += src_step, dst += pw) {
            float prev = (1 - a) * (b1 * p[0] + b2 * p[src_step]);
            for (int j = 0; j < pw; j++) {
                const float t = a12 * p[j + 1] + a22 * p[j + 1 + src_step];
                dst[j] = prev + t;
                prev = (float)(t * s);
            }
        }
        return;
    }
    const float cx = cx_f - (pw - 1) * 0.5f, cy = cy_f - (ph - 1) * 0.5f;
    ipx = (int)floorf(cx); ipy = (int)floorf(cy);
    const float a = cx - ipx, b = cy - ipy;
    const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
    const float b1 = 1.f - b, b2 = b;
    if (0 <= ipx && ipx < sw - pw && 0 <= ipy && ipy < sh - ph) {
        const uint8_t *p = src + (long long)ipy * src_step + ipx;
        for (int i = 0; i < ph; i++, p += src_step, dst += pw)
            for (int j = 0; j < pw; j++)
                dst[j] = p[j] * a11 + p[j + 1] * a12 + p[j + src_step] * a21 + p[j + src_step + 1] * a22;
        return;
    }
    int rx, ry, rw, rh;
    const uint8_t *p = src;
    if (ipx >= 0) { p += ipx; rx = 0; }
    else { rx = -ipx; if (rx > pw) rx = pw; }
    if (ipx < sw - pw) rw = pw;
    else { rw = sw - ipx - 1; if (rw < 0) { p += rw; rw = 0; } }
    if (ipy >= 0) { p += (long long)ipy * src_step; ry = 0; }
    else ry = -ipy;
    if (ipy < sh - ph) rh = ph;
    else { rh = sh - ipy - 1; if (rh < 0) { p += (long long)rh * src_step; rh = 0; } }
    p -= rx;
    for (int i = 0; i < ph; i++, dst += pw) {
        const uint8_t *p2 = p + src_step;
        if (i < ry || i >= rh) p2 -= src_step;
        float s0 = p[rx] * b1 + p2[rx] * b2;
        for (int j = 0; j < rx; j++) dst[j] = s0;
        for (int j = rx; j < rw; j++) dst[j] = p[j] * a11 + p[j + 1] * a12 + p2[j] * a21 + p2[j + 1] * a22;
        s0 = p[rw] * b1 + p2[rw] * b2;
        for (int j = rw; j < pw; j++) dst[j] = s0;
        if (i < rh) p = p2;
    }
}

// one pixel (i, j) of getRectSubPix(u8 -> f32), patch PW x PW around (cx, cy): the same three code paths as
// d_get_rect_subpix, evaluated per pixel (the fast path's `prev` recurrence only couples neighbouring columns:
// dst[j] = (j ? float(t[j-1] * s) : (1-a)(b1 p[0] + b2 p'[0])) + t[j],  t[j] = a12 p[j+1] + a22 p'[j+1])
// What of getRectSubPix depends on the patch centre only: computed once per Gauss-Newton trip, not once per pixel (the fp64 division
// of the fast path's `s` alone is ~40 instructions)
struct RectCtx {
    int fast, inner, ipx, ipy;
    float a12, a22, b1, b2, a1m, a11, a21;
    double s;
};
template <int PW>
__device__ __forceinline__ RectCtx d_rect_ctx(int sw, int sh, float cx_f, float cy_f)
{
    RectCtx c;
    const int pw = PW, ph = PW;
    const double cxd = (double)cx_f - (pw - 1) * 0.5, cyd = (double)cy_f - (ph - 1) * 0.5;
    int ipx = (int)floor(cxd), ipy = (int)floor(cyd);
    c.fast = (0 <= ipx && ipx + pw < sw && 0 <= ipy && ipy + ph < sh) ? 1 : 0;
    c.inner = 0; c.a11 = c.a21 = 0.f; c.s = 0; c.a1m = 0.f;
    if (c.fast) {
        float a = (float)(cxd - ipx), b = (float)(cyd - ipy);
        a = a > 0.0001f ? a : 0.0001f;
        c.a12 = a * (1.f - b); c.a22 = a * b; c.b1 = 1.f - b; c.b2 = b;
        c.s = (1. - (double)a) / (double)a;
        c.a1m = 1 - a;
        c.ipx = ipx; c.ipy = ipy;
        return c;
    }
    const float cx = cx_f - (pw - 1) * 0.5f, cy = cy_f - (ph - 1) * 0.5f;
    ipx = (int)floorf(cx); ipy = (int)floorf(cy);
    const float a = cx - ipx, b = cy - ipy;
    c.a11 = (1.f - a) * (1.f - b); c.a12 = a * (1.f - b); c.a21 = (1.f - a) * b; c.a22 = a * b;
    c.b1 = 1.f - b; c.b2 = b;
    c.inner = (0 <= ipx && ipx < sw - pw && 0 <= ipy && ipy < sh - ph) ? 1 : 0;
    c.ipx = ipx; c.ipy = ipy;
    return c;
}
// d_get_rect_subpix, evaluated per pixel (the fast path's `prev` recurrence only couples neighbouring columns:
// dst[j] = (j ? float(t[j-1] * s) : (1-a)(b1 p[0] + b2 p'[0])) + t[j],  t[j] = a12 p[j+1] + a22 p'[j+1])
__device__ __forceinline__ float d_rect_px(const RectCtx &c, const uint8_t *__restrict__ src, int src_step, int sw, int sh, int i, int j)
{
    if (c.fast) {
        const uint8_t *p = src + (long long)(c.ipy + i) * src_step + c.ipx;
        const float t = c.a12 * p[j + 1] + c.a22 * p[j + 1 + src_step];
        float prev;
        if (j == 0) prev = c.a1m * (c.b1 * p[0] + c.b2 * p[src_step]);
        else { const float tp = c.a12 * p[j] + c.a22 * p[j + src_step]; prev = (float)(tp * c.s); }
        return prev + t;
    }
    // rows / columns of adjustRect == replicated border; columns left of the image or without a right neighbour
    // use the vertical-only weights
    int y0 = c.ipy + i, y1 = y0 + 1;
    y0 = min(max(y0, 0), sh - 1); y1 = min(max(y1, 0), sh - 1);
    const uint8_t *r0 = src + (long long)y0 * src_step, *r1 = src + (long long)y1 * src_step;
    const int x = c.ipx + j;
    if (c.inner) return r0[x] * c.a11 + r0[x + 1] * c.a12 + r1[x] * c.a21 + r1[x + 1] * c.a22;
    if (x < 0) return r0[0] * c.b1 + r1[0] * c.b2;
    if (x >= sw - 1) return r0[sw - 1] * c.b1 + r1[sw - 1] * c.b2;
    return r0[x] * c.a11 + r0[x + 1] * c.a12 + r1[x] * c.a21 + r1[x + 1] * c.a22;
}
template <int PW>
__device__ __forceinline__ float d_rect_subpix_px(const uint8_t *__restrict__ src, int src_step, int sw, int sh, float cx_f, float cy_f, int i, int j)
{
    const RectCtx c = d_rect_ctx<PW>(sw, sh, cx_f, cy_f);
    return d_rect_px(c, src, src_step, sw, sh, i, j);
}

// One WAVEFRONT per point: the (WINW+2)^2 patch and the per-pixel gradient products are evaluated by all lanes,
// the five sums are accumulated by lanes 0..4 (one sum each) in the reference's raster order (double), so the result is
// bit-identical to the serial loop while an iteration costs ~1 us instead of ~10.
template <int HALF>
__global__ __launch_bounds__(64) void k_corner_subpix(SubpixParams P, const uint8_t *__restrict__ img, float2 *__restrict__ xy,
                                                      const int *__restrict__ n_dev)
{
    constexpr int WINW = 2 * HALF + 1, SW = WINW + 2, NPIX = WINW * WINW;
    __shared__ float sub[SW * SW];
    __shared__ double prod[NPIX][5];
    const int pt = blockIdx.x, lane = threadIdx.x;
    img += (long long)blockIdx.y * P.img_item_stride; xy += (long long)blockIdx.y * P.xy_item_stride;
    // n_dev: the point count still lives on the device (written by k_grid_select): the launch covers the capacity and the
    // surplus wavefronts leave here -- the detectors then need ONE host synchronisation instead of two
    if (pt >= (n_dev ? n_dev[(long long)blockIdx.y * P.ndev_item_stride] : P.n)) return;
    const float2 cT = xy[pt];
    float cIx = cT.x, cIy = cT.y;
    int iter = 0;
    bool go = true;
    // The patch's source pixels stay in REGISTERS between iterations: a lane's patch pixels (one per pass of 64) need the bytes
    // p[j], p[j+1] of two rows, and those change only when the patch's integer origin moves -- after the first step or two of a
    // converging point it does not, so most iterations issue no global load at all.  In a batch the kernel is bound by its fp64 issue
    // slots and this changes nothing (profiles/r6_detect_notes.txt); for ONE image (or the eleven of a lock-step step) a wavefront runs
    // alone on its SIMD and the two load round trips per iteration were the longest link of its dependent chain.  Same float operations
    // in the same order as d_rect_px.
    constexpr int NPASS = (SW * SW + 63) / 64;
    float q00[NPASS], q01[NPASS], q10[NPASS], q11[NPASS];
    int pj[NPASS], poff[NPASS];
#pragma unroll
    for (int k = 0; k < NPASS; k++) {
        const int e = min(lane + 64 * k, SW * SW - 1), i = e / SW;
        pj[k] = e - i * SW; poff[k] = i * P.stride + pj[k];
        q00[k] = q01[k] = q10[k] = q11[k] = 0.f;
    }
    int c_ipx = INT_MIN, c_ipy = INT_MIN;
    while (go) {
        const RectCtx rc = d_rect_ctx<SW>(P.w, P.h, cIx, cIy);
        if (__builtin_amdgcn_readfirstlane(rc.fast)) {
            const int ipx = __builtin_amdgcn_readfirstlane(rc.ipx), ipy = __builtin_amdgcn_readfirstlane(rc.ipy);
            if (ipx != c_ipx || ipy != c_ipy) {
                const uint8_t *p0 = img + (long long)ipy * P.stride + ipx;
#pragma unroll
                for (int k = 0; k < NPASS; k++) {
                    const uint8_t *p = p0 + poff[k];
                    q00[k] = (float)p[0]; q01[k] = (float)p[1]; q10[k] = (float)p[P.stride]; q11[k] = (float)p[P.stride + 1];
                }
                c_ipx = ipx; c_ipy = ipy;
            }
#pragma unroll
            for (int k = 0; k < NPASS; k++) {
                const float t = rc.a12 * q01[k] + rc.a22 * q11[k];
                float prev;
                if (pj[k] == 0) prev = rc.a1m * (rc.b1 * q00[k] + rc.b2 * q10[k]);
                else { const float tp = rc.a12 * q00[k] + rc.a22 * q10[k]; prev = (float)(tp * rc.s); }
                if (lane + 64 * k < SW * SW) sub[lane + 64 * k] = prev + t;
            }
        } else {
            c_ipx = INT_MIN;                                              // (the border paths read through d_rect_px every time)
            for (int e = lane; e < SW * SW; e += 64) { const int i = e / SW, j = e - i * SW; sub[e] = d_rect_px(rc, img, P.stride, P.w, P.h, i, j); }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int e = lane; e < NPIX; e += 64) {
            const int i = e / WINW, j = e - i * WINW;
            const float *sp = sub + (i + 1) * SW + 1;
            const double py = i - HALF, px = j - HALF;
            const double m = (double)(float)(P.e[i] * P.e[j]);
            const double tgx = sp[j + 1] - sp[j - 1];
            const double tgy = sp[j + SW] - sp[j - SW];
            const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
            prod[e][0] = gxx; prod[e][1] = gxy; prod[e][2] = gyy;
            prod[e][3] = gxx * px + gxy * py;
            prod[e][4] = gxy * px + gyy * py;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float nx = cIx, ny = cIy;
        int cont = 0;
        // the five sums in the reference's raster order, one lane each (a wave64 fp64 add occupies the SIMD for 8 cycles
        // whether one lane is active or five: 49 dependent adds instead of 245)
        double acc = 0;
        // (the reads of a window row are issued together, then added in order: one LDS round trip per row instead of one per term)
        if (lane < 5) {
#pragma unroll
            for (int r = 0; r < WINW; r++) {
                double v[WINW];
#pragma unroll
                for (int k = 0; k < WINW; k++) v[k] = prod[r * WINW + k][lane];
#pragma unroll
                for (int k = 0; k < WINW; k++) acc += v[k];
            }
        }
        const double a = __shfl(acc, 0, 64), b = __shfl(acc, 1, 64), c = __shfl(acc, 2, 64), bb1 = __shfl(acc, 3, 64), bb2 = __shfl(acc, 4, 64);
        if (lane == 0) {
            const double det = a * c - b * b;
            if (!(fabs(det) <= DBL_EPSILON * DBL_EPSILON)) {
                const double scale = 1.0 / det;
                const float c2x = (float)(cIx + c * scale * bb1 - b * scale * bb2);
                const float c2y = (float)(cIy - b * scale * bb1 + a * scale * bb2);
                const double err = (double)((c2x - cIx) * (c2x - cIx) + (c2y - cIy) * (c2y - cIy));
                nx = c2x; ny = c2y;
                const bool outside = nx < 0 || nx >= P.w || ny < 0 || ny >= P.h;
                cont = (!outside && ++iter < P.max_iters && err > P.eps2) ? 1 : 0;
            }
        }
        cIx = __shfl(nx, 0, 64); cIy = __shfl(ny, 0, 64);
        go = __shfl(cont, 0, 64) != 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (fabs((double)(cIx - cT.x)) > HALF || fabs((double)(cIy - cT.y)) > HALF) { cIx = cT.x; cIy = cT.y; }
    if (lane == 0) xy[pt] = make_float2(cIx, cIy);
}


// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
static int launch_subpix(ov2_ctx *ctx, const uint8_t *img_d, int w, int h, int stride, float2 *xy_d, int n,
                         int half_win, int max_iter, double eps, const int *n_dev = nullptr,
                         int items = 1, long long img_item_stride = 0, int xy_item_stride = 0, int ndev_item_stride = 0)
{
    if (n <= 0) return OV2_OK;
    OV2_REQUIRE(half_win >= 1 && half_win <= SP_MAX_HALF, OV2_EUNSUPPORTED, "cornerSubPix half window must be in [1,5]");
    OV2_REQUIRE(w >= half_win * 2 + 5 && h >= half_win * 2 + 5, OV2_EINVAL, "image too small for cornerSubPix");
    SubpixParams P;
    P.w = w; P.h = h; P.stride = stride; P.n = n; P.half_win = half_win;
    P.img_item_stride = img_item_stride; P.xy_item_stride = xy_item_stride; P.ndev_item_stride = ndev_item_stride;
    P.max_iters = max_iter < 1 ? 1 : (max_iter > 100 ? 100 : max_iter);
    if (eps < 0.) eps = 0.;
    P.eps2 = eps * eps;
    for (int i = 0; i < 2 * half_win + 1; i++) {
        const float x = (float)(i - half_win) / half_win;
        P.e[i] = expf(-x * x);
    }
    dim3 grid(n, items), block(64);                           // one wavefront per point (and batch item)
    switch (half_win) {
    case 1: hipLaunchKernelGGL(k_corner_subpix<1>, grid, block, 0, ctx->stream, P, img_d, xy_d, n_dev); break;
    case 2: hipLaunchKernelGGL(k_corner_subpix<2>, grid, block, 0, ctx->stream, P, img_d, xy_d, n_dev); break;
    case 3: hipLaunchKernelGGL(k_corner_subpix<3>, grid, block, 0, ctx->stream, P, img_d, xy_d, n_dev); break;
    case 4: hipLaunchKernelGGL(k_corner_subpix<4>, grid, block, 0, ctx->stream, P, img_d, xy_d, n_dev); break;
    default: hipLaunchKernelGGL(k_corner_subpix<5>, grid, block, 0, ctx->stream, P, img_d, xy_d, n_dev); break;
    }
    OV2_HIP_CHECK(hipGetLastError());
    return OV2_OK;
}

// The dynamic-LDS limit is a per-function, process-wide attribute: raise it ONCE for every instance to the hardware maximum
// (less the kernel's static LDS) instead of per call to that call's need -- two contexts on two threads would race on it.
static hipError_t det_raise_lds_limits()
{
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    std::call_once(once, [] {
        auto raise = [](const void *fn) {
            hipFuncAttributes fa;
            hipError_t e = hipFuncGetAttributes(&fa, fn);
            if (e != hipSuccess) return e;
            return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)fa.sharedSizeBytes);
        };
        const void *fns[] = {(const void *)k_mineig_cells, (const void *)k_mineig_strip,
                             (const void *)k_grid_select<0, 36, 1, 1024>, (const void *)k_grid_select<0, 52, 1, 1024>, (const void *)k_grid_select<0, 32, 2, 1024>,
                             (const void *)k_grid_select<1, 36, 1, 1024>, (const void *)k_grid_select<1, 52, 1, 1024>, (const void *)k_grid_select<1, 32, 2, 1024>,
                             (const void *)k_grid_select<0, 36, 1, DET_SEL_THREADS_BATCH>, (const void *)k_grid_select<0, 52, 1, DET_SEL_THREADS_BATCH>,
                             (const void *)k_grid_select<0, 32, 2, DET_SEL_THREADS_BATCH>, (const void *)k_grid_select<1, 36, 1, DET_SEL_THREADS_BATCH>,
                             (const void *)k_grid_select<1, 52, 1, DET_SEL_THREADS_BATCH>, (const void *)k_grid_select<1, 32, 2, DET_SEL_THREADS_BATCH>};
        for (const void *fn : fns) if (err == hipSuccess) err = raise(fn);
    });
    return err;
}

// Dynamic LDS of k_grid_select: exclusion mask + cell tables (+ the sort areas of the libstdc++ tie order, FAST only).
// *slots = sort areas that fit (<= 16); returns 0 when the image does not fit at all.
static size_t det_select_lds(int w, int h, int cell, int mode, int fast_tie, int *slots)
{
    const int nw = w / cell, nh = h / cell, ncells = nw * nh, wpr = (w + 31) / 32;
    const size_t base = (size_t)h * wpr * 4 + 64 * 4 + (size_t)ncells * 16 + (size_t)nh * 4 + (((size_t)(nh + 1) * (nw + 1) + 3) & ~(size_t)3) + (size_t)ncells * 8;
    const size_t limit = 160 * 1024 - 256;                           // less the kernel's static LDS (locks, counters)
    *slots = 0;
    if (base > limit) return 0;
    if (mode != 0 || fast_tie != OV2_FAST_TIE_LIBSTDCXX) return base;
    const size_t area = (size_t)(DET_SORT_CAP + 72) * 4;
    const size_t fit = (limit - base) / area;
    if (fit == 0) return 0;
    *slots = fit > 16 ? 16 : (int)fit;
    return base + (size_t)*slots * area;
}

// The three launches of a detection (response + candidates per cell, selection, sub-pixel refinement) for `items` images
// that lie `B.img_stride` bytes apart; every pointer addresses item 0.  Asynchronous on the context's stream.
static int enqueue_detect(ov2_ctx *ctx, int mode, const uint8_t *im, int w, int h, int im_stride, int cell, int items, DetBatch B,
                          int fast_th, int mask_mode, const int roi[4], double quality, int ncur, const float2 *cur_d,
                          uint8_t *maps_d, CellCand *cand_d, float2 *out_d, SelectOut *so_d, int do_subpix)
{
    const int nw = w / cell, nh = h / cell, ncells = nw * nh, npx = cell * cell, wpr = (w + 31) / 32;
    int sort_slots = 0;
    const size_t sel_lds = det_select_lds(w, h, cell, mode, ctx->det_fast_tie, &sort_slots);
    B.ncells = ncells; B.cur = cur_d; B.ncur_all = ncur;
    if (mode == 0) {
        int th = fast_th < 0 ? 0 : (fast_th > 255 ? 255 : fast_th);
        hipLaunchKernelGGL(k_fast_cells, dim3(ncells * items), dim3(256), 0, ctx->stream, im, w, h, im_stride, cell, nw, th, maps_d, mask_mode, cand_d, B);
    } else if (ctx->det_strip == 1 || (ctx->det_strip < 0 && (long long)ncells * items >= 2048)) {
        // batches: the free cells of an image side by side, several per work-group (k_mineig_strip); the cells per group / wavefronts
        // per group that leave the fewest lanes idle: a wavefront owns at most 60 of its 64 columns
        int best_n = 1, best_k = (cell + 59) / 60;
        double best_e = (double)cell / (64.0 * best_k);
        for (int nc = 2; nc <= STRIP_MAX_CELLS; nc++) {
            const int k = (nc * cell + 59) / 60;
            if (k > 8) break;
            const double e = (double)nc * cell / (64.0 * k);
            if (e > best_e + 1e-9) { best_e = e; best_n = nc; best_k = k; }
        }
        const int owned = (best_n * cell + best_k - 1) / best_k;
        const size_t lds = (((size_t)cell * 64 * best_k + 15) & ~(size_t)15) + (size_t)best_k * cell * 65 * 4;
        int *free_d = (int *)(cand_d + (size_t)items * ncells);          // (the callers reserve it behind the candidates)
        hipLaunchKernelGGL(k_free_cells, dim3(items), dim3(64), (size_t)(ncells + 31) / 32 * 4, ctx->stream, cell, nw, B, free_d);
        hipLaunchKernelGGL(k_mineig_strip, dim3((ncells + best_n - 1) / best_n, items), dim3(64 * best_k), lds, ctx->stream, im, w, h, im_stride, cell, nw,
                           (float *)maps_d, ctx->sobel_dy_order, cell / 4, cand_d, B, best_n, owned, (const int *)free_d);
    } else {
        const size_t lds = (size_t)npx * 5 + 16;                      // lambda_min (float) + blurred cell (byte) per pixel
        hipLaunchKernelGGL(k_mineig_cells, dim3(ncells * items), dim3(64), lds, ctx->stream, im, w, h, im_stride, cell, nw, (float *)maps_d, ctx->sobel_dy_order, cell / 4, cand_d, B);
    }
    SelectParams P;
    P.w = w; P.h = h; P.cs = cell; P.nwcells = nw; P.nhcells = nh; P.radius = cell / 4; P.mask_words_per_row = wpr;
    P.mode = mode; P.mask_mode = mask_mode; P.ncur = ncur;
    P.fast_tie = ctx->det_fast_tie; P.sort_slots = sort_slots > 0 ? sort_slots : 1;
    P.roi_x = roi ? roi[0] : 0; P.roi_y = roi ? roi[1] : 0; P.roi_w = roi ? roi[2] : w; P.roi_h = roi ? roi[3] : h;
    P.quality = quality;
    // One image: sixteen wavefronts, a row of cells each (latency).  Batches: EIGHT (rows w, w + 8 -- the anti-diagonal dependence keeps
    // at most nwcells / 2 rows busy at a time anyway), so that two work-groups share a CU's sixteen wavefront slots (106 VGPRs: four
    // per SIMD) instead of one
    const bool sel_batch = items >= 64;
#define OV2_LAUNCH_SELECT(MD, MR, CH)                                                                                               \
    do { if (sel_batch) hipLaunchKernelGGL((k_grid_select<MD, MR, CH, DET_SEL_THREADS_BATCH>), dim3(items), dim3(DET_SEL_THREADS_BATCH), sel_lds, ctx->stream, P, cur_d, \
                       (const uint8_t *)maps_d, (const float *)maps_d, (const CellCand *)cand_d, out_d, so_d, B);                   \
         else hipLaunchKernelGGL((k_grid_select<MD, MR, CH, 1024>), dim3(items), dim3(1024), sel_lds, ctx->stream, P, cur_d, (const uint8_t *)maps_d,     \
                       (const float *)maps_d, (const CellCand *)cand_d, out_d, so_d, B); } while (0)
    if (mode == 0) { if (cell <= 36) OV2_LAUNCH_SELECT(0, 36, 1); else if (cell <= 52) OV2_LAUNCH_SELECT(0, 52, 1); else OV2_LAUNCH_SELECT(0, 32, 2); }
    else { if (cell <= 36) OV2_LAUNCH_SELECT(1, 36, 1); else if (cell <= 52) OV2_LAUNCH_SELECT(1, 52, 1); else OV2_LAUNCH_SELECT(1, 32, 2); }
#undef OV2_LAUNCH_SELECT
    OV2_HIP_CHECK(hipGetLastError());
    // sub-pixel refinement over the CAPACITY of the output list (the kernel reads the count on the device and the
    // surplus wavefronts leave at once)
    if (do_subpix) {
        const int cap = mode == 0 ? ncells : 2 * ncells;
        const int rc = launch_subpix(ctx, im, w, h, im_stride, out_d, cap, 3, 30, 0.01, (const int *)so_d,
                                     items, B.img_stride, B.out_stride, (int)(sizeof(SelectOut) / sizeof(int)));
        if (rc) return rc;
    }
    return OV2_OK;
}

// img_h: host image (uploaded first) -- or img_d: an image already in HBM (pyramid level 0: no upload), row pitch `stride`
static int detect_common(ov2_ctx *ctx, int mode, const uint8_t *img_h, const uint8_t *img_d, int w, int h, int stride, int cell,
                         const float *cur_xy_h, int ncur, int fast_th, int mask_mode, const int roi[4], double quality,
                         int do_subpix, float *out_xy_h, int *out_n, SelectOut *so_h)
{
    OV2_REQUIRE(ctx && out_xy_h && out_n, OV2_EINVAL, "NULL argument");
    *out_n = 0;
    memset(so_h, 0, sizeof(*so_h));
    if ((!img_h && !img_d) || w <= 0 || h <= 0) return OV2_OK;            // empty image -> empty vector (:291-294 / :446-449)
    OV2_REQUIRE(stride >= w, OV2_EINVAL, "stride < width");
    OV2_REQUIRE(cell >= 8 && cell <= DET_MAX_CELL, OV2_EUNSUPPORTED, "cell size must be in [8,64]");
    OV2_REQUIRE(ncur >= 0 && (ncur == 0 || cur_xy_h), OV2_EINVAL, "bad current keypoints");
    OV2_REQUIRE(w < 65536 && h < 32768, OV2_EUNSUPPORTED, "image too large");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    OV2_HIP_CHECK(det_raise_lds_limits());
    const int nw = w / cell, nh = h / cell, ncells = nw * nh;
    if (ncells == 0) return OV2_OK;
    const int npx = cell * cell;
    { int slots_; OV2_REQUIRE(det_select_lds(w, h, cell, mode, ctx->det_fast_tie, &slots_) != 0, OV2_EUNSUPPORTED, "image too large for the LDS-resident exclusion mask"); }

    // device scratch: [img w*h][maps][cur 8*ncur][out 8*2*ncells][SelectOut]
    const size_t o_img = 0;
    const size_t map_bytes = (size_t)ncells * npx * (mode == 0 ? 1 : 4);
    const size_t o_map = img_d ? 0 : (((size_t)w * h + 255) & ~(size_t)255);
    const size_t o_cur = (o_map + map_bytes + 255) & ~(size_t)255;
    const size_t o_out = (o_cur + 8 * (size_t)(ncur > 0 ? ncur : 1) + 255) & ~(size_t)255;
    const size_t o_so = o_out + 16 * (size_t)ncells;
    const size_t o_cand = (o_so + sizeof(SelectOut) + 255) & ~(size_t)255;
    const size_t total = o_cand + sizeof(CellCand) * (size_t)ncells + 4 * ((size_t)ncells + 1);        // candidates, then the free-cell list (k_free_cells)
    int rc = ctx->reserve_device(total);  if (rc) return rc;
    rc = ctx->reserve_host(16 * (size_t)ncells + sizeof(SelectOut)); if (rc) return rc;
    uint8_t *ds = (uint8_t *)ctx->d_scratch;
    const uint8_t *im = img_d;                                  // what the kernels read, and its pitch
    int im_stride = stride;
    if (!img_d) {
        rc = ctx->upload_image(ds + o_img, (size_t)w, img_h, (size_t)stride, (size_t)w, (size_t)h);
        if (rc) return rc;
        im = ds + o_img; im_stride = w;
    }
    if (ncur > 0) OV2_HIP_CHECK(hipMemcpyAsync(ds + o_cur, cur_xy_h, 8 * (size_t)ncur, hipMemcpyHostToDevice, ctx->stream));

    DetBatch B;
    memset(&B, 0, sizeof(B));
    rc = enqueue_detect(ctx, mode, im, w, h, im_stride, cell, 1, B, fast_th, mask_mode, roi, quality, ncur, (const float2 *)(ds + o_cur),
                        ds + o_map, (CellCand *)(ds + o_cand), (float2 *)(ds + o_out), (SelectOut *)(ds + o_so), do_subpix);
    if (rc) return rc;
    // ONE copy of (points, counters) and ONE synchronisation
    uint8_t *hs = (uint8_t *)ctx->h_scratch;
    OV2_HIP_CHECK(hipMemcpyAsync(hs, ds + o_out, 16 * (size_t)ncells + sizeof(SelectOut), hipMemcpyDeviceToHost, ctx->stream));
    OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    memcpy(so_h, hs + 16 * (size_t)ncells, sizeof(SelectOut));
    if (ctx->debug)
        fprintf(stderr, "[ov2 det] select ticks (100MHz): init %llu prologue %llu sweep %llu compaction %llu; cells on the full-scan path: %d\n",
                so_h->dbg[0], so_h->dbg[1], so_h->dbg[2], so_h->dbg[3], so_h->nslow);
    const int n = so_h->n;
    if (n > 0) memcpy(out_xy_h, hs, 8 * (size_t)n);
    *out_n = n;
    return OV2_OK;
}

extern "C" {

int ov2_detect_grid_fast(ov2_ctx *ctx, const uint8_t *img_h, int w, int h, int stride, int cell,
                         const float *cur_xy_h, int ncur, int *fast_th_inout, int mask_mode,
                         int do_subpix, float *out_xy_h, int *out_n)
{
    OV2_REQUIRE(fast_th_inout != nullptr, OV2_EINVAL, "fast_th_inout == NULL");
    OV2_REQUIRE(mask_mode == OV2_MASK_AS_EXECUTED || mask_mode == OV2_MASK_INTENDED, OV2_EINVAL, "bad mask_mode");
    SelectOut so;
    const int th = *fast_th_inout;
    const int rc = detect_common(ctx, 0, img_h, nullptr, w, h, stride, cell, cur_xy_h, ncur, th, mask_mode, nullptr, 0.0,
                                 do_subpix, out_xy_h, out_n, &so);
    if (rc != OV2_OK) return rc;
    if (!img_h || w <= 0 || h <= 0) return OV2_OK;
    // threshold adaptation (:546-552), int *= double truncates
    if ((double)so.nbkps < 0.5 * (double)so.nbempty && so.nbempty > 10) *fast_th_inout = (int)(th * 0.66);
    else if (so.nbkps == so.nbempty) *fast_th_inout = (int)(th * 1.5);
    return OV2_OK;
}

int ov2_detect_singlescale(ov2_ctx *ctx, const uint8_t *img_h, int w, int h, int stride, int cell,
                           const float *cur_xy_h, int ncur, const int roi[4], double *quality_inout,
                           int do_subpix, float *out_xy_h, int *out_n)
{
    OV2_REQUIRE(quality_inout != nullptr && roi != nullptr, OV2_EINVAL, "quality_inout / roi == NULL");
    SelectOut so;
    const double q = *quality_inout;
    const int rc = detect_common(ctx, 1, img_h, nullptr, w, h, stride, cell, cur_xy_h, ncur, 0, 0, roi, q,
                                 do_subpix, out_xy_h, out_n, &so);
    if (rc != OV2_OK) return rc;
    if (!img_h || w <= 0 || h <= 0) return OV2_OK;
    // :418-423 (nbkps there is the size after the secondary top-up)
    const int ncells = (w / cell) * (h / cell);
    if ((double)so.n < 0.33 * (double)(ncells - so.nboccup)) *quality_inout = q / 2.;
    else if ((double)so.n > 0.9 * (double)(ncells - so.nboccup)) *quality_inout = q * 1.5;
    return OV2_OK;
}

// Device-resident forms: the image is level 0 of batch item `item` of a pyramid that is already in HBM -- the (equalised)
// cur_img_ the reference hands to extractKeypoints at a keyframe (src/map_manager.cpp:312-320) is exactly what
// preprocessImage left in level 0 of cur_pyr_, so a keyframe costs no image upload.
static int pyr_level0(ov2_ctx *ctx, const ov2_pyr *pyr, int item, const uint8_t **img, int *w, int *h, int *stride)
{
    OV2_REQUIRE(ctx && pyr, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(item >= 0 && item < pyr->d.batch, OV2_EINVAL, "batch item out of range");
    const PyrLevelDesc &L0 = pyr->d.lv[0];
    *img = pyr->d.base + (long long)item * pyr->d.item_stride + L0.img_roi;
    *w = L0.w; *h = L0.h; *stride = L0.img_pitch;
    return ov2_pyr_wait_ready(ctx, pyr);
}

int ov2_detect_grid_fast_d(ov2_ctx *ctx, const ov2_pyr *pyr, int item, int cell, const float *cur_xy_h, int ncur,
                           int *fast_th_inout, int mask_mode, int do_subpix, float *out_xy_h, int *out_n)
{
    OV2_REQUIRE(fast_th_inout != nullptr, OV2_EINVAL, "fast_th_inout == NULL");
    OV2_REQUIRE(mask_mode == OV2_MASK_AS_EXECUTED || mask_mode == OV2_MASK_INTENDED, OV2_EINVAL, "bad mask_mode");
    const uint8_t *img; int w, h, stride;
    int rc = pyr_level0(ctx, pyr, item, &img, &w, &h, &stride);
    if (rc != OV2_OK) return rc;
    SelectOut so;
    const int th = *fast_th_inout;
    rc = detect_common(ctx, 0, nullptr, img, w, h, stride, cell, cur_xy_h, ncur, th, mask_mode, nullptr, 0.0, do_subpix, out_xy_h, out_n, &so);
    if (rc != OV2_OK) return rc;
    if ((double)so.nbkps < 0.5 * (double)so.nbempty && so.nbempty > 10) *fast_th_inout = (int)(th * 0.66);     // :546-552
    else if (so.nbkps == so.nbempty) *fast_th_inout = (int)(th * 1.5);
    return OV2_OK;
}

int ov2_detect_singlescale_d(ov2_ctx *ctx, const ov2_pyr *pyr, int item, int cell, const float *cur_xy_h, int ncur,
                             const int roi[4], double *quality_inout, int do_subpix, float *out_xy_h, int *out_n)
{
    OV2_REQUIRE(quality_inout != nullptr && roi != nullptr, OV2_EINVAL, "quality_inout / roi == NULL");
    const uint8_t *img; int w, h, stride;
    int rc = pyr_level0(ctx, pyr, item, &img, &w, &h, &stride);
    if (rc != OV2_OK) return rc;
    SelectOut so;
    const double q = *quality_inout;
    rc = detect_common(ctx, 1, nullptr, img, w, h, stride, cell, cur_xy_h, ncur, 0, 0, roi, q, do_subpix, out_xy_h, out_n, &so);
    if (rc != OV2_OK) return rc;
    const int ncells = (w / cell) * (h / cell);
    if ((double)so.n < 0.33 * (double)(ncells - so.nboccup)) *quality_inout = q / 2.;                         // :418-423
    else if ((double)so.n > 0.9 * (double)(ncells - so.nboccup)) *quality_inout = q * 1.5;
    return OV2_OK;
}

// Batched device-resident detection: every batch item of the pyramid in ONE call (chunks of 256 items share the response /
// candidate scratch; launches of successive chunks are stream-ordered, one host synchronisation at the end).
static int detect_batch(ov2_ctx *ctx, int mode, const ov2_pyr *pyr, int cell, const float *cur_xy_d, int cur_cap, const int *ncur_d,
                        int *fast_th_inout, int mask_mode, const int roi[4], double *quality_inout, int do_subpix,
                        float *out_xy_d, int out_cap, int *out_n_h)
{
    OV2_REQUIRE(ctx && pyr && out_xy_d && out_n_h, OV2_EINVAL, "NULL argument");
    const uint8_t *img; int w, h, stride;
    int rc = pyr_level0(ctx, pyr, 0, &img, &w, &h, &stride);
    if (rc != OV2_OK) return rc;
    const int items = pyr->d.batch;
    OV2_REQUIRE(cell >= 8 && cell <= DET_MAX_CELL, OV2_EUNSUPPORTED, "cell size must be in [8,64]");
    OV2_REQUIRE(w < 65536 && h < 32768, OV2_EUNSUPPORTED, "image too large");
    OV2_REQUIRE(cur_cap >= 0 && (ncur_d == nullptr || cur_xy_d != nullptr), OV2_EINVAL, "bad current keypoints");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    OV2_HIP_CHECK(det_raise_lds_limits());
    const int nw = w / cell, nh = h / cell, ncells = nw * nh, npx = cell * cell;
    for (int i = 0; i < items; i++) out_n_h[i] = 0;
    if (ncells == 0) return OV2_OK;
    const int cap = mode == 0 ? ncells : 2 * ncells;
    OV2_REQUIRE(out_cap >= cap, OV2_EINVAL, "out_cap too small: (w/cell)*(h/cell) points per item for FAST, twice that for single scale");
    { int slots_; OV2_REQUIRE(det_select_lds(w, h, cell, mode, ctx->det_fast_tie, &slots_) != 0, OV2_EUNSUPPORTED, "image too large for the LDS-resident exclusion mask"); }
    // Passes of `chunk` images share the response / candidate scratch.  More than one pass: TWO scratch sets of half a chunk, the passes
    // alternating between the context's stream and an auxiliary one -- the selection sweep of a pass (a dependency chain: two work-groups
    // per CU, the vector units nearly idle) then runs beside the response kernel of the next pass (issue-bound) instead of between two of them
    const bool two = DET_TWO_STREAMS && items > DET_CHUNK / 2;
    const int chunk = two ? DET_CHUNK / 2 : std::min(items, DET_CHUNK);
    const size_t map_bytes = (size_t)ncells * npx * (mode == 0 ? 1 : 4);
    const size_t set_bytes = (((size_t)chunk * map_bytes + 255) & ~(size_t)255) +
                             (((size_t)chunk * (ncells * sizeof(CellCand) + 4 * ((size_t)ncells + 1)) + 255) & ~(size_t)255);   // maps; candidates + free-cell lists
    const size_t o_cand_in_set = ((size_t)chunk * map_bytes + 255) & ~(size_t)255;
    const size_t o_so = (two ? 2 : 1) * set_bytes;
    const size_t o_par = (o_so + (size_t)items * sizeof(SelectOut) + 255) & ~(size_t)255;       // per-item threshold / quality
    const size_t total = o_par + (size_t)items * 8;
    rc = ctx->reserve_device(total); if (rc) return rc;
    rc = ctx->reserve_host((size_t)items * sizeof(SelectOut)); if (rc) return rc;
    uint8_t *ds = (uint8_t *)ctx->d_scratch;
    if (mode == 0) OV2_HIP_CHECK(hipMemcpyAsync(ds + o_par, fast_th_inout, 4 * (size_t)items, hipMemcpyHostToDevice, ctx->stream));
    else OV2_HIP_CHECK(hipMemcpyAsync(ds + o_par, quality_inout, 8 * (size_t)items, hipMemcpyHostToDevice, ctx->stream));
    hipStream_t main_stream = ctx->stream;
    if (two) {
        if (!ctx->det_aux_stream) OV2_HIP_CHECK(hipStreamCreateWithFlags(&ctx->det_aux_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) if (!ctx->det_ev[i]) OV2_HIP_CHECK(hipEventCreateWithFlags(&ctx->det_ev[i], hipEventDisableTiming));
        OV2_HIP_CHECK(hipEventRecord(ctx->det_ev[0], main_stream));                     // what precedes this call on the context's stream (the pyramid, the parameters)
        OV2_HIP_CHECK(hipStreamWaitEvent(ctx->det_aux_stream, ctx->det_ev[0], 0));
    }
    int pass = 0;
    for (int c0 = 0; c0 < items; c0 += chunk, pass++) {
        const int n = std::min(chunk, items - c0);
        const size_t o_set = two ? (size_t)(pass & 1) * set_bytes : 0;
        DetBatch B;
        memset(&B, 0, sizeof(B));
        B.img_stride = (long long)pyr->d.item_stride; B.cur_stride = cur_cap; B.out_stride = out_cap;
        B.ncur = ncur_d ? ncur_d + c0 : nullptr;
        if (mode == 0) B.fast_th = (const int *)(ds + o_par) + c0; else B.quality = (const double *)(ds + o_par) + c0;
        if (two && (pass & 1)) ctx->stream = ctx->det_aux_stream;                       // (the launchers enqueue on the context's stream; a context belongs to one thread)
        rc = enqueue_detect(ctx, mode, img + (long long)c0 * pyr->d.item_stride, w, h, stride, cell, n, B, 0, mask_mode, roi, 0.0, 0,
                            (const float2 *)cur_xy_d + (long long)c0 * cur_cap, ds + o_set, (CellCand *)(ds + o_set + o_cand_in_set),
                            (float2 *)out_xy_d + (long long)c0 * out_cap, (SelectOut *)(ds + o_so) + c0, do_subpix);
        ctx->stream = main_stream;
        if (rc) break;
    }
    if (two) {                                                                          // join (also on an error: nothing of this call may outlive it on the auxiliary stream)
        const hipError_t e1 = hipEventRecord(ctx->det_ev[1], ctx->det_aux_stream);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(main_stream, ctx->det_ev[1], 0) : e1;
        if (e2 != hipSuccess) { (void)hipStreamSynchronize(ctx->det_aux_stream); if (!rc) { ov2_set_error("detect_batch join: %s", hipGetErrorString(e2)); rc = OV2_EHIP; } }
    }
    if (rc) return rc;
    SelectOut *so = (SelectOut *)ctx->h_scratch;
    OV2_HIP_CHECK(hipMemcpyAsync(so, ds + o_so, (size_t)items * sizeof(SelectOut), hipMemcpyDeviceToHost, ctx->stream));
    OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < items; i++) {
        out_n_h[i] = so[i].n;
        if (mode == 0) {                                                 // threshold adaptation (:546-552), int *= double truncates
            const int th = fast_th_inout[i];
            if ((double)so[i].nbkps < 0.5 * (double)so[i].nbempty && so[i].nbempty > 10) fast_th_inout[i] = (int)(th * 0.66);
            else if (so[i].nbkps == so[i].nbempty) fast_th_inout[i] = (int)(th * 1.5);
        } else {                                                         // :418-423
            const double q = quality_inout[i];
            if ((double)so[i].n < 0.33 * (double)(ncells - so[i].nboccup)) quality_inout[i] = q / 2.;
            else if ((double)so[i].n > 0.9 * (double)(ncells - so[i].nboccup)) quality_inout[i] = q * 1.5;
        }
    }
    return OV2_OK;
}

int ov2_detect_singlescale_batch_d(ov2_ctx *ctx, const ov2_pyr *pyr, int cell, const float *cur_xy_d, int cur_cap, const int *ncur_d,
                                   const int roi[4], double *quality_inout, int do_subpix, float *out_xy_d, int out_cap, int *out_n_h)
{
    OV2_REQUIRE(quality_inout != nullptr && roi != nullptr, OV2_EINVAL, "quality_inout / roi == NULL");
    return detect_batch(ctx, 1, pyr, cell, cur_xy_d, cur_cap, ncur_d, nullptr, 0, roi, quality_inout, do_subpix, out_xy_d, out_cap, out_n_h);
}

int ov2_detect_grid_fast_batch_d(ov2_ctx *ctx, const ov2_pyr *pyr, int cell, const float *cur_xy_d, int cur_cap, const int *ncur_d,
                                 int *fast_th_inout, int mask_mode, int do_subpix, float *out_xy_d, int out_cap, int *out_n_h)
{
    OV2_REQUIRE(fast_th_inout != nullptr, OV2_EINVAL, "fast_th_inout == NULL");
    OV2_REQUIRE(mask_mode == OV2_MASK_AS_EXECUTED || mask_mode == OV2_MASK_INTENDED, OV2_EINVAL, "bad mask_mode");
    return detect_batch(ctx, 0, pyr, cell, cur_xy_d, cur_cap, ncur_d, fast_th_inout, mask_mode, nullptr, nullptr, do_subpix, out_xy_d, out_cap, out_n_h);
}

int ov2_corner_subpix(ov2_ctx *ctx, const uint8_t *img_h, int w, int h, int stride,
                      float *xy_inout_h, int n, int half_win, int max_iter, double eps)
{
    OV2_REQUIRE(ctx && img_h && w > 0 && h > 0 && stride >= w, OV2_EINVAL, "bad image");
    if (n <= 0) return OV2_OK;
    OV2_REQUIRE(xy_inout_h != nullptr, OV2_EINVAL, "xy == NULL");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t o_xy = ((size_t)w * h + 255) & ~(size_t)255;
    int rc = ctx->reserve_device(o_xy + 8 * (size_t)n); if (rc) return rc;
    uint8_t *ds = (uint8_t *)ctx->d_scratch;
    rc = ctx->upload_image(ds, (size_t)w, img_h, (size_t)stride, (size_t)w, (size_t)h); if (rc) return rc;
    OV2_HIP_CHECK(hipMemcpyAsync(ds + o_xy, xy_inout_h, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_subpix(ctx, ds, w, h, w, (float2 *)(ds + o_xy), n, half_win, max_iter, eps);
    if (rc) return rc;
    OV2_HIP_CHECK(hipMemcpyAsync(xy_inout_h, ds + o_xy, 8 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return OV2_OK;
}

} // extern "C"
