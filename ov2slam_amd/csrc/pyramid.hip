// pyramid.hip -- device-resident optical-flow pyramid for gfx950.
//
// Replaces cv::buildOpticalFlowPyramid(img, pyr, Size(win,win), maxLevel,
// withDerivatives=true, BORDER_REFLECT_101, BORDER_CONSTANT) as called at
// /root/reference/src/visual_front_end.cpp:1172, :53 and src/mapper.cpp:81.
// Arithmetic (all integer, bit-exact vs the oracle):
//   level l>0 : 5x5 [1 4 6 4 1]^2 pyrDown, (sum + 128) >> 8, REFLECT_101
//   derivative: Scharr 3/10/3, int16 (dx,dy) interleaved, REFLECT_101 on the image --
//               NOT materialised in HBM: the LK kernel evaluates it in registers from the image
//               rows it fetches anyway (lk.hip); k_scharr_level exists only for ov2_pyr_download.
// Layout: see PyrDesc in common.hpp.  Every level image carries a REFLECT_101
// border of >= win pixels so that (a) LK windows that hang over the image edge
// read legal memory exactly like OpenCV's padded Mats and (b) the 5x5 / 3x3
// stencils of the next kernel need no border logic when reading.
#include "common.hpp"
#include "xcd_map.hpp"
#include <limits.h>

#pragma clang fp contract(off)

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// ---- level kernel ----------------------------------------------------------------------------
// One launch per level L: a 256x32 tile of level L (+2 halo) is staged in LDS with aligned 8-byte
// loads; each thread then owns 8 pixels x 4 rows:
//   (i)   L == 0 only: the level-0 copy (aligned dword stores straight from the tile),
//   (iii) level L+1 = pyrDown(level L): a 4x2 block of outputs (two dword stores).
// Only ROI pixels are written here; the REFLECT_101 borders are filled by k_pyr_border (tiny).
// HBM traffic per level-0 pixel: 1 B read, 1 B + 0.25 B written -- the reference's pyramid also
// stores 4 B/px of derivatives, which this design never materialises.
// Knock-out timing (PYR_KO) of the first version (128x32 tiles, 4x4 pixels per thread, 2-byte stores) showed
// three additive costs -- load misses 80 us, 2-byte stores 50 us, LDS traffic + launch 46 us of 176 us at
// 1024 x 752x480 -- and no arithmetic cost at all: a latency-bound kernel.  Hence wider work per thread (twice
// the bytes in flight, dword stores, 3.5 instead of 5.25 LDS dwords per output).
#ifndef PYR_KO
#define PYR_KO 0                     // knock-out timing experiments (1: loads hit one line, 2: no stores, 4: no arithmetic)
#endif
#define PT_W 256
#define PT_H 32                      // (64 = two row sets per thread, twice the bytes in flight: measured 3 % slower)
#define PT_LDS_DW 66                 // (PT_W + 8) / 4 dwords per tile row: columns x0-4 .. x0+259
#define PT_ROWS (PT_H + 4)           // rows y0-2 .. y0+33

typedef uint32_t pyr_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t pyr_u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

// FUSE_BORDER: the work-group also writes the part of level L+1's REFLECT_101 border that mirrors its own outputs
// (rows: every thread stores its dwords a second time at the mirrored row; columns: the first / last work-group of
// a row rebuilds the left / right border dwords from the output tile kept in LDS).  The host falls back to
// k_pyr_border when the last work-group column holds fewer than win + 1 output columns.
template <bool FROM_RAW, bool FUSE_BORDER>
__global__ __launch_bounds__(256) void k_pyr_level(PyrDesc P, int level, const uint8_t *__restrict__ raw, int raw_stride,
                                                   long long raw_item_stride)
{
    __shared__ __attribute__((aligned(8))) uint32_t tile[PT_ROWS][PT_LDS_DW];
    const PyrLevelDesc L = P.lv[level];
    // XCD-aware work-group -> (image, tile) map (1-D launch; consecutive ids are dealt round-robin over the 8 XCDs): all
    // tiles of an image get ids of one residue mod 8 and consecutive rank, so that the halo rows / columns neighbouring
    // tiles share and the two halves of the output lines they split meet in ONE L2
    const int gx = (L.w + PT_W - 1) / PT_W, gy = (L.h + PT_H - 1) / PT_H, tiles = gx * gy;
    int b, tno;
    ov2_xcd_map(blockIdx.x, tiles, P.batch, &b, &tno);
    const int tby = tno / gx, tbx = tno - tby * gx;
    uint8_t *item = P.base + (long long)b * P.item_stride;
    const int x0 = tbx * PT_W, y0 = tby * PT_H;
    const int tid = threadIdx.x;

    // ---- stage the tile ----
    const uint8_t *src = FROM_RAW ? raw + (long long)b * raw_item_stride : item + L.img_roi;
    const bool aligned_src = FROM_RAW ? (((raw_stride | (int)(size_t)src) & 3) == 0) : true;
    // all loads of the tile are in flight before the first LDS store (a load -> store loop is latency-bound)
    constexpr int NE = PT_ROWS * (PT_LDS_DW / 2), NPT = (NE + 255) / 256;
    pyr_u32x2 stage[NPT];
#pragma unroll
    for (int i = 0; i < NPT; i++) {
        const int e = tid + 256 * i;
        const int row = e / (PT_LDS_DW / 2), dc = e - row * (PT_LDS_DW / 2);
        const int gy = y0 - 2 + row, gx = x0 - 4 + 8 * dc;
        pyr_u32x2 v = {0u, 0u};
        if (e < NE) {
            if (FROM_RAW) {
                const int sy = reflect101(gy, L.h);
                const uint8_t *rp = src + sy * raw_stride;
                if (aligned_src && gx >= 0 && gx + 7 < L.w) v = *(const pyr_u32x2_a4 *)(rp + gx);
                else {
                    uint32_t lo = 0, hi = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        lo |= (uint32_t)rp[reflect101(gx + k, L.w)] << (8 * k);
                        hi |= (uint32_t)rp[reflect101(gx + 4 + k, L.w)] << (8 * k);
                    }
                    v.x = lo; v.y = hi;
                }
            } else {
                // padded source: its REFLECT_101 border supplies the halo; clamp what lies beyond it (never consumed:
                // the stencils reach 2 columns / rows past the ROI, the clamps start at win >= 3)
                const int cy = min(max(gy, -P.win), L.h + P.win - 1);
                const int cx = min(gx, L.w + P.win) & ~3;                       // [cx, cx+8) stays inside the row (pitch slack 8)
                if (PYR_KO & 1) v = *(const pyr_u32x2_a4 *)(src + (tid & 15) * 8);
                else v = *(const pyr_u32x2_a4 *)(src + cy * L.img_pitch + cx);
            }
        }
        stage[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NPT; i++) {
        const int e = tid + 256 * i;
        if (e < NE) ((pyr_u32x2 *)&tile[0][0])[e] = stage[i];
    }
    __syncthreads();

    const int tx = tid & 31;
    const bool has_next = level + 1 < P.n_levels;
    const PyrLevelDesc N = P.lv[has_next ? level + 1 : level];
    const int win = P.win;
    __shared__ uint32_t otile[FUSE_BORDER ? PT_H / 2 : 1][FUSE_BORDER ? PT_W / 8 + 1 : 1];      // the work-group's outputs (border source)
    uint8_t *nroi = item + N.img_roi;
    // rows of the top / bottom border that mirror ROI row Y: -Y for 1 <= Y <= win, 2 (h-1) - Y for h-1-win <= Y <= h-2
    auto mirror_rows = [&](int Yo, int &m0, int &m1) {
        m0 = (Yo >= 1 && Yo <= win) ? -Yo : INT_MIN;
        m1 = (Yo >= N.h - 1 - win && Yo <= N.h - 2) ? 2 * (N.h - 1) - Yo : INT_MIN;
    };
#pragma unroll
    for (int set = 0; set < PT_H / 32; set++) {
    const int ty = (tid >> 5) + 8 * set;
    const int x = x0 + 8 * tx, y = y0 + 4 * ty;          // this thread: pixels (x..x+7, y..y+3); tile rows 4ty+2 .., dwords 2tx+1, 2tx+2
    const bool inside = x < L.w && y < L.h;
    // ---- (i) level-0 copy ----
    if (FROM_RAW && inside) {
        uint8_t *roi = item + L.img_roi;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int yy = y + rr;
            if (yy >= L.h) break;
#pragma unroll
            for (int hx = 0; hx < 2; hx++) {
                const int xx = x + 4 * hx;
                const uint32_t v = tile[4 * ty + 2 + rr][2 * tx + 1 + hx];
                uint8_t *d = roi + yy * L.img_pitch + xx;
                if (xx + 3 < L.w) *(uint32_t *)d = v;
                else for (int j = 0; j < 4 && xx + j < L.w; j++) d[j] = (uint8_t)(v >> (8 * j));
            }
        }
    }
    // ---- (iii) next level: 4x2 outputs (X..X+3, Y..Y+1), X = x/2, Y = y/2 ----
    if (has_next && inside) {
        const int X = x >> 1, Y = y >> 1;
        // horizontal 5-tap sums of the seven rows y-2 .. y+4 at centre columns x, x+2, x+4, x+6, on packed bytes:
        // v_dot4_u32_u8 with the weights (1,4,6,4) over the 4 leftmost taps, a second one (weight 1) for the fifth.
        // hp[r][k] = (h[2k], h[2k+1]) as two u16 (<= 16 * 255); the vertical pass stays in packed 16 bits:
        // 16 * 4080 + 128 < 2^16.
        uint32_t hp[7][2];
#pragma unroll
        for (int r = 0; r < 7; r++) {
            const pyr_u32x2 w0 = *(const pyr_u32x2 *)&tile[4 * ty + r][2 * tx], w1 = *(const pyr_u32x2 *)&tile[4 * ty + r][2 * tx + 2];
            const uint32_t a = w0.x, q0 = w0.y, q1 = w1.x, c = w1.y;              // columns x-4.., x.., x+4.., x+8..
            const uint32_t l0 = __builtin_amdgcn_alignbyte(q0, a, 2), l2 = __builtin_amdgcn_alignbyte(q1, q0, 2);
            const uint32_t h0 = __builtin_amdgcn_udot4(l0, 0x04060401u, __builtin_amdgcn_udot4(q0, 0x00010000u, 0u, false), false);
            const uint32_t h1 = __builtin_amdgcn_udot4(q0, 0x04060401u, __builtin_amdgcn_udot4(q1, 0x00000001u, 0u, false), false);
            const uint32_t h2 = __builtin_amdgcn_udot4(l2, 0x04060401u, __builtin_amdgcn_udot4(q1, 0x00010000u, 0u, false), false);
            const uint32_t h3 = __builtin_amdgcn_udot4(q1, 0x04060401u, __builtin_amdgcn_udot4(c, 0x00000001u, 0u, false), false);
            hp[r][0] = (PYR_KO & 4) ? q0 : (h0 | (h1 << 16));
            hp[r][1] = (PYR_KO & 4) ? q1 : (h2 | (h3 << 16));
        }
        typedef unsigned short pu16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int oy = 0; oy < 2; oy++) {
            if (Y + oy >= N.h) break;
            const int r = 2 * oy;                                                  // rows y-2+2oy .. y+2+2oy
            uint32_t vv[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const pu16x2 c2 = __builtin_bit_cast(pu16x2, hp[r + 2][k]), c1 = __builtin_bit_cast(pu16x2, hp[r + 1][k]), c3 = __builtin_bit_cast(pu16x2, hp[r + 3][k]);
                const pu16x2 c0 = __builtin_bit_cast(pu16x2, hp[r][k]), c4 = __builtin_bit_cast(pu16x2, hp[r + 4][k]);
                const pu16x2 v = (c2 * (unsigned short)6 + (c1 + c3) * (unsigned short)4 + c0 + c4 + (unsigned short)128) >> (unsigned short)8;
                vv[k] = (PYR_KO & 4) ? hp[r][k] ^ hp[r + 4][k] : __builtin_bit_cast(uint32_t, v);
            }
            const uint32_t out = __builtin_amdgcn_perm(vv[1], vv[0], 0x06040200u);    // bytes (v0, v1, v2, v3)
            if (FUSE_BORDER) otile[2 * ty + oy][tx] = out;
            if ((PYR_KO & 2) && out != 0x12345678u) continue;
            uint8_t *d = nroi + (Y + oy) * N.img_pitch + X;                         // X % 4 == 0, ROI origin 16-byte aligned
            if (X + 3 < N.w) {
                *(uint32_t *)d = out;
                if (FUSE_BORDER) {
                    int m0, m1;
                    mirror_rows(Y + oy, m0, m1);
                    if (m0 != INT_MIN) *(uint32_t *)(nroi + m0 * N.img_pitch + X) = out;
                    if (m1 != INT_MIN) *(uint32_t *)(nroi + m1 * N.img_pitch + X) = out;
                }
            } else for (int j = 0; j < 4 && X + j < N.w; j++) d[j] = (uint8_t)(out >> (8 * j));    // (its mirrors: right-border items below)
        }
    }
    }
    if (!has_next) return;
    if (FUSE_BORDER) {
        // left / right border dwords of this work-group's output rows (and of the border rows mirroring them), same
        // dword set as k_pyr_border: columns [-PB_LEFT, 0) and [w & ~3, (w + win + 3) & ~3)
        const bool first = tbx == 0, last = tbx == gx - 1;
        if (!first && !last) return;
        __syncthreads();
        const int PB_LEFT = (win + 3) & ~3, nl = first ? PB_LEFT >> 2 : 0;
        const int rbeg = N.w & ~3, nr = last ? (((N.w + win + 3) & ~3) - rbeg) >> 2 : 0;
        const int X0 = x0 >> 1, Y0 = y0 >> 1;
        for (int e = tid; e < (PT_H / 2) * (nl + nr); e += 256) {
            const int row = e / (nl + nr), dw = e - row * (nl + nr);
            const int Yo = Y0 + row;
            if (Yo >= N.h) break;
            const int c0 = dw < nl ? 4 * dw - PB_LEFT : rbeg + 4 * (dw - nl);
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int c = c0 + k;
                c = c < -win ? -win : (c > N.w + win - 1 ? N.w + win - 1 : c);            // padding bytes: any value
                const int sc = reflect101(c, N.w) - X0;                                      // 0 <= sc < PT_W / 2 (host-checked)
                v |= ((otile[row][sc >> 2] >> (8 * (sc & 3))) & 0xFFu) << (8 * k);
            }
            int m0, m1;
            mirror_rows(Yo, m0, m1);
            *(uint32_t *)(nroi + Yo * N.img_pitch + c0) = v;
            if (m0 != INT_MIN) *(uint32_t *)(nroi + m0 * N.img_pitch + c0) = v;
            if (m1 != INT_MIN) *(uint32_t *)(nroi + m1 * N.img_pitch + c0) = v;
        }
    }
}

// ---- REFLECT_101 border of one level: thread per border DWORD ---------------------------------
// border = padded rect [-win, w+win) x [-win, h+win) minus the ROI.  Work items are aligned dwords:
//   * 2*win border rows, each from column -round_up(win, 4) to round_up(w+win, 4)      (plain dword copies of the
//     mirrored ROI row, except for the few edge dwords that also mirror columns),
//   * h ROI rows x (round_up(win, 4)/4 left dwords + the dwords covering [w & ~3, w+win) on the right).
// Bytes written outside [-win, w+win) lie in the row padding (img_padx >= 16, pitch slack >= 8) and are
// never consumed; bytes of a right-edge dword that belong to the ROI are rewritten with their own value.
__global__ __launch_bounds__(256) void k_pyr_border(PyrDesc P, int level)
{
    const PyrLevelDesc L = P.lv[level];
    const int win = P.win, PB_LEFT = (P.win + 3) & ~3;
    const int wend = (L.w + win + 3) & ~3;
    const int ndw_row = (PB_LEFT + wend) >> 2;                       // dwords of a full border row
    const int rbeg = L.w & ~3, ndw_r = (wend - rbeg) >> 2, ndw_lr = PB_LEFT / 4 + ndw_r;
    const int n_tb = 2 * win * ndw_row, n_lr = L.h * ndw_lr;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_tb + n_lr) return;
    int yt, c0;
    if (e < n_tb) {
        const int row = e / ndw_row;
        c0 = (e - row * ndw_row) * 4 - PB_LEFT;
        yt = row < win ? row - win : L.h + (row - win);
    } else {
        const int f = e - n_tb;
        yt = f / ndw_lr;
        const int d = f - yt * ndw_lr;
        c0 = d < PB_LEFT / 4 ? 4 * d - PB_LEFT : rbeg + 4 * (d - PB_LEFT / 4);
    }
    uint8_t *roi = P.base + (long long)blockIdx.z * P.item_stride + L.img_roi;
    const uint8_t *srow = roi + reflect101(yt, L.h) * L.img_pitch;
    uint32_t v;
    if (c0 >= 0 && c0 + 3 < L.w) v = *(const uint32_t *)(srow + c0);
    else {
        v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int c = c0 + k;
            c = c < -win ? -win : (c > L.w + win - 1 ? L.w + win - 1 : c);       // padding bytes: any value
            v |= (uint32_t)srow[reflect101(c, L.w)] << (8 * k);
        }
    }
    *(uint32_t *)(roi + yt * L.img_pitch + c0) = v;
}

// ---- Scharr derivative of one level of one item, on demand (ov2_pyr_download only) -----------
__global__ __launch_bounds__(256) void k_scharr_level(PyrDesc P, int level, int b, uint32_t *__restrict__ out)
{
    const PyrLevelDesc L = P.lv[level];
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= L.w || y >= L.h) return;
    const uint8_t *s = P.base + (long long)b * P.item_stride + L.img_roi + (long long)y * L.img_pitch + x;
    const int p = L.img_pitch;
    const int a00 = s[-p - 1], a01 = s[-p], a02 = s[-p + 1];
    const int a10 = s[-1],                 a12 = s[1];
    const int a20 = s[p - 1],  a21 = s[p],  a22 = s[p + 1];
    const int t0l = (a00 + a20) * 3 + a10 * 10, t0r = (a02 + a22) * 3 + a12 * 10;
    const int t1l = a20 - a00, t1c = a21 - a01, t1r = a22 - a02;
    const int dx = t0r - t0l;
    const int dy = (t1l + t1r) * 3 + t1c * 10;
    out[(long long)y * L.w + x] = ((uint32_t)(uint16_t)(int16_t)dx) | (((uint32_t)(uint16_t)(int16_t)dy) << 16);
}

int ov2_launch_pyr_build(ov2_ctx *ctx, ov2_pyr *p, const uint8_t *img_d, int stride, size_t img_batch_stride, int from_level)
{
    const PyrDesc &P = p->d;
    auto border = [&](int l) {
        const PyrLevelDesc &L = P.lv[l];
        const int wend = (L.w + P.win + 3) & ~3, PB_LEFT = (P.win + 3) & ~3;
        const int n = 2 * P.win * ((PB_LEFT + wend) >> 2) + L.h * (PB_LEFT / 4 + ((wend - (L.w & ~3)) >> 2));
        hipLaunchKernelGGL(k_pyr_border, dim3((n + 255) / 256, 1, P.batch), dim3(256), 0, ctx->stream, P, l);
    };
    for (int l = from_level; l < P.n_levels; l++) {
        const PyrLevelDesc &L = P.lv[l];
        const int gx = (L.w + PT_W - 1) / PT_W, gy = (L.h + PT_H - 1) / PT_H;
        dim3 grid(gx * gy * P.batch);                       // 1-D: the kernel decodes (image, tile) XCD-aware
        // the level kernel writes level l+1's border itself when the last work-group column owns >= win + 1 output columns
        const bool has_next = l + 1 < P.n_levels;
        const bool fuse = has_next && P.lv[l + 1].w - (PT_W / 2) * (gx - 1) >= P.win + 1;
        if (l == 0 && img_d) {
            if (fuse) hipLaunchKernelGGL((k_pyr_level<true, true>), grid, dim3(256), 0, ctx->stream, P, 0, img_d, stride, (long long)img_batch_stride);
            else hipLaunchKernelGGL((k_pyr_level<true, false>), grid, dim3(256), 0, ctx->stream, P, 0, img_d, stride, (long long)img_batch_stride);
            border(0);
        } else if (has_next) {
            // l == 0: level 0 and its border were written in place by the producer (ov2_pyr_build_clahe_d): pyrDown from the padded image
            if (fuse) hipLaunchKernelGGL((k_pyr_level<false, true>), grid, dim3(256), 0, ctx->stream, P, l, (const uint8_t *)nullptr, 0, 0LL);
            else hipLaunchKernelGGL((k_pyr_level<false, false>), grid, dim3(256), 0, ctx->stream, P, l, (const uint8_t *)nullptr, 0, 0LL);
        }
        if (has_next && !fuse) border(l + 1);        // level l+1 was just produced
    }
    OV2_HIP_CHECK(hipGetLastError());
    return OV2_OK;
}

// ---- producer / consumer ordering across contexts ---------------------------------
int ov2_pyr_mark_ready(ov2_ctx *ctx, ov2_pyr *p)
{
    OV2_REQUIRE(p->parent == nullptr, OV2_EINVAL, "an item view (ov2_pyr_item_view) is read-only: build the batch pyramid it aliases");
    OV2_HIP_CHECK(hipEventRecord(p->ready, ctx->stream));
    p->producer = ctx->stream;
    p->built = true;
    return OV2_OK;
}

int ov2_pyr_wait_ready(ov2_ctx *ctx, const ov2_pyr *p)
{
    if (p->parent) p = p->parent;                 // an item view: the batch pyramid's hand-off state
    if (p->producer != ctx->stream && p->ready) OV2_HIP_CHECK(hipStreamWaitEvent(ctx->stream, p->ready, 0));
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
        L.der_padx = 0; L.der_pitch = 0; L.der_roi = -1;    // derivatives are not stored (see header)
        const long long img_bytes = (long long)(lh + 2 * win) * L.img_pitch;
        off = round_up(off, 256);
        L.img_roi = off + (long long)L.pady * L.img_pitch + L.img_padx;
        off += img_bytes;
        D.n_levels = l + 1;
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
        if (lw <= win || lh <= win) break;     // buildOpticalFlowPyramid stops early
    }
    D.item_stride = round_up(off, 4096);
    p->bytes = (size_t)D.item_stride * (size_t)batch;
    hipError_t e = hipMalloc((void **)&D.base, p->bytes);
    if (e != hipSuccess) { delete p; ov2_set_error("hipMalloc(%zu): %s", p->bytes, hipGetErrorString(e)); return OV2_ENOMEM; }
    // alignment slack around the borders is read (never consumed) by the dword row loads: keep it defined
    e = hipMemsetAsync(D.base, 0, p->bytes, ctx->stream);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ready, hipEventDisableTiming);
    if (e == hipSuccess) { e = hipEventRecord(p->ready, ctx->stream); p->producer = ctx->stream; }
    if (e != hipSuccess) {
        if (p->ready) (void)hipEventDestroy(p->ready);
        (void)hipFree(D.base); delete p; ov2_set_error("ov2_pyr_create: %s", hipGetErrorString(e)); return OV2_EHIP;
    }
    *out = p;
    return OV2_OK;
}

int ov2_pyr_item_view(const ov2_pyr *p, int item, ov2_pyr **out)
{
    OV2_REQUIRE(p && out, OV2_EINVAL, "NULL argument");
    *out = nullptr;
    OV2_REQUIRE(p->parent == nullptr, OV2_EINVAL, "views of views are not supported");
    OV2_REQUIRE(item >= 0 && item < p->d.batch, OV2_EINVAL, "batch item out of range");
    ov2_pyr *v = new (std::nothrow) ov2_pyr();
    OV2_REQUIRE(v != nullptr, OV2_ENOMEM, "out of host memory");
    v->d = p->d;
    v->d.base = p->d.base + (long long)item * p->d.item_stride;
    v->d.batch = 1;
    v->w = p->w; v->h = p->h; v->max_level = p->max_level; v->device = p->device;
    v->bytes = (size_t)p->d.item_stride;
    v->parent = p;
    *out = v;
    return OV2_OK;
}

void ov2_pyr_destroy(ov2_pyr *p)
{
    if (!p) return;
    if (p->parent) { delete p; return; }          // a view owns nothing
    (void)hipSetDevice(p->device);
    if (p->ready) { (void)hipEventSynchronize(p->ready); (void)hipEventDestroy(p->ready); }   // a consumer may still be reading
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
    const int rc = ov2_launch_pyr_build(ctx, p, img_d, stride, img_batch_stride);
    return rc != OV2_OK ? rc : ov2_pyr_mark_ready(ctx, p);
}

int ov2_pyr_build_h(ov2_ctx *ctx, ov2_pyr *p, const uint8_t *img_h, int stride, size_t img_batch_stride)
{
    OV2_REQUIRE(ctx && p && img_h, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(stride >= p->w, OV2_EINVAL, "stride < width");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    // device staging copy with a 16-byte-aligned pitch: the kernels then take their aligned-dword paths whatever the width
    const size_t pitch = ((size_t)p->w + 15) & ~(size_t)15, item = pitch * (size_t)p->h;
    const int rc = ctx->reserve_device(item * (size_t)p->d.batch);
    if (rc != OV2_OK) return rc;
    if (p->d.batch > 1) OV2_REQUIRE(img_batch_stride >= (size_t)stride * (size_t)p->h, OV2_EINVAL, "img_batch_stride too small");
    if (p->d.batch == 1) {
        const int rcu = ctx->upload_image(ctx->d_scratch, pitch, img_h, (size_t)stride, (size_t)p->w, (size_t)p->h);
        if (rcu != OV2_OK) return rcu;
    } else
    for (int b = 0; b < p->d.batch; b++) {
        OV2_HIP_CHECK(hipMemcpy2DAsync((uint8_t *)ctx->d_scratch + item * b, pitch,
                                       img_h + img_batch_stride * b, (size_t)stride, (size_t)p->w, (size_t)p->h,
                                       hipMemcpyHostToDevice, ctx->stream));
    }
    const int rc2 = ov2_launch_pyr_build(ctx, p, (const uint8_t *)ctx->d_scratch, (int)pitch, item);
    return rc2 != OV2_OK ? rc2 : ov2_pyr_mark_ready(ctx, p);
}

static int pyr_download_impl(ov2_ctx *ctx, const ov2_pyr *p, int b, int level, uint8_t *img_h, int16_t *deriv_h, int padded)
{
    OV2_REQUIRE(ctx && p, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(level >= 0 && level < p->d.n_levels && b >= 0 && b < p->d.batch, OV2_EINVAL, "bad level/batch index");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    if (int rcw = ov2_pyr_wait_ready(ctx, p)) return rcw;
    const PyrLevelDesc &L = p->d.lv[level];
    const int pad = padded ? p->d.win : 0;
    const int ow = L.w + 2 * pad, oh = L.h + 2 * pad;
    const uint8_t *item = p->d.base + (long long)b * p->d.item_stride;
    if (img_h) {
        const uint8_t *src = item + L.img_roi - (long long)pad * L.img_pitch - pad;
        const int rcd = ctx->download_image(img_h, (size_t)ow, src, (size_t)L.img_pitch, (size_t)ow, (size_t)oh);
        if (rcd != OV2_OK) return rcd;
    }
    if (deriv_h) {
        // evaluate the derivative of this level on demand; border = BORDER_CONSTANT(0) like the reference's Mats
        const size_t bytes = (size_t)L.w * L.h * 4;
        const int rc = ctx->reserve_device(bytes);
        if (rc != OV2_OK) return rc;
        hipLaunchKernelGGL(k_scharr_level, dim3((L.w + 255) / 256, L.h), dim3(256), 0, ctx->stream, p->d, level, b, (uint32_t *)ctx->d_scratch);
        OV2_HIP_CHECK(hipGetLastError());
        if (pad) memset(deriv_h, 0, (size_t)ow * oh * 4);
        const int rcd = ctx->download_image((uint8_t *)deriv_h + ((size_t)pad * ow + pad) * 4, (size_t)ow * 4, ctx->d_scratch, (size_t)L.w * 4,
                                            (size_t)L.w * 4, (size_t)L.h);
        if (rcd != OV2_OK) return rcd;
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
