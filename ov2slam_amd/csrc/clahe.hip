// clahe.hip -- cv::CLAHE::apply (CV_8UC1) for gfx950.
//
// Replaces ptracker_->pclahe_->apply(img_raw, cur_img_) at /root/reference/src/visual_front_end.cpp:1159
// and src/mapper.cpp:76 (handle created at src/ov2slam.cpp:85-89: clip = fclahe_val, tiles = (w/50, h/50)).
// Two kernels, integer histogram work + fp32 interpolation without FMA contraction (bit-exact vs oracle):
//   k_clahe_lut   : one workgroup per (tile, image): 256-bin histogram in LDS (ds_add), clip,
//                   redistribution, inclusive scan, LUT = saturate(cvRound(cdf * 255 / tile_area))
//   k_clahe_apply : 4 pixels per thread (dword load/store), bilinear blend of the four tile LUTs
#include "common.hpp"
#include "xcd_map.hpp"
#include <math.h>
#include <mutex>
#include <type_traits>

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
    int border;          // > 0: dst is a padded pyramid level -- also write its REFLECT_101 border of this many pixels
    int batch, gx_lut;   // images in the launch, work-groups per image of the LUT kernel (1-D XCD-aware launches)
    int ysplit;          // apply kernel: work-groups per row of interpolation cells (1 in batch mode; a single image is cut
                         // into ~60 short row bands so that it does not run on 10 CUs only -- latency)
    // strip kernel (k_clahe_apply_pyr): column strips per image, level 1 of the pyramid next to the level-0 destination
    int nstrips, nlut;   // nlut: LUT wavefronts of the fused form
    long long l1_delta;  // byte offset of level 1's ROI pixel (0, 0) from the dst ROI pointer
    int l1_pitch, l1_w, l1_h;
};

// One WAVEFRONT per (tile, image), four tiles per workgroup, no workgroup barriers: 16 lanes cover one
// tile row as aligned dwords (<= 64 bytes), so a wavefront histograms 4 rows per trip.
//   * (round 5: FOUR staggered copies again for the wavefronts that histogram most of the tiles.  The measurement below was taken on
//     band-limited noise, where a wavefront's 64 pixels hit ~60 different bins; on a constant, an over- or an under-exposed frame they
//     hit a handful, the ds_adds serialise and the fused pre-processing took 3.0x / 2.5x / 1.45x as long (tools/pre_micro.py .. entropy).
//     Four copies: 1.30x / 1.20x / 0.98x, +1.5 % on noise.  Adding a dword that many lanes share once with the lane count as weight
//     was built as well: tested per row group it costs 15 % on noise, tested per tile it gains nothing over the copies -- not kept;
//     two copies leave 1.5x, eight lose to their LDS footprint: profiles/r5_clahe_histogram_variants.txt.)
//   * ONE 256-bin histogram per wavefront.  Rounds 1-2 kept 8 staggered copies against same-address / same-bank collisions
//     of neighbouring pixels; measured in round 3 (OV2SLAM_HIP_LIB A/B builds, tools/pre_micro.py): 16, 8, 4, 2 and 1 copies run
//     within noise of each other (16: slower, occupancy), and so does a build with 27 % fewer vector instructions: the kernel is
//     bound by the rate of the LDS atomics themselves (one ds_add_u32 per pixel, ~10 cycles per wave64 instruction, 83 % of the
//     CUs' LDS-unit time).  One copy is the simplest form: the address is the bin, one b128 store per lane to clear, one b128
//     load to read it back;
//   * bytes of an edge dword that lie outside the tile are added with weight 0 instead of being branched around; all loads of
//     a tile are in flight before the first ds_add (the loop is latency-bound otherwise);
//   * clip / redistribute / scan / LUT run on 4 bins per lane with DPP row scans and readlane, no division
//     per bin.
__device__ __forceinline__ void clahe_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef CLAHE_KO
#define CLAHE_KO 0        // knock-out timing experiments. lut: 1 loads hit one line, 2 no ds_add, 4 no clip/scan tail;
#endif                    // apply: 8 loads hit, 16 no stores, 32 no LUT look-ups, 64 no blend
#define CH_WAVE_DW 256               // dwords of a one-copy histogram
#define CH_CSTRIDE 264               // dwords between the copies of a multi-copy histogram: bin b of copy c sits in LDS bank (b + 8 c) % 64
#ifndef CH_NCOPY
#define CH_NCOPY 4                   // copies of the wavefronts that histogram most tiles (k_clahe_lut; the LUT wavefronts of the fused kernel)
#endif
                                     // (A/B builds: tools/build_variant.sh .. -DCH_NCOPY=..; profiles/r5_clahe_histogram_variants.txt)
#define CH_MULTI_DW (CH_NCOPY * CH_CSTRIDE)
typedef uint32_t c_u32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ int c_dpp0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }   // out-of-row sources read 0

// The tiles t, t + tstride, .. < t_end of ONE image, by one wavefront with its own 256-bin histogram `hw` in LDS; store_lut(t, packed)
// receives the lane's four LUT bytes (bins 4 lane .. 4 lane + 3) of tile t.  Shared by k_clahe_lut (LUTs to HBM) and the fused
// strip kernel (LUTs stay in LDS).
// ncopy (wave-uniform, 1 or CH_NCOPY): copies of the histogram, CH_CSTRIDE dwords apart; a lane adds into copy (lane ^ lane >> 4) % ncopy
// (horizontal and vertical neighbours -- similar grey levels -- go to different copies and different banks).  Low-entropy frames:
// with one copy every equal pair of the 64 pixels of a ds_add serialises; four copies bound the serialisation at 16 lanes.
template <bool SRC_ALIGNED, class StoreLut>          // true: rows and base are 4-byte aligned (phase 0 everywhere: cheap addressing)
__device__ __forceinline__ void clahe_lut_tiles(const ClaheParams &P, const uint8_t *img, uint32_t *hw, int lane, int t, int t_end, int tstride,
                                                StoreLut store_lut, int ncopy = 1)
{
    const int ntiles = t_end;
    const int sub = lane >> 4, l16 = lane & 15;
    uint32_t *hist = hw + ((lane ^ (lane >> 4)) & (ncopy - 1)) * CH_CSTRIDE;        // this lane's copy
    const bool fast_geom = P.tw <= 61 && P.th <= 64;
    // Rows are fetched as ALIGNED dwords whatever the alignment of the image (KITTI: 1241-byte rows): dword l16 of
    // the row segment starts `ph` bytes before the first tile byte, ph = (address of that byte) & 3, per row.
    const uint8_t *img_end = img + (long long)(P.h - 1) * P.stride + P.w;          // one past the last pixel of this image
    const bool same_phase = SRC_ALIGNED || (P.stride & 3) == 0;

    // a wavefront walks over several tiles; the 16 row-dwords of the NEXT tile are requested before the
    // current one is histogrammed, so the global round trip hides behind the LDS work
    // Tiles of the last column hang over the right image edge (REFLECT_101 padding of the reference's copyMakeBorder):
    // padded column w + k is column w - 2 - k, i.e. the columns (2w - 2 - x_end, w - 2] of such a tile count TWICE and
    // the columns >= w not at all -- per-byte weights 0 / 1 / 2 instead of a byte-wise walk (which, at 1/15 of the
    // tiles, took half of this kernel's time: ~56 dependent byte loads per lane).
    // (needs the mirror sources of a tile's padded columns inside the same tile: x_begin + x_end <= 2w - 1)
    auto tile_fast = [&](int t) { const int xb = (t % P.tiles_x) * P.tw; return fast_geom && 2 * xb + P.tw <= 2 * P.w - 1; };
    // (knock-out timing: with the loads hitting one cache line AND without the ds_adds the kernel still took 80 % of
    // its time -- it is bound by the ~1000 instructions a wavefront issues per tile, a third of them the address
    // arithmetic and predicates of these 16 loads; hence the wave-uniform fast path below)
    const uint32_t lane_off = (uint32_t)(sub * P.stride + 4 * l16);
    auto tile_load = [&](int t, uint32_t (&vv)[16], uint32_t &phs) {
        const int ty = t / P.tiles_x, tx = t - ty * P.tiles_x;
        const int x_begin = tx * P.tw;
        if (SRC_ALIGNED && !(CLAHE_KO & 1) && (ty * P.th + 64 <= P.h) && (long long)P.h * P.stride < (1ll << 31)) {
            // all 64 candidate rows lie inside the image: one scalar base, one 32-bit lane offset, no per-row tests
            // (rows >= th are fetched and ignored; lanes right of the tile stay masked: the row may end there)
            const uint32_t ph = (uint32_t)(x_begin & 3);
            phs = ph * 0x55555555u;
            const uint8_t *tp = img + (long long)(ty * P.th) * P.stride + (x_begin & ~3);
            // lanes right of the tile re-read the row's first dword (the row may end at the tile edge); never counted
            uint32_t off = (4 * l16 < P.tw + (int)ph && (x_begin & ~3) + 4 * l16 < P.w) ? lane_off : (uint32_t)(sub * P.stride);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                vv[i] = *(const uint32_t *)(tp + off);
                off += 4u * (uint32_t)P.stride;
            }
            return;
        }
        const int ybase = ty * P.th + sub;
        phs = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            int y = ybase + 4 * i;
            y = y >= P.h ? 2 * P.h - 2 - y : y;                     // bottom REFLECT_101 padding (< one tile high)
            y = y < 0 ? 0 : y;
            uint32_t v = 0;
            if (SRC_ALIGNED) {
                const uint32_t ph = (uint32_t)(x_begin & 3);          // rows and base aligned: the phase is the tile's column phase
                phs |= ph << (2 * i);
                if (CLAHE_KO & 1) { if (4 * i + sub < P.th && 4 * l16 < P.tw + (int)ph && (x_begin & ~3) + 4 * l16 < P.w) v = *(const uint32_t *)(img + 4 * l16); }
                else
                if (4 * i + sub < P.th && 4 * l16 < P.tw + (int)ph && (x_begin & ~3) + 4 * l16 < P.w) v = *(const uint32_t *)(img + y * P.stride + (x_begin & ~3) + 4 * l16);
            } else {
                const uint8_t *rp = img + (long long)y * P.stride + x_begin;
                const uint32_t ph = (uint32_t)((size_t)rp & 3);
                phs |= ph << (2 * i);
                const uint8_t *ap = rp - ph + 4 * l16;
                if (4 * i + sub < P.th && 4 * l16 < P.tw + (int)ph && x_begin - (int)ph + 4 * l16 < P.w) {  // this dword holds at least one byte of the tile row
                    if (ap + 4 <= img_end) v = *(const uint32_t *)ap;
                    else for (int k = 0; k < 4; k++) if (ap + k < img_end) v |= (uint32_t)ap[k] << (8 * k);
                }
            }
            vv[i] = v;
        }
    };

    // Software pipeline over the wavefront's tiles (round 4: one register set instead of two, and the NEXT tile's atomics in flight
    // during the current tile's clip / scan / LUT arithmetic -- a single LUT wavefront of the fused kernel is a serial chain, the
    // ~300 instructions of that tail used to run with the LDS idle and the histogram with the VALU idle):
    //   histogram(t): clear, then every ds_add of tile t is ISSUED (its pixels come from `cur`, read at issue);
    //   then the loads of the tile after it go into the same registers; nobody waits for either before the next iteration.
    uint32_t cur[16], cur_ph = 0;
    auto histogram = [&](int th) {
        const int ty = th / P.tiles_x, tx = th - ty * P.tiles_x;
        const int x_begin = tx * P.tw;                              // tile columns in padded coordinates
        if (ncopy == 1) ((c_u32x4 *)hw)[lane] = (c_u32x4)(0u);
        else for (int q = lane; q < CH_MULTI_DW / 4; q += 64) ((c_u32x4 *)hw)[q] = (c_u32x4)(0u);
        clahe_wave_sync();
        if (tile_fast(th)) {
            // byte k of this lane's dword is tile column 4*l16 + k - ph; bytes outside [0, tw) are added with weight 0
            // weight of tile column p (image column x_begin + p): see tile_fast
            const int dbl_lo = 2 * P.w - 2 - (x_begin + P.tw);      // columns in (dbl_lo, w - 2] also stand for a padded column
            auto wgt = [&](int p) { const int c = x_begin + p; return ((unsigned)p < (unsigned)P.tw && c < P.w) ? ((c > dbl_lo && c <= P.w - 2) ? 2u : 1u) : 0u; };
            const int p_same = 4 * l16 - (int)(cur_ph & 3);
            uint32_t w0 = wgt(p_same), w1 = wgt(p_same + 1), w2 = wgt(p_same + 2), w3 = wgt(p_same + 3);
            bool mine = p_same < P.tw;                              // this lane's dword holds at least one tile byte
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (!same_phase) {                                   // wave-uniform: rows of an unaligned image differ in phase
                    const int p0 = 4 * l16 - (int)((cur_ph >> (2 * i)) & 3);
                    w0 = wgt(p0); w1 = wgt(p0 + 1); w2 = wgt(p0 + 2); w3 = wgt(p0 + 3);
                    mine = p0 < P.tw;
                }
                if (mine && 4 * i + sub < P.th) {
                    const uint32_t v = cur[i];
                    if (CLAHE_KO & 2) { if (v == 0x12345678u) hw[0] = 1; continue; }
                    atomicAdd(&hist[v & 0xFF], w0);
                    atomicAdd(&hist[(v >> 8) & 0xFF], w1);
                    atomicAdd(&hist[(v >> 16) & 0xFF], w2);
                    atomicAdd(&hist[v >> 24], w3);
                }
            }
        } else {
            for (int ly = sub; ly < P.th; ly += 4) {
                const uint8_t *row = img + (long long)c_reflect101(ty * P.th + ly, P.h) * P.stride;
                for (int lx = l16; lx < P.tw; lx += 16) atomicAdd(&hist[row[c_reflect101(x_begin + lx, P.w)]], 1u);
            }
        }
    };
    if (t < ntiles) {
        if (tile_fast(t)) tile_load(t, cur, cur_ph);
        histogram(t);
        if (t + tstride < ntiles && tile_fast(t + tstride)) tile_load(t + tstride, cur, cur_ph);
    }
#pragma nounroll
    for (; t < ntiles; t += tstride) {
        clahe_wave_sync();                                            // the atomics of tile t have landed
        // lane owns bins 4*lane .. 4*lane+3
        int hv[4];
        {
            c_u32x4 q = *(const c_u32x4 *)(hw + 4 * lane);
            for (int c = 1; c < ncopy; c++) q += *(const c_u32x4 *)(hw + c * CH_CSTRIDE + 4 * lane);
            hv[0] = (int)q.x; hv[1] = (int)q.y; hv[2] = (int)q.z; hv[3] = (int)q.w;
        }
        clahe_wave_sync();                                            // the histogram may be cleared for the next tile
        const int tn = t + tstride;
        if (tn < ntiles) {
            histogram(tn);
            if (tn + tstride < ntiles && tile_fast(tn + tstride)) tile_load(tn + tstride, cur, cur_ph);
        }
        if (P.clip > 0 && !(CLAHE_KO & 4)) {
            int over = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { over += max(hv[k] - P.clip, 0); hv[k] = min(hv[k], P.clip); }
            over += c_dpp0<0xB1>(over); over += c_dpp0<0x4E>(over); over += c_dpp0<0x124>(over); over += c_dpp0<0x128>(over);   // row totals
            const int clipped = __builtin_amdgcn_readlane(over, 0) + __builtin_amdgcn_readlane(over, 16) +
                                __builtin_amdgcn_readlane(over, 32) + __builtin_amdgcn_readlane(over, 48);
            const int batch = clipped >> 8, residual = clipped & 255;
            // serial loop `for (i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++` : bin gets +1 iff
            // bin % step == 0 && bin / step < residual, step = max(256 / residual, 1)
            int step = residual ? (int)(256.0f / (float)residual) : 1;          // exact: |256 - r n| >= 1
            int q = (int)((float)(4 * lane) / (float)step), r = 4 * lane - q * step;    // exact for the same reason
#pragma unroll
            for (int k = 0; k < 4; k++) {
                hv[k] += batch + ((residual != 0 && r == 0 && q < residual) ? 1 : 0);
                r++;
                if (r == step) { r = 0; q++; }
            }
        }
        // inclusive scan over the 256 bins: in-lane prefix, DPP scan inside the 16-lane rows, row offsets by readlane
        hv[1] += hv[0]; hv[2] += hv[1]; hv[3] += hv[2];
        int v = hv[3];
        v += c_dpp0<0x111>(v); v += c_dpp0<0x112>(v); v += c_dpp0<0x114>(v); v += c_dpp0<0x118>(v);      // row_shr:1,2,4,8 (zero fill)
        const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
        const int rowoff = sub == 0 ? 0 : (sub == 1 ? r0 : (sub == 2 ? r0 + r1 : r0 + r1 + r2));
        const int base = v + rowoff - hv[3];
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int r = __float2int_rn((float)(hv[k] + base) * P.lut_scale);   // saturate_cast<uchar>(float)
            r = r < 0 ? 0 : (r > 255 ? 255 : r);
            packed |= (uint32_t)r << (8 * k);
        }
        store_lut(t, packed);
    }
}

template <bool SRC_ALIGNED>
__global__ __launch_bounds__(256, 4) void k_clahe_lut(ClaheParams P, const uint8_t *__restrict__ src, uint8_t *__restrict__ lut)
{
    // four wavefronts = four neighbouring tiles per work-group (no barrier between them): tiles that share cache lines run at
    // the same time on one CU.  (One wavefront per work-group makes a bin's LDS address a single SDWA shift, but the tiles of a
    // row then run at different times and the kernel fetches 1.44x the bytes from HBM -- same 635 us, LDS-atomic bound either way.)
    __shared__ __attribute__((aligned(16))) uint32_t hist_all[4][CH_MULTI_DW];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // scalar: all tile arithmetic on the SALU
    const int ntiles = P.tiles_x * P.tiles_y;
    int b, bx;
    ov2_xcd_map(blockIdx.x, P.gx_lut, P.batch, &b, &bx);    // the work-groups of an image share image lines and its LUTs
    uint8_t *lb = lut + (long long)b * ntiles * 256 + 4 * lane;
    clahe_lut_tiles<SRC_ALIGNED>(P, src + (long long)b * P.src_item_stride, hist_all[wave], lane, bx * 4 + wave, ntiles, 4 * P.gx_lut,
                                 [&](int t, uint32_t packed) { *(uint32_t *)(lb + (long long)t * 256) = packed; }, CH_NCOPY);
}

// One workgroup per (row of interpolation cells, image): the rows whose two surrounding tile rows are
// (cy-1, cy).  LDS holds, for every cell column, the four surrounding tile LUTs packed as one dword per gray
// value.  A thread owns one dword column (4 pixels: horizontal weights and cell column are per-thread
// constants) and walks down the rows of the cell row: per pixel one LDS look-up, four byte->float
// conversions, the bilinear blend in packed fp32 (two pixels per v_pk_mul/add_f32, same operation order
// and rounding as the scalar formula), round-to-even and a byte insert.
#define CLAHE_MAX_CELLS 48
#define CA_UNROLL 6
typedef float c_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float c_ub(uint32_t q, int k) { return (float)((q >> (8 * k)) & 0xFFu); }    // v_cvt_f32_ubyteK

template <bool SRC_ALIGNED>
__global__ __launch_bounds__(512) void k_clahe_apply(ClaheParams P, const uint8_t *__restrict__ src, const uint8_t *__restrict__ lut,
                                                     uint8_t *__restrict__ dst)
{
    extern __shared__ __align__(16) unsigned char clahe_smem[];
    const int ncx = P.tiles_x + 1;
    uint32_t *lut4 = (uint32_t *)clahe_smem;                            // ncx * 256
    int cy, b;
    ov2_xcd_map(blockIdx.x, (P.tiles_y + 1) * P.ysplit, P.batch, &b, &cy);
    const int part = cy % P.ysplit;
    cy /= P.ysplit;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int ty1 = max(cy - 1, 0), ty2 = min(cy, P.tiles_y - 1);
    const uint8_t *L = lut + (long long)b * P.tiles_x * P.tiles_y * 256;
    // stage: item = (cell column, 4 consecutive gray values): four dword loads, a 4x4 byte transpose, one b128 store
    for (int e = tid; e < ncx * 64; e += nthr) {
        const int cx = e >> 6, v4 = (e & 63) * 4;
        const int tx1 = max(cx - 1, 0), tx2 = min(cx, P.tiles_x - 1);
        const uint32_t a = *(const uint32_t *)(L + (ty1 * P.tiles_x + tx1) * 256 + v4), bb = *(const uint32_t *)(L + (ty1 * P.tiles_x + tx2) * 256 + v4);
        const uint32_t c = *(const uint32_t *)(L + (ty2 * P.tiles_x + tx1) * 256 + v4), d = *(const uint32_t *)(L + (ty2 * P.tiles_x + tx2) * 256 + v4);
        // entry k = a.byte[k] | b.byte[k] << 8 | c.byte[k] << 16 | d.byte[k] << 24
        const uint32_t ab01 = __builtin_amdgcn_perm(bb, a, 0x05010400u), ab23 = __builtin_amdgcn_perm(bb, a, 0x07030602u);   // (a0 b0 a1 b1), (a2 b2 a3 b3)
        const uint32_t cd01 = __builtin_amdgcn_perm(d, c, 0x05010400u), cd23 = __builtin_amdgcn_perm(d, c, 0x07030602u);
        c_u32x4 o;
        o.x = __builtin_amdgcn_perm(cd01, ab01, 0x05040100u); o.y = __builtin_amdgcn_perm(cd01, ab01, 0x07060302u);
        o.z = __builtin_amdgcn_perm(cd23, ab23, 0x05040100u); o.w = __builtin_amdgcn_perm(cd23, ab23, 0x07060302u);
        *(c_u32x4 *)(lut4 + (cx << 8) + v4) = o;
    }
    __syncthreads();
    // rows of this cell row: fy + 1 == cy with fy = floor(y * inv_th - 0.5), decided by the float formula like the reference
    int y0 = max(0, (int)floorf(((float)cy - 0.5f) * (float)P.th) - 1), y1 = min(P.h, (int)ceilf(((float)cy + 0.5f) * (float)P.th) + 2);
    while (y0 < y1 && (int)floorf((float)y0 * P.inv_th - 0.5f) + 1 != cy) y0++;
    while (y1 > y0 && (int)floorf((float)(y1 - 1) * P.inv_th - 0.5f) + 1 != cy) y1--;
    if (P.ysplit > 1) {                                                 // this work-group's band of the cell row
        const int chunk = (y1 - y0 + P.ysplit - 1) / P.ysplit;
        y0 = min(y1, y0 + part * chunk);
        y1 = min(y1, y0 + chunk);
    }
    const uint8_t *simg = src + (long long)b * P.src_item_stride;
    uint8_t *dimg = dst + (long long)b * P.dst_item_stride;
    // the source may have any alignment (KITTI: 1241-byte rows): rows are read as aligned dwords + v_alignbyte;
    // only the destination decides between dword and byte stores
    const bool dst_aligned = ((P.dst_stride | (int)(size_t)dimg) & 3) == 0;
    constexpr bool src_aligned = SRC_ALIGNED;
    const uint8_t *simg_end = simg + (long long)(P.h - 1) * P.stride + P.w;    // one past the last source pixel
    const int ndw = (P.w + 3) >> 2;
    for (int dwi = tid; dwi < ndw; dwi += nthr) {
        const int xb = 4 * dwi;
        float xa[4], xa1[4];
        const uint32_t *lutc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = min(xb + k, P.w - 1);
            const float txf = (float)x * P.inv_tw - 0.5f;
            const int fx = (int)floorf(txf);
            xa[k] = txf - (float)fx; xa1[k] = 1.0f - xa[k];
            lutc[k] = lut4 + ((fx + 1) << 8);
        }
        const c_f32x2 XA01 = {xa[0], xa[1]}, XA23 = {xa[2], xa[3]}, XB01 = {xa1[0], xa1[1]}, XB23 = {xa1[2], xa1[3]};
        const bool full = dst_aligned && xb + 3 < P.w;
        // the loads of the NEXT CA_UNROLL rows are issued before the current ones are consumed: with ~8 waves per
        // SIMD the loaded HBM round trip (several us at this traffic) is not covered by the other waves alone
        auto load_rows = [&](int yb, uint32_t (&inr)[CA_UNROLL]) {
#pragma unroll
            for (int u = 0; u < CA_UNROLL; u++) {
                const int y = min(yb + u, y1 - 1);
                const uint8_t *sp = simg + (long long)y * P.stride + xb;
                uint32_t in = 0;
                if (CLAHE_KO & 8) in = *(const uint32_t *)(simg + (tid & 15) * 4);
                else
                // aligned rows: the dword at xb lies inside the row's stride even when w % 4 != 0 (stride % 4 == 0 > w); no
                // branch here, so that the compiler can count the loads in flight (vmcnt(CA_UNROLL) instead of vmcnt(0))
                if (src_aligned) in = *(const uint32_t *)sp;
                else {
                    const uint32_t ph = (uint32_t)((size_t)sp & 3);
                    const uint8_t *ap = sp - ph;
                    if (xb + 3 < P.w && ap + 8 <= simg_end) {
                        const uint32_t lo = *(const uint32_t *)ap, hi = *(const uint32_t *)(ap + 4);
                        in = __builtin_amdgcn_alignbyte(hi, lo, ph);
                    } else for (int k = 0; k < 4; k++) if (xb + k < P.w) in |= (uint32_t)sp[k] << (8 * k);
                }
                inr[u] = in;
            }
        };
        uint32_t inr[CA_UNROLL], nxt[CA_UNROLL];
        if (y0 < y1) load_rows(y0, inr);
        for (int yb = y0; yb < y1; yb += CA_UNROLL) {
            load_rows(yb + CA_UNROLL, nxt);                         // unconditional (row index clamped): one path, exact vmcnt
#pragma unroll
            for (int u = 0; u < CA_UNROLL; u++) {
                const int y = yb + u;
                if (y >= y1) break;
                const float tyf = (float)y * P.inv_th - 0.5f;
                const float ya = tyf - (float)(cy - 1), ya1 = 1.0f - ya;
                const c_f32x2 YA = {ya, ya}, YB = {ya1, ya1};
                uint8_t *drow = dimg + y * P.dst_stride;
                const uint32_t in = inr[u];
                uint32_t q0, q1, q2, q3;
                if (CLAHE_KO & 32) { q0 = in; q1 = in >> 1; q2 = in >> 2; q3 = in >> 3; }
                else { q0 = lutc[0][in & 0xFF]; q1 = lutc[1][(in >> 8) & 0xFF]; q2 = lutc[2][(in >> 16) & 0xFF]; q3 = lutc[3][in >> 24]; }
                // res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya, pixels (0,1) and (2,3) side by side
                const c_f32x2 A11 = {c_ub(q0, 0), c_ub(q1, 0)}, A12 = {c_ub(q0, 1), c_ub(q1, 1)}, A21 = {c_ub(q0, 2), c_ub(q1, 2)}, A22 = {c_ub(q0, 3), c_ub(q1, 3)};
                const c_f32x2 B11 = {c_ub(q2, 0), c_ub(q3, 0)}, B12 = {c_ub(q2, 1), c_ub(q3, 1)}, B21 = {c_ub(q2, 2), c_ub(q3, 2)}, B22 = {c_ub(q2, 3), c_ub(q3, 3)};
                const c_f32x2 r01 = (A11 * XB01 + A12 * XA01) * YB + (A21 * XB01 + A22 * XA01) * YA;
                const c_f32x2 r23 = (B11 * XB23 + B12 * XA23) * YB + (B21 * XB23 + B22 * XA23) * YA;
                // cvRound + saturate_cast<uchar>: round to nearest even, then a saturating byte insert (exact on integers)
                uint32_t out = __builtin_amdgcn_cvt_pk_u8_f32(rintf(r01.x), 0, 0u);
                out = __builtin_amdgcn_cvt_pk_u8_f32(rintf(r01.y), 1, out);
                out = __builtin_amdgcn_cvt_pk_u8_f32(rintf(r23.x), 2, out);
                out = __builtin_amdgcn_cvt_pk_u8_f32(rintf(r23.y), 3, out);
                if (CLAHE_KO & 64) out = q0 ^ q1 ^ q2 ^ q3;
                if ((CLAHE_KO & 16) && out != 0x12345678u) continue;
                if (full) *(uint32_t *)(drow + xb) = out;
                else for (int k = 0; k < 4; k++) if (xb + k < P.w) drow[xb + k] = (uint8_t)(out >> (8 * k));
            }
#pragma unroll
            for (int u = 0; u < CA_UNROLL; u++) inr[u] = nxt[u];
        }
    }
    if (P.border > 0) {
        // REFLECT_101 border of the padded destination (pyramid level 0), written by the workgroup that owns
        // the mirrored source rows while they are still in L2: left/right dwords of rows [y0, y1), and the
        // mirror images of rows 1..border / h-1-border..h-2 above / below the image (full padded width).
        __threadfence_block();
        __syncthreads();
        const int win = P.border, pbl = (win + 3) & ~3, wend = (P.w + win + 3) & ~3, rbeg = P.w & ~3;
        const int ndw_lr = pbl / 4 + ((wend - rbeg) >> 2), ndw_row = (pbl + wend) >> 2;
        auto put = [&](int yt, int ysrc, int c0) {
            const uint8_t *srow = dimg + ysrc * P.dst_stride;
            uint32_t v;
            if (c0 >= 0 && c0 + 3 < P.w) v = *(const uint32_t *)(srow + c0);
            else {
                v = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int c = c0 + k;
                    c = c < -win ? -win : (c > P.w + win - 1 ? P.w + win - 1 : c);       // padding bytes: any value
                    v |= (uint32_t)srow[c_reflect101(c, P.w)] << (8 * k);
                }
            }
            *(uint32_t *)(dimg + yt * P.dst_stride + c0) = v;
        };
        // left/right: 8 threads per row (ndw_lr <= 8 for win <= 15; wider borders take more passes over d)
        for (int row = tid >> 3; row < y1 - y0; row += nthr >> 3)
            for (int d = tid & 7; d < ndw_lr; d += 8)
                put(y0 + row, y0 + row, d < pbl / 4 ? 4 * d - pbl : rbeg + 4 * (d - pbl / 4));
        // rows above the image mirror rows 1..win, rows below mirror h-2..h-1-win: only the owners of those rows work
        for (int k = max(1, y0); k <= min(win, y1 - 1); k++)
            for (int d = tid; d < ndw_row; d += nthr) put(-k, k, 4 * d - pbl);
        for (int k = max(1, P.h - y1); k <= min(win, P.h - 1 - y0); k++)
            for (int d = tid; d < ndw_row; d += nthr) put(P.h - 1 + k, P.h - 1 - k, 4 * d - pbl);
    }
}

// ---- strip kernel: CLAHE apply + pyramid level 1 + both REFLECT_101 borders in one walk (batch mode) -------------------
// One work-group per image, one WAVEFRONT per column strip, walking the image top to bottom.  A lane owns one dword column
// (4 pixels) like k_clahe_apply; the strip's first / last lane is a halo column (it computes the same dword as the neighbouring
// strip's core lane and stores that identical copy) so that every core lane finds its horizontal neighbours in the lanes next
// to it (v_mov_b32_dpp wave_shr:1 / wave_shl:1) -- 752 pixels = 188 dwords = 63 + 62 + 63 core columns + 4 halo columns =
// 3 x 64 lanes.  What the separate kernels re-read from HBM stays in registers here:
//   * level 0 is stored once (plus its mirror rows / border dwords, built from the lane's and its neighbour's dword);
//   * pyrDown: the 5-tap horizontal sums of a row (v_dot4_u32_u8 on the lane's dword and its neighbours') roll through a
//     five-row register window, every second row emits two level-1 pixels per lane (same integer arithmetic as k_pyr_level);
//   * the packed four-LUT table of all cell columns is staged once per image and row of cells and shared by the strips
//     (two barriers per row of cells -- the only work-group synchronisation), not once per 192-thread work-group.
// Width: any (round 4; KITTI's 1241).  With r = w % 4 != 0 the last dword column holds r image pixels; its other bytes ARE border
// pixels (x >= w mirrors 2 (w - 1) - x), so the last lane patches its dword from its own and its left neighbour's bytes and stores
// it whole; the mirror axis of the right border then sits mid-dword, which moves every right-border dword's source window by 2 r
// bytes (one r-dependent byte selector) and adds one border lane.  Level 1 likewise when its width (w + 1) / 2 is odd.  Source rows
// may be unaligned (hardware handles unaligned dword loads).
// Geometry: dword-aligned destinations, h >= 8, at most CS_MAX_STRIPS strips, tiles_x + 1 <= 40 (host-checked; anything else takes
// the separate kernels).  tests/test_gpu_clahe.py: named geometries vs the oracle + a random sweep vs the
// separate kernels.
__device__ __forceinline__ uint32_t c_wave_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true); }   // lane i <- lane i-1
__device__ __forceinline__ uint32_t c_wave_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true); }   // lane i <- lane i+1

#define CS_UNROLL 6
#ifndef CS_KO
#define CS_KO 0           // knock-out timing experiments (tools/build_variant.sh; profiles/archive/r4_strip_kernel_knockouts.txt): 1 loads hit one
#endif                    // row, 2 no level-0 stores, 4 no level-1 stores, 8 no LUT look-ups, 16 no blend, 32 no pyrDown sums / level-1
                          // rows, 64 / 128 no border stores of level 0 / 1, 256 distinct slack dwords, 512 / 1024 (fused form) the LUT
                          // wavefront / the shared first row of tiles computes nothing
#define CS_MAX_STRIPS 8
// UNAL: w % 4 != 0 or an odd level-1 width (the generic right edge); false keeps the w % 4 == 0 instance free of its per-lane
// selectors (they cost three spilled registers at this kernel's 80-VGPR budget)
// FUSED (round 4): the work-group has one more wavefront, which computes the tile LUTs (clahe_lut_tiles: histogram, clip, scan) of the
// NEXT row of tiles while the strips walk the current row of cells -- the LDS-atomic-bound half of CLAHE runs beside the store- and
// latency-bound half on the same CU, the LUTs never leave LDS (a ring of two rows of tiles), the source rows are fetched from HBM once
// (the strips find them in L2 half a tile later), and k_clahe_lut's launch is gone.  The first row of tiles is shared by all wavefronts
// before the walk starts.  Every wavefront passes the same 2 (cell rows) barriers.
template <bool UNAL, bool FUSED>
__global__ __launch_bounds__(64 * (CS_MAX_STRIPS + 2), 6) void k_clahe_apply_pyr(ClaheParams P, const uint8_t *__restrict__ src, const uint8_t *__restrict__ lut,
                                                                                uint8_t *__restrict__ dst)
{
    extern __shared__ __align__(16) unsigned char clahe_smem[];
    uint32_t *lut4 = (uint32_t *)clahe_smem;                            // (tiles_x + 1 cell columns) x 256 packed LUT quadruples
    uint8_t *ring = clahe_smem + (size_t)(P.tiles_x + 1) * 1024;          // FUSED: LUT bytes of two rows of tiles (row ty in slot ty & 1)
    // one work-group per image, one wavefront per column strip: the strips share the LUT table (two barriers per row of cells),
    // everything else is wave-private
    const int b = blockIdx.x, s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, ndw = (P.w + 3) >> 2, wr = UNAL ? (P.w & 3) : 0;
    // FUSED: this wavefront's histogram -- one copy for the strips (they histogram the first row of tiles only), CH_NCOPY for the LUT wavefronts
    uint32_t *hist_w = (uint32_t *)(ring + (size_t)2 * P.tiles_x * 256) + (s < P.nstrips ? s * CH_WAVE_DW : P.nstrips * CH_WAVE_DW + (s - P.nstrips) * CH_MULTI_DW);
    const int hist_copies = s < P.nstrips ? 1 : CH_NCOPY;
    auto lut_row = [&](int ty, int tx0, int txstep) {                   // FUSED: LUTs of the tiles tx0, tx0 + txstep, .. of tile row ty
        uint8_t *slot = ring + (size_t)(ty & 1) * P.tiles_x * 256 - (size_t)ty * P.tiles_x * 256 + 4 * lane;
        clahe_lut_tiles<!UNAL>(P, src + (long long)b * P.src_item_stride, hist_w, lane, ty * P.tiles_x + tx0, (ty + 1) * P.tiles_x, txstep,
                               [&](int t, uint32_t packed) { *(uint32_t *)(slot + (size_t)t * 256) = packed; }, hist_copies);
    };
    typedef uint32_t cs_u32_a1 __attribute__((aligned(1)));             // source rows of any alignment (KITTI: 1241-byte rows)
    const int cmin = 0, ncell = P.tiles_x + 1;
    const uint8_t *L = lut + (long long)b * P.tiles_x * P.tiles_y * 256;
    auto cell_row = [&](int y) { return (int)floorf((float)y * P.inv_th - 0.5f) + 1; };
    int cy = -1, y_switch = 0;                                          // rows [.., y_switch) belong to the row of cells cy
    auto build = [&](int cyv) {                                         // the packed table of the row of cells cyv, by the whole work-group
        const int ty1 = max(cyv - 1, 0), ty2 = min(cyv, P.tiles_y - 1);
        const uint8_t *L1 = FUSED ? ring + (size_t)(ty1 & 1) * P.tiles_x * 256 : L + (size_t)ty1 * P.tiles_x * 256;
        const uint8_t *L2 = FUSED ? ring + (size_t)(ty2 & 1) * P.tiles_x * 256 : L + (size_t)ty2 * P.tiles_x * 256;
        __syncthreads();                                                // the previous table's look-ups are done (FUSED: and tile row cyv's LUTs)
        for (int e = threadIdx.x; e < ncell * 64; e += blockDim.x) {
            const int cx = cmin + (e >> 6), v4 = (e & 63) * 4;
            const int tx1 = max(cx - 1, 0), tx2 = min(cx, P.tiles_x - 1);
            const uint32_t a = *(const uint32_t *)(L1 + tx1 * 256 + v4), bb = *(const uint32_t *)(L1 + tx2 * 256 + v4);
            const uint32_t c = *(const uint32_t *)(L2 + tx1 * 256 + v4), dd = *(const uint32_t *)(L2 + tx2 * 256 + v4);
            const uint32_t ab01 = __builtin_amdgcn_perm(bb, a, 0x05010400u), ab23 = __builtin_amdgcn_perm(bb, a, 0x07030602u);
            const uint32_t cd01 = __builtin_amdgcn_perm(dd, c, 0x05010400u), cd23 = __builtin_amdgcn_perm(dd, c, 0x07030602u);
            c_u32x4 o;
            o.x = __builtin_amdgcn_perm(cd01, ab01, 0x05040100u); o.y = __builtin_amdgcn_perm(cd01, ab01, 0x07060302u);
            o.z = __builtin_amdgcn_perm(cd23, ab23, 0x05040100u); o.w = __builtin_amdgcn_perm(cd23, ab23, 0x07060302u);
            *(c_u32x4 *)(lut4 + ((e >> 6) << 8) + v4) = o;
        }
        __syncthreads();
    };
    if (FUSED) {
        if (!(CS_KO & 1024)) lut_row(0, s, P.nstrips + P.nlut);              // the first row of tiles: all wavefronts
        if (s >= P.nstrips) {                                           // the LUT wavefront: one row of tiles ahead of the strips
            const int n_cell_rows = cell_row(P.h - 1) + 1;              // (the strips stage every row of cells once, in order)
#ifndef CS_LUT_PRIO
#define CS_LUT_PRIO 3
#endif
            __builtin_amdgcn_s_setprio(CS_LUT_PRIO);                    // one wavefront against 3: it must not wait for issue slots
            for (int c = 0; c < n_cell_rows; c++) {
                build(c);
                if (c + 1 < P.tiles_y && !(CS_KO & 512)) lut_row(c + 1, s - P.nstrips, P.nlut);
            }
            return;
        }
    }
    // core columns of strip s: 63 (first), 62 (middle), <= 63 (last); lane 0 of every strip but the first is the left halo
    const int core0 = s == 0 ? 0 : 63 + 62 * (s - 1);
    const int core1 = s == P.nstrips - 1 ? ndw : 63 + 62 * s;
    const int d_first = s == 0 ? 0 : core0 - 1;
    const int d_raw = d_first + lane;
    const int d = min(d_raw, ndw - 1);                                  // lanes beyond the image recompute (and re-store) the last column
    const bool core = d_raw >= core0 && d_raw < core1;
    const int xb = 4 * d;
    float xa[4], xa1[4];
    const uint32_t *lutc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float txf = (float)min(xb + k, P.w - 1) * P.inv_tw - 0.5f;     // (bytes beyond the image are patched below: any table)
        const int fx = (int)floorf(txf);
        xa[k] = txf - (float)fx; xa1[k] = 1.0f - xa[k];
        lutc[k] = lut4 + ((fx + 1 - cmin) << 8);
    }
    const c_f32x2 XA01 = {xa[0], xa[1]}, XA23 = {xa[2], xa[3]}, XB01 = {xa1[0], xa1[1]}, XB23 = {xa1[2], xa1[3]};
    uint8_t *dimg = dst + (long long)b * P.dst_item_stride;
    uint8_t *l1 = dimg + P.l1_delta;
    const int win = P.border;
    // Border columns this lane writes besides its own: level 0 in dwords, level 1 in pixel pairs -- pixel -k is pixel k, pixel
    // w - 1 + k is pixel w - 1 - k.  The kernel is bound by the SCALAR unit if one lets it (a wavefront walking 480 rows does
    // all its row bookkeeping on the SALU, which issues once per four cycles like the VALU): so no store is predicated -- halo
    // lanes store their (identical) copy of the neighbouring strip's dword, lanes without a border column / level-1 pair aim
    // at the row's alignment slack right of the border (never consumed) -- row pointers advance by one scalar add, and the
    // switch to the next row of cells is one scalar compare against a precomputed row number.
    const int l1odd = UNAL ? (P.l1_w & 1) : 0;
    const int nb0 = (win + 3) >> 2, nb1 = (win + 1) >> 1;              // border lanes on the left; on the right one more when the
    const int nbr0 = nb0 + (wr ? 1 : 0), nbr1 = nb1 + l1odd;            // mirror axis sits inside the last dword / pixel pair
    const bool bl0 = core && d < nb0, br0 = core && d >= ndw - nbr0;
    const bool bl1 = core && d >= 1 && d <= nb1, br1 = core && d >= ndw - nbr1;
    const bool edge_strip = s == 0 || core1 > ndw - max(nbr0, nbr1);    // (uniform) only these strips hold border lanes
    const bool first_col = d_raw == 0, last_col = d_raw == ndw - 1;
    const bool dup = d_raw > ndw - 1;                                   // lanes beyond the image recompute the last column: their stores go to the slack
    const uint32_t pad0 = 4u * nb0, pad1 = 2u * nb1;
    // stores of lanes without a border column: the alignment slack right of the border (never consumed, inside the row pitch)
    const uint32_t slack0 = pad0 + 4u * (uint32_t)((P.w + win + 3) >> 2) + ((CS_KO & 256) ? 4u * lane : 0u), slack1 = pad1 + 2u * (uint32_t)((P.l1_w + win + 3) >> 1) + ((CS_KO & 256) ? 2u * lane : 0u);
    const uint32_t o0 = dup ? slack0 : pad0 + 4u * d;
    // right border dword of this lane: pixels p[top], p[top-1], p[top-2], p[top-3] at x0 = 2 (w - 1) - top, top = xb + 2 (w even)
    // or xb (w odd): the choice that makes x0 a multiple of 4; both windows lie inside (left neighbour's dword, own dword)
    const uint32_t o0b = bl0 ? pad0 - 4u * (d + 1) : (br0 ? pad0 + (uint32_t)(2 * P.w - 2 - xb - ((wr & 1) ? 0 : 2)) : slack0);
    const uint32_t sel_br = (wr & 1) ? 0x01020304u : 0x03040506u;
    // the last column's dword = its wr image pixels followed by the border pixels p[w], p[w+1].. = p[w-2], p[w-3]..; identity elsewhere
    const uint32_t sel_fix = (last_col && wr) ? (wr == 1 ? 0x01020304u : (wr == 2 ? 0x03040504u : 0x05060504u)) : 0x07060504u;
    // p[xb + 4] of the last column (byte 0): the mirror image of pixel 2 (w - 1) - xb - 4
    const uint32_t sel_rt = wr == 0 ? 0x0c0c0c06u : (wr == 1 ? 0x0c0c0c00u : (wr == 2 ? 0x0c0c0c02u : 0x0c0c0c04u));
    const uint32_t o1 = core ? pad1 + 2u * d : slack1;
    const uint32_t o1b = bl1 ? pad1 - 2u * d : (br1 ? pad1 + (uint32_t)P.l1_w + 2u * (ndw - 1 - d) - (uint32_t)l1odd : slack1);
    const bool fix1 = last_col && l1odd;                               // the last pair = (last level-1 pixel, first border pixel)
    uint8_t *d0m = dimg - pad0, *d1m = l1 - pad1;
    // source dword of this lane: the last column of a width that is no multiple of 4 reads the row's last four bytes and shifts
    const uint32_t xo = (d == ndw - 1 && wr) ? (uint32_t)(P.w - 4) : (uint32_t)xb;
    const uint32_t in_sh = (d == ndw - 1 && wr) ? 8u * (4u - (uint32_t)wr) : 0u;

    auto stage = [&](int y) {
        cy = cell_row(y);
        int yn = max(y + 1, (int)(((float)cy + 0.5f) * (float)P.th) - 2);
        while (yn < P.h && cell_row(yn) == cy) yn++;
        y_switch = yn;
        build(cy);
    };
    // a level-0 row: every lane's dword and, in the edge strips, the border dword
#ifndef CS_BUF
#define CS_BUF 1          // 0: 64-bit lane addresses (the form the CS_KO store knock-outs are written for)
#endif
#if CS_BUF
    // buffer addressing (round 4): one descriptor per image, the row as a scalar offset, the lane's column as a 32-bit offset -- no
    // 64-bit address arithmetic per store (the kernel's time is within noise of the pointer form, 1 % in the median; it needs 6-8
    // registers less, and the instance for unaligned widths no longer spills)
    uint8_t *dbase = min(d0m - (long long)win * P.dst_stride, d1m - (long long)win * P.l1_pitch) - 256;
    const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(dbase, 0, 0x7fffffff, 0x00020000);
    const uint8_t *sbase = src + (long long)b * P.src_item_stride;
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void *)sbase, 0, 0x7fffffff, 0x00020000);
    auto put0 = [&](uint8_t *row, uint32_t v, uint32_t bv) {
        const int so = (int)(row - dbase);
        __builtin_amdgcn_raw_buffer_store_b32(v, rs_dst, (int)o0, so, 0);
        if (edge_strip) __builtin_amdgcn_raw_buffer_store_b32(bv, rs_dst, (int)o0b, so, 0);
    };
    auto put1 = [&](uint8_t *row, uint32_t v, uint32_t bv) {
        const int so = (int)(row - dbase);
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, rs_dst, (int)o1, so, 0);
        if (edge_strip) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bv, rs_dst, (int)o1b, so, 0);
    };
#else
    auto put0 = [&](uint8_t *row, uint32_t v, uint32_t bv) {
        if ((CS_KO & 2) && v != 0x12345678u) return;
        *(uint32_t *)(row + o0) = v;
        if (edge_strip && !(CS_KO & 64)) *(uint32_t *)(row + o0b) = bv;
    };
    auto put1 = [&](uint8_t *row, uint32_t v, uint32_t bv) {               // a level-1 row: pixel pairs
        if ((CS_KO & 4) && v != 0x12345678u) return;
        *(uint16_t *)(row + o1) = (uint16_t)v;
        if (edge_strip && !(CS_KO & 128)) *(uint16_t *)(row + o1b) = (uint16_t)bv;
    };
#endif
    typedef unsigned short cu16x2 __attribute__((ext_vector_type(2)));
    uint8_t *r1 = d1m;                                                  // level-1 row of the next emit
    // level-1 row Y from the horizontal sums of level-0 rows 2Y-2 .. 2Y+2 (packed pairs, <= 16 * 255 each); rows are emitted in order.
    // The REFLECT_101 border mirrors rows 1 .. win above the image and rows h1-1-win .. h1-2 below it: the same stores once more
    auto emit = [&](int Y, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t c4, auto mirror_c) {
        constexpr bool MIRROR = decltype(mirror_c)::value;
        if (CS_KO & 32) { if (c4 == 0x12345678u) *(uint32_t *)r1 = c4; return; }
        const cu16x2 s0 = __builtin_bit_cast(cu16x2, c0), s1 = __builtin_bit_cast(cu16x2, c1), s2 = __builtin_bit_cast(cu16x2, c2),
                     s3 = __builtin_bit_cast(cu16x2, c3), s4 = __builtin_bit_cast(cu16x2, c4);
        const cu16x2 v = (s2 * (unsigned short)6 + (s1 + s3) * (unsigned short)4 + s0 + s4 + (unsigned short)128) >> (unsigned short)8;
        uint32_t o = __builtin_amdgcn_perm(0u, __builtin_bit_cast(uint32_t, v), 0x0c0c0200u);       // (v0, v1) as two bytes
        uint32_t bv = 0;
        if (edge_strip) {
            bv = __builtin_amdgcn_perm(c_wave_shr1(o), o, 0x0c0c0500u);    // border pair: (own first pixel, left neighbour's second)
            if (UNAL && fix1) o = bv;                                              // odd level-1 width: the last pair's second pixel is border
        }
        put1(r1, o, bv);
        r1 += P.l1_pitch;
        if (MIRROR) {
            if (Y >= 1 && Y <= win) put1(d1m - (long long)Y * P.l1_pitch, o, bv);
            if (Y >= P.l1_h - 1 - win && Y <= P.l1_h - 2) put1(d1m + (long long)(2 * (P.l1_h - 1) - Y) * P.l1_pitch, o, bv);
        }
    };

    uint32_t hA = 0, hB = 0, hC = 0, hD = 0, hE = 0;                      // horizontal sums of rows y-4 .. y
    const uint8_t *srow = src + (long long)b * P.src_item_stride;        // (uniform) source row of the next prefetch
    uint8_t *r0 = d0m;                                                  // level-0 row of the next store
    uint32_t inr[CS_UNROLL], nxt[CS_UNROLL];
#pragma unroll
    for (int u = 0; u < CS_UNROLL; u++) inr[u] = *(const cs_u32_a1 *)(srow + (long long)min(u, P.h - 1) * P.stride + xo) >> (UNAL ? in_sh : 0u);
    srow += (long long)CS_UNROLL * P.stride;
    // one row: CLAHE blend of the lane's four pixels, level-0 stores, horizontal pyrDown sums into the rolling window.
    // MIRROR: the row may be one the REFLECT_101 border mirrors (-y for 1 <= y <= win, 2 (h-1) - y for h-1-win <= y <= h-2)
    auto do_row = [&](int y, uint32_t in, auto mirror_c) {
        constexpr bool MIRROR = decltype(mirror_c)::value;
        const float tyf = (float)y * P.inv_th - 0.5f;
        const float ya = tyf - (float)(cy - 1), ya1 = 1.0f - ya;
        const c_f32x2 YA = {ya, ya}, YB = {ya1, ya1};
        uint32_t q0, q1, q2, q3;
        if (CS_KO & 8) { q0 = in * 0x01010101u; q1 = in ^ 0x55u; q2 = in + 0x01020304u; q3 = in >> 3; }
        else { q0 = lutc[0][in & 0xFF]; q1 = lutc[1][(in >> 8) & 0xFF]; q2 = lutc[2][(in >> 16) & 0xFF]; q3 = lutc[3][in >> 24]; }
        const c_f32x2 A11 = {c_ub(q0, 0), c_ub(q1, 0)}, A12 = {c_ub(q0, 1), c_ub(q1, 1)}, A21 = {c_ub(q0, 2), c_ub(q1, 2)}, A22 = {c_ub(q0, 3), c_ub(q1, 3)};
        const c_f32x2 B11 = {c_ub(q2, 0), c_ub(q3, 0)}, B12 = {c_ub(q2, 1), c_ub(q3, 1)}, B21 = {c_ub(q2, 2), c_ub(q3, 2)}, B22 = {c_ub(q2, 3), c_ub(q3, 3)};
        const c_f32x2 r01 = (A11 * XB01 + A12 * XA01) * YB + (A21 * XB01 + A22 * XA01) * YA;
        const c_f32x2 r23 = (B11 * XB23 + B12 * XA23) * YB + (B21 * XB23 + B22 * XA23) * YA;
        // cvRound + saturate_cast<uchar>: 0 <= res < 255.5 (a convex combination of bytes), so adding 1.5 * 2^23 rounds to the
        // nearest even integer in the float's low mantissa byte -- two packed adds and three byte permutes for four pixels
        const c_f32x2 MAGIC = {12582912.0f, 12582912.0f};
        const c_f32x2 m01 = r01 + MAGIC, m23 = r23 + MAGIC;
        const uint32_t u0 = __builtin_bit_cast(uint32_t, (float)m01.x), u1 = __builtin_bit_cast(uint32_t, (float)m01.y);
        const uint32_t u2 = __builtin_bit_cast(uint32_t, (float)m23.x), u3 = __builtin_bit_cast(uint32_t, (float)m23.y);
        uint32_t out = __builtin_amdgcn_perm(__builtin_amdgcn_perm(u3, u2, 0x0c0c0400u), __builtin_amdgcn_perm(u1, u0, 0x0c0c0400u), 0x05040100u);
        if (CS_KO & 16) out = q0 + q1 + q2 + q3;
        // neighbours: pixels xb-2, xb-1 (left lane's bytes 2, 3) and xb+4 (right lane's byte 0); REFLECT_101 at the image edge
        uint32_t lf = c_wave_shr1(out), rt = c_wave_shl1(out);
        if (first_col) lf = __builtin_amdgcn_perm(out, out, 0x01020000u);      // (.., .., p2, p1)
        if (edge_strip) {
            if (UNAL) out = __builtin_amdgcn_perm(out, lf, sel_fix);           // (identity except in the last column of a width % 4 != 0)
        }
        if (UNAL) { if (last_col) rt = __builtin_amdgcn_perm(out, lf, sel_rt); }   // p[xb + 4] mirrored about w - 1
        else if (last_col) rt = out >> 16;                                      // p(w) = p(w-2)
        uint32_t bv = 0;                                                        // border dword of the edge lanes:
        if (edge_strip)                                                         // (p[4d+4], p[4d+3], p[4d+2], p[4d+1]) left, p[top .. top-3] right
            bv = bl0 ? __builtin_amdgcn_perm(out, rt, 0x05060700u) : __builtin_amdgcn_perm(out, lf, sel_br);
        put0(r0, out, bv);
        r0 += P.dst_stride;
        if (MIRROR) {
            if (y >= 1 && y <= win) put0(d0m - (long long)y * P.dst_stride, out, bv);
            if (y >= P.h - 1 - win && y <= P.h - 2) put0(d0m + (long long)(2 * (P.h - 1) - y) * P.dst_stride, out, bv);
        }
        // pyrDown, horizontal: sums centred on pixels xb and xb + 2
        if (CS_KO & 32) { hE += out; return; }
        const uint32_t l0 = __builtin_amdgcn_alignbyte(out, lf, 2);
        const uint32_t h0 = __builtin_amdgcn_udot4(l0, 0x04060401u, __builtin_amdgcn_udot4(out, 0x00010000u, 0u, false), false);
        const uint32_t h1 = __builtin_amdgcn_udot4(out, 0x04060401u, __builtin_amdgcn_udot4(rt, 0x00000001u, 0u, false), false);
        hA = hB; hB = hC; hC = hD; hD = hE; hE = h0 | (h1 << 16);
    };
    // Rows go in groups of six (the prefetch distance).  Almost every group is FAST: no row of it is mirrored by a border, the
    // table of its row of cells is already staged, all six rows exist -- straight-line code, so that the look-ups and loads of
    // one row overlap the arithmetic of another.  The few others (top / bottom of the image, a switch of tables inside the
    // group) take the rolled loop below with every test per row.
    for (int yb = 0; yb < P.h; yb += CS_UNROLL) {
        if (yb + 2 * CS_UNROLL <= P.h) {
#pragma unroll
#if CS_BUF
            for (int u = 0; u < CS_UNROLL; u++) { nxt[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_src, (int)xo, (int)(srow - sbase), 0) >> (UNAL ? in_sh : 0u); srow += P.stride; }
#else
            for (int u = 0; u < CS_UNROLL; u++) { nxt[u] = *(const cs_u32_a1 *)(srow + xo) >> (UNAL ? in_sh : 0u); if (!(CS_KO & 1)) srow += P.stride; }
#endif
        } else {
#pragma unroll
            for (int u = 0; u < CS_UNROLL; u++)
                nxt[u] = *(const cs_u32_a1 *)(src + (long long)b * P.src_item_stride + (long long)min(yb + CS_UNROLL + u, P.h - 1) * P.stride + xo) >> (UNAL ? in_sh : 0u);
        }
        if (yb == y_switch) stage(yb);
        // (level-0 rows yb .. yb+5 and the level-1 rows (yb-2)/2 .. (yb+2)/2 they complete are not mirrored by a border)
        const bool fast = yb > 2 * win + 2 && yb + CS_UNROLL - 1 < P.h - 1 - win && yb + CS_UNROLL - 4 < 2 * (P.l1_h - 1 - win) &&
                          y_switch >= yb + CS_UNROLL && yb + CS_UNROLL <= P.h;
        if (fast) {
#pragma unroll
            for (int u = 0; u < CS_UNROLL; u++) {
                do_row(yb + u, inr[u], std::false_type());
                if ((u & 1) == 0) emit((yb + u - 2) >> 1, hA, hB, hC, hD, hE, std::false_type());   // yb is even, > 2
            }
        } else {
#pragma nounroll
            for (int u = 0; u < CS_UNROLL; u++) {
                const int y = yb + u;
                if (y >= P.h) break;
                if (y == y_switch) stage(y);
                do_row(y, inr[0], std::true_type());
                if (!(y & 1)) {
                    if (y == 2) emit(0, hE, hD, hC, hD, hE, std::true_type());        // rows -2, -1 mirror rows 2, 1
                    else if (y > 2) emit((y - 2) >> 1, hA, hB, hC, hD, hE, std::true_type());
                }
#pragma unroll
                for (int k = 0; k + 1 < CS_UNROLL; k++) inr[k] = inr[k + 1];
            }
        }
#pragma unroll
        for (int u = 0; u < CS_UNROLL; u++) inr[u] = nxt[u];
    }
    // the last level-1 row: its window hangs over the bottom edge (row h mirrors h-2, row h+1 mirrors h-3)
    if (P.h & 1) emit((P.h - 1) >> 1, hC, hD, hE, hD, hC, std::true_type());
    else emit((P.h - 2) >> 1, hB, hC, hD, hE, hD, std::true_type());
}

int ov2_launch_clahe(ov2_ctx *ctx, const uint8_t *src_d, int w, int h, int stride, size_t src_batch_stride, int batch,
                     double clip_limit, int tiles_x, int tiles_y, uint8_t *dst_d, int dst_stride, size_t dst_batch_stride,
                     uint8_t *lut_d, int border, const PyrDesc *pyr, int *level1_done)
{
    if (level1_done) *level1_done = 0;
    // geometry checks first: nothing is enqueued when the call is going to fail
    OV2_REQUIRE(tiles_x + 1 <= CLAHE_MAX_CELLS && (size_t)(tiles_x + 1) * 1024 <= 160 * 1024, OV2_EUNSUPPORTED,
                "CLAHE: too many tile columns / too wide an image for the LDS tables");
    // the dynamic-LDS limit is a per-function, process-wide attribute: raise it ONCE to the hardware maximum instead of
    // per call to that call's need (two contexts on two threads would otherwise race on it)
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [] {
        attr_err = hipFuncSetAttribute((const void *)k_clahe_apply<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (attr_err == hipSuccess)
            attr_err = hipFuncSetAttribute((const void *)k_clahe_apply<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    OV2_HIP_CHECK(attr_err);
    ClaheParams P;
    P.border = border;
    P.nstrips = 0; P.nlut = 0; P.l1_delta = 0; P.l1_pitch = P.l1_w = P.l1_h = 0;
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
    // ~5 tiles per wavefront (prefetch pipeline) once the batch alone fills the GPU, one tile per wavefront otherwise (latency)
    const int tiles_per_wave = (long long)batch * tiles_x * tiles_y >= 32768 ? 5 : 1;
    const bool src_al = ((stride | (int)(size_t)src_d | (int)src_batch_stride) & 3) == 0;
    P.ysplit = (long long)batch * (tiles_y + 1) >= 256 ? 1 : (P.th >= 48 ? 6 : (P.th >= 16 ? 3 : 1));
    P.batch = batch; P.gx_lut = (tiles_x * tiles_y + 4 * tiles_per_wave - 1) / (4 * tiles_per_wave);
    // Batch mode, destination = level 0 of a pyramid: the strip kernel also writes level 1 and both borders in the same walk, and
    // in its fused form computes the LUTs as well (OV2_OPT_CLAHE_STRIPS: 2 forces the fused kernel for any batch, 1 the strip kernel
    // after k_clahe_lut, 0 the separate kernels -- A/B runs, parity tests of all paths)
    bool lut_launched = false;
    auto launch_lut = [&] {
        if (!lut_launched) hipLaunchKernelGGL(src_al ? k_clahe_lut<true> : k_clahe_lut<false>, dim3(P.gx_lut * batch), dim3(256), 0, ctx->stream, P, src_d, lut_d);
        lut_launched = true;
    };
    if (pyr && pyr->n_levels >= 2 && border == pyr->win && w >= 64 && h >= 8 && 2 * ((border + 3) / 4) + 1 <= (w + 3) / 4 &&
        (((size_t)dst_d | (size_t)dst_stride | dst_batch_stride) & 3) == 0 && (pyr->lv[1].img_pitch & 1) == 0) {
        const int ndw = (w + 3) / 4, nstrips = ndw <= 64 ? 1 : 2 + (ndw - 126 + 61) / 62;
        const bool want = ctx->clahe_strips >= 0 ? ctx->clahe_strips >= 1 : (long long)batch * nstrips >= 1024;
        if (want && nstrips <= CS_MAX_STRIPS && tiles_x + 1 <= 40) {
            static std::once_flag strip_once;
            static hipError_t strip_err = hipSuccess;
            std::call_once(strip_once, [] {
                const void *fn[4] = {(const void *)k_clahe_apply_pyr<false, false>, (const void *)k_clahe_apply_pyr<true, false>,
                                     (const void *)k_clahe_apply_pyr<false, true>, (const void *)k_clahe_apply_pyr<true, true>};
                for (int i = 0; i < 4 && strip_err == hipSuccess; i++) strip_err = hipFuncSetAttribute(fn[i], hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
            });
            OV2_HIP_CHECK(strip_err);
            P.nstrips = nstrips;
            P.l1_delta = pyr->lv[1].img_roi - pyr->lv[0].img_roi; P.l1_pitch = pyr->lv[1].img_pitch; P.l1_w = pyr->lv[1].w; P.l1_h = pyr->lv[1].h;
            const bool unal = (w & 3) != 0 || (pyr->lv[1].w & 1) != 0;
            // fused form: the aligned instance takes aligned sources only (its LUT wavefront uses the aligned-dword tile loads)
            // One LUT wavefront keeps up with the strips of a 752-pixel image (15 tiles per row); wider images get one per ~8 tiles of a row (KITTI, 24 tiles per row: 3 -- with one the
            // pre-processing of configs[2] took 4.3 instead of 2.6 ms), as far as the work-group's ten wavefronts go; auto mode
            // leaves geometries with more than 16 tiles per LUT wavefront to the two-kernel form
            const int nlut = tiles_x <= 16 ? 1 : std::max(1, std::min((tiles_x + 7) / 8, CS_MAX_STRIPS + 2 - nstrips));   // (EuRoC, 15: two are slower than one)
            const bool fused = (ctx->clahe_strips == 2 || (ctx->clahe_strips < 0 && tiles_x <= 16 * nlut)) && (unal || src_al);
            if (fused) {
                P.nlut = nlut;
                const size_t lds = (size_t)(tiles_x + 1) * 1024 + (size_t)2 * tiles_x * 256 + ((size_t)nstrips * CH_WAVE_DW + (size_t)P.nlut * CH_MULTI_DW) * 4;
                hipLaunchKernelGGL((unal ? k_clahe_apply_pyr<true, true> : k_clahe_apply_pyr<false, true>), dim3(batch), dim3(64 * (nstrips + P.nlut)), lds, ctx->stream, P, src_d, lut_d, dst_d);
            } else {
                launch_lut();
                hipLaunchKernelGGL((unal ? k_clahe_apply_pyr<true, false> : k_clahe_apply_pyr<false, false>), dim3(batch), dim3(64 * nstrips), (size_t)(tiles_x + 1) * 1024, ctx->stream, P, src_d, lut_d, dst_d);
            }
            OV2_HIP_CHECK(hipGetLastError());
            if (level1_done) *level1_done = 1;
            return OV2_OK;
        }
    }
    launch_lut();
    const size_t apply_lds = (size_t)(tiles_x + 1) * 1024;
    // one thread per dword column; several column passes only for images wider than 2048 pixels
    const int ndw = (w + 3) / 4, passes = (ndw + 511) / 512;
    const int apply_threads = (((ndw + passes - 1) / passes) + 63) / 64 * 64;
    hipLaunchKernelGGL(src_al ? k_clahe_apply<true> : k_clahe_apply<false>, dim3((tiles_y + 1) * P.ysplit * batch), dim3(apply_threads), apply_lds, ctx->stream, P, src_d, lut_d, dst_d);
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
    return ov2_launch_clahe(ctx, src_d, w, h, stride, src_batch_stride, batch, clip_limit, tiles_x, tiles_y, dst_d, dst_stride,
                            dst_batch_stride, (uint8_t *)ctx->d_scratch, 0, nullptr, nullptr);
}

int ov2_pyr_build_clahe_d(ov2_ctx *ctx, ov2_pyr *p, const uint8_t *img_d, int stride, size_t img_batch_stride,
                          double clip_limit, int tiles_x, int tiles_y)
{
    OV2_REQUIRE(ctx && p && img_d, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(stride >= p->w && tiles_x > 0 && tiles_y > 0 && tiles_x <= p->w && tiles_y <= p->h, OV2_EINVAL, "bad geometry");
    OV2_REQUIRE(p->d.batch == 1 || img_batch_stride >= (size_t)stride * (size_t)p->h, OV2_EINVAL, "img_batch_stride too small");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t lut_bytes = (size_t)p->d.batch * tiles_x * tiles_y * 256;
    int rc = ctx->reserve_device(lut_bytes);
    if (rc != OV2_OK) return rc;
    // the equalised image is written straight into the pyramid's padded level-0 slot (no intermediate image,
    // no level-0 copy); borders and the coarser levels follow from there
    const PyrLevelDesc &L0 = p->d.lv[0];
    int l1_done = 0;
    rc = ov2_launch_clahe(ctx, img_d, p->w, p->h, stride, img_batch_stride, p->d.batch, clip_limit, tiles_x, tiles_y,
                          p->d.base + L0.img_roi, L0.img_pitch, (size_t)p->d.item_stride, (uint8_t *)ctx->d_scratch, p->d.win,
                          &p->d, &l1_done);
    if (rc != OV2_OK) return rc;
    rc = ov2_launch_pyr_build(ctx, p, nullptr, 0, 0, l1_done);
    if (rc != OV2_OK) return rc;
    return ov2_pyr_mark_ready(ctx, p);
}

// host-image form of ov2_pyr_build_clahe_d (batch-1 pyramids): one H2D of the raw frame, nothing comes back
int ov2_pyr_build_clahe_h(ov2_ctx *ctx, ov2_pyr *p, const uint8_t *img_h, int stride, double clip_limit, int tiles_x, int tiles_y)
{
    OV2_REQUIRE(ctx && p && img_h, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(p->d.batch == 1, OV2_EINVAL, "host-buffer entry point takes batch=1 pyramids");
    OV2_REQUIRE(stride >= p->w && tiles_x > 0 && tiles_y > 0 && tiles_x <= p->w && tiles_y <= p->h, OV2_EINVAL, "bad geometry");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    // device staging copy with a 16-byte-aligned pitch (aligned-dword kernel instances whatever the width)
    const size_t pitch = ((size_t)p->w + 15) & ~(size_t)15;
    const size_t img = (pitch * p->h + 255) & ~(size_t)255, lut_bytes = (size_t)tiles_x * tiles_y * 256;
    int rc = ctx->reserve_device(img + lut_bytes);
    if (rc != OV2_OK) return rc;
    uint8_t *ds = (uint8_t *)ctx->d_scratch;
    rc = ctx->upload_image(ds, pitch, img_h, (size_t)stride, (size_t)p->w, (size_t)p->h);
    if (rc != OV2_OK) return rc;
    const PyrLevelDesc &L0 = p->d.lv[0];
    int l1_done = 0;
    rc = ov2_launch_clahe(ctx, ds, p->w, p->h, (int)pitch, 0, 1, clip_limit, tiles_x, tiles_y, p->d.base + L0.img_roi, L0.img_pitch,
                          (size_t)p->d.item_stride, ds + img, p->d.win, &p->d, &l1_done);
    if (rc != OV2_OK) return rc;
    rc = ov2_launch_pyr_build(ctx, p, nullptr, 0, 0, l1_done);
    if (rc != OV2_OK) return rc;
    return ov2_pyr_mark_ready(ctx, p);
}

// preprocessImage of `n_items` host images into items [0, n_items) of a batch pyramid in one enqueue (the right images of the keyframes a
// lock-step batch reaches together, src/mapper.cpp:74-81 per keyframe): one repack into pinned memory, ONE H2D, the batched kernels.
// clip_limit < 0: no CLAHE (use_clahe: 0), plain cv::buildOpticalFlowPyramid.  Items beyond n_items keep their contents.
int ov2_pyr_build_clahe_hb(ov2_ctx *ctx, ov2_pyr *p, int n_items, const uint8_t *const *img_h, int stride, double clip_limit, int tiles_x, int tiles_y)
{
    OV2_REQUIRE(ctx && p && img_h, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(p->parent == nullptr, OV2_EINVAL, "an item view is read-only");
    OV2_REQUIRE(n_items >= 1 && n_items <= p->d.batch, OV2_EINVAL, "n_items out of range");
    const bool use_clahe = clip_limit >= 0.0;
    OV2_REQUIRE(stride >= p->w && (!use_clahe || (tiles_x > 0 && tiles_y > 0 && tiles_x <= p->w && tiles_y <= p->h)), OV2_EINVAL, "bad geometry");
    for (int b = 0; b < n_items; b++) OV2_REQUIRE(img_h[b] != nullptr, OV2_EINVAL, "NULL image");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t pitch = ((size_t)p->w + 15) & ~(size_t)15;
    const size_t img = (pitch * p->h + 255) & ~(size_t)255, lut_bytes = use_clahe ? (size_t)n_items * tiles_x * tiles_y * 256 : 0;
    int rc = ctx->reserve_device(img * (size_t)n_items + lut_bytes);
    if (rc != OV2_OK) return rc;
    uint8_t *ds = (uint8_t *)ctx->d_scratch;
    rc = ctx->upload_images(ds, pitch, img, img_h, n_items, (size_t)stride, (size_t)p->w, (size_t)p->h);
    if (rc != OV2_OK) return rc;
    ov2_pyr q = *p;                                   // items [0, n_items): the launchers size their grids from d.batch
    q.d.batch = n_items;
    if (use_clahe) {
        const PyrLevelDesc &L0 = q.d.lv[0];
        int l1_done = 0;
        rc = ov2_launch_clahe(ctx, ds, p->w, p->h, (int)pitch, img, n_items, clip_limit, tiles_x, tiles_y, q.d.base + L0.img_roi, L0.img_pitch,
                              (size_t)q.d.item_stride, ds + img * (size_t)n_items, q.d.win, &q.d, &l1_done);
        if (rc != OV2_OK) return rc;
        rc = ov2_launch_pyr_build(ctx, &q, nullptr, 0, 0, l1_done);
    } else
        rc = ov2_launch_pyr_build(ctx, &q, ds, (int)pitch, img);
    if (rc != OV2_OK) return rc;
    return ov2_pyr_mark_ready(ctx, p);
}

int ov2_clahe_h(ov2_ctx *ctx, const uint8_t *src_h, int w, int h, int stride, double clip_limit, int tiles_x, int tiles_y,
                uint8_t *dst_h, int dst_stride)
{
    OV2_REQUIRE(ctx && src_h && dst_h, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(w > 0 && h > 0 && stride >= w && dst_stride >= w && tiles_x > 0 && tiles_y > 0, OV2_EINVAL, "bad geometry");
    OV2_REQUIRE(tiles_x <= w && tiles_y <= h, OV2_EINVAL, "more tiles than pixels");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t pitch = ((size_t)w + 15) & ~(size_t)15;          // aligned staging pitch for both device images
    const size_t img = (pitch * h + 255) & ~(size_t)255, lut_bytes = (size_t)tiles_x * tiles_y * 256;
    int rc = ctx->reserve_device(2 * img + lut_bytes);
    if (rc != OV2_OK) return rc;
    uint8_t *ds = (uint8_t *)ctx->d_scratch;
    rc = ctx->upload_image(ds, pitch, src_h, (size_t)stride, (size_t)w, (size_t)h);
    if (rc != OV2_OK) return rc;
    const int rc2 = ov2_launch_clahe(ctx, ds, w, h, (int)pitch, 0, 1, clip_limit, tiles_x, tiles_y, ds + img, (int)pitch, 0, ds + 2 * img, 0, nullptr, nullptr);
    if (rc2 != OV2_OK) return rc2;
    rc = ctx->download_image(dst_h, (size_t)dst_stride, ds + img, pitch, (size_t)w, (size_t)h);      // (synchronises)
    if (rc != OV2_OK) return rc;
    return OV2_OK;
}

} // extern "C"
