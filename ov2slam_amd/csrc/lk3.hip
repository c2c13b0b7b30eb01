// lk3.hip -- forward/backward pyramidal Lucas-Kanade, 3 lanes per keypoint (gfx950, wave64).
//
// Same arithmetic and the same reference semantics as lk.hip (cv::calcOpticalFlowPyrLK inside
// FeatureTracker::fbKltTracking, /root/reference/src/feature_tracker.cpp:35-137), different work
// mapping.  PMC counters showed the row-per-lane kernel (16 lanes per keypoint, 10 active) to be
// VALU-issue bound with ~56 % of every Gauss-Newton iteration being per-keypoint scalar work
// (weights, bounds, reductions, convergence tests) replicated over 16 lanes.  Here a keypoint owns
// 3 lanes and each lane owns 3 window rows (WIN = 9): 5 keypoints per 16-lane DPP row, 20 per
// wavefront, 80 per workgroup -- the scalar part is amortised over 27 pixels per lane.
//   * Both image neighbourhoods of a level are staged once per level in per-wavefront LDS:
//     12 x 16 B of the template image (rows ipy-1.., for the on-the-fly Scharr derivative) and
//     16 x 16 B of the search image; the three lanes of a keypoint fetch them cooperatively with
//     aligned dword loads.  No workgroup barrier anywhere (wave-private LDS regions).
//   * Reductions over the 3 lanes of a keypoint: DPP row_shr:1/2 + row_shl:1/2 inside the 16-lane row,
//     on exact integer partial sums (16-bit halves), so results stay bit-identical to the oracle.
//   * Forward levels and the backward level run through ONE inlined instance of the level body
//     (a small step machine) to keep the code inside the instruction cache.
#include "common.hpp"
#include "xcd_map.hpp"
#include <float.h>
#include <math.h>
#include <stdlib.h>

#pragma clang fp contract(off)

struct LK3Params {
    int max_level, max_iter;
    double eps2;
    float min_eig_th;
    int flags;
    float err_th, fb_dist;
    int do_fb;
    int n_max;
};

#define L3_WIN 9
#define L3_RPL 3                      // window rows per lane
#define L3_KPW 20                     // keypoints per wavefront
#ifndef L3_WAVES
#define L3_WAVES 1                    // wavefronts per workgroup (no workgroup-level cooperation: small groups balance best)
#endif
#ifndef L3_MIN_WAVES_PER_EU
#define L3_MIN_WAVES_PER_EU 3
#endif
#define L3_KPB (L3_KPW * L3_WAVES)     // keypoints per workgroup
#define L3_IROWS (L3_WIN + 3)         // template neighbourhood rows  (12)
#define L3_JROWS 16                   // search neighbourhood rows
#define L3_NBH_R 3                    // (16 - (WIN+1)) / 2
#define L3_STRIDE 116                 // dwords per keypoint slot: 48 (I) + 64 (J) + 4 pad; 16-byte multiple so that rows move as b128

struct L3Lv { int w, h, img_pitch, pady; long long img_roi; };      // the level fields this kernel needs (kept in SGPRs)

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // global_load_dwordx4 needs dword alignment only
// LDS reads at BYTE granularity (gfx950 serves unaligned ds_read_b96): the twelve bytes p[0..11] of a staged row from the window's
// own first column -- no v_alignbyte to shift the row into place, one LDS instruction per row instead of two
#ifndef L3_UNALIGNED_LDS
#define L3_UNALIGNED_LDS 1
#endif
typedef uint32_t u32x3_a1 __attribute__((ext_vector_type(3), aligned(1)));
#ifndef L3_UNALIGNED_GLOBAL
#define L3_UNALIGNED_GLOBAL 1
#endif
typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));

// ---- packed 16-bit helpers (carrier type: uint32_t = two 16-bit fields, element 0 in the low half) ----
// Every quantity of the LK inner loops fits 16 bits: pixels (8), bilinear weights (<= 2^14), Scharr sums
// (<= 4080), interpolated patch values (<= 8160) and their differences.  v_pk_* does two pixels per
// instruction, v_dot2_i32_i16 does half a bilinear tap set (2 multiplies + accumulate) per instruction,
// all exact in integers.
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b)); }
__device__ __forceinline__ uint32_t pk_mul(uint32_t a, short c) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) * c); }
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false); }
// a.lo * b.lo + a.hi * b.hi without an accumulator: the VOP3P encoding with the inline constant 0 (the compiler
// only selects the two-address v_dot2c form, which costs a v_mov to clear the destination first)
__device__ __forceinline__ int dot2z(uint32_t a, uint32_t b) { int r; asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b)); return r; }
// a.lo * b.lo + a.hi * b.hi + 256: the rounding constant of the bilinear sums rides in src2 (a scalar register: 256 is no inline
// constant), so (S + 256) >> 9 needs no separate add
__device__ __forceinline__ int dot2r(uint32_t a, uint32_t b)
{
    int r;
    const int c256 = __builtin_amdgcn_readfirstlane(256);
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c256));
    return r;
}
#ifndef L3_DOT2H_ASM
#define L3_DOT2H_ASM 0
#endif
#define DOT2H(a, b) (L3_DOT2H_ASM ? dot2h((a), (b)) : dot2((a), (b), 1 << 15))
// a.lo * b.lo + a.hi * b.hi + 2^15 (scalar-register accumulator like dot2r): the derivative taps, whose inputs carry a factor 4
__device__ __forceinline__ int dot2h(uint32_t a, uint32_t b)
{
    int r;
    const int c = __builtin_amdgcn_readfirstlane(1 << 15);
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
// (S0 >> 16, S1 >> 16) of two signed sums as an i16 pair: their high halves, one byte permute
__device__ __forceinline__ uint32_t hi16_pair(int s0, int s1) { return __builtin_amdgcn_perm((uint32_t)s1, (uint32_t)s0, 0x07060302u); }
// (S0 >> 9, S1 >> 9) as two u16 for sums 0 <= S < 2^24 that already hold their rounding constant: bytes 1-2 of each
// (S >> 8, 16 bits) by one byte permute, then one packed shift
__device__ __forceinline__ uint32_t shr9_pair(int s0, int s1)
{
    const uint32_t hi8 = __builtin_amdgcn_perm((uint32_t)s1, (uint32_t)s0, 0x06050201u);
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, hi8) >> (unsigned short)1));
}
__device__ __forceinline__ uint32_t pack_lo16(int lo, int hi) { return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u); }   // (lo & 0xFFFF) | (hi << 16)
// ((S0 + 256) >> 9, (S1 + 256) >> 9) as two u16 for bilinear sums S (-256 <= S < 2^23): (S + 256) >> 9 == ((S >> 8) + 1) >> 1,
// so one byte permute (S >> 8 of both), one packed add and one packed shift replace two 32-bit shifts, a pack and
// the two accumulator initialisations (v_dot2c needs its rounding constant moved into the destination first)
__device__ __forceinline__ uint32_t round9_pair(int s0, int s1)
{
    const uint32_t hi8 = __builtin_amdgcn_perm((uint32_t)s1, (uint32_t)s0, 0x06050201u);
    return __builtin_bit_cast(uint32_t, (u16x2)((__builtin_bit_cast(u16x2, hi8) + (unsigned short)1) >> (unsigned short)1));
}
__device__ __forceinline__ uint32_t odd_pair(uint32_t e_next, uint32_t e) { return __builtin_amdgcn_alignbyte(e_next, e, 2); }             // (e.hi, e_next.lo)
__device__ __forceinline__ uint32_t bytes01(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c010c00u); }                                 // (b0, b1) as two u16
__device__ __forceinline__ uint32_t bytes23(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c030c02u); }                                 // (b2, b3)

template <int CTRL>
__device__ __forceinline__ int l3_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

// sum over the 3 lanes of a keypoint group (lanes 3g, 3g+1, 3g+2 of a 16-lane row), result in all three
__device__ __forceinline__ int l3_sum3(int v, int sub)
{
    int u = v + l3_dpp<0x111>(v);          // row_shr:1  (lane i gets i-1)
    u += l3_dpp<0x112>(v);                 // row_shr:2  -> total at sub == 2
    const int b1 = l3_dpp<0x101>(u);       // row_shl:1  (lane i gets i+1)
    const int b2 = l3_dpp<0x102>(u);       // row_shl:2
    return sub == 2 ? u : (sub == 1 ? b1 : b2);
}

// exact sum of int32 partials (|p| < 2^31) over the group, as a double
__device__ __forceinline__ double l3_sum3_exact(int p, int sub)
{
    const int slo = l3_sum3(p & 0xFFFF, sub);
    const int shi = l3_sum3(p >> 16, sub);
    return (double)shi * 65536.0 + (double)slo;
}

// (float)(exact sum of three int32 partials, |p| < 2^30 each) over the group, in all three lanes: lanes 0 + 1 add up in int32
// (|.| < 2^31), lane 2 joins in fp64 (exact) and rounds ONCE to fp32 -- the rounding of the oracle's (float)int64 -- and the
// float travels back through two DPP moves.  10 instructions instead of the 20 of the 16-bit-halves form above.
__device__ __forceinline__ float l3_sum3_f32(int p, int sub)
{
    const int t = p + l3_dpp<0x111>(p);            // lane 1: p0 + p1
    const int q = l3_dpp<0x111>(t);                // lane 2: p0 + p1
    const float f = (float)((double)q + (double)p);                 // valid in lane 2
    const int fi = __builtin_bit_cast(int, f);
    const int f1 = l3_dpp<0x101>(fi), f2 = l3_dpp<0x102>(fi);      // lane 1 <- lane 2, lane 0 <- lane 2
    return __builtin_bit_cast(float, sub == 2 ? fi : (sub == 1 ? f1 : f2));
}

__device__ __forceinline__ int l3_round(float v) { return __float2int_rn(v); }
__device__ __forceinline__ int l3_floor(float v) { return (int)floorf(v); }
__device__ __forceinline__ int l3_m24(int a, int b) { return __mul24(a, b); }

__device__ __forceinline__ void l3_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// cooperative fetch (3 lanes) of the 16 x 16-byte search neighbourhood with origin (jx0, jy0) into slot[48..111].
// One row = one 16-byte + one 4-byte request (a lane's row never coalesces with its neighbours': the number of
// L1 line look-ups, not bytes, bounds this phase).  Rows are addressed from one 64-bit base with a constant
// stride; the clamp into the padded buffer is only evaluated when some keypoint of the wavefront needs it.
#ifndef OV2_LK3_KO
#define OV2_LK3_KO 0
#endif
template <bool CLAMP>
__device__ __forceinline__ void l3_fetch_J_rows(uint32_t *slot, const uint8_t *jroi, const L3Lv &LJ, int xa, uint32_t sh, int jy0, int sub)
{
    const uint8_t *p0 = jroi + (long long)(jy0 + sub) * LJ.img_pitch + xa;
    const long long step = 3LL * LJ.img_pitch;
    uint32_t *dst = slot + 48 + 4 * sub;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (k < 5 || sub == 0) {                                    // rows sub + 3k < 16
            const uint8_t *p = p0 + k * step;
            if (CLAMP) {
                int y = jy0 + sub + 3 * k;
                y = y < -LJ.pady ? -LJ.pady : (y > LJ.h + LJ.pady - 1 ? LJ.h + LJ.pady - 1 : y);   // rows beyond the buffer are never consumed
                p = jroi + (long long)y * LJ.img_pitch + xa;
            }
            if (OV2_LK3_KO & 2) p = jroi + sub * 64 + k * 32;
#if L3_UNALIGNED_GLOBAL
            // the row's sixteen bytes from its own first column: ONE request at byte granularity (the memory pipeline serves
            // unaligned dwordx4), nothing to shift
            *(u32x4 *)(dst + 12 * k) = *(const u32x4_a1 *)(p + sh);
#else
            const u32x4 lo = *(const u32x4_a4 *)p;
            const uint32_t hi = *(const uint32_t *)(p + 16);
            u32x4 o;
            o.x = __builtin_amdgcn_alignbyte(lo.y, lo.x, sh); o.y = __builtin_amdgcn_alignbyte(lo.z, lo.y, sh);
            o.z = __builtin_amdgcn_alignbyte(lo.w, lo.z, sh); o.w = __builtin_amdgcn_alignbyte(hi, lo.w, sh);
            *(u32x4 *)(dst + 12 * k) = o;
#endif
        }
    }
}

__device__ __forceinline__ void l3_fetch_J(uint32_t *slot, const uint8_t *jroi, const L3Lv &LJ, int jx0, int jy0, int sub)
{
    const int xa = jx0 & ~3;
    const uint32_t sh = (uint32_t)(jx0 - xa);
    const bool outside = jy0 < -LJ.pady || jy0 + L3_JROWS - 1 > LJ.h + LJ.pady - 1;
    if (OV2_LK3_KO & 16) return;
    if (__builtin_amdgcn_ballot_w64(outside) == 0) l3_fetch_J_rows<false>(slot, jroi, LJ, xa, sh, jy0, sub);
    else l3_fetch_J_rows<true>(slot, jroi, LJ, xa, sh, jy0, sub);
}

// the three window rows of a lane as 16-bit pairs: slots 0..3 = pixels (0,1)(2,3)(4,5)(6,7), slot 4 = (8, -)
// (the interpolated template values I are not kept: sum (J - I) dI = sum J dI - sum I dI, and the second sum is a per-lane
// constant of the visit -- 15 VGPRs and 15 packed subtractions per Gauss-Newton trip less)
struct L3Tmpl { uint32_t X[L3_RPL][5], Y[L3_RPL][5]; int cIX, cIY; };

struct L3State {
    float nx, ny;      // nextPts[i] as OpenCV carries it between levels
    int status;
    float err;
    int iters, visits;
    // L3Slot -- what the keypoint's LDS slot holds after a level: region A (slot[0..47]) = 12 rows x 16 B of the level's
    // template image from (ax0, ay0), region B (slot[48..111]) = 16 x 16 B of its search image from (bx0, by0)
};
// what the keypoint's LDS slot holds after a level (see L3State comment)
struct L3Slot { int ax0, ay0, bx0, by0; };

// One pyramid level for the keypoint owned by this 3-lane group; lane `sub` owns window rows 3 sub .. 3 sub + 2.
__device__ __forceinline__ void l3_level(const uint8_t *__restrict__ itemI, const L3Lv &LI,
                                         const uint8_t *__restrict__ itemJ, const L3Lv &LJ,
                                         const LK3Params &prm, int level, bool scale_from_input, bool reuse,
                                         float px0, float py0, int sub, uint32_t *slot, L3State &st, L3Slot &sl)
{
    constexpr int WIN = L3_WIN;
    const float halfWin = (float)(WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const float W14 = (float)(1 << 14);
    const float lvl_scale = (float)(1. / (double)(1 << level));

    float prevx = px0 * lvl_scale, prevy = py0 * lvl_scale;
    float nextx, nexty;
    if (scale_from_input) { nextx = st.nx * lvl_scale; nexty = st.ny * lvl_scale; }   // top level, USE_INITIAL_FLOW
    else { nextx = st.nx * 2.f; nexty = st.ny * 2.f; }
    st.nx = nextx; st.ny = nexty;

    prevx -= halfWin; prevy -= halfWin;
    const int ipx = l3_floor(prevx), ipy = l3_floor(prevy);
    if (ipx < -WIN || ipx >= LI.w || ipy < -WIN || ipy >= LI.h) {
        if (level == 0) { st.status = 0; st.err = 0.f; }
        return;
    }
    st.visits++;
    float a = prevx - (float)ipx, b = prevy - (float)ipy;
    int iw00 = l3_round((1.f - a) * (1.f - b) * W14);
    int iw01 = l3_round(a * (1.f - b) * W14);
    int iw10 = l3_round((1.f - a) * b * W14);
    int iw11 = (1 << 14) - iw00 - iw01 - iw10;
    uint32_t W01 = pack_lo16(iw00, iw01), W23 = pack_lo16(iw10, iw11);

    // ---- stage both neighbourhoods (all global loads of this level are issued here) ----
    l3_lds_sync();                                                  // previous level's reads are done
    const uint8_t *iroi = itemI + LI.img_roi, *jroi = itemJ + LJ.img_roi;
    // 12 rows, one 16-byte request each, rows sub + 3k from one base pointer (clamp only near the image border)
    auto fetch_I = [&](uint32_t *dst, int ixa) {
        const bool outside = ipy - 1 < -LI.pady || ipy - 1 + L3_IROWS - 1 > LI.h + LI.pady - 1;
        const bool clamp_rows = __builtin_amdgcn_ballot_w64(outside) != 0;
        const uint8_t *p0 = iroi + (long long)(ipy - 1 + sub) * LI.img_pitch + ixa;
        const long long step = 3LL * LI.img_pitch;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint8_t *p = p0 + k * step;
            if (clamp_rows) {
                int y = ipy - 1 + sub + 3 * k;
                y = y < -LI.pady ? -LI.pady : (y > LI.h + LI.pady - 1 ? LI.h + LI.pady - 1 : y);     // only feeds derivatives of out-of-image rows (= 0)
                p = iroi + (long long)y * LI.img_pitch + ixa;
            }
            if (OV2_LK3_KO & 1) p = iroi + sub * 64 + k * 16;
            if (!(OV2_LK3_KO & 8)) *(u32x4 *)(dst + 4 * sub + 12 * k) = *(const u32x4_a4 *)p;
        }
    };
    const uint32_t *tsrc;                                           // template source rows (16 B each), first needed byte at `ish`
    uint32_t ish;
    int jx0, jy0, jb;                                               // search block: origin, LDS offset of its first row
    if (!reuse) {
        const int ixa = (ipx - 1) & ~3;
        ish = (uint32_t)((ipx - 1) - ixa);
        fetch_I(slot, ixa);
        tsrc = slot;
        sl.ax0 = ixa; sl.ay0 = ipy - 1;
        const float sx = nextx - halfWin, sy = nexty - halfWin;
        const int cx0 = l3_floor(fminf(fmaxf(sx, (float)(-WIN)), (float)(LJ.w - 1)));
        const int cy0 = l3_floor(fminf(fmaxf(sy, (float)(-WIN)), (float)(LJ.h - 1)));
        jx0 = cx0 - L3_NBH_R; jy0 = cy0 - L3_NBH_R; jb = 48;
        l3_fetch_J(slot, jroi, LJ, jx0, jy0, sub);
    } else {
        // Backward level 0 straight after forward level 0: the two images swap roles and the slot already holds
        // them.  Region B (search block of the forward level, around the tracked position) contains the
        // 12 x 12 template neighbourhood unless the last forward step left its 3-pixel margin; region A (template
        // neighbourhood of the forward level, around the keypoint) is the search block of a backward track that
        // stays within about a pixel of the keypoint -- the only tracks that pass the fb test.  No global load on
        // the common path: the cache-line fills of these 28 rows were the larger part of a level-0 visit.
        const int sx = ipx - 1 - sl.bx0, sy = ipy - 1 - sl.by0;
        const bool inB = (unsigned)sx <= 3u && (unsigned)sy <= (unsigned)(L3_JROWS - L3_IROWS);
        tsrc = slot + 48 + 4 * sy; ish = (uint32_t)sx;
        if (!inB) {                                                 // refetch into region B (its content is of no use now)
            const int ixa = (ipx - 1) & ~3;
            ish = (uint32_t)((ipx - 1) - ixa);
            fetch_I(slot + 48, ixa);
            tsrc = slot + 48;
        }
        jx0 = sl.ax0; jy0 = sl.ay0; jb = 0;
    }
    l3_lds_sync();

    // ---- template: I (5 fractional bits), dIx, dIy of window rows r0 .. r0+2 ----
    // image rows ipy+r0-1 .. ipy+r0+4 = neighbourhood rows r0 .. r0+5; byte k <-> column ipx-1+k (12 bytes).
    // Scharr (calcSharrDeriv + copyMakeBorder(BORDER_CONSTANT 0)) is evaluated on the fly at the 4 x 10 integer
    // positions the bilinear footprints of these rows touch, two columns per packed instruction.
    L3Tmpl T;
    T.cIX = 0; T.cIY = 0;
    int s11 = 0, s12 = 0, s22 = 0;
    {
        const int r0 = 3 * sub;
        // Streaming order (keeps ~100 VGPRs live instead of ~140): image rows are unpacked from LDS as they are
        // needed, derivative row d is formed from image rows d, d+1, d+2, and window row j = d-1 is finished as
        // soon as derivative rows j, j+1 exist.  E[m][i] = (p[2i], p[2i+1]) of image row m;
        // DX[d][i] = (dx[2i], dx[2i+1]) with column c <-> ipx + c.
        auto load_row = [&](int m, uint32_t (&e)[6]) {
#if L3_UNALIGNED_LDS
            const u32x3_a1 w = *(const u32x3_a1 *)((const uint8_t *)(tsrc + 4 * (r0 + m)) + ish);
            const uint32_t d0 = w.x, d1 = w.y, d2 = w.z;
#else
            const u32x4 w = *(const u32x4 *)(tsrc + 4 * (r0 + m));
            const uint32_t d0 = __builtin_amdgcn_alignbyte(w.y, w.x, ish), d1 = __builtin_amdgcn_alignbyte(w.z, w.y, ish), d2 = __builtin_amdgcn_alignbyte(w.w, w.z, ish);
#endif
            e[0] = bytes01(d0); e[1] = bytes23(d0); e[2] = bytes01(d1); e[3] = bytes23(d1); e[4] = bytes01(d2); e[5] = bytes23(d2);
        };
        // derivative positions outside the image are 0 (rare: only for windows overlapping the image border)
        const bool border = ipx < 0 || ipx + WIN >= LI.w || ipy + r0 < 0 || ipy + r0 + 3 >= LI.h;
        const bool any_border = __builtin_amdgcn_ballot_w64(border) != 0;
        auto deriv_row = [&](int d, const uint32_t (&ea)[6], const uint32_t (&eb)[6], const uint32_t (&ec)[6], uint32_t (&dx)[5], uint32_t (&dy)[5]) {
            uint32_t T0[6], T1[6];
#pragma unroll
            // The derivatives are produced with a factor 4 (Scharr weights 12 / 40 instead of 3 / 10: |4 d| <= 16320 still fits 16
            // bits), so that the bilinear tap sum 4 S = sum w (4 d) descales with the high half of (4 S + 2^15):
            // (S + 2^13) >> 14 == (4 S + 2^15) >> 16 -- one byte permute per pixel pair instead of two shifts and a pack
            for (int i = 0; i < 6; i++) {
                T0[i] = pk_add(pk_mul(pk_add(ea[i], ec[i]), 12), pk_mul(eb[i], 40));
                T1[i] = pk_sub(ec[i], ea[i]);
            }
#pragma unroll
            for (int i = 0; i < 5; i++) {
                dx[i] = pk_sub(T0[i + 1], T0[i]);
                dy[i] = pk_add(pk_mul(pk_add(T1[i], T1[i + 1]), 12), pk_mul(odd_pair(T1[i + 1], T1[i]), 40));
            }
            if (any_border) {
                const int Yd = ipy + r0 + d;
                const bool yin = Yd >= 0 && Yd < LI.h;
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    const int X = ipx + 2 * i;
                    const uint32_t mk = ((yin && X >= 0 && X < LI.w) ? 0x0000FFFFu : 0u) | ((yin && X + 1 >= 0 && X + 1 < LI.w) ? 0xFFFF0000u : 0u);
                    dx[i] &= mk; dy[i] &= mk;
                }
            }
        };
        auto window_row = [&](int j, const uint32_t (&et)[6], const uint32_t (&eb)[6], const uint32_t (&dxt)[5], const uint32_t (&dyt)[5],
                              const uint32_t (&dxb)[5], const uint32_t (&dyb)[5]) {
            // two window pixels at a time, packed at once (short live ranges: the first form kept three 10-entry sum arrays alive
            // and spilled once the derivative sums stayed 32 bits wide until the pack)
            auto taps = [&](int x, int &iv, int &xv, int &yv) {
                const int k = x + 1;                                // pixel pair (p[x+1], p[x+2]) of image rows j+1 (top), j+2 (bottom)
                const uint32_t pt = (k & 1) ? odd_pair(et[(k >> 1) + 1], et[k >> 1]) : et[k >> 1];
                const uint32_t pb = (k & 1) ? odd_pair(eb[(k >> 1) + 1], eb[k >> 1]) : eb[k >> 1];
                iv = dot2(pt, W01, dot2r(pb, W23));                 // + 256: shifted and packed by shr9_pair
                const uint32_t xt = (x & 1) ? odd_pair(dxt[(x >> 1) + 1], dxt[x >> 1]) : dxt[x >> 1];   // (d[x], d[x+1]) of rows j, j+1
                const uint32_t xb = (x & 1) ? odd_pair(dxb[(x >> 1) + 1], dxb[x >> 1]) : dxb[x >> 1];
                const uint32_t yt = (x & 1) ? odd_pair(dyt[(x >> 1) + 1], dyt[x >> 1]) : dyt[x >> 1];
                const uint32_t yb = (x & 1) ? odd_pair(dyb[(x >> 1) + 1], dyb[x >> 1]) : dyb[x >> 1];
                xv = dot2(xt, W01, DOT2H(xb, W23));                 // 4 S + 2^15: descaled by hi16_pair
                yv = dot2(yt, W01, DOT2H(yb, W23));
            };
#pragma unroll
            for (int t = 0; t < 5; t++) {
                int i0, x0, y0, i1 = 256, x1 = 0, y1 = 0;           // (the pad pixel's derivatives are 0: its I never counts)
                taps(2 * t, i0, x0, y0);
                if (2 * t + 1 < WIN) taps(2 * t + 1, i1, x1, y1);
                const uint32_t Ipair = shr9_pair(i0, i1);
                T.X[j][t] = hi16_pair(x0, x1);
                T.Y[j][t] = hi16_pair(y0, y1);
                s11 = dot2(T.X[j][t], T.X[j][t], s11);
                s12 = dot2(T.X[j][t], T.Y[j][t], s12);
                s22 = dot2(T.Y[j][t], T.Y[j][t], s22);
                T.cIX = dot2(Ipair, T.X[j][t], T.cIX);              // |I dI| <= 8160 * 4080, 27 pixels: < 2^30
                T.cIY = dot2(Ipair, T.Y[j][t], T.cIY);
            }
        };
        uint32_t E0[6], E1[6], E2[6], E3[6], DXa[5], DYa[5], DXb[5], DYb[5];
        if (OV2_LK3_KO & 4) {
            for (int j = 0; j < L3_RPL; j++) for (int t = 0; t < 5; t++) { T.X[j][t] = slot[4 * r0 + j * 5 + t] >> 1; T.Y[j][t] = T.X[j][t] >> 1; }
            s11 = s22 = 1 << 28; s12 = 0;
        } else {
        load_row(0, E0); load_row(1, E1); load_row(2, E2);
        deriv_row(0, E0, E1, E2, DXa, DYa);
        load_row(3, E3);
        deriv_row(1, E1, E2, E3, DXb, DYb);
        window_row(0, E1, E2, DXa, DYa, DXb, DYb);                  // window row 0: image rows 1, 2; derivative rows 0, 1
        load_row(4, E0);                                            // E0 <- image row 4
        deriv_row(2, E2, E3, E0, DXa, DYa);
        window_row(1, E2, E3, DXb, DYb, DXa, DYa);                  // image rows 2, 3; derivative rows 1, 2
        load_row(5, E1);                                            // E1 <- image row 5
        deriv_row(3, E3, E0, E1, DXb, DYb);
        window_row(2, E3, E0, DXa, DYa, DXb, DYb);                  // image rows 3, 4; derivative rows 2, 3
        }
    }
    // per-lane partials: 27 * 4080^2 = 4.5e8 < 2^31
    const float A11 = l3_sum3_f32(s11, sub) * FLT_SCALE;
    const float A12 = l3_sum3_f32(s12, sub) * FLT_SCALE;
    const float A22 = l3_sum3_f32(s22, sub) * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
    if (prm.flags & OV2_LK_GET_MIN_EIGENVALS) st.err = minEig;
    if (minEig < prm.min_eig_th || D < FLT_EPSILON) {
        if (level == 0) st.status = 0;
        return;
    }
    D = 1.f / D;
    nextx -= halfWin; nexty -= halfWin;
    // Gauss-Newton loop in single-exit form: `active` per keypoint, one wave-uniform back edge (the four early
    // exits of the reference loop -- out of bounds, eps, oscillation, max count -- only clear `active`)
    float pdx = 0.f, pdy = 0.f;
    int it = 0;
    bool active = prm.max_iter > 0;
    while (__builtin_amdgcn_ballot_w64(active) != 0) {
        if (active) {
            const int inx = l3_floor(nextx), iny = l3_floor(nexty);
            const bool oob = inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h;
            if (oob) {
                if (level == 0) st.status = 0;
                active = false;
            } else {
                st.iters++;
                a = nextx - (float)inx; b = nexty - (float)iny;
                iw00 = l3_round((1.f - a) * (1.f - b) * W14);
                iw01 = l3_round(a * (1.f - b) * W14);
                iw10 = l3_round((1.f - a) * b * W14);
                iw11 = (1 << 14) - iw00 - iw01 - iw10;
                W01 = pack_lo16(iw00, iw01); W23 = pack_lo16(iw10, iw11);
                int ox = inx - jx0, oy = iny - jy0;
                // region B holds 16 rows (row offsets 0..6 leave the 10 the window needs), a reused region A 12 (0..2)
                if ((unsigned)ox > (unsigned)(2 * L3_NBH_R) || (unsigned)oy > (jb ? (unsigned)(2 * L3_NBH_R) : (unsigned)(L3_IROWS - L3_WIN - 1))) {
                    jx0 = inx - L3_NBH_R; jy0 = iny - L3_NBH_R; jb = 48;       // drifted: re-centre the block (always into region B)
                    l3_lds_sync();
                    l3_fetch_J(slot, jroi, LJ, jx0, jy0, sub);
                    l3_lds_sync();
                    ox = L3_NBH_R; oy = L3_NBH_R;
                }
                // source rows iny + 3 sub + m (m = 0..3), bytes [inx, inx + WIN]
                uint32_t P[4][9];                                              // P[m][x] = (p[x], p[x+1])
#if L3_UNALIGNED_LDS
                const uint8_t *s0 = (const uint8_t *)(slot + jb + 4 * (oy + 3 * sub)) + ox;
#else
                const uint32_t sh = (uint32_t)(ox & 3);
                const uint32_t *s0 = slot + jb + 4 * (oy + 3 * sub) + (ox >> 2);
#endif
#pragma unroll
                for (int m = 0; m < 4; m++) {
#if L3_UNALIGNED_LDS
                    const u32x3_a1 w = *(const u32x3_a1 *)(s0 + 16 * m);
                    const uint32_t d0 = w.x, d1 = w.y, d2 = w.z;
#else
                    const uint32_t a0 = s0[4 * m], a1 = s0[4 * m + 1], a2 = s0[4 * m + 2], a3 = s0[4 * m + 3];
                    const uint32_t d0 = __builtin_amdgcn_alignbyte(a1, a0, sh), d1 = __builtin_amdgcn_alignbyte(a2, a1, sh), d2 = __builtin_amdgcn_alignbyte(a3, a2, sh);
#endif
                    P[m][0] = bytes01(d0); P[m][2] = bytes23(d0); P[m][4] = bytes01(d1); P[m][6] = bytes23(d1); P[m][8] = bytes01(d2);
                    P[m][1] = odd_pair(P[m][2], P[m][0]); P[m][3] = odd_pair(P[m][4], P[m][2]);
                    P[m][5] = odd_pair(P[m][6], P[m][4]); P[m][7] = odd_pair(P[m][8], P[m][6]);
                }
                // sum (J - I) dI = sum J dI - sum I dI: the accumulators start at minus the template's constants; every partial
                // sum is  sum_done (J - I) dI - sum_rest I dI,  |.| <= 27 * 8160 * 4080 = 9e8 < 2^30
                int sb1 = -T.cIX, sb2 = -T.cIY;
#pragma unroll
                for (int j = 0; j < L3_RPL; j++) {
                    int v[WIN + 1];
#pragma unroll
                    for (int x = 0; x < WIN; x++) v[x] = dot2(P[j][x], W01, dot2r(P[j + 1][x], W23));
                    v[WIN] = 0;
#pragma unroll
                    for (int t = 0; t < 5; t++) {
                        const uint32_t Jpair = shr9_pair(v[2 * t], v[2 * t + 1]);
                        sb1 = dot2(Jpair, T.X[j][t], sb1);
                        sb2 = dot2(Jpair, T.Y[j][t], sb2);
                    }
                }
                const float b1 = l3_sum3_f32(sb1, sub) * FLT_SCALE;
                const float b2 = l3_sum3_f32(sb2, sub) * FLT_SCALE;
                const float dx = (A12 * b2 - A22 * b1) * D;
                const float dy = (A12 * b1 - A11 * b2) * D;
                nextx += dx; nexty += dy;
                const bool conv_eps = (double)dx * (double)dx + (double)dy * (double)dy <= prm.eps2;
                const bool conv_osc = !conv_eps && it > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01;
                st.nx = nextx + halfWin - (conv_osc ? dx * 0.5f : 0.f);
                st.ny = nexty + halfWin - (conv_osc ? dy * 0.5f : 0.f);
                pdx = dx; pdy = dy;
                it++;
                active = !conv_eps && !conv_osc && it < prm.max_iter;
            }
        }
    }
    sl.bx0 = jx0; sl.by0 = jy0;
}

__global__ __launch_bounds__(64 * L3_WAVES, L3_MIN_WAVES_PER_EU) void k_fb_klt3(PyrDesc P, PyrDesc C, LK3Params prm, int nbx,
                                                 const float2 *__restrict__ kps, float2 *__restrict__ priors,
                                                 uint8_t *__restrict__ status, float *__restrict__ err_out,
                                                 int *__restrict__ iters_out, const int *__restrict__ n_per_item,
                                                 unsigned long long *__restrict__ stats)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[L3_KPB * L3_STRIDE];
    __shared__ unsigned int s_stats[2];
    // XCD-aware work-group -> (batch item, keypoint block) map: the dispatcher deals consecutive work-group
    // ids round-robin over the 8 XCDs, each with its own L2.  All blocks of one item (one image pair) are
    // given ids of the same residue mod 8 and consecutive rank, so the image lines they share are fetched
    // into one L2 only, and at about the same time.
    int b, bx;
    ov2_xcd_map(blockIdx.x, nbx, P.batch, &b, &bx);
    const int n = n_per_item ? n_per_item[b] : prm.n_max;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, g = l16 / 3, sub = l16 - 3 * g;
    const int kslot = wave * L3_KPW + (lane >> 4) * 5 + g;
    const int i = bx * L3_KPB + kslot;
    if (stats) {
        if (threadIdx.x < 2) s_stats[threadIdx.x] = 0;
        __syncthreads();
    }
    if (l16 < 15 && i < n) {                                     // the 3 lanes of a keypoint take the same side
        uint32_t *slot = lds + kslot * L3_STRIDE;
        const long long gi = (long long)b * prm.n_max + i;
        const uint8_t *itemP = P.base + (long long)b * P.item_stride;
        const uint8_t *itemC = C.base + (long long)b * C.item_stride;
        const float2 kp = kps[gi];
        const float2 pr = priors[gi];
        L3State st;
        st.nx = (prm.flags & OV2_LK_USE_INITIAL_FLOW) ? pr.x : kp.x;
        st.ny = (prm.flags & OV2_LK_USE_INITIAL_FLOW) ? pr.y : kp.y;
        st.status = 1; st.err = 0.f; st.iters = 0; st.visits = 0;
        L3Slot sl = {0, 0, 0, 0};
        int ok = 1;
        float fx = 0.f, fy = 0.f;
        int acc_iters = 0, acc_visits = 0;
        float err_fwd = 0.f;
        // step machine: steps 0..max_level = forward levels max_level..0, step max_level+1 = backward level 0
        const int nsteps = prm.max_level + 1 + (prm.do_fb ? 1 : 0);
        for (int step = 0; step < nsteps; step++) {
            const bool bwd = step > prm.max_level;
            if (bwd) {
                // forward pass finished: filter (feature_tracker.cpp:79-101), then set up the backward track (:113-116)
                fx = st.nx; fy = st.ny;
                ok = st.status;
                if (ok && st.err > prm.err_th) ok = 0;
                const float W0 = (float)C.lv[0].w, H0 = (float)C.lv[0].h;
                if (ok && !(1.f <= fx && fx < W0 - 1.f && 1.f <= fy && fy < H0 - 1.f)) ok = 0;       // inBorder :216-221
                acc_iters = st.iters; acc_visits = st.visits; err_fwd = st.err;
                if (!ok) break;
                st.nx = kp.x; st.ny = kp.y; st.status = 1; st.err = 0.f; st.iters = 0; st.visits = 0;
                // (sl.ax0 .. sl.by0 keep describing the slot: forward level 0 staged both regions, or ok would be 0)
            }
            const int level = bwd ? 0 : prm.max_level - step;
            const bool top = bwd || step == 0;
            const uint8_t *itemI = bwd ? itemC : itemP, *itemJ = bwd ? itemP : itemC;
            L3Lv LI, LJ;
            {
                const PyrLevelDesc &a = P.lv[level], &c = C.lv[level];
                LI.w = bwd ? c.w : a.w; LI.h = bwd ? c.h : a.h; LI.img_pitch = bwd ? c.img_pitch : a.img_pitch;
                LI.pady = bwd ? c.pady : a.pady; LI.img_roi = bwd ? c.img_roi : a.img_roi;
                LJ.w = bwd ? a.w : c.w; LJ.h = bwd ? a.h : c.h; LJ.img_pitch = bwd ? a.img_pitch : c.img_pitch;
                LJ.pady = bwd ? a.pady : c.pady; LJ.img_roi = bwd ? a.img_roi : c.img_roi;
            }
            l3_level(itemI, LI, itemJ, LJ, prm, level, top, bwd, bwd ? fx : kp.x, bwd ? fy : kp.y, sub, slot, st, sl);
        }
        if (prm.do_fb) {
            if (ok) {
                acc_iters += st.iters; acc_visits += st.visits;
                if (!st.status) ok = 0;
                else {
                    const float ddx = kp.x - st.nx, ddy = kp.y - st.ny;      // cv::norm(Point2f) (:128)
                    const double nrm = sqrt((double)ddx * (double)ddx + (double)ddy * (double)ddy);
                    if (nrm > (double)prm.fb_dist) ok = 0;
                }
            }
        } else {
            fx = st.nx; fy = st.ny; ok = st.status; acc_iters = st.iters; acc_visits = st.visits; err_fwd = st.err;
        }
        if (sub == 0) {
            priors[gi] = make_float2(fx, fy);
            status[gi] = (uint8_t)ok;
            if (err_out) err_out[gi] = err_fwd;
            if (iters_out) iters_out[gi] = acc_iters;
            if (stats) {
                atomicAdd(&s_stats[0], (unsigned int)acc_iters);
                atomicAdd(&s_stats[1], (unsigned int)acc_visits);
            }
        }
    }
    if (stats) {
        __syncthreads();
        if (threadIdx.x < 2 && s_stats[threadIdx.x])
            atomicAdd(&stats[(blockIdx.x & (LK_STAT_SLOTS - 1)) * LK_STAT_STRIDE + threadIdx.x], (unsigned long long)s_stats[threadIdx.x]);
    }
}

// launcher used by lk.hip's dispatch for WIN == 9
int ov2_launch_fb_klt3(hipStream_t s, const PyrDesc &P, const PyrDesc &C, int max_level, int max_iter, double eps2,
                       float min_eig_th, int flags, float err_th, float fb_dist, int do_fb, int n_max,
                       const float2 *kps, float2 *priors, uint8_t *status, float *err, int *iters,
                       const int *n_per_item, unsigned long long *stats)
{
    LK3Params prm;
    prm.max_level = max_level; prm.max_iter = max_iter; prm.eps2 = eps2; prm.min_eig_th = min_eig_th; prm.flags = flags;
    prm.err_th = err_th; prm.fb_dist = fb_dist; prm.do_fb = do_fb; prm.n_max = n_max;
    const int nbx = (n_max + L3_KPB - 1) / L3_KPB;
    dim3 grid(nbx * P.batch);
    hipLaunchKernelGGL(k_fb_klt3, grid, dim3(64 * L3_WAVES), 0, s, P, C, prm, nbx, kps, priors, status, err, iters, n_per_item, stats);
    return 0;
}
