// lk.hip -- forward/backward pyramidal Lucas-Kanade for gfx950 (wave64).
//
// Replaces cv::calcOpticalFlowPyrLK as called by FeatureTracker::fbKltTracking
// (/root/reference/src/feature_tracker.cpp:66-69 forward, :113-116 backward) and
// the filtering between / after the two calls (:79-101, :119-134).
//
// Work mapping (CDNA4): one keypoint per 16-lane DPP row, 4 keypoints per
// wavefront, 16 per 256-thread workgroup.  Lane r of a row owns window row r
// (r < WIN) and source row r (r <= WIN):
//   * it loads its source row as (WIN+1+3) bytes -> aligned dwords straight from the
//     padded level image in HBM/L2 (one row = one or two cache lines), the row below
//     comes from lane r+1 through a DPP row_shl:1 move -- no LDS, no bank conflicts;
//   * the template patch (I, dIx, dIy: WIN int16 each) lives in VGPRs for the whole
//     Gauss-Newton loop;
//   * the 2x2 normal matrix and the mismatch vector are exact integer sums reduced
//     with a 4-step DPP butterfly inside the row (quad_perm, quad_perm, row_ror:4,
//     row_ror:8), so the result does not depend on reduction order and matches the
//     oracle (OpenCV's `int64 acctype` variant) bit for bit.
// All levels (coarse -> fine), the forward pass, the status/err/border filter, the
// backward pass and the forward-backward distance test run in ONE launch.
#include "common.hpp"
#include "lk_params.hpp"
#include <float.h>
#include <math.h>

#pragma clang fp contract(off)

#define DPP_ROW_SHL1   0x101
#define DPP_ROW_SHL2   0x102
#define DPP_ROW_SHL3   0x103
#define DPP_QP_1032    0xB1
#define DPP_QP_2301    0x4E
#define DPP_ROW_ROR4   0x124
#define DPP_ROW_ROR8   0x128

template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// all-reduce (sum) of an int32 inside each 16-lane row
__device__ __forceinline__ int row_allreduce_i32(int v)
{
    v += dpp_mov<DPP_QP_1032>(v);
    v += dpp_mov<DPP_QP_2301>(v);
    v += dpp_mov<DPP_ROW_ROR4>(v);
    v += dpp_mov<DPP_ROW_ROR8>(v);
    return v;
}

// exact sum of per-lane int32 partials (|p| < 2^31) as a double, via two int32 butterflies
__device__ __forceinline__ double row_allreduce_exact(int p)
{
    const int lo = p & 0xFFFF;       // [0, 65535]
    const int hi = p >> 16;          // arithmetic shift: p == hi*65536 + lo
    const int slo = row_allreduce_i32(lo);
    const int shi = row_allreduce_i32(hi);
    return (double)shi * 65536.0 + (double)slo;
}

__device__ __forceinline__ int cv_round(float v) { return __float2int_rn(v); }
__device__ __forceinline__ int cv_floor(float v) { return (int)floorf(v); }
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
// every integer product on this path has operands below 2^23 (pixels, 14-bit weights, 13/14-bit
// derivatives and differences, row indices, pitches): use the full-rate 24-bit multiplier
// (v_mul_i32_i24 / v_mad_i32_i24) instead of the quarter-rate v_mul_lo_u32.
__device__ __forceinline__ int m24(int a, int b) { return __mul24(a, b); }

template <int WIN>
struct RowBytes {
    static constexpr int ND = (WIN + 1 + 3 + 3) / 4;   // aligned dwords covering WIN+1 bytes at any byte phase
    static constexpr int NR = (WIN + 1 + 3) / 4;       // dwords after re-alignment
    uint32_t d[NR];
    __device__ __forceinline__ int px(int k) const { return (int)((d[k >> 2] >> ((k & 3) * 8)) & 0xFFu); }
};

// load bytes [x, x+WIN] of one padded image row as aligned dwords and shift them so
// that byte 0 is pixel x.
template <int WIN>
__device__ __forceinline__ RowBytes<WIN> load_row(const uint8_t *roi_row, int x)
{
    constexpr int ND = RowBytes<WIN>::ND, NR = RowBytes<WIN>::NR;
    const int xa = x & ~3;                 // floor to a multiple of 4 (two's complement: works for x < 0)
    const int sh = x - xa;                 // 0..3
    const uint32_t *p = (const uint32_t *)(roi_row + xa);
    uint32_t raw[ND];
#pragma unroll
    for (int i = 0; i < ND; i++) raw[i] = p[i];
    RowBytes<WIN> r;
#pragma unroll
    for (int i = 0; i < NR; i++) r.d[i] = __builtin_amdgcn_alignbyte(raw[i + 1 < ND ? i + 1 : i], raw[i], (uint32_t)sh);
    return r;
}

template <int WIN>
__device__ __forceinline__ RowBytes<WIN> row_from_next_lane(const RowBytes<WIN> &r)
{
    RowBytes<WIN> o;
#pragma unroll
    for (int i = 0; i < RowBytes<WIN>::NR; i++) o.d[i] = (uint32_t)dpp_mov<DPP_ROW_SHL1>((int)r.d[i]);
    return o;
}


// ---- template rows for on-the-fly Scharr ---------------------------------------------------
// bytes [x-1, x+WIN+2) of one padded image row (WIN+3 bytes), re-phased so that byte 0 is column x-1
template <int WIN>
struct TRow {
    static constexpr int NB = WIN + 3;                 // bytes
    static constexpr int NR = (NB + 3) / 4;            // dwords after re-alignment
    static constexpr int ND = (NB + 3 + 3) / 4;        // aligned dwords covering NB bytes at any phase
    uint32_t d[NR];
    __device__ __forceinline__ int px(int k) const { return (int)((d[k >> 2] >> ((k & 3) * 8)) & 0xFFu); }
};

template <int WIN>
__device__ __forceinline__ TRow<WIN> trow_load(const uint8_t *roi, const PyrLevelDesc &L, int x, int y)
{
    constexpr int ND = TRow<WIN>::ND, NR = TRow<WIN>::NR;
    // rows beyond the padded buffer only feed derivatives of out-of-image rows (defined as 0): clamp
    y = y < -L.pady ? -L.pady : (y > L.h + L.pady - 1 ? L.h + L.pady - 1 : y);
    const uint8_t *rp = roi + m24(y, L.img_pitch);
    const int x1 = x - 1, xa = x1 & ~3, sh = x1 - xa;
    const uint32_t *p = (const uint32_t *)(rp + xa);
    uint32_t raw[ND];
#pragma unroll
    for (int i = 0; i < ND; i++) raw[i] = p[i];
    TRow<WIN> r;
#pragma unroll
    for (int i = 0; i < NR; i++) r.d[i] = __builtin_amdgcn_alignbyte(raw[i + 1 < ND ? i + 1 : i], raw[i], (uint32_t)sh);
    return r;
}

template <int WIN, int CTRL>
__device__ __forceinline__ TRow<WIN> trow_dpp(const TRow<WIN> &r)
{
    TRow<WIN> o;
#pragma unroll
    for (int i = 0; i < TRow<WIN>::NR; i++) o.d[i] = (uint32_t)dpp_mov<CTRL>((int)r.d[i]);
    return o;
}

// ---- search-image neighbourhood held in registers -----------------------------------------
// Lane r of the 16-lane row holds bytes [jx0, jx0+16) of image row jy0 + r: a 16x16 block that
// contains every (WIN+1)^2 bilinear footprint whose origin lies within +-NBH_R pixels of the
// position the level started from.  A Gauss-Newton iteration then needs NO global load: it pulls
// its two source rows from lanes r+oy, r+oy+1 (ds_bpermute inside the row / DPP) and re-phases the
// columns with v_alignbyte.  Only when the track drifts further than NBH_R pixels is the block
// re-fetched.  (SQ_WAIT_ANY was 56 % of the wave cycles with one global round trip per iteration.)
template <int WIN> struct Nbh { static constexpr int R = (15 - WIN) / 2; static constexpr bool USE = (15 - WIN) / 2 >= 2; };

struct NbhRegs { uint32_t d[4]; };

__device__ __forceinline__ NbhRegs nbh_load(const uint8_t *jroi, const PyrLevelDesc &LJ, int jx0, int jy0, int r)
{
    // rows beyond the padded image are never consumed by a legal window: clamp them into the buffer
    int row = jy0 + r;
    row = row < -LJ.pady ? -LJ.pady : (row > LJ.h + LJ.pady - 1 ? LJ.h + LJ.pady - 1 : row);
    const uint8_t *rp = jroi + m24(row, LJ.img_pitch);
    const int xa = jx0 & ~3, sh = jx0 - xa;
    const uint32_t *p = (const uint32_t *)(rp + xa);
    uint32_t raw[5];
#pragma unroll
    for (int i = 0; i < 5; i++) raw[i] = p[i];
    NbhRegs n;
#pragma unroll
    for (int i = 0; i < 4; i++) n.d[i] = __builtin_amdgcn_alignbyte(raw[i + 1], raw[i], (uint32_t)sh);
    return n;
}

// window rows for lane r at offset (ox, oy) inside the neighbourhood
template <int WIN>
__device__ __forceinline__ RowBytes<WIN> nbh_window_row(const NbhRegs &n, int ox, int oy, int r)
{
    int src = r + oy; src = src > 15 ? 15 : src;
    uint32_t g[5];
#pragma unroll
    for (int k = 0; k < 4; k++) g[k] = (uint32_t)__shfl((int)n.d[k], src, 16);
    g[4] = 0;
    const int q = ox >> 2;
    const uint32_t sh = (uint32_t)(ox & 3);
    RowBytes<WIN> o;
#pragma unroll
    for (int i = 0; i < RowBytes<WIN>::NR; i++) {
        // dword offset q = ox / 4 is 0..2 (ox <= 2 R <= 10); g[] lives in registers, so select instead of indexing
        auto pick = [&](int k) -> uint32_t {
            const uint32_t a0 = k <= 4 ? g[k] : 0u, a1 = k + 1 <= 4 ? g[k + 1] : 0u, a2 = k + 2 <= 4 ? g[k + 2] : 0u;
            return q == 0 ? a0 : (q == 1 ? a1 : a2);
        };
        o.d[i] = __builtin_amdgcn_alignbyte(pick(i + 1), pick(i), sh);
    }
    return o;
}

struct LKPointState {
    float nx, ny;     // nextPts[i] as OpenCV stores it between levels
    int status;
    float err;
    int iters;
    int visits;
};

// One pyramid level for the keypoint owned by this 16-lane row.
// I = template level (image + derivative), J = search level (image only).
// FACC (OV2_OPT_LK_ACC = OV2_LK_ACC_FLOAT_UI4): the sums of the normal matrix and of the mismatch vector in FLOAT accumulators, in the
// order an x86 OpenCV 4.x build executes them (`typedef float acctype` + 128-bit universal intrinsics, no FMA; public lkpyramid.cpp,
// restated by oracle/frontend.c ORC_LK_ACC_FLOAT_UI4): the first 4 * (WIN / 4) columns of a window row feed four lane accumulators per
// sum (lane = x & 3) row after row, the remaining columns a scalar accumulator; for the mismatch vector the first 8 * (WIN / 8)
// columns feed eight lanes with (float)(d[p] g[p] + d[p + 4] g[p + 4]) (v_dotprod adds the pair exactly), the rest a scalar.  Float
// addition does not associate: every lane of the keypoint's 16-lane row replays the chain in order, pulling row y's terms from lane y.
template <int WIN, bool FACC>
__device__ __forceinline__ void lk_level(const uint8_t *__restrict__ itemI, const PyrLevelDesc &LI,
                                         const uint8_t *__restrict__ itemJ, const PyrLevelDesc &LJ,
                                         const LKParams &prm, int level, int top_level, bool use_initial,
                                         float px0, float py0, int r, LKPointState &st)
{
    const float halfWin = (float)(WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const float W14 = (float)(1 << 14);
    const float lvl_scale = (float)(1. / (double)(1 << level));

    float prevx = px0 * lvl_scale, prevy = py0 * lvl_scale;
    float nextx, nexty;
    if (level == top_level) {
        if (use_initial) { nextx = st.nx * lvl_scale; nexty = st.ny * lvl_scale; }
        else { nextx = prevx; nexty = prevy; }
    } else {
        nextx = st.nx * 2.f; nexty = st.ny * 2.f;
    }
    st.nx = nextx; st.ny = nexty;

    prevx -= halfWin; prevy -= halfWin;
    const int ipx = cv_floor(prevx), ipy = cv_floor(prevy);
    if (ipx < -WIN || ipx >= LI.w || ipy < -WIN || ipy >= LI.h) {
        if (level == 0) { st.status = 0; st.err = 0.f; }
        return;
    }
    st.visits++;
    float a = prevx - (float)ipx, b = prevy - (float)ipy;
    int iw00 = cv_round((1.f - a) * (1.f - b) * W14);
    int iw01 = cv_round(a * (1.f - b) * W14);
    int iw10 = cv_round((1.f - a) * b * W14);
    int iw11 = (1 << 14) - iw00 - iw01 - iw10;

    const int rr = r <= WIN ? r : WIN;          // idle lanes (r > WIN) re-read the last row; results are masked
    const bool row_active = r < WIN;

    // search-image neighbourhood around the position this level starts from (loads issued together
    // with the template loads below: one memory round trip per level)
    const uint8_t *jroi = itemJ + LJ.img_roi;
    int jx0 = 0, jy0 = 0;
    NbhRegs jn;
    jn.d[0] = jn.d[1] = jn.d[2] = jn.d[3] = 0;
    if (Nbh<WIN>::USE) {
        // clamp the start so that an out-of-image search position (rejected below) cannot index outside the buffer
        const float sx = nextx - halfWin, sy = nexty - halfWin;
        int cx0 = cv_floor(fminf(fmaxf(sx, (float)(-WIN)), (float)(LJ.w - 1)));
        int cy0 = cv_floor(fminf(fmaxf(sy, (float)(-WIN)), (float)(LJ.h - 1)));
        jx0 = cx0 - Nbh<WIN>::R; jy0 = cy0 - Nbh<WIN>::R;
        jn = nbh_load(jroi, LJ, jx0, jy0, r);
    }

    // ---- template patch: I (5 fractional bits), dIx, dIy; exact sums of products ----
    // The Scharr derivative image of the reference's pyramid is NOT materialised: lane q fetches image
    // row ipy-1+q (columns ipx-1 .. ipx+WIN+1), the three rows below arrive through DPP row_shl:1..3,
    // and the derivative at the (WIN+1)^2 integer positions of the bilinear footprint is evaluated in
    // registers -- same integers as calcSharrDeriv + copyMakeBorder(BORDER_CONSTANT 0).
    int Iw[WIN], dIx[WIN], dIy[WIN];
    double A11d, A12d, A22d;
    {
        const uint8_t *iroi = itemI + LI.img_roi;
        const int rq = r <= WIN + 2 ? r : WIN + 2;                 // lanes 0..WIN+2 fetch rows ipy-1 .. ipy+WIN+1
        const TRow<WIN> R0 = trow_load<WIN>(iroi, LI, ipx, ipy - 1 + rq);
        const TRow<WIN> R1 = trow_dpp<WIN, DPP_ROW_SHL1>(R0);      // row ipy + r
        const TRow<WIN> R2 = trow_dpp<WIN, DPP_ROW_SHL2>(R0);      // row ipy + r + 1
        const TRow<WIN> R3 = trow_dpp<WIN, DPP_ROW_SHL3>(R0);      // row ipy + r + 2
        // derivative rows Y0 = ipy + r (rows R0,R1,R2) and Y1 = Y0 + 1 (rows R1,R2,R3); byte k <-> column ipx-1+k
        const int Y0 = ipy + rr;
        const bool y0in = Y0 >= 0 && Y0 < LI.h, y1in = Y0 + 1 >= 0 && Y0 + 1 < LI.h;
        int t0a[WIN + 3], t1a[WIN + 3], t0b[WIN + 3], t1b[WIN + 3];
#pragma unroll
        for (int k = 0; k < WIN + 3; k++) {
            const int p0 = R0.px(k), p1 = R1.px(k), p2 = R2.px(k), p3 = R3.px(k);
            t0a[k] = (p0 + p2) * 3 + p1 * 10; t1a[k] = p2 - p0;
            t0b[k] = (p1 + p3) * 3 + p2 * 10; t1b[k] = p3 - p1;
        }
        int dx0[WIN + 1], dy0[WIN + 1], dx1[WIN + 1], dy1[WIN + 1];
#pragma unroll
        for (int c = 0; c <= WIN; c++) {
            const int X = ipx + c;
            const bool xin = X >= 0 && X < LI.w;
            const bool in0 = xin && y0in, in1 = xin && y1in;
            dx0[c] = in0 ? t0a[c + 2] - t0a[c] : 0;
            dy0[c] = in0 ? (t1a[c] + t1a[c + 2]) * 3 + t1a[c + 1] * 10 : 0;
            dx1[c] = in1 ? t0b[c + 2] - t0b[c] : 0;
            dy1[c] = in1 ? (t1b[c] + t1b[c + 2]) * 3 + t1b[c + 1] * 10 : 0;
        }
        int s11 = 0, s12 = 0, s22 = 0;
#pragma unroll
        for (int x = 0; x < WIN; x++) {
            const int ival = descale(m24(R1.px(x + 1), iw00) + m24(R1.px(x + 2), iw01) + m24(R2.px(x + 1), iw10) + m24(R2.px(x + 2), iw11), 14 - 5);
            const int ixval = descale(m24(dx0[x], iw00) + m24(dx0[x + 1], iw01) + m24(dx1[x], iw10) + m24(dx1[x + 1], iw11), 14);
            const int iyval = descale(m24(dy0[x], iw00) + m24(dy0[x + 1], iw01) + m24(dy1[x], iw10) + m24(dy1[x + 1], iw11), 14);
            Iw[x] = ival; dIx[x] = ixval; dIy[x] = iyval;
            if (row_active) { s11 += m24(ixval, ixval); s12 += m24(ixval, iyval); s22 += m24(iyval, iyval); }
        }
        // per-lane partials: WIN * 4080^2 < 2^31 for WIN <= 15
        if (!FACC) {
            A11d = row_allreduce_exact(s11);
            A12d = row_allreduce_exact(s12);
            A22d = row_allreduce_exact(s22);
        } else { A11d = A12d = A22d = 0; }
    }
    float A11, A12, A22;
    if (!FACC) { A11 = (float)A11d * FLT_SCALE; A12 = (float)A12d * FLT_SCALE; A22 = (float)A22d * FLT_SCALE; }
    else {
        constexpr int SIMD_A = 4 * (WIN / 4);
        float q11[4] = {0.f, 0.f, 0.f, 0.f}, q12[4] = {0.f, 0.f, 0.f, 0.f}, q22[4] = {0.f, 0.f, 0.f, 0.f}, f11 = 0.f, f12 = 0.f, f22 = 0.f;
#pragma unroll 1
        for (int y = 0; y < WIN; y++) {
#pragma unroll
            for (int x = 0; x < WIN; x++) {
                const int ixv = __shfl(dIx[x], y, 16), iyv = __shfl(dIy[x], y, 16);
                if (x < SIMD_A) {                                   // qA = qA + fx * fx  (v_muladd without FMA in baseline builds)
                    const float fx = (float)ixv, fy = (float)iyv;
                    q22[x & 3] = q22[x & 3] + fy * fy; q12[x & 3] = q12[x & 3] + fx * fy; q11[x & 3] = q11[x & 3] + fx * fx;
                } else {                                            // iA11 += (itemtype)(ixval * ixval)
                    f11 += (float)(ixv * ixv); f12 += (float)(ixv * iyv); f22 += (float)(iyv * iyv);
                }
            }
        }
        if (SIMD_A > 0) {                                           // v_reduce_sum: (a0 + a2) + (a1 + a3)
            f11 += (q11[0] + q11[2]) + (q11[1] + q11[3]); f12 += (q12[0] + q12[2]) + (q12[1] + q12[3]); f22 += (q22[0] + q22[2]) + (q22[1] + q22[3]);
        }
        A11 = f11 * FLT_SCALE; A12 = f12 * FLT_SCALE; A22 = f22 * FLT_SCALE;
    }
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
    if (prm.flags & OV2_LK_GET_MIN_EIGENVALS) st.err = minEig;
    if (minEig < prm.min_eig_th || D < FLT_EPSILON) {
        if (level == 0) st.status = 0;
        return;
    }
    D = 1.f / D;
    nextx -= halfWin; nexty -= halfWin;
    float pdx = 0.f, pdy = 0.f;
    for (int j = 0; j < prm.max_iter; j++) {
        const int inx = cv_floor(nextx), iny = cv_floor(nexty);
        if (inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h) {
            if (level == 0) st.status = 0;
            break;
        }
        st.iters++;
        a = nextx - (float)inx; b = nexty - (float)iny;
        iw00 = cv_round((1.f - a) * (1.f - b) * W14);
        iw01 = cv_round(a * (1.f - b) * W14);
        iw10 = cv_round((1.f - a) * b * W14);
        iw11 = (1 << 14) - iw00 - iw01 - iw10;
        RowBytes<WIN> r0;
        if (Nbh<WIN>::USE) {
            int ox = inx - jx0, oy = iny - jy0;
            if ((unsigned)ox > (unsigned)(2 * Nbh<WIN>::R) || (unsigned)oy > (unsigned)(2 * Nbh<WIN>::R)) {
                jx0 = inx - Nbh<WIN>::R; jy0 = iny - Nbh<WIN>::R;            // drifted: re-centre the block
                jn = nbh_load(jroi, LJ, jx0, jy0, r);
                ox = Nbh<WIN>::R; oy = Nbh<WIN>::R;
            }
            r0 = nbh_window_row<WIN>(jn, ox, oy, r);
        } else {
            r0 = load_row<WIN>(jroi + m24(iny + rr, LJ.img_pitch), inx);
        }
        const RowBytes<WIN> r1 = row_from_next_lane<WIN>(r0);
        int sb1 = 0, sb2 = 0;
        int dv[WIN];
#pragma unroll
        for (int x = 0; x < WIN; x++) {
            const int diff = descale(m24(r0.px(x), iw00) + m24(r0.px(x + 1), iw01) + m24(r1.px(x), iw10) + m24(r1.px(x + 1), iw11), 14 - 5) - Iw[x];
            dv[x] = diff;
            sb1 += m24(diff, dIx[x]);
            sb2 += m24(diff, dIy[x]);
        }
        float b1, b2;
        if (!FACC) {
            if (!row_active) { sb1 = 0; sb2 = 0; }
            // |diff * dI| <= 8160*4080 -> per-lane partial < WIN * 3.33e7 < 2^31 for WIN <= 15
            b1 = (float)row_allreduce_exact(sb1) * FLT_SCALE;
            b2 = (float)row_allreduce_exact(sb2) * FLT_SCALE;
        } else {
            // this lane's (= window row's) terms, then the chains over the rows in order
            constexpr int SIMD_B = 8 * (WIN / 8), NT = WIN - SIMD_B;
            float tq[8], tx[NT > 0 ? NT : 1], ty[NT > 0 ? NT : 1];
            if (SIMD_B > 0) {
#pragma unroll
                for (int m = 0; m < 4; m++) {                       // lanes (x, y, x, y) of qb0 / qb1: pixels m and m + 4
                    tq[2 * m] = (float)(dv[m] * dIx[m] + dv[m + 4] * dIx[m + 4]);
                    tq[2 * m + 1] = (float)(dv[m] * dIy[m] + dv[m + 4] * dIy[m + 4]);
                }
            }
#pragma unroll
            for (int x = SIMD_B; x < WIN; x++) { tx[x - SIMD_B] = (float)(dv[x] * dIx[x]); ty[x - SIMD_B] = (float)(dv[x] * dIy[x]); }
            float qb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, fb1 = 0.f, fb2 = 0.f;
#pragma unroll 1
            for (int y = 0; y < WIN; y++) {
                if (SIMD_B > 0) {
#pragma unroll
                    for (int m = 0; m < 8; m++) qb[m] = qb[m] + __shfl(tq[m], y, 16);
                }
#pragma unroll
                for (int x = 0; x < NT; x++) { fb1 += __shfl(tx[x], y, 16); fb2 += __shfl(ty[x], y, 16); }
            }
            if (SIMD_B > 0) {                                       // qb0 = lanes 0..3, qb1 = lanes 4..7: s = qb0 + qb1; ib1 += s0 + s2, ib2 += s1 + s3
                const float s0 = qb[0] + qb[4], s1 = qb[1] + qb[5], s2 = qb[2] + qb[6], s3 = qb[3] + qb[7];
                fb1 += s0 + s2; fb2 += s1 + s3;
            }
            b1 = fb1 * FLT_SCALE; b2 = fb2 * FLT_SCALE;
        }
        const float dx = (A12 * b2 - A22 * b1) * D;
        const float dy = (A12 * b1 - A11 * b2) * D;
        nextx += dx; nexty += dy;
        st.nx = nextx + halfWin; st.ny = nexty + halfWin;
        if ((double)dx * (double)dx + (double)dy * (double)dy <= prm.eps2) break;
        if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
            st.nx -= dx * 0.5f; st.ny -= dy * 0.5f;
            break;
        }
        pdx = dx; pdy = dy;
    }
}

// fbKltTracking for the keypoint owned by this 16-lane row: forward levels max_level..0 (feature_tracker.cpp:66-69),
// the status / err / border filter (:79-101), the backward track at level 0 from the original keypoint (:113-116) and
// the forward-backward distance test (:119-134).  Returns the status; (fx, fy) = vpriorkps[i] after the forward call.
template <int WIN, bool FACC>
__device__ __forceinline__ int fb_track_point(const uint8_t *__restrict__ itemP, const uint8_t *__restrict__ itemC,
                                              const PyrDesc &P, const PyrDesc &C, const LKParams &prm, int max_level,
                                              float2 kp, float2 pr, int r, float &fx, float &fy, LKPointState &st)
{
    st.nx = pr.x; st.ny = pr.y; st.status = 1; st.err = 0.f; st.iters = 0; st.visits = 0;
    // forward: prev -> cur, levels max_level..0   (feature_tracker.cpp:66-69)
    for (int level = max_level; level >= 0; level--)
        lk_level<WIN, FACC>(itemP, P.lv[level], itemC, C.lv[level], prm, level, max_level,
                      (prm.flags & OV2_LK_USE_INITIAL_FLOW) != 0, kp.x, kp.y, r, st);
    fx = st.nx; fy = st.ny;
    int ok = st.status;
    if (prm.do_fb) {
        // feature_tracker.cpp:79-101
        if (ok && st.err > prm.err_th) ok = 0;
        const float W0 = (float)C.lv[0].w, H0 = (float)C.lv[0].h;
        if (ok && !(1.f <= fx && fx < W0 - 1.f && 1.f <= fy && fy < H0 - 1.f)) ok = 0;   // inBorder :216-221
        if (ok) {
            // backward: cur -> prev at level 0, initial guess = original keypoint (:113-116)
            LKPointState sb;
            sb.nx = kp.x; sb.ny = kp.y; sb.status = 1; sb.err = 0.f; sb.iters = 0; sb.visits = 0;
            lk_level<WIN, FACC>(itemC, C.lv[0], itemP, P.lv[0], prm, 0, 0, true, fx, fy, r, sb);
            st.iters += sb.iters; st.visits += sb.visits;
            if (!sb.status) ok = 0;
            else {
                const float ddx = kp.x - sb.nx, ddy = kp.y - sb.ny;      // cv::norm(Point2f) (:128)
                const double nrm = sqrt((double)ddx * (double)ddx + (double)ddy * (double)ddy);
                if (nrm > (double)prm.fb_dist) ok = 0;
            }
        }
    }
    return ok;
}

template <int WIN, bool FACC>
__global__ __launch_bounds__(256) void k_fb_klt(PyrDesc P, PyrDesc C, LKParams prm,
                                                const float2 *__restrict__ kps, float2 *__restrict__ priors,
                                                uint8_t *__restrict__ status, float *__restrict__ err_out,
                                                int *__restrict__ iters_out, const int *__restrict__ n_per_item,
                                                unsigned long long *__restrict__ stats)
{
    const int b = blockIdx.y;
    const int n = n_per_item ? n_per_item[b] : prm.n_max;
    const int r = threadIdx.x & 15;
    const int i = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    __shared__ unsigned int s_stats[2];
    if (stats) {                                         // uniform branch (kernel argument)
        if (threadIdx.x < 2) s_stats[threadIdx.x] = 0;
        __syncthreads();
    }
    if (i < n) {                                         // whole 16-lane rows take the same side
        const long long gi = (long long)b * prm.n_max + i;
        const uint8_t *itemP = P.base + (long long)b * P.item_stride;
        const uint8_t *itemC = C.base + (long long)b * C.item_stride;
        LKPointState st;
        float fx, fy;
        const int ok = fb_track_point<WIN, FACC>(itemP, itemC, P, C, prm, prm.max_level, kps[gi], priors[gi], r, fx, fy, st);
        if (r == 0) {
            priors[gi] = make_float2(fx, fy);
            status[gi] = (uint8_t)ok;
            if (err_out) err_out[gi] = st.err;
            if (iters_out) iters_out[gi] = st.iters;
            if (stats) {                                 // workgroup-level pre-reduction in LDS
                atomicAdd(&s_stats[0], (unsigned int)st.iters);
                atomicAdd(&s_stats[1], (unsigned int)st.visits);
            }
        }
    }
    if (stats) {
        __syncthreads();
        if (threadIdx.x < 2 && s_stats[threadIdx.x])
            atomicAdd(&stats[((blockIdx.y * gridDim.x + blockIdx.x) & (LK_STAT_SLOTS - 1)) * LK_STAT_STRIDE + threadIdx.x],
                      (unsigned long long)s_stats[threadIdx.x]);
    }
}

// ---- VisualFrontEnd::kltTracking in ONE launch (src/visual_front_end.cpp:132-275) ------------------------------
// The reference tracks the keypoints that carry a 3-D prior on `lvl_prior` (= 1) pyramid levels first (:186-218),
// then everything else on `lvl_full` (= nklt_pyr_lvl_) levels (:239-268) -- and every keypoint the first call lost is
// appended to the second call with the forward result of the first call as its prior (:213-217: v3dpriors was
// updated in place by fbKltTracking, feature_tracker.cpp:66).  Keypoints are independent inside fbKltTracking, so
// the two calls and the retry are one pass over the keypoints: flag bit 0 = "has a 3-D prior".
// status bit 0 = tracked, bit 1 = the prior-pass failed and the result comes from the full-pyramid retry.  The one
// cross-keypoint decision of the reference (:225-230: fewer than a third of the prior tracks good -> retry from the
// keypoints themselves) is taken by the host from the bit-1 count (ov2_tracker_klt, track.hip).
// One wavefront per work-group (1 or 4 keypoints, see the launcher): a single frame has ~300 keypoints on 1024 SIMDs.
template <int WIN, bool FACC>
__global__ __launch_bounds__(64) void k_track_klt(PyrDesc P, PyrDesc C, LKParams prm, int lvl_prior, int lvl_full,
                                                  const int *__restrict__ n_dev, const float2 *__restrict__ kps,
                                                  const float2 *__restrict__ priors, const uint8_t *__restrict__ flags,
                                                  float2 *__restrict__ out_xy, uint8_t *__restrict__ status,
                                                  int *__restrict__ iters_out, const float *__restrict__ sad_x, float sad_up)
{
    const int item = blockIdx.y;                                     // batch item (lock-step tracker, trackb.hip); 0 for one camera
    const int n = n_dev ? n_dev[item] : prm.n_max;
    const int r = threadIdx.x & 15;
    // blockDim.x = 64: four keypoints per wavefront; blockDim.x = 16: one (A/B switch of the launcher)
    const int il = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    if (il >= n) return;
    const int i = item * prm.n_max + il;
    const uint8_t *pb = P.base + (long long)item * P.item_stride, *cb = C.base + (long long)item * C.item_stride;
    const float2 kp = kps[i];
    float2 pr = priors[i];
    const bool has_prior = (flags[i] & 1) != 0;
    // sad_x != NULL: MapManager::stereoMatching (src/map_manager.cpp:367-611) -- same two-call structure, but (i) a keypoint
    // without a 3-D prior starts from its own position with x replaced by the getLineMinSAD prior of the coarsest level when
    // that lies left of it (:433-437); (ii) as in kltTracking, a 3-D prior track that fails is retried on the full pyramid
    // from the first call's FORWARD RESULT: `vpriors.push_back(v3dpriors.at(i))` (:533-538) runs after fbKltTracking has
    // overwritten v3dpriors in place (calcOpticalFlowPyrLK writes nextPts into it, feature_tracker.cpp:66)
    const bool stereo = sad_x != nullptr;
    if (stereo && !has_prior) {
        pr = kp;
        const float xp = sad_x[i] * sad_up;
        if (xp >= 0.f && xp <= kp.x) pr.x = xp;
    }
    int max_level = has_prior ? lvl_prior : lvl_full;
    int ok = 0, retried = 0, iters = 0;
    float fx = 0.f, fy = 0.f;
    for (int attempt = 0; attempt < 2; attempt++) {
        LKPointState st;
        ok = fb_track_point<WIN, FACC>(pb, cb, P, C, prm, max_level, kp, pr, r, fx, fy, st);
        iters += st.iters;
        if (ok || !has_prior || attempt == 1) break;
        pr = make_float2(fx, fy);                                         // visual_front_end.cpp:213-217, map_manager.cpp:533-538
        max_level = lvl_full; retried = 1;
    }
    if (r == 0) {
        out_xy[i] = make_float2(fx, fy);
        status[i] = (uint8_t)(ok | (retried << 1));
        if (iters_out) iters_out[i] = iters;
    }
}

// folds (and clears) the per-slot partial sums of one LK launch into the caller's {iterations, patch builds}
__global__ __launch_bounds__(LK_STAT_SLOTS) void k_lk_stats_fold(unsigned long long *__restrict__ slots, long long *__restrict__ stats)
{
    __shared__ unsigned long long s[2][LK_STAT_SLOTS / 64];
    unsigned long long a = slots[threadIdx.x * LK_STAT_STRIDE], b = slots[threadIdx.x * LK_STAT_STRIDE + 1];
    if (a) slots[threadIdx.x * LK_STAT_STRIDE] = 0;
    if (b) slots[threadIdx.x * LK_STAT_STRIDE + 1] = 0;
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); }
    if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = a; s[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x < 2) {
        unsigned long long t = 0;
        for (int w = 0; w < LK_STAT_SLOTS / 64; w++) t += s[threadIdx.x][w];
        if (t) atomicAdd((unsigned long long *)&stats[threadIdx.x], t);
    }
}

// ---- host side -------------------------------------------------------------------
// lk3.hip: 3-lanes-per-keypoint kernel for the reference's window (9)
int ov2_launch_fb_klt3(hipStream_t s, const PyrDesc &P, const PyrDesc &C, int max_level, int max_iter, double eps2,
                       float min_eig_th, int flags, float err_th, float fb_dist, int do_fb, int n_max,
                       const float2 *kps, float2 *priors, uint8_t *status, float *err, int *iters,
                       const int *n_per_item, unsigned long long *stat_slots);

// Kernel choice for the reference's window (9): the 3-lanes-per-keypoint kernel needs >= ~3000 wavefronts of
// 20 keypoints to fill the 1024 SIMDs (offline batch-of-sequences mode); below that -- the single-sequence
// drop-in case: a few hundred keypoints -- the row-per-lane kernel has 5x more, 4x shorter wavefronts and
// the lower latency.  ov2_ctx_set_option(OV2_OPT_LK_IMPL) pins one of them (parity tests, A/B measurements).
static bool lk_use_row_kernel(const ov2_ctx *ctx, long long points)
{
    if (ctx->lk_impl == OV2_LK_IMPL_ROW) return true;
    if (ctx->lk_impl == OV2_LK_IMPL_LANE3) return false;
    return points < 65536;
}

template <int WIN>
static void launch_fb_klt(hipStream_t s, dim3 grid, const PyrDesc &P, const PyrDesc &C, const LKParams &prm,
                          const float2 *kps, float2 *priors, uint8_t *status, float *err, int *iters,
                          const int *n_per_item, unsigned long long *stats, int lk_acc)
{
    if (lk_acc == OV2_LK_ACC_FLOAT_UI4) hipLaunchKernelGGL((k_fb_klt<WIN, true>), grid, dim3(256), 0, s, P, C, prm, kps, priors, status, err, iters, n_per_item, stats);
    else hipLaunchKernelGGL((k_fb_klt<WIN, false>), grid, dim3(256), 0, s, P, C, prm, kps, priors, status, err, iters, n_per_item, stats);
}

static int lk_dispatch(ov2_ctx *ctx, const ov2_pyr *prev, const ov2_pyr *cur, LKParams prm,
                       const float2 *kps_d, float2 *priors_d, uint8_t *status_d, float *err_d, int *iters_d,
                       const int *n_per_item_d, long long *stats_d)
{
    const PyrDesc &P = prev->d, &C = cur->d;
    OV2_REQUIRE(P.n_levels == C.n_levels && P.batch == C.batch && P.win == C.win, OV2_EINVAL,
                "prev/cur pyramids differ in geometry");
    OV2_REQUIRE(P.lv[0].w == C.lv[0].w && P.lv[0].h == C.lv[0].h, OV2_EINVAL, "prev/cur image size differs");
    OV2_REQUIRE(prm.win == P.win, OV2_EINVAL, "LK window differs from the window the pyramid was padded for");
    if (int rcw = ov2_pyr_wait_ready(ctx, prev)) return rcw;      // pyramids built on another context's stream
    if (int rcw = ov2_pyr_wait_ready(ctx, cur)) return rcw;
    dim3 grid((prm.n_max + 15) / 16, P.batch);
    unsigned long long *slots = nullptr;
    if (stats_d) {
        if (int rc = ctx->reserve_stat_slots()) return rc;
        slots = ctx->stat_slots;
    }
    // (the float-accumulator mode lives in the row kernel: the 3-lanes-per-keypoint kernel sums exact integers only)
    if (prm.win == 9 && ctx->lk_acc == OV2_LK_ACC_INT64 && !lk_use_row_kernel(ctx, (long long)prm.n_max * P.batch)) {
        ov2_launch_fb_klt3(ctx->stream, P, C, prm.max_level, prm.max_iter, prm.eps2, prm.min_eig_th, prm.flags, prm.err_th,
                           prm.fb_dist, prm.do_fb, prm.n_max, kps_d, priors_d, status_d, err_d, iters_d, n_per_item_d, slots);
        if (slots) hipLaunchKernelGGL(k_lk_stats_fold, dim3(1), dim3(LK_STAT_SLOTS), 0, ctx->stream, slots, stats_d);
        OV2_HIP_CHECK(hipGetLastError());
        return OV2_OK;
    }
    switch (prm.win) {
    case 5:  launch_fb_klt<5>(ctx->stream, grid, P, C, prm, kps_d, priors_d, status_d, err_d, iters_d, n_per_item_d, slots, ctx->lk_acc); break;
    case 7:  launch_fb_klt<7>(ctx->stream, grid, P, C, prm, kps_d, priors_d, status_d, err_d, iters_d, n_per_item_d, slots, ctx->lk_acc); break;
    case 9:  launch_fb_klt<9>(ctx->stream, grid, P, C, prm, kps_d, priors_d, status_d, err_d, iters_d, n_per_item_d, slots, ctx->lk_acc); break;
    case 11: launch_fb_klt<11>(ctx->stream, grid, P, C, prm, kps_d, priors_d, status_d, err_d, iters_d, n_per_item_d, slots, ctx->lk_acc); break;
    case 13: launch_fb_klt<13>(ctx->stream, grid, P, C, prm, kps_d, priors_d, status_d, err_d, iters_d, n_per_item_d, slots, ctx->lk_acc); break;
    default:
        ov2_set_error("LK window %d has no kernel instance (supported: 5,7,9,11,13; the reference ships 9)", prm.win);
        return OV2_EUNSUPPORTED;
    }
    if (slots) hipLaunchKernelGGL(k_lk_stats_fold, dim3(1), dim3(LK_STAT_SLOTS), 0, ctx->stream, slots, stats_d);
    OV2_HIP_CHECK(hipGetLastError());
    return OV2_OK;
}

static LKParams make_params(const ov2_pyr *pyr, int win, int max_level, int max_iter, float eps, int flags,
                            float err_th, float fb_dist, int do_fb, int n_max)
{
    LKParams prm;
    prm.win = win;
    // calcOpticalFlowPyrLK: maxLevel = min(maxLevel, levels in the pyramid)
    prm.max_level = max_level > pyr->d.n_levels - 1 ? pyr->d.n_levels - 1 : max_level;
    if (prm.max_level < 0) prm.max_level = 0;
    prm.max_iter = max_iter < 0 ? 0 : (max_iter > 100 ? 100 : max_iter);
    double e = (double)eps;                      // TermCriteria::epsilon is a double holding the float
    if (e < 0.) e = 0.;
    if (e > 10.) e = 10.;
    prm.eps2 = e * e;
    prm.min_eig_th = 1e-4f;
    prm.flags = flags;
    prm.err_th = err_th; prm.fb_dist = fb_dist; prm.do_fb = do_fb; prm.n_max = n_max;
    return prm;
}

// launcher of the fused kltTracking kernel (used by track.hip); all pointers are device memory, *n_dev <= n_max
int ov2_launch_track_klt(hipStream_t s, const ov2_pyr *prev, const ov2_pyr *cur, int win, int lvl_prior, int lvl_full,
                         int max_iter, float eps, float err_th, float fb_dist, int n_max, const int *n_dev,
                         const float *kps, const float *priors, const uint8_t *flags, float *out_xy, uint8_t *status, int *iters,
                         const float *sad_x, float sad_up, int track_impl, int items, int lk_acc)
{
    const PyrDesc &P = prev->d, &C = cur->d;
    OV2_REQUIRE(P.n_levels == C.n_levels && items >= 1 && P.batch >= items && C.batch >= items && P.win == C.win && win == P.win, OV2_EINVAL,
                "tracker pyramids differ in geometry");
    LKParams prm = make_params(prev, win, lvl_full, max_iter, eps, OV2_LK_USE_INITIAL_FLOW | OV2_LK_GET_MIN_EIGENVALS,
                               err_th, fb_dist, 1, n_max);
    const int lf = prm.max_level;                                  // clamped to the pyramid like feature_tracker.cpp:50-52
    const int lp = lvl_prior > P.n_levels - 1 ? P.n_levels - 1 : (lvl_prior < 0 ? 0 : lvl_prior);
    // window 9 (the reference's): the wavefront-per-keypoint kernel (lkw.hip) -- the 81 window pixels over all 64 lanes, the next
    // level's search block requested while the current level iterates.  Other windows (and OV2_OPT_TRACK_IMPL = ROW): the
    // row-per-lane kernel, four keypoints per wavefront
    if (win == 9 && track_impl == OV2_TRACK_IMPL_WAVE)
        return ov2_launch_track_klt_w(s, P, C, prm, lp, lf, n_max, n_dev, kps, priors, flags, out_xy, status, iters, sad_x, sad_up, items, lk_acc);
    constexpr int kpw = 4;
    dim3 grid((n_max + kpw - 1) / kpw, items), block(16 * kpw);
#define OV2_TK(W) do { if (lk_acc == OV2_LK_ACC_FLOAT_UI4) hipLaunchKernelGGL((k_track_klt<W, true>), grid, block, 0, s, P, C, prm, lp, lf, n_dev, (const float2 *)kps, \
                                     (const float2 *)priors, flags, (float2 *)out_xy, status, iters, sad_x, sad_up); \
                       else hipLaunchKernelGGL((k_track_klt<W, false>), grid, block, 0, s, P, C, prm, lp, lf, n_dev, (const float2 *)kps, \
                                     (const float2 *)priors, flags, (float2 *)out_xy, status, iters, sad_x, sad_up); } while (0)
    switch (win) {
    case 5:  OV2_TK(5); break;
    case 7:  OV2_TK(7); break;
    case 9:  OV2_TK(9); break;
    case 11: OV2_TK(11); break;
    case 13: OV2_TK(13); break;
    default:
        ov2_set_error("LK window %d has no kernel instance (supported: 5,7,9,11,13; the reference ships 9)", win);
        return OV2_EUNSUPPORTED;
    }
#undef OV2_TK
    OV2_HIP_CHECK(hipGetLastError());
    return OV2_OK;
}

extern "C" {

int ov2_fb_klt_d(ov2_ctx *ctx, const ov2_pyr *prev, const ov2_pyr *cur,
                 int win, int nbpyrlvl, int max_iter, float eps, float err_th, float fb_dist,
                 const float *kps_xy_d, float *prior_xy_inout_d, int n_max, const int *n_per_item_d,
                 uint8_t *status_d, long long *stats_d)
{
    OV2_REQUIRE(ctx && prev && cur, OV2_EINVAL, "NULL argument");
    if (n_max <= 0) return OV2_OK;
    OV2_REQUIRE(kps_xy_d && prior_xy_inout_d && status_d, OV2_EINVAL, "NULL point buffer");
    OV2_REQUIRE(nbpyrlvl >= 0, OV2_EINVAL, "nbpyrlvl < 0");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    LKParams prm = make_params(prev, win, nbpyrlvl, max_iter, eps,
                               OV2_LK_USE_INITIAL_FLOW | OV2_LK_GET_MIN_EIGENVALS, err_th, fb_dist, 1, n_max);
    return lk_dispatch(ctx, prev, cur, prm, (const float2 *)kps_xy_d, (float2 *)prior_xy_inout_d, status_d,
                       nullptr, nullptr, n_per_item_d, stats_d);
}

// host-buffer wrappers: stage through the context's scratch, one H2D + one D2H
static int lk_host_call(ov2_ctx *ctx, const ov2_pyr *prev, const ov2_pyr *cur, LKParams prm,
                        const float *kps_h, float *priors_h, int n, uint8_t *status_h, float *err_h, int *iters_h,
                        long long *stats_h)
{
    OV2_REQUIRE(prev->d.batch == 1 && cur->d.batch == 1, OV2_EINVAL, "host-buffer LK entry points take batch=1 pyramids");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    // layout: [kps 8n][priors 8n][err 4n][iters 4n][stats 16][status n]
    const size_t o_kps = 0, o_pri = 8 * (size_t)n, o_err = 16 * (size_t)n, o_it = 20 * (size_t)n;
    const size_t o_stats = 24 * (size_t)n, o_st = o_stats + 16, total = o_st + (size_t)n;
    int rc = ctx->reserve_device(total);  if (rc) return rc;
    rc = ctx->reserve_host(total);        if (rc) return rc;
    uint8_t *hs = (uint8_t *)ctx->h_scratch, *ds = (uint8_t *)ctx->d_scratch;
    memcpy(hs + o_kps, kps_h, 8 * (size_t)n);
    memcpy(hs + o_pri, priors_h, 8 * (size_t)n);
    memset(hs + o_stats, 0, 16);
    OV2_HIP_CHECK(hipMemcpyAsync(ds, hs, o_err, hipMemcpyHostToDevice, ctx->stream));
    OV2_HIP_CHECK(hipMemsetAsync(ds + o_stats, 0, 16, ctx->stream));
    rc = lk_dispatch(ctx, prev, cur, prm, (const float2 *)(ds + o_kps), (float2 *)(ds + o_pri), ds + o_st,
                     (float *)(ds + o_err), (int *)(ds + o_it), nullptr, (long long *)(ds + o_stats));
    if (rc) return rc;
    OV2_HIP_CHECK(hipMemcpyAsync(hs + o_pri, ds + o_pri, total - o_pri, hipMemcpyDeviceToHost, ctx->stream));
    OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    memcpy(priors_h, hs + o_pri, 8 * (size_t)n);
    memcpy(status_h, hs + o_st, (size_t)n);
    if (err_h) memcpy(err_h, hs + o_err, 4 * (size_t)n);
    if (iters_h) memcpy(iters_h, hs + o_it, 4 * (size_t)n);
    if (stats_h) memcpy(stats_h, hs + o_stats, 16);
    return OV2_OK;
}

int ov2_fb_klt(ov2_ctx *ctx, const ov2_pyr *prev, const ov2_pyr *cur,
               int win, int nbpyrlvl, int max_iter, float eps, float err_th, float fb_dist,
               const float *kps_xy_h, float *prior_xy_inout_h, int n,
               uint8_t *status_h, long long stats[2])
{
    OV2_REQUIRE(ctx && prev && cur, OV2_EINVAL, "NULL argument");
    if (n <= 0) return OV2_OK;                       // feature_tracker.cpp:43-46
    OV2_REQUIRE(kps_xy_h && prior_xy_inout_h && status_h, OV2_EINVAL, "NULL point buffer");
    OV2_REQUIRE(nbpyrlvl >= 0, OV2_EINVAL, "nbpyrlvl < 0");
    LKParams prm = make_params(prev, win, nbpyrlvl, max_iter, eps,
                               OV2_LK_USE_INITIAL_FLOW | OV2_LK_GET_MIN_EIGENVALS, err_th, fb_dist, 1, n);
    return lk_host_call(ctx, prev, cur, prm, kps_xy_h, prior_xy_inout_h, n, status_h, nullptr, nullptr, stats);
}

int ov2_lk_track(ov2_ctx *ctx, const ov2_pyr *prev, const ov2_pyr *next,
                 int win, int max_level, int max_iter, float eps, int flags,
                 const float *prev_xy_h, float *next_xy_inout_h, int n,
                 uint8_t *status_h, float *err_h, int *iters_h)
{
    OV2_REQUIRE(ctx && prev && next, OV2_EINVAL, "NULL argument");
    if (n <= 0) return OV2_OK;
    OV2_REQUIRE(prev_xy_h && next_xy_inout_h && status_h, OV2_EINVAL, "NULL point buffer");
    OV2_REQUIRE(max_level >= 0, OV2_EINVAL, "max_level < 0");
    LKParams prm = make_params(prev, win, max_level, max_iter, eps, flags, 0.f, 0.f, 0, n);
    return lk_host_call(ctx, prev, next, prm, prev_xy_h, next_xy_inout_h, n, status_h, err_h, iters_h, nullptr);
}

} // extern "C"
