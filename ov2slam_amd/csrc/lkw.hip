// lkw.hip -- VisualFrontEnd::kltTracking for ONE camera frame: a whole wavefront per keypoint (gfx950, wave64).
//
// Same arithmetic and the same reference semantics as lk.hip / lk3.hip (cv::calcOpticalFlowPyrLK inside
// FeatureTracker::fbKltTracking, /root/reference/src/feature_tracker.cpp:35-137; both calls and the retry of
// VisualFrontEnd::kltTracking, src/visual_front_end.cpp:132-275; the stereo variant of MapManager::stereoMatching,
// src/map_manager.cpp:497-565), different work mapping.  A single frame has ~300 keypoints on 1024 SIMDs: nothing is
// throughput-bound, the frame's latency is every keypoint's own dependent chain -- ~5 level visits of (global round trip ->
// template build -> ~4 Gauss-Newton trips).  k_track_klt (lk.hip) gives a keypoint 16 lanes (lane = window row, 9 pixels per
// lane): ~240 dependent wave-instructions per trip.  Here the 81 window pixels are spread over all 64 lanes (pixels p and
// p + 64), so a trip is ~2 pixels of work per lane + two wave reductions:
//   * both neighbourhoods of a level are staged in LDS by the whole wavefront (one dword per lane and row segment), the
//     Scharr derivative is evaluated once on the 10 x 10 integer grid the bilinear footprints touch (one position per lane),
//   * integer partial sums are reduced exactly: 4 DPP steps inside each 16-lane row (|row sum| < 2^31), the four row totals
//     through v_readlane in fp64 -- bit-identical to the oracle's int64 accumulation,
//   * one keypoint per wavefront: every branch (level skip, re-centring fetch, convergence) is wave-uniform.
// The search block of the NEXT level is requested while the current level iterates (its position is predicted from the current
// estimate; the +-3-pixel margin of the block absorbs the rest, and the ordinary re-centring fetch is the fall-back).
#include "common.hpp"
#include "lk_params.hpp"
#include "keypoint_dev.hpp"
#include <float.h>
#include <math.h>

#pragma clang fp contract(off)

#define W_WIN 9
#define W_NPIX (W_WIN * W_WIN)        // 81
#define W_IROWS 12                    // template neighbourhood rows ipy-1 .. ipy+10, 16 bytes each from the aligned column
#define W_JROWS 16                    // search neighbourhood rows, 20 bytes each
#define W_R 3                         // margin of the search block around the start position
#define W_GRID 10                     // derivative grid (WIN + 1)^2
#define W_LDS_INT (W_IROWS * 4 + W_JROWS * 5 + W_GRID * W_GRID + 4)   // dwords of the integer-sum kernel
#define W_FACC_PIX 84                 // float-accumulator mode: packed template gradients + the trip's differences, one dword per window pixel each

template <int CTRL>
__device__ __forceinline__ int w_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ int w_round(float v) { return __float2int_rn(v); }
__device__ __forceinline__ int w_floor(float v) { return (int)floorf(v); }
__device__ __forceinline__ int w_descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ int w_m24(int a, int b) { return __mul24(a, b); }
__device__ __forceinline__ void w_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// exact sum of the 64 per-lane partials as a double (|16-lane row sum| < 2^31), the same value in every lane
__device__ __forceinline__ double w_sum_exact(int p)
{
    p += w_dpp<0xB1>(p);              // quad_perm 1,0,3,2
    p += w_dpp<0x4E>(p);              // quad_perm 2,3,0,1
    p += w_dpp<0x124>(p);             // row_ror:4
    p += w_dpp<0x128>(p);             // row_ror:8
    return ((double)__builtin_amdgcn_readlane(p, 0) + (double)__builtin_amdgcn_readlane(p, 16)) +
           ((double)__builtin_amdgcn_readlane(p, 32) + (double)__builtin_amdgcn_readlane(p, 48));
}

struct WState { float nx, ny; int status; float err; int iters; };

// stage the 16 x 20-byte search block with origin (jx0, jy0) (rows clamped into the padded buffer: rows beyond it are never consumed)
__device__ __forceinline__ void w_fetch_J(const uint8_t *jroi, const PyrLevelDesc &LJ, int jx0, int jy0, int lane, uint32_t (&v)[2])
{
    const int xa = jx0 & ~3;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int e = lane + 64 * k;
        v[k] = 0;
        if (e < W_JROWS * 5) {
            const int row = e / 5, dwc = e - row * 5;
            int y = jy0 + row;
            y = y < -LJ.pady ? -LJ.pady : (y > LJ.h + LJ.pady - 1 ? LJ.h + LJ.pady - 1 : y);
            v[k] = *(const uint32_t *)(jroi + (long long)y * LJ.img_pitch + xa + 4 * dwc);
        }
    }
}
__device__ __forceinline__ void w_store_J(uint32_t *Jb, int lane, const uint32_t (&v)[2])
{
    Jb[lane] = v[0];
    if (lane + 64 < W_JROWS * 5) Jb[lane + 64] = v[1];
}

__device__ __forceinline__ float w_lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// One pyramid level for the keypoint owned by this wavefront.  pre_* : a search block requested earlier (by the previous level's
// visit) for origin (pre_x0, pre_y0), pre_ok != 0 when the registers pre_v hold it.
// FACC (OV2_OPT_LK_ACC = OV2_LK_ACC_FLOAT_UI4, see lk.hip): float accumulators in the order of an x86 OpenCV 4.x build.  Window 9: the
// normal matrix has 4 lane accumulators per sum over columns 0..7 (lane = x & 3) and a scalar one for column 8; the mismatch vector 8 lane
// accumulators fed with (float)(d[p] g[p] + d[p + 4] g[p + 4]) and two scalars for column 8.  Each accumulator is ONE lane of the wavefront
// here (15 resp. 10 lanes busy), walking the rows in order over values the other lanes left in LDS; the folds go through v_readlane.
template <bool FACC>
__device__ __forceinline__ void w_level(const uint8_t *__restrict__ itemI, const PyrLevelDesc &LI, const uint8_t *__restrict__ itemJ,
                                        const PyrLevelDesc &LJ, const PyrLevelDesc *LJnext, const LKParams &prm, int level, int top_level,
                                        bool use_initial, float px0, float py0, int lane, uint32_t *lds, WState &st,
                                        bool &pre_ok, int &pre_x0, int &pre_y0, uint32_t (&pre_v)[2])
{
    constexpr int WIN = W_WIN;
    uint32_t *Ib = lds, *Jb = lds + W_IROWS * 4, *Dg = Jb + W_JROWS * 5;       // 48 + 80 + 100 dwords
    const uint8_t *Ibb = (const uint8_t *)Ib, *Jbb = (const uint8_t *)Jb;
    const float halfWin = (float)(WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const float W14 = (float)(1 << 14);
    const float lvl_scale = (float)(1. / (double)(1 << level));
    const bool had_pre = pre_ok;
    pre_ok = false;

    float prevx = px0 * lvl_scale, prevy = py0 * lvl_scale;
    float nextx, nexty;
    if (level == top_level) {
        if (use_initial) { nextx = st.nx * lvl_scale; nexty = st.ny * lvl_scale; }
        else { nextx = prevx; nexty = prevy; }
    } else { nextx = st.nx * 2.f; nexty = st.ny * 2.f; }
    st.nx = nextx; st.ny = nexty;

    prevx -= halfWin; prevy -= halfWin;
    const int ipx = w_floor(prevx), ipy = w_floor(prevy);
    if (ipx < -WIN || ipx >= LI.w || ipy < -WIN || ipy >= LI.h) {
        if (level == 0) { st.status = 0; st.err = 0.f; }
        return;
    }
    float a = prevx - (float)ipx, b = prevy - (float)ipy;
    int iw00 = w_round((1.f - a) * (1.f - b) * W14);
    int iw01 = w_round(a * (1.f - b) * W14);
    int iw10 = w_round((1.f - a) * b * W14);
    int iw11 = (1 << 14) - iw00 - iw01 - iw10;

    // ---- stage both neighbourhoods: every global load of the visit is issued here ----
    const uint8_t *iroi = itemI + LI.img_roi, *jroi = itemJ + LJ.img_roi;
    const int ixa = (ipx - 1) & ~3, ish = (ipx - 1) - ixa;
    uint32_t iv = 0;
    if (lane < W_IROWS * 4) {
        const int row = lane >> 2, dwc = lane & 3;
        int y = ipy - 1 + row;
        y = y < -LI.pady ? -LI.pady : (y > LI.h + LI.pady - 1 ? LI.h + LI.pady - 1 : y);   // only feeds derivatives of out-of-image rows (= 0)
        iv = *(const uint32_t *)(iroi + (long long)y * LI.img_pitch + ixa + 4 * dwc);
    }
    const float sx = nextx - halfWin, sy = nexty - halfWin;
    int jx0 = w_floor(fminf(fmaxf(sx, (float)(-WIN)), (float)(LJ.w - 1))) - W_R;
    int jy0 = w_floor(fminf(fmaxf(sy, (float)(-WIN)), (float)(LJ.h - 1))) - W_R;
    uint32_t jv[2];
    // the block requested during the previous level serves when this level's first window lies inside it with room to move
    const int fx = w_floor(sx) - pre_x0, fy = w_floor(sy) - pre_y0;
    const bool use_pre = had_pre && (unsigned)(fx - 1) <= (unsigned)(2 * W_R - 2) && (unsigned)(fy - 1) <= (unsigned)(2 * W_R - 2);
    if (use_pre) { jx0 = pre_x0; jy0 = pre_y0; jv[0] = pre_v[0]; jv[1] = pre_v[1]; }
    else w_fetch_J(jroi, LJ, jx0, jy0, lane, jv);
    w_sync();                                                     // the previous visit's LDS reads are done
    if (lane < W_IROWS * 4) Ib[lane] = iv;
    w_store_J(Jb, lane, jv);
    w_sync();

    // ---- Scharr derivative (calcSharrDeriv + copyMakeBorder(BORDER_CONSTANT 0)) on the 10 x 10 integer grid ----
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int q = lane + 64 * k;
        if (q < W_GRID * W_GRID) {
            const int r = q / W_GRID, c = q - r * W_GRID;
            const uint8_t *p0 = Ibb + r * 16 + ish + c;                // rows r, r+1, r+2 <-> image rows Y-1, Y, Y+1; bytes <-> columns X-1, X, X+1
            int t0[3], t1[3];
#pragma unroll
            for (int kk = 0; kk < 3; kk++) {
                const int u = p0[kk], m = p0[16 + kk], d = p0[32 + kk];
                t0[kk] = (u + d) * 3 + m * 10;
                t1[kk] = d - u;
            }
            int dx = t0[2] - t0[0], dy = (t1[0] + t1[2]) * 3 + t1[1] * 10;
            const int X = ipx + c, Y = ipy + r;
            if (!(X >= 0 && X < LI.w && Y >= 0 && Y < LI.h)) { dx = 0; dy = 0; }
            Dg[q] = ((uint32_t)dx & 0xFFFFu) | ((uint32_t)dy << 16);
        }
    }
    w_sync();

    // ---- template: I (5 fractional bits), dIx, dIy of this lane's pixels p = lane, lane + 64 ----
    int Iw[2], Ix[2], Iy[2];
    int s11 = 0, s12 = 0, s22 = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int p = lane + 64 * k;
        Iw[k] = Ix[k] = Iy[k] = 0;
        if (p < W_NPIX) {
            const int y = p / WIN, x = p - y * WIN;
            const uint8_t *ps = Ibb + (y + 1) * 16 + ish + 1 + x;
            Iw[k] = w_descale(w_m24(ps[0], iw00) + w_m24(ps[1], iw01) + w_m24(ps[16], iw10) + w_m24(ps[17], iw11), 14 - 5);
            const uint32_t d00 = Dg[y * W_GRID + x], d01 = Dg[y * W_GRID + x + 1], d10 = Dg[(y + 1) * W_GRID + x], d11 = Dg[(y + 1) * W_GRID + x + 1];
            const int x00 = (int)(short)(d00 & 0xFFFFu), x01 = (int)(short)(d01 & 0xFFFFu), x10 = (int)(short)(d10 & 0xFFFFu), x11 = (int)(short)(d11 & 0xFFFFu);
            const int y00 = (int)d00 >> 16, y01 = (int)d01 >> 16, y10 = (int)d10 >> 16, y11 = (int)d11 >> 16;
            Ix[k] = w_descale(w_m24(x00, iw00) + w_m24(x01, iw01) + w_m24(x10, iw10) + w_m24(x11, iw11), 14);
            Iy[k] = w_descale(w_m24(y00, iw00) + w_m24(y01, iw01) + w_m24(y10, iw10) + w_m24(y11, iw11), 14);
            s11 += w_m24(Ix[k], Ix[k]); s12 += w_m24(Ix[k], Iy[k]); s22 += w_m24(Iy[k], Iy[k]);
        }
    }
    // per-lane partials <= 2 * 4080^2, 16-lane rows < 2^31
    float A11, A12, A22;
    uint32_t *Gp = lds + W_LDS_INT;                                // [FACC] packed (dIx, dIy) per window pixel
    int *Dv = (int *)(Gp + W_FACC_PIX);                            // [FACC] the trip's I - J differences per window pixel
    if (!FACC) { A11 = (float)w_sum_exact(s11) * FLT_SCALE; A12 = (float)w_sum_exact(s12) * FLT_SCALE; A22 = (float)w_sum_exact(s22) * FLT_SCALE; }
    else {
#pragma unroll
        for (int k = 0; k < 2; k++) { const int p = lane + 64 * k; if (p < W_NPIX) Gp[p] = ((uint32_t)Ix[k] & 0xFFFFu) | ((uint32_t)Iy[k] << 16); }
        w_sync();
        // lanes 0..11: lane accumulator (lane & 3) of A11 / A12 / A22 (lane >> 2) over columns x = lane & 3 and x + 4; lanes 12..14: column 8
        const bool tail = lane >= 12;
        const int typ = tail ? lane - 12 : lane >> 2, kk = lane & 3;
        float acc = 0.f;
        if (lane < 15) {
#pragma unroll
            for (int y = 0; y < WIN; y++) {
                if (!tail) {
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) {                 // qA = qA + fx * fx, no FMA
                        const uint32_t g = Gp[y * WIN + kk + 4 * hh];
                        const float fx = (float)(int)(short)(g & 0xFFFFu), fy = (float)((int)g >> 16);
                        const float v = typ == 0 ? fx * fx : (typ == 1 ? fx * fy : fy * fy);
                        acc = acc + v;
                    }
                } else {                                            // iA11 += (itemtype)(ixval * ixval)
                    const uint32_t g = Gp[y * WIN + 8];
                    const int ixv = (int)(short)(g & 0xFFFFu), iyv = (int)g >> 16;
                    acc += (float)(typ == 0 ? ixv * ixv : (typ == 1 ? ixv * iyv : iyv * iyv));
                }
            }
        }
        // v_reduce_sum: (a0 + a2) + (a1 + a3), added to the scalar accumulator
        const float f11 = w_lane_f(acc, 12) + ((w_lane_f(acc, 0) + w_lane_f(acc, 2)) + (w_lane_f(acc, 1) + w_lane_f(acc, 3)));
        const float f12 = w_lane_f(acc, 13) + ((w_lane_f(acc, 4) + w_lane_f(acc, 6)) + (w_lane_f(acc, 5) + w_lane_f(acc, 7)));
        const float f22 = w_lane_f(acc, 14) + ((w_lane_f(acc, 8) + w_lane_f(acc, 10)) + (w_lane_f(acc, 9) + w_lane_f(acc, 11)));
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
    // request the NEXT level's search block now: its first window will sit near twice the current estimate (block margin +-3 px
    // there = +-1.5 px here); the loads complete while this level iterates
    if (LJnext != nullptr) {
        const float qx = (nextx + halfWin) * 2.f - halfWin, qy = (nexty + halfWin) * 2.f - halfWin;
        pre_x0 = w_floor(fminf(fmaxf(qx, (float)(-WIN)), (float)(LJnext->w - 1))) - W_R;
        pre_y0 = w_floor(fminf(fmaxf(qy, (float)(-WIN)), (float)(LJnext->h - 1))) - W_R;
        w_fetch_J(itemJ + LJnext->img_roi, *LJnext, pre_x0, pre_y0, lane, pre_v);
        pre_ok = true;
    }
    float pdx = 0.f, pdy = 0.f;
    for (int j = 0; j < prm.max_iter; j++) {
        const int inx = w_floor(nextx), iny = w_floor(nexty);
        if (inx < -WIN || inx >= LJ.w || iny < -WIN || iny >= LJ.h) {
            if (level == 0) st.status = 0;
            break;
        }
        st.iters++;
        a = nextx - (float)inx; b = nexty - (float)iny;
        iw00 = w_round((1.f - a) * (1.f - b) * W14);
        iw01 = w_round(a * (1.f - b) * W14);
        iw10 = w_round((1.f - a) * b * W14);
        iw11 = (1 << 14) - iw00 - iw01 - iw10;
        int ox = inx - jx0, oy = iny - jy0;
        if ((unsigned)ox > (unsigned)(2 * W_R) || (unsigned)oy > (unsigned)(2 * W_R)) {   // drifted: re-centre the block (wave-uniform)
            jx0 = inx - W_R; jy0 = iny - W_R;
            w_fetch_J(jroi, LJ, jx0, jy0, lane, jv);
            w_sync();
            w_store_J(Jb, lane, jv);
            w_sync();
            ox = W_R; oy = W_R;
        }
        const int jsh = jx0 & 3;
        int sb1 = 0, sb2 = 0;
        if (FACC) w_sync();                                         // the previous trip's reads of Dv are done
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int p = lane + 64 * k;
            if (p < W_NPIX) {
                const int y = p / WIN, x = p - y * WIN;
                const uint8_t *ps = Jbb + (oy + y) * 20 + jsh + ox + x;
                const int diff = w_descale(w_m24(ps[0], iw00) + w_m24(ps[1], iw01) + w_m24(ps[20], iw10) + w_m24(ps[21], iw11), 14 - 5) - Iw[k];
                if (FACC) Dv[p] = diff;
                sb1 += w_m24(diff, Ix[k]);
                sb2 += w_m24(diff, Iy[k]);
            }
        }
        float b1, b2;
        if (!FACC) {
            // |diff * dI| <= 8160 * 4080: per-lane partial < 6.7e7, 16-lane rows < 2^31
            b1 = (float)w_sum_exact(sb1) * FLT_SCALE;
            b2 = (float)w_sum_exact(sb2) * FLT_SCALE;
        } else {
            w_sync();
            // lanes 0..7 = (qb0[0..3], qb1[0..3]): pixel pair (lane >> 1, + 4), component lane & 1 (x, y); lanes 8, 9: column 8
            const int comp = lane & 1, pair = (lane >> 1) & 3;
            float acc = 0.f;
            if (lane < 10) {
#pragma unroll
                for (int y = 0; y < WIN; y++) {
                    if (lane < 8) {
                        const int p0 = y * WIN + pair;
                        const uint32_t g0 = Gp[p0], g4 = Gp[p0 + 4];
                        const int c0 = comp ? (int)g0 >> 16 : (int)(short)(g0 & 0xFFFFu), c4 = comp ? (int)g4 >> 16 : (int)(short)(g4 & 0xFFFFu);
                        acc = acc + (float)(Dv[p0] * c0 + Dv[p0 + 4] * c4);      // v_dotprod: the pair's products added exactly, then v_cvt_f32
                    } else {
                        const uint32_t g = Gp[y * WIN + 8];
                        const int c = comp ? (int)g >> 16 : (int)(short)(g & 0xFFFFu);
                        acc += (float)(Dv[y * WIN + 8] * c);                    // ib += (itemtype)(diff * dI)
                    }
                }
            }
            const float s0 = w_lane_f(acc, 0) + w_lane_f(acc, 4), s1 = w_lane_f(acc, 1) + w_lane_f(acc, 5);
            const float s2 = w_lane_f(acc, 2) + w_lane_f(acc, 6), s3 = w_lane_f(acc, 3) + w_lane_f(acc, 7);
            const float fb1 = w_lane_f(acc, 8) + (s0 + s2), fb2 = w_lane_f(acc, 9) + (s1 + s3);
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

// fbKltTracking for this wavefront's keypoint (feature_tracker.cpp:35-137): forward levels, filter, backward level 0, fb test
// pb / cb: the batch item of the previous / current pyramid this keypoint belongs to
template <bool FACC>
__device__ __forceinline__ int w_fb_track_point(const uint8_t *__restrict__ pb, const uint8_t *__restrict__ cb, const PyrDesc &P, const PyrDesc &C,
                                                const LKParams &prm, int max_level, float2 kp, float2 pr,
                                                int lane, uint32_t *lds, float &fx, float &fy, int &iters)
{
    WState st;
    st.nx = pr.x; st.ny = pr.y; st.status = 1; st.err = 0.f; st.iters = 0;
    bool pre_ok = false; int pre_x0 = 0, pre_y0 = 0; uint32_t pre_v[2] = {0u, 0u};
    for (int level = max_level; level >= 0; level--)
        w_level<FACC>(pb, P.lv[level], cb, C.lv[level], level > 0 ? &C.lv[level - 1] : nullptr, prm, level, max_level,
                (prm.flags & OV2_LK_USE_INITIAL_FLOW) != 0, kp.x, kp.y, lane, lds, st, pre_ok, pre_x0, pre_y0, pre_v);
    fx = st.nx; fy = st.ny;
    iters = st.iters;
    int ok = st.status;
    if (prm.do_fb) {
        if (ok && st.err > prm.err_th) ok = 0;                                              // :79-101
        const float W0 = (float)C.lv[0].w, H0 = (float)C.lv[0].h;
        if (ok && !(1.f <= fx && fx < W0 - 1.f && 1.f <= fy && fy < H0 - 1.f)) ok = 0;      // inBorder :216-221
        if (ok) {
            WState sb;                                                                     // backward: cur -> prev at level 0 from the keypoint (:113-116)
            sb.nx = kp.x; sb.ny = kp.y; sb.status = 1; sb.err = 0.f; sb.iters = 0;
            pre_ok = false;
            w_level<FACC>(cb, C.lv[0], pb, P.lv[0], nullptr, prm, 0, 0, true, fx, fy, lane, lds, sb, pre_ok, pre_x0, pre_y0, pre_v);
            iters += sb.iters;
            if (!sb.status) ok = 0;
            else {
                const float ddx = kp.x - sb.nx, ddy = kp.y - sb.ny;                        // cv::norm(Point2f) (:128)
                const double nrm = sqrt((double)ddx * (double)ddx + (double)ddy * (double)ddy);
                if (nrm > (double)prm.fb_dist) ok = 0;
            }
        }
    }
    return ok;
}

// VisualFrontEnd::kltTracking / the LK part of MapManager::stereoMatching in ONE launch: same contract as k_track_klt (lk.hip)
template <bool FACC>
__global__ __launch_bounds__(64) void k_track_klt_w(PyrDesc P, PyrDesc C, LKParams prm, int lvl_prior, int lvl_full,
                                                    const int *__restrict__ n_dev, const float2 *__restrict__ kps,
                                                    const float2 *__restrict__ priors, const uint8_t *__restrict__ flags,
                                                    float2 *__restrict__ out_xy, uint8_t *__restrict__ status,
                                                    int *__restrict__ iters_out, const float *__restrict__ sad_x, float sad_up)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[W_LDS_INT + (FACC ? 2 * W_FACC_PIX : 0)];
    // blockIdx.y: batch item (lock-step tracker, trackb.hip: n_max point slots and one count per item; 0 for one camera)
    const int item = blockIdx.y;
    const int n = n_dev ? n_dev[item] : prm.n_max;
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= n) return;
    const int i = item * prm.n_max + blockIdx.x;
    const uint8_t *pb = P.base + (long long)item * P.item_stride, *cb = C.base + (long long)item * C.item_stride;
    const float2 kp = kps[i];
    float2 pr = priors[i];
    const bool has_prior = (flags[i] & 1) != 0;
    const bool stereo = sad_x != nullptr;
    if (stereo && !has_prior) {                                    // map_manager.cpp:433-437
        pr = kp;
        const float xp = sad_x[i] * sad_up;
        if (xp >= 0.f && xp <= kp.x) pr.x = xp;
    }
    int max_level = has_prior ? lvl_prior : lvl_full;
    int ok = 0, retried = 0, iters = 0;
    float fx = 0.f, fy = 0.f;
    for (int attempt = 0; attempt < 2; attempt++) {
        int it = 0;
        ok = w_fb_track_point<FACC>(pb, cb, P, C, prm, max_level, kp, pr, lane, lds, fx, fy, it);
        iters += it;
        if (ok || !has_prior || attempt == 1) break;
        pr = make_float2(fx, fy);                                   // visual_front_end.cpp:213-217, map_manager.cpp:533-538
        max_level = lvl_full; retried = 1;
    }
    if (lane == 0) {
        out_xy[i] = make_float2(fx, fy);
        status[i] = (uint8_t)(ok | (retried << 1));
        if (iters_out) iters_out[i] = iters;
    }
}

int ov2_launch_track_klt_w(hipStream_t s, const PyrDesc &P, const PyrDesc &C, const LKParams &prm, int lp, int lf, int n_max, const int *n_dev,
                           const float *kps, const float *priors, const uint8_t *flags, float *out_xy, uint8_t *status, int *iters,
                           const float *sad_x, float sad_up, int items, int lk_acc)
{
    if (lk_acc == OV2_LK_ACC_FLOAT_UI4)
        hipLaunchKernelGGL(k_track_klt_w<true>, dim3(n_max, items), dim3(64), 0, s, P, C, prm, lp, lf, n_dev, (const float2 *)kps, (const float2 *)priors, flags,
                           (float2 *)out_xy, status, iters, sad_x, sad_up);
    else
        hipLaunchKernelGGL(k_track_klt_w<false>, dim3(n_max, items), dim3(64), 0, s, P, C, prm, lp, lf, n_dev, (const float2 *)kps, (const float2 *)priors, flags,
                           (float2 *)out_xy, status, iters, sad_x, sad_up);
    OV2_HIP_CHECK(hipGetLastError());
    return OV2_OK;
}
