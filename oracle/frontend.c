/*
 * frontend.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Image pyramid + pyramidal Lucas-Kanade + forward/backward KLT, restating
 *   - cv::buildOpticalFlowPyramid  (call sites src/visual_front_end.cpp:1172, :53,
 *                                   src/mapper.cpp:81)
 *   - cv::calcOpticalFlowPyrLK     (call sites src/feature_tracker.cpp:66-69, :113-116)
 *   - FeatureTracker::fbKltTracking (src/feature_tracker.cpp:35-137)
 *   - FeatureTracker::inBorder      (src/feature_tracker.cpp:216-221)
 * OpenCV itself is a third-party dependency that is NOT under /root/reference
 * (find_package(OpenCV), CMakeLists.txt:74-78, version unpinned), so its public
 * 3.4/4.x algorithm (modules/video/src/lkpyramid.cpp, modules/imgproc/src/
 * pyramids.cpp) is restated here: PARITY UNPINNED (no reference tests exist).
 *
 * Canonicalisation choices (documented in DESIGN.md):
 *   - LK accumulators follow OpenCV's `typedef int64 acctype; typedef int itemtype`
 *     variant (exact integer sums, one rounding to float).  The default build's
 *     float/SIMD accumulation order is compiler/ISA dependent; the int64 variant
 *     is order independent, which is what makes a parallel GPU reduction
 *     bit-reproducible.  Stock OpenCV builds use `float` accumulators instead; their
 *     summation orders (scalar raster order, the 3.4 SSE2 intrinsics, the 4.x
 *     universal-intrinsics code) are restated as orc_set_lk_acc_mode(ORC_LK_ACC_FLOAT_*)
 *     so that the distance between the canonical variant and what a real build
 *     executes can be MEASURED (tests/test_oracle_frontend.py, bench.py `parity`).
 *   - all float expressions are evaluated without FMA contraction
 *     (-ffp-contract=off) in source order.
 */
#include "ov2_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

/* ---- helpers -------------------------------------------------------- */
static inline int reflect101(int p, int len)
{
    /* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

static inline int cv_round_f(float v) { return (int)lrintf(v); } /* round-half-even */
static inline int cv_floor_f(float v) { return (int)floorf(v); }

#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

/* ---- pyrDown -------------------------------------------------------- */
/* cv::pyrDown for CV_8UC1, BORDER_REFLECT_101: separable [1 4 6 4 1],
 * dst = (sum + 128) >> 8  (imgproc/src/pyramids.cpp, PyrDownInvoker / FixPtCast<uchar,8>) */
typedef struct { const uint8_t *src; int sw, sh, sstride; uint8_t *dst; int dw, dh, dstride; } pyrdown_job;

static void pyr_down_rows(int y0, int y1, void *ctx)
{
    const pyrdown_job *jb = (const pyrdown_job *)ctx;
    const uint8_t *src = jb->src; uint8_t *dst = jb->dst;
    const int sw = jb->sw, sh = jb->sh, sstride = jb->sstride, dw = jb->dw, dstride = jb->dstride;
    int *rows = (int *)malloc(sizeof(int) * (size_t)dw * 5);
    /* interior columns need no border handling: 2x-2 >= 0 and 2x+2 <= sw-1 */
    int xi0 = 1, xi1 = (sw - 3) / 2;           /* inclusive range of interior x */
    if (xi1 > dw - 1) xi1 = dw - 1;
    for (int y = y0; y < y1; y++) {
        for (int k = 0; k < 5; k++) {
            int sy = reflect101(2 * y - 2 + k, sh);
            const uint8_t *s = src + (size_t)sy * sstride;
            int *row = rows + (size_t)k * dw;
            for (int x = 0; x < dw; x++) {
                if (x >= xi0 && x <= xi1) {
                    const uint8_t *q = s + 2 * x;
                    row[x] = q[0] * 6 + (q[-1] + q[1]) * 4 + q[-2] + q[2];
                    continue;
                }
                int x0 = reflect101(2 * x - 2, sw), x1 = reflect101(2 * x - 1, sw);
                int x2 = reflect101(2 * x, sw), x3 = reflect101(2 * x + 1, sw);
                int x4 = reflect101(2 * x + 2, sw);
                row[x] = s[x2] * 6 + (s[x1] + s[x3]) * 4 + s[x0] + s[x4];
            }
        }
        uint8_t *d = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; x++) {
            int v = rows[2 * dw + x] * 6 + (rows[dw + x] + rows[3 * dw + x]) * 4 + rows[x] + rows[4 * dw + x];
            d[x] = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows);
}

void orc_pyr_down_u8(const uint8_t *src, int sw, int sh, int sstride,
                     uint8_t *dst, int dw, int dh, int dstride)
{
    pyrdown_job jb = {src, sw, sh, sstride, dst, dw, dh, dstride};
    orc_parallel_for(dh, pyr_down_rows, &jb, 8);          /* rows in parallel, like cv::pyrDown's ParallelLoopBody */
}

/* ---- Scharr derivative (calcSharrDeriv in video/src/lkpyramid.cpp) ---- */
typedef struct { const uint8_t *src; int w, h, sstride; int16_t *dst; int dstride_elems; } scharr_job;

static void scharr_rows(int y0, int y1, void *ctx)
{
    const scharr_job *jb = (const scharr_job *)ctx;
    const uint8_t *src = jb->src; int16_t *dst = jb->dst;
    const int w = jb->w, h = jb->h, sstride = jb->sstride, dstride_elems = jb->dstride_elems;
    int *t0 = (int *)malloc(sizeof(int) * (size_t)(w + 2) * 2);
    int *t1 = t0 + (w + 2);
    for (int y = y0; y < y1; y++) {
        const uint8_t *r0 = src + (size_t)(y > 0 ? y - 1 : (h > 1 ? 1 : 0)) * sstride;
        const uint8_t *r1 = src + (size_t)y * sstride;
        const uint8_t *r2 = src + (size_t)(y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0)) * sstride;
        for (int x = 0; x < w; x++) {
            t0[x + 1] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
            t1[x + 1] = r2[x] - r0[x];
        }
        int xl = (w > 1 ? 1 : 0), xr = (w > 1 ? w - 2 : 0);
        t0[0] = t0[xl + 1]; t0[w + 1] = t0[xr + 1];
        t1[0] = t1[xl + 1]; t1[w + 1] = t1[xr + 1];
        int16_t *d = dst + (size_t)y * dstride_elems;
        for (int x = 0; x < w; x++) {
            d[2 * x]     = (int16_t)(t0[x + 2] - t0[x]);
            d[2 * x + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
    free(t0);
}

void orc_scharr_u8(const uint8_t *src, int w, int h, int sstride,
                   int16_t *dst, int dstride_elems)
{
    scharr_job jb = {src, w, h, sstride, dst, dstride_elems};
    orc_parallel_for(h, scharr_rows, &jb, 16);            /* calcSharrDeriv runs under cv::parallel_for_ too */
}

/* ---- buildOpticalFlowPyramid(img, pyr, win, maxLevel, withDerivatives=true,
 *      pyrBorder=REFLECT_101, derivBorder=CONSTANT) ------------------------ */
static void fill_border_reflect101(orc_level *L)
{
    int pad = L->pad, w = L->w, h = L->h;
    for (int y = -pad; y < h + pad; y++) {
        int sy = reflect101(y, h);
        uint8_t *drow = L->img + (size_t)(y + pad) * L->img_pitch + pad;
        const uint8_t *srow = L->img + (size_t)(sy + pad) * L->img_pitch + pad;
        for (int x = -pad; x < w + pad; x++) {
            if (y >= 0 && y < h && x >= 0 && x < w) continue;
            drow[x] = srow[reflect101(x, w)];
        }
    }
}

static int pyr_build_impl(const uint8_t *img, int w, int h, int stride, int win, int max_level, orc_pyr *out, int reuse)
{
    if (!img || !out || w <= 0 || h <= 0 || win <= 2 || max_level < 0 ||
        max_level >= ORC_MAX_LEVELS)
        return -1;
    if (!reuse) memset(out, 0, sizeof(*out));
    out->win = win;
    int lw = w, lh = h;
    for (int level = 0; level <= max_level; level++) {
        orc_level *L = &out->lv[level];
        if (!reuse) {
            L->w = lw; L->h = lh; L->pad = win;
            L->img_pitch = lw + 2 * win;
            L->der_pitch = (lw + 2 * win) * 2;
            L->img = (uint8_t *)calloc((size_t)(lh + 2 * win) * L->img_pitch, 1);
            L->der = (int16_t *)calloc((size_t)(lh + 2 * win) * L->der_pitch, sizeof(int16_t));
            if (!L->img || !L->der) { orc_pyr_free(out); return -2; }
        }
        uint8_t *roi = L->img + (size_t)win * L->img_pitch + win;
        if (level == 0) {
            for (int y = 0; y < lh; y++) memcpy(roi + (size_t)y * L->img_pitch, img + (size_t)y * stride, (size_t)lw);
        } else {
            orc_level *P = &out->lv[level - 1];
            orc_pyr_down_u8(P->img + (size_t)P->pad * P->img_pitch + P->pad, P->w, P->h, P->img_pitch,
                            roi, lw, lh, L->img_pitch);
        }
        fill_border_reflect101(L);
        orc_scharr_u8(roi, lw, lh, L->img_pitch,
                      L->der + (size_t)win * L->der_pitch + 2 * win, L->der_pitch);
        out->n_levels = level + 1;
        /* size of the next level; stop early like the reference */
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
        if (lw <= win || lh <= win) break;
    }
    return 0;
}

int orc_pyr_build(const uint8_t *img, int w, int h, int stride, int win,
                  int max_level, orc_pyr *out)
{
    return pyr_build_impl(img, w, h, stride, win, max_level, out, 0);
}

/* Rebuild into the buffers of a pyramid of the same geometry: cv::buildOpticalFlowPyramid re-uses the Mats of the
 * output vector when their sizes match (Mat::create is a no-op), which is the per-frame case of the reference
 * (cur_pyr_ / prev_pyr_ are swapped, visual_front_end.cpp:1169-1172).  The derivative border stays zero: only the
 * ROI is rewritten.                                                                                             */
int orc_pyr_rebuild(const uint8_t *img, int w, int h, int stride, orc_pyr *p)
{
    if (!p || p->n_levels <= 0 || p->lv[0].w != w || p->lv[0].h != h) return -1;
    return pyr_build_impl(img, w, h, stride, p->win, p->n_levels - 1, p, 1);
}

void orc_pyr_free(orc_pyr *p)
{
    if (!p) return;
    for (int i = 0; i < ORC_MAX_LEVELS; i++) {
        free(p->lv[i].img); free(p->lv[i].der);
        p->lv[i].img = NULL; p->lv[i].der = NULL;
    }
    p->n_levels = 0;
}

int orc_pyr_level_size(const orc_pyr *p, int level, int *w, int *h)
{
    if (!p || level < 0 || level >= p->n_levels) return -1;
    *w = p->lv[level].w; *h = p->lv[level].h;
    return 0;
}

int orc_pyr_copy_level(const orc_pyr *p, int level, uint8_t *img_out, int16_t *der_out)
{
    if (!p || level < 0 || level >= p->n_levels) return -1;
    const orc_level *L = &p->lv[level];
    for (int y = 0; y < L->h; y++) {
        if (img_out)
            memcpy(img_out + (size_t)y * L->w, L->img + (size_t)(y + L->pad) * L->img_pitch + L->pad, (size_t)L->w);
        if (der_out)
            memcpy(der_out + (size_t)y * L->w * 2, L->der + (size_t)(y + L->pad) * L->der_pitch + 2 * L->pad,
                   sizeof(int16_t) * 2 * (size_t)L->w);
    }
    return 0;
}

int orc_pyr_copy_level_padded(const orc_pyr *p, int level, uint8_t *img_out, int16_t *der_out)
{
    if (!p || level < 0 || level >= p->n_levels) return -1;
    const orc_level *L = &p->lv[level];
    int pw = L->w + 2 * L->pad, ph = L->h + 2 * L->pad;
    for (int y = 0; y < ph; y++) {
        if (img_out) memcpy(img_out + (size_t)y * pw, L->img + (size_t)y * L->img_pitch, (size_t)pw);
        if (der_out) memcpy(der_out + (size_t)y * pw * 2, L->der + (size_t)y * L->der_pitch, sizeof(int16_t) * 2 * (size_t)pw);
    }
    return 0;
}

/* ---- calcOpticalFlowPyrLK ------------------------------------------- */
/* Accumulator variant of LKTrackerInvoker (lkpyramid.cpp: `typedef float acctype; typedef float itemtype;` in stock
 * builds, `int64` / `int` behind the same typedefs): process-wide switch, read once per orc_lk_track call.            */
static int g_lk_acc_mode = ORC_LK_ACC_INT64;
void orc_set_lk_acc_mode(int mode) { if (mode >= ORC_LK_ACC_INT64 && mode <= ORC_LK_ACC_FLOAT_UI4) g_lk_acc_mode = mode; }
int orc_get_lk_acc_mode(void) { return g_lk_acc_mode; }

typedef struct {
    const orc_pyr *prev, *next;
    const float *prev_xy; float *next_xy;
    uint8_t *status; float *err;
    int win, max_level, max_count; double epsilon; int flags; double min_eig_threshold;
    int *iters_out;
    int level;
    int begin, end;
    int acc_mode;
} lk_job;

#define LK_MAX_WIN 31

/* LKTrackerInvoker::operator() for one level and a range of points
 * (video/src/lkpyramid.cpp).  W_BITS = 14, FLT_SCALE = 2^-20.             */
/* diagnostics (tools/lk_trip_hist.py): histogram of Gauss-Newton trips per level visit; off unless switched on */
static int orc_lk_trip_hist_on = 0;
static int orc_lk_trip_hist[64];
void orc_lk_trip_hist_enable(int on) { orc_lk_trip_hist_on = on; if (on) for (int k = 0; k < 64; k++) orc_lk_trip_hist[k] = 0; }
void orc_lk_trip_hist_get(int *out64) { for (int k = 0; k < 64; k++) out64[k] = orc_lk_trip_hist[k]; }
/* diagnostics (tools/lk_lockstep.py): per-keypoint, per-level trip counts of the NEXT orc_lk_track call: buf[i * 8 + level] */
static int *orc_lk_trip_log = 0;
void orc_lk_trip_log_set(int *buf) { orc_lk_trip_log = buf; }

static void lk_level_range(const lk_job *jb)
{
    const int win = jb->win, level = jb->level;
    const orc_level *I = &jb->prev->lv[level];
    const orc_level *J = &jb->next->lv[level];
    const float halfWin = (float)(win - 1) * 0.5f;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const int stepI = I->img_pitch, stepJ = J->img_pitch, dstep = I->der_pitch;
    const uint8_t *Ibase = I->img + (size_t)I->pad * stepI + I->pad;
    const uint8_t *Jbase = J->img + (size_t)J->pad * stepJ + J->pad;
    const int16_t *Dbase = I->der + (size_t)I->pad * dstep + 2 * I->pad;
    int16_t Iwin[LK_MAX_WIN * LK_MAX_WIN], dIx[LK_MAX_WIN * LK_MAX_WIN], dIy[LK_MAX_WIN * LK_MAX_WIN];

    for (int i = jb->begin; i < jb->end; i++) {
        const float lvl_scale = (float)(1. / (double)(1 << level));
        float prevx = jb->prev_xy[2 * i] * lvl_scale, prevy = jb->prev_xy[2 * i + 1] * lvl_scale;
        float nextx, nexty;
        if (level == jb->max_level) {
            if (jb->flags & ORC_LK_USE_INITIAL_FLOW) {
                nextx = jb->next_xy[2 * i] * lvl_scale; nexty = jb->next_xy[2 * i + 1] * lvl_scale;
            } else { nextx = prevx; nexty = prevy; }
        } else {
            nextx = jb->next_xy[2 * i] * 2.f; nexty = jb->next_xy[2 * i + 1] * 2.f;
        }
        jb->next_xy[2 * i] = nextx; jb->next_xy[2 * i + 1] = nexty;

        prevx -= halfWin; prevy -= halfWin;
        int ipx = cv_floor_f(prevx), ipy = cv_floor_f(prevy);
        if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
            if (level == 0) {
                if (jb->status) jb->status[i] = 0;
                if (jb->err) jb->err[i] = 0.f;
            }
            continue;
        }
        float a = prevx - (float)ipx, b = prevy - (float)ipy;
        int iw00 = cv_round_f((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
        int iw01 = cv_round_f(a * (1.f - b) * (float)(1 << W_BITS));
        int iw10 = cv_round_f((1.f - a) * b * (float)(1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

        int64_t iA11 = 0, iA12 = 0, iA22 = 0;
        /* float-accumulator variants (stock OpenCV).  The SIMD loops cover the first 4*(win/4) columns of a window row
         * four at a time: lane k of qA** sums the pixels x = k (mod 4) of those columns over all rows; the remaining
         * columns go to the scalar float accumulators fA**; the lanes are folded in once after the last row.          */
        const int accm = jb->acc_mode;
        const int simdA = accm >= ORC_LK_ACC_FLOAT_SSE34 ? 4 * (win / 4) : 0;
        float fA11 = 0.f, fA12 = 0.f, fA22 = 0.f, qA11[4] = {0.f, 0.f, 0.f, 0.f}, qA12[4] = {0.f, 0.f, 0.f, 0.f}, qA22[4] = {0.f, 0.f, 0.f, 0.f};
        for (int y = 0; y < win; y++) {
            const uint8_t *src = Ibase + (ptrdiff_t)(y + ipy) * stepI + ipx;
            const int16_t *dsrc = Dbase + (ptrdiff_t)(y + ipy) * dstep + 2 * ipx;
            for (int x = 0; x < win; x++, dsrc += 2) {
                int ival = DESCALE(src[x] * iw00 + src[x + 1] * iw01 + src[x + stepI] * iw10 + src[x + stepI + 1] * iw11, W_BITS - 5);
                int ixval = DESCALE(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[dstep] * iw10 + dsrc[dstep + 2] * iw11, W_BITS);
                int iyval = DESCALE(dsrc[1] * iw00 + dsrc[3] * iw01 + dsrc[dstep + 1] * iw10 + dsrc[dstep + 3] * iw11, W_BITS);
                Iwin[y * win + x] = (int16_t)ival;
                dIx[y * win + x] = (int16_t)ixval;
                dIy[y * win + x] = (int16_t)iyval;
                if (accm == ORC_LK_ACC_INT64) {
                    iA11 += (int)(ixval * ixval);
                    iA12 += (int)(ixval * iyval);
                    iA22 += (int)(iyval * iyval);
                } else if (x < simdA) {                       /* fx = cvt(t), qA = qA + fx * fx  (no FMA in SSE2 / baseline builds) */
                    const float fx = (float)ixval, fy = (float)iyval;
                    qA22[x & 3] = qA22[x & 3] + fy * fy;
                    qA12[x & 3] = qA12[x & 3] + fx * fy;
                    qA11[x & 3] = qA11[x & 3] + fx * fx;
                } else {                                      /* iA11 += (itemtype)(ixval*ixval) */
                    fA11 += (float)(ixval * ixval);
                    fA12 += (float)(ixval * iyval);
                    fA22 += (float)(iyval * iyval);
                }
            }
        }
        float A11, A12, A22;
        if (accm == ORC_LK_ACC_INT64) { A11 = (float)iA11 * FLT_SCALE; A12 = (float)iA12 * FLT_SCALE; A22 = (float)iA22 * FLT_SCALE; }
        else {
            if (accm == ORC_LK_ACC_FLOAT_SSE34) {             /* iA11 += A11buf[0] + A11buf[1] + A11buf[2] + A11buf[3] */
                fA11 += ((qA11[0] + qA11[1]) + qA11[2]) + qA11[3];
                fA12 += ((qA12[0] + qA12[1]) + qA12[2]) + qA12[3];
                fA22 += ((qA22[0] + qA22[1]) + qA22[2]) + qA22[3];
            } else if (accm == ORC_LK_ACC_FLOAT_UI4) {        /* iA11 += v_reduce_sum(qA11): (a0 + a2) + (a1 + a3) */
                fA11 += (qA11[0] + qA11[2]) + (qA11[1] + qA11[3]);
                fA12 += (qA12[0] + qA12[2]) + (qA12[1] + qA12[3]);
                fA22 += (qA22[0] + qA22[2]) + (qA22[1] + qA22[3]);
            }
            A11 = fA11 * FLT_SCALE; A12 = fA12 * FLT_SCALE; A22 = fA22 * FLT_SCALE;
        }
        float D = A11 * A22 - A12 * A12;
        float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
        if (jb->err && (jb->flags & ORC_LK_GET_MIN_EIGENVALS)) jb->err[i] = minEig;
        if (minEig < (float)jb->min_eig_threshold || D < FLT_EPSILON) { /* LKTrackerInvoker stores the threshold as float */
            if (level == 0 && jb->status) jb->status[i] = 0;
            continue;
        }
        D = 1.f / D;
        nextx -= halfWin; nexty -= halfWin;
        float pdx = 0.f, pdy = 0.f;
        int j, visit_trips = 0;
        for (j = 0; j < jb->max_count; j++) {
            int inx = cv_floor_f(nextx), iny = cv_floor_f(nexty);
            if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
                if (level == 0 && jb->status) jb->status[i] = 0;
                break;
            }
            if (jb->iters_out) jb->iters_out[i]++;
            visit_trips++;
            a = nextx - (float)inx; b = nexty - (float)iny;
            iw00 = cv_round_f((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
            iw01 = cv_round_f(a * (1.f - b) * (float)(1 << W_BITS));
            iw10 = cv_round_f((1.f - a) * b * (float)(1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int64_t ib1 = 0, ib2 = 0;
            /* float variants: the SIMD loop takes the first 8*(win/8) columns of a row eight at a time into qb0 / qb1 =
             * (x, y, x, y) lanes.  3.4 SSE2: every product It*Ix is converted to float on its own (mullo / mulhi ->
             * cvtepi32_ps: rounds above 2^24) -- pixels {0,4} -> qb0[0..1], {1,5} -> qb0[2..3], {2,6} -> qb1[0..1],
             * {3,7} -> qb1[2..3]; 4.x universal intrinsics: v_dotprod adds the products of pixels p and p+4 exactly in
             * int32 before v_cvt_f32.  Remaining columns: ib += (itemtype)(diff * dI) in float.                         */
            const int simdB = accm >= ORC_LK_ACC_FLOAT_SSE34 ? 8 * (win / 8) : 0;
            float fb1 = 0.f, fb2 = 0.f, qb0[4] = {0.f, 0.f, 0.f, 0.f}, qb1[4] = {0.f, 0.f, 0.f, 0.f};
            for (int y = 0; y < win; y++) {
                const uint8_t *Jp = Jbase + (ptrdiff_t)(y + iny) * stepJ + inx;
                int dgrp[8];
                for (int x = 0; x < win; x++) {
                    int diff = DESCALE(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 + Jp[x + stepJ + 1] * iw11, W_BITS - 5)
                               - Iwin[y * win + x];
                    if (accm == ORC_LK_ACC_INT64) {
                        ib1 += (int)(diff * dIx[y * win + x]);
                        ib2 += (int)(diff * dIy[y * win + x]);
                    } else if (x >= simdB) {
                        fb1 += (float)(diff * dIx[y * win + x]);
                        fb2 += (float)(diff * dIy[y * win + x]);
                    } else {
                        dgrp[x & 7] = diff;
                        if ((x & 7) == 7) {
                            const int16_t *gx = &dIx[y * win + x - 7], *gy = &dIy[y * win + x - 7];
                            if (accm == ORC_LK_ACC_FLOAT_SSE34) {
                                for (int h = 0; h < 2; h++) {          /* v00 * diff0 (pixels 0..3), then v01 * diff1 (4..7) */
                                    const int p = 4 * h;
                                    qb0[0] = qb0[0] + (float)(dgrp[p] * gx[p]);         qb0[1] = qb0[1] + (float)(dgrp[p] * gy[p]);
                                    qb0[2] = qb0[2] + (float)(dgrp[p + 1] * gx[p + 1]); qb0[3] = qb0[3] + (float)(dgrp[p + 1] * gy[p + 1]);
                                    qb1[0] = qb1[0] + (float)(dgrp[p + 2] * gx[p + 2]); qb1[1] = qb1[1] + (float)(dgrp[p + 2] * gy[p + 2]);
                                    qb1[2] = qb1[2] + (float)(dgrp[p + 3] * gx[p + 3]); qb1[3] = qb1[3] + (float)(dgrp[p + 3] * gy[p + 3]);
                                }
                            } else {
                                qb0[0] = qb0[0] + (float)(dgrp[0] * gx[0] + dgrp[4] * gx[4]); qb0[1] = qb0[1] + (float)(dgrp[0] * gy[0] + dgrp[4] * gy[4]);
                                qb0[2] = qb0[2] + (float)(dgrp[1] * gx[1] + dgrp[5] * gx[5]); qb0[3] = qb0[3] + (float)(dgrp[1] * gy[1] + dgrp[5] * gy[5]);
                                qb1[0] = qb1[0] + (float)(dgrp[2] * gx[2] + dgrp[6] * gx[6]); qb1[1] = qb1[1] + (float)(dgrp[2] * gy[2] + dgrp[6] * gy[6]);
                                qb1[2] = qb1[2] + (float)(dgrp[3] * gx[3] + dgrp[7] * gx[7]); qb1[3] = qb1[3] + (float)(dgrp[3] * gy[3] + dgrp[7] * gy[7]);
                            }
                        }
                    }
                }
            }
            float b1, b2;
            if (accm == ORC_LK_ACC_INT64) { b1 = (float)ib1 * FLT_SCALE; b2 = (float)ib2 * FLT_SCALE; }
            else {
                if (simdB) {       /* 3.4: bbuf = qb0 + qb1; ib1 += bbuf[0] + bbuf[2].  4.x: the same sums after interleave / recombine / v_reduce_sum with zeros */
                    const float s0 = qb0[0] + qb1[0], s1 = qb0[1] + qb1[1], s2 = qb0[2] + qb1[2], s3 = qb0[3] + qb1[3];
                    fb1 += s0 + s2; fb2 += s1 + s3;
                }
                b1 = fb1 * FLT_SCALE; b2 = fb2 * FLT_SCALE;
            }
            float dx = (A12 * b2 - A22 * b1) * D;
            float dy = (A12 * b1 - A11 * b2) * D;
            nextx += dx; nexty += dy;
            jb->next_xy[2 * i] = nextx + halfWin; jb->next_xy[2 * i + 1] = nexty + halfWin;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= jb->epsilon) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                jb->next_xy[2 * i] -= dx * 0.5f; jb->next_xy[2 * i + 1] -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        /* (diagnostics, tools/lk_trip_hist.py) Gauss-Newton trips of this level visit: j + 1 when the loop left through a
         * break after computing a step, j when it ran out of iterations or left the image before computing one */
        if (orc_lk_trip_hist_on) {
            __atomic_fetch_add(&orc_lk_trip_hist[visit_trips < 63 ? visit_trips : 63], 1, __ATOMIC_RELAXED);
        }
        if (orc_lk_trip_log && level < 8) orc_lk_trip_log[i * 8 + level] = visit_trips;
    }
}

static void *lk_thread(void *arg);
static void lk_points(int b, int e, void *ctx)
{
    lk_job jb = *(const lk_job *)ctx;
    jb.begin = b; jb.end = e;
    lk_thread(&jb);
}

static void *lk_thread(void *arg)
{
    /* points are independent across levels, so a thread walks all levels for its own range */
    lk_job jb = *(const lk_job *)arg;
    for (int level = jb.max_level; level >= 0; level--) { jb.level = level; lk_level_range(&jb); }
    return NULL;
}

int orc_lk_track(const orc_pyr *prev, const orc_pyr *next,
                 const float *prev_xy, float *next_xy, int n,
                 uint8_t *status, float *err,
                 int win, int max_level, int max_count, double epsilon,
                 int flags, double min_eig_threshold,
                 int *iters_out, int nthreads)
{
    if (!prev || !next || n < 0 || win > LK_MAX_WIN || win != prev->win || win != next->win) return -1;
    if (prev->n_levels != next->n_levels) return -1;
    if (n == 0) return 0;
    /* levels available / criteria clamps (lkpyramid.cpp SparsePyrLKOpticalFlowImpl::calc) */
    if (max_level > prev->n_levels - 1) max_level = prev->n_levels - 1;
    if (max_count < 0) max_count = 0;
    if (max_count > 100) max_count = 100;
    if (epsilon < 0.) epsilon = 0.;
    if (epsilon > 10.) epsilon = 10.;
    epsilon *= epsilon;
    for (int i = 0; i < n; i++) { status[i] = 1; if (err) err[i] = 0.f; if (iters_out) iters_out[i] = 0; }
    lk_job jb;
    jb.prev = prev; jb.next = next; jb.prev_xy = prev_xy; jb.next_xy = next_xy;
    jb.status = status; jb.err = err; jb.win = win; jb.max_level = max_level;
    jb.max_count = max_count; jb.epsilon = epsilon; jb.flags = flags;
    jb.min_eig_threshold = min_eig_threshold; jb.iters_out = iters_out;
    jb.level = max_level; jb.begin = 0; jb.end = n; jb.acc_mode = g_lk_acc_mode;
    if (nthreads <= 1) lk_thread(&jb);
    else {
        /* points over the persistent pool (cv::parallel_for_ in LKTrackerInvoker); the pool is grown on demand */
        if (orc_get_num_threads() < nthreads) orc_set_num_threads(nthreads);
        orc_parallel_for(n, lk_points, &jb, 4);
    }
    return 0;
}

/* ---- FeatureTracker::fbKltTracking (src/feature_tracker.cpp:35-137) ---- */
int orc_fb_klt(const orc_pyr *prevpyr, const orc_pyr *curpyr,
               int win, int nbpyrlvl, int max_iter, float eps_px,
               float ferr, float fmax_fbklt_dist,
               const float *kps_xy, float *prior_xy, int n,
               uint8_t *status_out, long long *iters_total, long long *lvl_visits,
               int nthreads)
{
    if (iters_total) *iters_total = 0;
    if (lvl_visits) *lvl_visits = 0;
    if (n <= 0) return 0;                                   /* :43-46 */
    if (prevpyr->n_levels != curpyr->n_levels) return -1;    /* assert :41 */
    /* :50-52: vector has 2 Mats per level */
    if (2 * prevpyr->n_levels < 2 * (nbpyrlvl + 1)) nbpyrlvl = (2 * prevpyr->n_levels) / 2 - 1;

    uint8_t *st = (uint8_t *)malloc((size_t)n);
    float *er = (float *)malloc(sizeof(float) * (size_t)n);
    int *its = (int *)malloc(sizeof(int) * (size_t)n);
    int *idx = (int *)malloc(sizeof(int) * (size_t)n);
    float *newk = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    float *backk = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    const int flags = ORC_LK_USE_INITIAL_FLOW + ORC_LK_GET_MIN_EIGENVALS;
    /* TermCriteria(COUNT+EPS, nmax_iter, fmax_px_precision): epsilon is a double
     * holding the float value (include/feature_tracker.hpp:39-40) */
    const double epsilon = (double)eps_px;

    orc_lk_track(prevpyr, curpyr, kps_xy, prior_xy, n, st, er, win, nbpyrlvl, max_iter, epsilon,
                 flags, 1e-4, its, nthreads);                                   /* :66-69 */
    int eff_lvls = (nbpyrlvl > prevpyr->n_levels - 1 ? prevpyr->n_levels - 1 : nbpyrlvl) + 1;
    if (lvl_visits) *lvl_visits += (long long)n * eff_lvls;
    if (iters_total) for (int i = 0; i < n; i++) *iters_total += its[i];

    const int W0 = curpyr->lv[0].w, H0 = curpyr->lv[0].h;
    int m = 0;
    for (int i = 0; i < n; i++) {                                               /* :79-101 */
        if (!st[i]) { status_out[i] = 0; continue; }
        if (er[i] > ferr) { status_out[i] = 0; continue; }
        float x = prior_xy[2 * i], y = prior_xy[2 * i + 1];
        /* inBorder :216-221, BORDER_SIZE = 1 */
        if (!(1.f <= x && x < (float)W0 - 1.f && 1.f <= y && y < (float)H0 - 1.f)) { status_out[i] = 0; continue; }
        newk[2 * m] = x; newk[2 * m + 1] = y;
        backk[2 * m] = kps_xy[2 * i]; backk[2 * m + 1] = kps_xy[2 * i + 1];
        status_out[i] = 1; idx[m] = i; m++;
    }
    if (m > 0) {
        orc_lk_track(curpyr, prevpyr, newk, backk, m, st, er, win, 0, max_iter, epsilon,
                     flags, 1e-4, its, nthreads);                               /* :113-116 */
        if (lvl_visits) *lvl_visits += m;
        if (iters_total) for (int i = 0; i < m; i++) *iters_total += its[i];
        for (int k = 0; k < m; k++) {                                           /* :119-134 */
            int i = idx[k];
            if (!st[k]) { status_out[i] = 0; continue; }
            /* cv::norm(Point2f) = sqrt((double)x*x + (double)y*y) */
            float ddx = kps_xy[2 * i] - backk[2 * k], ddy = kps_xy[2 * i + 1] - backk[2 * k + 1];
            double nrm = sqrt((double)ddx * ddx + (double)ddy * ddy);
            if (nrm > (double)fmax_fbklt_dist) status_out[i] = 0;
        }
    }
    free(st); free(er); free(its); free(idx); free(newk); free(backk);
    return 0;
}
