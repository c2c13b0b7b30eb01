/*
 * detect.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Keypoint grid detection, restating
 *   - FeatureExtractor::detectGridFAST    (src/feature_extractor.cpp:443-570)
 *   - FeatureExtractor::detectSingleScale (src/feature_extractor.cpp:288-440)
 * and the OpenCV operators they call (OpenCV is NOT under /root/reference; public
 * 3.4/4.x algorithms restated -- PARITY UNPINNED, see ov2_oracle.h):
 *   cv::FAST(TYPE_9_16, nonmax)  features2d/src/fast.cpp, fast_score.cpp
 *   KeyPointsFilter::runByPixelsMask  features2d/src/keypoint.cpp
 *   cv::circle(FILLED)           imgproc/src/drawing.cpp (Circle)
 *   cv::GaussianBlur 3x3 (8-bit fixed point), cv::cornerMinEigenVal(3,3)
 *                                imgproc/src/smooth*.cpp, filter.simd.hpp, corner.cpp, box_filter.simd.hpp
 *   cv::minMaxLoc                core/src/minmax.cpp (first maximum, row-major)
 *   cv::cornerSubPix             imgproc/src/cornersubpix.cpp, samplers.cpp (getRectSubPix)
 *
 * Canonicalisation: the reference runs the per-cell lambdas under
 * cv::parallel_for_ while mutating the shared mask; the oracle is the serial raster
 * order i = 0..nbcells-1.  Ties in FAST response are broken by raster order (what
 * std::sort's insertion sort gives for <= 16 keypoints per cell).
 */
#include "ov2_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ---- FAST-9/16 --------------------------------------------------------- */
static const int fast_off[16][2] = {
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* cornerScore<16> (fast_score.cpp) */
static int fast_corner_score(const uint8_t *ptr, const int pixel[25], int threshold)
{
    const int N = 25;
    int d[25], v = ptr[0];
    for (int k = 0; k < N; k++) d[k] = v - ptr[pixel[k]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = imin(d[k + 1], d[k + 2]);
        a = imin(a, d[k + 3]);
        if (a <= a0) continue;
        a = imin(a, d[k + 4]); a = imin(a, d[k + 5]); a = imin(a, d[k + 6]);
        a = imin(a, d[k + 7]); a = imin(a, d[k + 8]);
        a0 = imax(a0, imin(a, d[k]));
        a0 = imax(a0, imin(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = imax(d[k + 1], d[k + 2]);
        b = imax(b, d[k + 3]); b = imax(b, d[k + 4]); b = imax(b, d[k + 5]);
        if (b >= b0) continue;
        b = imax(b, d[k + 6]); b = imax(b, d[k + 7]); b = imax(b, d[k + 8]);
        b0 = imin(b0, imax(b, d[k]));
        b0 = imin(b0, imax(b, d[k + 9]));
    }
    return -b0 - 1;
}

int orc_fast9_16(const uint8_t *img, int w, int h, int stride, int threshold,
                 int nonmax, int *xs, int *ys, int *scores, int cap)
{
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = fast_off[k][0] + fast_off[k][1] * stride;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    threshold = imin(imax(threshold, 0), 255);
    if (w < 7 || h < 7) return 0;
    uint8_t *score = (uint8_t *)calloc((size_t)w * h, 1);   /* 0 = not a corner */
    uint8_t *is_corner = (uint8_t *)calloc((size_t)w * h, 1);
    for (int y = 3; y < h - 3; y++) {
        for (int x = 3; x < w - 3; x++) {
            const uint8_t *ptr = img + (size_t)y * stride + x;
            int v = ptr[0], corner = 0;
            /* darker arc: >= 9 contiguous circle pixels < v - t */
            int vt = v - threshold, count = 0;
            for (int k = 0; k < N; k++) {
                if (ptr[pixel[k]] < vt) { if (++count > K) { corner = 1; break; } }
                else count = 0;
            }
            if (!corner) {
                vt = v + threshold; count = 0;
                for (int k = 0; k < N; k++) {
                    if (ptr[pixel[k]] > vt) { if (++count > K) { corner = 1; break; } }
                    else count = 0;
                }
            }
            if (corner) {
                is_corner[(size_t)y * w + x] = 1;
                score[(size_t)y * w + x] = (uint8_t)fast_corner_score(ptr, pixel, threshold);
            }
        }
    }
    int n = 0;
    for (int y = 3; y < h - 3; y++) {
        for (int x = 3; x < w - 3; x++) {
            if (!is_corner[(size_t)y * w + x]) continue;
            int s = score[(size_t)y * w + x];
            if (nonmax) {
                const uint8_t *c = score + (size_t)y * w + x;
                if (!(s > c[1] && s > c[-1] && s > c[-w - 1] && s > c[-w] && s > c[-w + 1] &&
                      s > c[w - 1] && s > c[w] && s > c[w + 1]))
                    continue;
            }
            if (n < cap) { xs[n] = x; ys[n] = y; scores[n] = s; }
            n++;
        }
    }
    free(score); free(is_corner);
    return n < cap ? n : cap;
}

/* ---- cv::circle(mask, c, r, 0, FILLED) --------------------------------- */
static void hline0(uint8_t *mask, int w, int h, int y, int x0, int x1)
{
    if ((unsigned)y >= (unsigned)h) return;
    if (x0 < 0) x0 = 0;
    if (x1 > w - 1) x1 = w - 1;
    for (int x = x0; x <= x1; x++) mask[(size_t)y * w + x] = 0;
}

void orc_circle_fill0(uint8_t *mask, int w, int h, int cx, int cy, int radius)
{
    /* drawing.cpp Circle(): midpoint circle, fill variant; the `inside` fast path and the
     * clipped path draw the same pixel set, so only the clipped one is restated. */
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        int y11 = cy - dy, y12 = cy + dy, y21 = cy - dx, y22 = cy + dx;
        int x11 = cx - dx, x12 = cx + dx, x21 = cx - dy, x22 = cx + dy;
        if (x11 < w && x12 >= 0 && y21 < h && y22 >= 0) {
            hline0(mask, w, h, y11, x11, x12);
            hline0(mask, w, h, y12, x11, x12);
            if (x21 < w && x22 >= 0) {
                hline0(mask, w, h, y21, x21, x22);
                hline0(mask, w, h, y22, x21, x22);
            }
        }
        dy++;
        err += plus;
        plus += 2;
        int m = (err <= 0) - 1;
        err -= minus & m;
        dx += m;
        minus -= m & 2;
    }
}

static inline int cv_round_f(float v) { return (int)lrintf(v); }

/* ---- version-dependent arithmetic of the detector's OpenCV calls (tools/detect_variant_campaign.py) ----------------
 * The canonical choices (what the HIP kernels implement bit for bit) next to the variants a different OpenCV build would run; the
 * switches exist to MEASURE how far apart they are on the keypoint sets, like orc_set_lk_acc_mode does for LK -- nothing is pinned
 * against a real build (none in this image).
 *   blur:   GaussianBlur 3x3 on 8U.  FIXED (canonical): the fixed-point paths (ufixedpoint16 in 3.4.2+ / 4.x, FixedPtCastEx before)
 *           round the exact sum s / 16 half UP, (s + 8) >> 4.  HALF_EVEN: a float-kernel build, cvRound(s / 16.f): ties to even.
 *   subpix: getRectSubPix 8U -> 32F inside cornerSubPix.  FAST (canonical): the two-tap running form of the optimised path;
 *           GENERIC: the four-tap form of getRectSubPix_Cn_ for every patch (builds without that path);
 *           FLOAT_ACC: the five gradient sums accumulated in float instead of double (2.4-era cvFindCornerSubPix).          */
static int g_blur_mode = ORC_BLUR_FIXED, g_subpix_mode = ORC_SUBPIX_FAST;
void orc_set_blur_mode(int m) { g_blur_mode = m == ORC_BLUR_HALF_EVEN ? ORC_BLUR_HALF_EVEN : ORC_BLUR_FIXED; }
int orc_get_blur_mode(void) { return g_blur_mode; }
void orc_set_subpix_mode(int m) { g_subpix_mode = (m >= ORC_SUBPIX_FAST && m <= ORC_SUBPIX_FLOAT_ACC) ? m : ORC_SUBPIX_FAST; }
int orc_get_subpix_mode(void) { return g_subpix_mode; }

/* ---- cornerSubPix ------------------------------------------------------ */
/* getRectSubPix(src u8 -> f32 patch pw x ph, center) (samplers.cpp) */
static void get_rect_subpix_8u32f(const uint8_t *src, int src_step, int sw, int sh,
                                  float *dst, int pw, int ph, float cx_f, float cy_f)
{
    /* getRectSubPix_8u32f fast path uses double centre arithmetic */
    double cxd = (double)cx_f - (pw - 1) * 0.5, cyd = (double)cy_f - (ph - 1) * 0.5;
    int ipx = (int)floor(cxd), ipy = (int)floor(cyd);
    if (g_subpix_mode != ORC_SUBPIX_GENERIC && 0 <= ipx && ipx + pw < sw && 0 <= ipy && ipy + ph < sh && pw > 0 && ph > 0) {
        float a = (float)(cxd - ipx), b = (float)(cyd - ipy);
        a = a > 0.0001f ? a : 0.0001f;
        float a12 = a * (1.f - b), a22 = a * b, b1 = 1.f - b, b2 = b;
        double s = (1. - (double)a) / (double)a;
        const uint8_t *p = src + (size_t)ipy * src_step + ipx;
        for (int i = 0; i < ph; i++, p += src_step, dst += pw) {
            float prev = (1 - a) * (b1 * p[0] + b2 * p[src_step]);
            for (int j = 0; j < pw; j++) {
                float t = a12 * p[j + 1] + a22 * p[j + 1 + src_step];
                dst[j] = prev + t;
                prev = (float)(t * s);
            }
        }
        return;
    }
    /* generic path getRectSubPix_Cn_<uchar,float,float>: float centre arithmetic, replicated border */
    float cx = cx_f - (pw - 1) * 0.5f, cy = cy_f - (ph - 1) * 0.5f;
    ipx = (int)floorf(cx); ipy = (int)floorf(cy);
    float a = cx - ipx, b = cy - ipy;
    float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
    float b1 = 1.f - b, b2 = b;
    if (0 <= ipx && ipx < sw - pw && 0 <= ipy && ipy < sh - ph) {
        const uint8_t *p = src + (size_t)ipy * src_step + ipx;
        for (int i = 0; i < ph; i++, p += src_step, dst += pw)
            for (int j = 0; j < pw; j++)
                dst[j] = p[j] * a11 + p[j + 1] * a12 + p[j + src_step] * a21 + p[j + src_step + 1] * a22;
        return;
    }
    /* adjustRect */
    int rx, ry, rw, rh;
    const uint8_t *p = src;
    if (ipx >= 0) { p += ipx; rx = 0; }
    else { rx = -ipx; if (rx > pw) rx = pw; }
    if (ipx < sw - pw) rw = pw;
    else { rw = sw - ipx - 1; if (rw < 0) { p += rw; rw = 0; } }
    if (ipy >= 0) { p += (ptrdiff_t)ipy * src_step; ry = 0; }
    else ry = -ipy;
    if (ipy < sh - ph) rh = ph;
    else { rh = sh - ipy - 1; if (rh < 0) { p += (ptrdiff_t)rh * src_step; rh = 0; } }
    p -= rx;
    for (int i = 0; i < ph; i++, dst += pw) {
        const uint8_t *p2 = p + src_step;
        if (i < ry || i >= rh) p2 -= src_step;
        float s0 = p[rx] * b1 + p2[rx] * b2;
        for (int j = 0; j < rx; j++) dst[j] = s0;
        for (int j = rx; j < rw; j++)
            dst[j] = p[j] * a11 + p[j + 1] * a12 + p2[j] * a21 + p2[j + 1] * a22;
        s0 = p[rw] * b1 + p2[rw] * b2;
        for (int j = rw; j < pw; j++) dst[j] = s0;
        if (i < rh) p = p2;
    }
}

void orc_corner_subpix(const uint8_t *img, int w, int h, int stride,
                       float *xy, int n, int half_win, int max_iter_in, double eps)
{
    const int win_w = half_win * 2 + 1, win_h = win_w;
    int max_iters = max_iter_in < 1 ? 1 : (max_iter_in > 100 ? 100 : max_iter_in);
    if (eps < 0.) eps = 0.;
    eps *= eps;
    float *mask = (float *)malloc(sizeof(float) * win_w * win_h);
    float *sub = (float *)malloc(sizeof(float) * (win_w + 2) * (win_h + 2));
    for (int i = 0; i < win_h; i++) {
        float y = (float)(i - half_win) / half_win;
        float vy = expf(-y * y);
        for (int j = 0; j < win_w; j++) {
            float x = (float)(j - half_win) / half_win;
            mask[i * win_w + j] = (float)(vy * expf(-x * x));
        }
    }
    const int sw = win_w + 2;
    for (int pt = 0; pt < n; pt++) {
        float cTx = xy[2 * pt], cTy = xy[2 * pt + 1], cIx = cTx, cIy = cTy;
        int iter = 0;
        double err = 0;
        do {
            double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
            get_rect_subpix_8u32f(img, stride, w, h, sub, win_w + 2, win_h + 2, cIx, cIy);
            const float *sp = sub + sw + 1;
            if (g_subpix_mode == ORC_SUBPIX_FLOAT_ACC) {
                float fa = 0, fb = 0, fc = 0, fb1 = 0, fb2 = 0;
                for (int i = 0, k = 0; i < win_h; i++, sp += sw) {
                    float py = (float)(i - half_win);
                    for (int j = 0; j < win_w; j++, k++) {
                        float m = mask[k], tgx = sp[j + 1] - sp[j - 1], tgy = sp[j + sw] - sp[j - sw];
                        float gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m, px = (float)(j - half_win);
                        fa += gxx; fb += gxy; fc += gyy;
                        fb1 += gxx * px + gxy * py;
                        fb2 += gxy * px + gyy * py;
                    }
                }
                a = fa; b = fb; c = fc; bb1 = fb1; bb2 = fb2;
            } else
            for (int i = 0, k = 0; i < win_h; i++, sp += sw) {
                double py = i - half_win;
                for (int j = 0; j < win_w; j++, k++) {
                    double m = mask[k];
                    double tgx = sp[j + 1] - sp[j - 1];
                    double tgy = sp[j + sw] - sp[j - sw];
                    double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                    double px = j - half_win;
                    a += gxx; b += gxy; c += gyy;
                    bb1 += gxx * px + gxy * py;
                    bb2 += gxy * px + gyy * py;
                }
            }
            double det = a * c - b * b;
            if (fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
            double scale = 1.0 / det;
            float c2x = (float)(cIx + c * scale * bb1 - b * scale * bb2);
            float c2y = (float)(cIy - b * scale * bb1 + a * scale * bb2);
            err = (double)((c2x - cIx) * (c2x - cIx) + (c2y - cIy) * (c2y - cIy));
            cIx = c2x; cIy = c2y;
            if (cIx < 0 || cIx >= w || cIy < 0 || cIy >= h) break;
        } while (++iter < max_iters && err > eps);
        if (fabs((double)(cIx - cTx)) > half_win || fabs((double)(cIy - cTy)) > half_win) { cIx = cTx; cIy = cTy; }
        xy[2 * pt] = cIx; xy[2 * pt + 1] = cIy;
    }
    free(mask); free(sub);
}

/* ---- shared grid prologue (feature_extractor.cpp:296-319 / :451-474) ----- */
typedef struct {
    int nhcells, nwcells, nbcells, nhalfcell;
    uint8_t *occ;      /* (nhcells+1) x (nwcells+1) */
    uint8_t *mask;     /* h x w, 1 = free (stands for the CV_32F ones mask) */
} grid_state;

static void grid_init(grid_state *g, int w, int h, int cell, const float *cur_xy, int ncur)
{
    g->nhalfcell = cell / 4;
    g->nhcells = h / cell; g->nwcells = w / cell;
    g->nbcells = g->nhcells * g->nwcells;
    g->occ = (uint8_t *)calloc((size_t)(g->nhcells + 1) * (g->nwcells + 1), 1);
    g->mask = (uint8_t *)malloc((size_t)w * h);
    memset(g->mask, 1, (size_t)w * h);
    for (int i = 0; i < ncur; i++) {
        float px = cur_xy[2 * i], py = cur_xy[2 * i + 1];
        /* voccupcells[px.y / ncellsize][px.x / ncellsize]: float division, truncation to size_t */
        int r = (int)(py / (float)cell), c = (int)(px / (float)cell);
        if (r >= 0 && r <= g->nhcells && c >= 0 && c <= g->nwcells) g->occ[r * (g->nwcells + 1) + c] = 1;
        /* cv::circle takes cv::Point: Point2f -> Point rounds (saturate_cast<int> = cvRound) */
        orc_circle_fill0(g->mask, w, h, cv_round_f(px), cv_round_f(py), g->nhalfcell);
    }
}

static void grid_free(grid_state *g) { free(g->occ); free(g->mask); }

/* ---- std::sort(vkps.begin(), vkps.end(), compare_response) as libstdc++ runs it (bits/stl_algo.h: __introsort_loop with the median-of-three
 * pivot moved to the front, __unguarded_partition, __final_insertion_sort with its 16-element threshold) on (x, y, score) records, comparator
 * a.score > b.score.  Only the order of EQUAL scores depends on these details -- which is the point (ORC_FAST_TIE_LIBSTDCXX). */
typedef struct { int x, y, s; } fkp;
static int g_fast_tie_mode = ORC_FAST_TIE_LIBSTDCXX, g_fast_tie_fallbacks = 0;
void orc_set_fast_tie_mode(int mode) { g_fast_tie_mode = mode == ORC_FAST_TIE_LIBSTDCXX ? ORC_FAST_TIE_LIBSTDCXX : ORC_FAST_TIE_SCAN_ORDER; }
int orc_get_fast_tie_mode(void) { return g_fast_tie_mode; }
int orc_fast_tie_sort_fallbacks(void) { return g_fast_tie_fallbacks; }
#define FK_LESS(a, b) ((a).s > (b).s)                      /* compare_response(first, second) = first.response > second.response */
static void fk_swap(fkp *a, fkp *b) { fkp t = *a; *a = *b; *b = t; }
static void fk_unguarded_linear_insert(fkp *last)
{
    fkp val = *last, *next = last - 1;
    while (FK_LESS(val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}
static void fk_insertion_sort(fkp *first, fkp *last)
{
    if (first == last) return;
    for (fkp *i = first + 1; i != last; ++i) {
        if (FK_LESS(*i, *first)) { fkp val = *i; memmove(first + 1, first, (size_t)(i - first) * sizeof(fkp)); *first = val; }
        else fk_unguarded_linear_insert(i);
    }
}
static void fk_move_median_to_first(fkp *result, fkp *a, fkp *b, fkp *c)
{
    if (FK_LESS(*a, *b)) {
        if (FK_LESS(*b, *c)) fk_swap(result, b);
        else if (FK_LESS(*a, *c)) fk_swap(result, c);
        else fk_swap(result, a);
    } else if (FK_LESS(*a, *c)) fk_swap(result, a);
    else if (FK_LESS(*b, *c)) fk_swap(result, c);
    else fk_swap(result, b);
}
static fkp *fk_unguarded_partition(fkp *first, fkp *last, fkp *pivot)
{
    for (;;) {
        while (FK_LESS(*first, *pivot)) ++first;
        --last;
        while (FK_LESS(*pivot, *last)) --last;
        if (!(first < last)) return first;
        fk_swap(first, last);
        ++first;
    }
}
static void fk_introsort_loop(fkp *first, fkp *last, int depth_limit)
{
    while (last - first > 16) {
        if (depth_limit == 0) { g_fast_tie_fallbacks++; fk_insertion_sort(first, last); return; }      /* (libstdc++: heap sort; never reached on cells) */
        --depth_limit;
        fkp *mid = first + (last - first) / 2;
        fk_move_median_to_first(first, first + 1, mid, last - 1);
        fkp *cut = fk_unguarded_partition(first + 1, last, first);
        fk_introsort_loop(cut, last, depth_limit);
        last = cut;
    }
}
static void fk_std_sort(fkp *first, int n)
{
    if (n <= 0) return;
    int lg = 0; for (int m = n; m > 1; m >>= 1) lg++;                         /* std::__lg(n) */
    fk_introsort_loop(first, first + n, 2 * lg);
    if (n > 16) { fk_insertion_sort(first, first + 16); for (fkp *i = first + 16; i != first + n; ++i) fk_unguarded_linear_insert(i); }
    else fk_insertion_sort(first, first + n);
}

/* ---- detectGridFAST ------------------------------------------------------ */
int orc_detect_grid_fast(const uint8_t *img, int w, int h, int stride, int cell,
                         const float *cur_xy, int ncur, int *fast_th_inout,
                         int mask_mode, int do_subpix, float *out_xy, int *out_n)
{
    *out_n = 0;
    if (!img || w <= 0 || h <= 0) return 0;            /* :446-449 empty image */
    if (cell < 7) return -1;
    grid_state g;
    grid_init(&g, w, h, cell, cur_xy, ncur);
    const int th = *fast_th_inout;
    int nboccup = 0, nbempty = 0, nbkps = 0;
    const int cap = cell * cell;
    int *xs = (int *)malloc(sizeof(int) * cap * 3), *ys = xs + cap, *sc = ys + cap;
    fkp *sorted = (fkp *)malloc(sizeof(fkp) * (size_t)cap);
    for (int i = 0; i < g.nbcells; i++) {
        int r = i / g.nwcells, c = i % g.nwcells;
        if (g.occ[r * (g.nwcells + 1) + c]) { nboccup++; continue; }
        nbempty++;
        int x0 = c * cell, y0 = r * cell;
        if (!(x0 + cell < w - 1 && y0 + cell < h - 1)) continue;            /* :510 */
        int n = orc_fast9_16(img + (size_t)y0 * stride + x0, cell, cell, stride, th, 1, xs, ys, sc, cap);
        int best = -1, best_score = -1, nkeep = 0;
        for (int k = 0; k < n; k++) {                       /* runByPixelsMask + sort by response (desc) */
            int lx = xs[k], ly = ys[k], keep;
            if (mask_mode == ORC_MASK_AS_EXECUTED)
                /* CV_32F mask read as bytes: byte lx of the ROI row belongs to float lx/4; the bytes of
                 * 1.0f (00 00 80 3F) are non-zero only at positions 2,3 */
                keep = ((lx & 3) >= 2) && g.mask[(size_t)(y0 + ly) * w + x0 + (lx >> 2)];
            else
                keep = g.mask[(size_t)(y0 + ly) * w + x0 + lx];
            if (keep && sc[k] > best_score) { best_score = sc[k]; best = k; }
            if (keep && g_fast_tie_mode == ORC_FAST_TIE_LIBSTDCXX) { sorted[nkeep].x = lx; sorted[nkeep].y = ly; sorted[nkeep].s = sc[k]; nkeep++; }
        }
        if (best < 0) continue;
        if (g_fast_tie_mode == ORC_FAST_TIE_LIBSTDCXX) {     /* the element std::sort leaves in front */
            fk_std_sort(sorted, nkeep);
            for (int k = 0; k < n; k++) if (xs[k] == sorted[0].x && ys[k] == sorted[0].y) { best = k; break; }
        }
        if (best_score >= 20) {                                              /* :521 */
            int px = xs[best] + x0, py = ys[best] + y0;
            orc_circle_fill0(g.mask, w, h, px, py, g.nhalfcell);             /* :527 */
            out_xy[2 * nbkps] = (float)px; out_xy[2 * nbkps + 1] = (float)py;
            nbkps++;
        }
    }
    /* :546-552 threshold adaptation (int *= double truncates) */
    if ((double)nbkps < 0.5 * (double)nbempty && nbempty > 10) *fast_th_inout = (int)(th * 0.66);
    else if (nbkps == nbempty) *fast_th_inout = (int)(th * 1.5);
    if (nbkps > 0 && do_subpix) orc_corner_subpix(img, w, h, stride, out_xy, nbkps, 3, 30, 0.01);
    *out_n = nbkps;
    free(xs); free(sorted);
    grid_free(&g);
    return 0;
}

/* ---- Sobel dy evaluation order (canonical choice 7) ------------------------------------
 * cv::Sobel(src, dy, CV_32F, 0, 1, 3, scale) applies `scale` to the SMOOTHING kernel when dx == 0 ("kx *= scale",
 * imgproc/src/deriv.cpp), so the row pass of the 8U -> 32F separable filter is the generic RowFilter<uchar, float>
 *   s(x) = ((p[x-1] * k0 + p[x] * k1) + p[x+1] * k2),  k = scale * (1, 2, 1),   every operation rounded to float,
 * and the column pass (-1, 0, 1) is the exact difference of two such rounded rows.  ORC_SOBEL_DY_OPENCV_ROWFILTER (the
 * default) restates that; ORC_SOBEL_DY_EXACT_SUM is round 1's order ((s2 - s0) * scale on exact integer row sums),
 * which differs from it by <= 1 ulp per pixel and can flip arg-max ties in the min-eigenvalue map.  Neither is pinned
 * against a real OpenCV build (none available): the switch exists so that a capture run can decide (tools/ref_capture). */
static int g_sobel_dy_order = ORC_SOBEL_DY_OPENCV_ROWFILTER;
void orc_set_sobel_dy_order(int order) { g_sobel_dy_order = order == ORC_SOBEL_DY_EXACT_SUM ? ORC_SOBEL_DY_EXACT_SUM : ORC_SOBEL_DY_OPENCV_ROWFILTER; }
int orc_get_sobel_dy_order(void) { return g_sobel_dy_order; }

/* ---- min-eigenvalue response of one cell ----------------------------------- */
void orc_cell_mineig(const uint8_t *img, int w, int h, int stride,
                     int x0, int y0, int cell, float *hmap)
{
    const int cs = cell;
    uint8_t *blur = (uint8_t *)malloc((size_t)cs * cs);
    float *dxm = (float *)malloc(sizeof(float) * cs * cs * 2), *dym = dxm + cs * cs;
    double *rows = (double *)malloc(sizeof(double) * cs * cs * 3);
    /* GaussianBlur(im(hroi), 3x3, sigma 0): ROI not isolated -> neighbours come from the parent image,
     * REFLECT_101 only at the true image border; 8-bit fixed point: (sum16 + 8) >> 4 */
    for (int j = 0; j < cs; j++)
        for (int i = 0; i < cs; i++) {
            int s = 0;
            for (int dy = -1; dy <= 1; dy++) {
                const uint8_t *row = img + (size_t)reflect101(y0 + j + dy, h) * stride;
                int wy = dy == 0 ? 2 : 1;
                s += wy * (row[reflect101(x0 + i - 1, w)] + 2 * row[reflect101(x0 + i, w)] + row[reflect101(x0 + i + 1, w)]);
            }
            if (g_blur_mode == ORC_BLUR_HALF_EVEN) blur[j * cs + i] = (uint8_t)lrintf((float)s / 16.f);     /* (s <= 4080: exact in float) */
            else blur[j * cs + i] = (uint8_t)((s + 8) >> 4);
        }
    /* cornerMinEigenVal(filtered, hmap, 3, 3): Sobel 3x3 with scale 1/(4*3*255), REFLECT_101 at the cell edges */
    const float f1 = (float)(1.0 / (4.0 * 3.0 * 255.0)), f0 = (float)(2.0 * (1.0 / (4.0 * 3.0 * 255.0)));
#define B(yy, xx) ((int)blur[reflect101((yy), cs) * cs + reflect101((xx), cs)])
    for (int y = 0; y < cs; y++)
        for (int x = 0; x < cs; x++) {
            float r0 = (float)(B(y - 1, x + 1) - B(y - 1, x - 1));
            float r1 = (float)(B(y, x + 1) - B(y, x - 1));
            float r2 = (float)(B(y + 1, x + 1) - B(y + 1, x - 1));
            dxm[y * cs + x] = (r0 + r2) * f1 + r1 * f0;
            if (g_sobel_dy_order == ORC_SOBEL_DY_EXACT_SUM) {
                float s0 = (float)(B(y - 1, x - 1) + 2 * B(y - 1, x) + B(y - 1, x + 1));
                float s2 = (float)(B(y + 1, x - 1) + 2 * B(y + 1, x) + B(y + 1, x + 1));
                dym[y * cs + x] = (s2 - s0) * f1;
            } else {
                float s0 = ((float)B(y - 1, x - 1) * f1 + (float)B(y - 1, x) * f0) + (float)B(y - 1, x + 1) * f1;
                float s2 = ((float)B(y + 1, x - 1) * f1 + (float)B(y + 1, x) * f0) + (float)B(y + 1, x + 1) * f1;
                dym[y * cs + x] = s2 - s0;
            }
        }
#undef B
    /* cov = (dx*dx, dx*dy, dy*dy) in float; boxFilter 3x3 un-normalised, REFLECT_101, sums in double
     * (RowSum<float,double> ksize 3 = a+b+c; ColumnSum<double,float> sliding), cast to float */
    for (int y = 0; y < cs; y++)
        for (int x = 0; x < cs; x++)
            for (int ch = 0; ch < 3; ch++) {
                double s = 0;
                for (int k = -1; k <= 1; k++) {
                    int xx = reflect101(x + k, cs);
                    float dx = dxm[y * cs + xx], dy = dym[y * cs + xx];
                    float v = ch == 0 ? dx * dx : (ch == 1 ? dx * dy : dy * dy);
                    s = (k == -1) ? (double)v : s + (double)v;
                }
                rows[((size_t)y * cs + x) * 3 + ch] = s;
            }
    /* sliding column sum exactly like ColumnSum: SUM = r(-1) + r(0); out_y = SUM + r(y+1); SUM = out_y - r(y-1) */
    {
        float *cov = (float *)malloc(sizeof(float) * cs * cs * 3);
        for (int x = 0; x < cs; x++)
            for (int ch = 0; ch < 3; ch++) {
                double SUM = 0;
                SUM += rows[((size_t)reflect101(-1, cs) * cs + x) * 3 + ch];
                SUM += rows[((size_t)0 * cs + x) * 3 + ch];
                for (int y = 0; y < cs; y++) {
                    double s0 = SUM + rows[((size_t)reflect101(y + 1, cs) * cs + x) * 3 + ch];
                    cov[((size_t)y * cs + x) * 3 + ch] = (float)s0;
                    SUM = s0 - rows[((size_t)reflect101(y - 1, cs) * cs + x) * 3 + ch];
                }
            }
        for (int k = 0; k < cs * cs; k++) {
            float a = cov[3 * k] * 0.5f, b = cov[3 * k + 1], c = cov[3 * k + 2] * 0.5f;
            hmap[k] = (a + c) - sqrtf((a - c) * (a - c) + b * b);
        }
        free(cov);
    }
    free(blur); free(dxm); free(rows);
}

/* first maximum (row-major) of hmap .* mask over the cell; returns value, writes location */
static float masked_argmax(const float *hmap, const uint8_t *mask, int w, int x0, int y0, int cs, int *mx, int *my)
{
    float best = -FLT_MAX; int bx = 0, by = 0;
    for (int y = 0; y < cs; y++)
        for (int x = 0; x < cs; x++) {
            /* hmap.mul(mask): mask is 0.f or 1.f */
            float v = hmap[y * cs + x] * (mask[(size_t)(y0 + y) * w + x0 + x] ? 1.f : 0.f);
            if (v > best) { best = v; bx = x; by = y; }
        }
    *mx = bx; *my = by;
    return best;
}

/* ---- detectSingleScale ----------------------------------------------------- */
int orc_detect_singlescale(const uint8_t *img, int w, int h, int stride, int cell,
                           const float *cur_xy, int ncur, const int roi[4],
                           double *quality_inout, int do_subpix, float *out_xy, int *out_n)
{
    *out_n = 0;
    if (!img || w <= 0 || h <= 0) return 0;            /* :291-294 */
    if (cell < 4 || cell > 64) return -1;
    grid_state g;
    grid_init(&g, w, h, cell, cur_xy, ncur);
    const double q = *quality_inout;
    int nboccup = 0;
    int *prim = (int *)malloc(sizeof(int) * 4 * (size_t)(g.nbcells > 0 ? g.nbcells : 1));
    int *sec = prim + 2 * g.nbcells;
    uint8_t *has = (uint8_t *)calloc((size_t)(g.nbcells > 0 ? g.nbcells : 1) * 2, 1);
    float *hmap = (float *)malloc(sizeof(float) * cell * cell);
    for (int i = 0; i < g.nbcells; i++) {
        int r = i / g.nwcells, c = i % g.nwcells;
        if (g.occ[r * (g.nwcells + 1) + c]) { nboccup++; continue; }
        int x0 = c * cell, y0 = r * cell;
        if (!(x0 + cell < w - 1 && y0 + cell < h - 1)) continue;            /* :350 */
        orc_cell_mineig(img, w, h, stride, x0, y0, cell, hmap);
        int mx, my;
        double dmax = (double)masked_argmax(hmap, g.mask, w, x0, y0, cell, &mx, &my);
        mx += x0; my += y0;
        if (mx < roi[0] || my < roi[1] || mx >= roi[0] + roi[2] || my >= roi[1] + roi[3]) continue;   /* :363-368 */
        if (dmax >= q) {
            prim[2 * i] = mx; prim[2 * i + 1] = my; has[2 * i] = 1;
            orc_circle_fill0(g.mask, w, h, mx, my, g.nhalfcell);
        }
        dmax = (double)masked_argmax(hmap, g.mask, w, x0, y0, cell, &mx, &my);
        mx += x0; my += y0;
        if (mx < roi[0] || my < roi[1] || mx >= roi[0] + roi[2] || my >= roi[1] + roi[3]) continue;   /* :379-384 */
        if (dmax >= q) {
            sec[2 * i] = mx; sec[2 * i + 1] = my; has[2 * i + 1] = 1;
            orc_circle_fill0(g.mask, w, h, mx, my, g.nhalfcell);
        }
    }
    int nb = 0;
    for (int i = 0; i < g.nbcells; i++)
        if (has[2 * i]) { out_xy[2 * nb] = (float)prim[2 * i]; out_xy[2 * nb + 1] = (float)prim[2 * i + 1]; nb++; }
    if (nb + nboccup < g.nbcells) {                                          /* :400-414 */
        int nbsec = g.nbcells - (nb + nboccup), k = 0;
        for (int i = 0; i < g.nbcells; i++)
            if (has[2 * i + 1]) {
                out_xy[2 * nb] = (float)sec[2 * i]; out_xy[2 * nb + 1] = (float)sec[2 * i + 1]; nb++;
                if (++k == nbsec) break;
            }
    }
    /* :418-423 */
    if ((double)nb < 0.33 * (double)(g.nbcells - nboccup)) *quality_inout = q / 2.;
    else if ((double)nb > 0.9 * (double)(g.nbcells - nboccup)) *quality_inout = q * 1.5;
    if (nb > 0 && do_subpix) orc_corner_subpix(img, w, h, stride, out_xy, nb, 3, 30, 0.01);
    *out_n = nb;
    free(prim); free(has); free(hmap);
    grid_free(&g);
    return 0;
}
