/*
 * stereo.c -- CPU ORACLE (test infrastructure) for the data-parallel front half of
 * MapManager::stereoMatching (/root/reference/src/map_manager.cpp:367-611):
 *   - FeatureTracker::getLineMinSAD  (src/feature_tracker.cpp:138-206): 1-D SAD scan along the row of a
 *     rectified pair, patches through cv::getRectSubPix (u8 -> u8), cost cv::norm(NORM_L1) / #pixels;
 *   - the epipolar gate of :568-590: |lunpx.y - runpx.y| (rectified) or
 *     MultiViewGeometry::computeSampsonDistance (src/multi_view_geometry.cpp:797-822), threshold 2.
 * cv::getRectSubPix is restated from the public OpenCV 3.4/4.x modules/imgproc/src/samplers.cpp
 * (getRectSubPix_Cn_<uchar, uchar, int, scale_fixpt, cast_8u> + adjustRect): "parity unpinned".
 */
#include "ov2_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <stddef.h>


static inline int fixpt(float a) { return (int)lrintf(a * (float)(1 << 16)); }          /* scale_fixpt: cvRound(a * 65536) */
static inline uint8_t cast8(int a) { return (uint8_t)((a + (1 << 15)) >> 16); }         /* cast_8u */

/* cv::getRectSubPix(src u8, Size(pw, ph), center, dst u8) */
void orc_get_rect_subpix_8u(const uint8_t *src, int src_step, int sw, int sh, uint8_t *dst, int pw, int ph, float cx_f, float cy_f)
{
    float cx = cx_f - (pw - 1) * 0.5f, cy = cy_f - (ph - 1) * 0.5f;
    int ipx = (int)floorf(cx), ipy = (int)floorf(cy);
    const float a = cx - ipx, b = cy - ipy;
    const int a11 = fixpt((1.f - a) * (1.f - b)), a12 = fixpt(a * (1.f - b)), a21 = fixpt((1.f - a) * b), a22 = fixpt(a * b);
    const int b1 = fixpt(1.f - b), b2 = fixpt(b);
    if (0 <= ipx && ipx < sw - pw && 0 <= ipy && ipy < sh - ph) {
        const uint8_t *p = src + (size_t)ipy * src_step + ipx;
        for (int i = 0; i < ph; i++, p += src_step, dst += pw)
            for (int j = 0; j < pw; j++)
                dst[j] = cast8(p[j] * a11 + p[j + 1] * a12 + p[j + src_step] * a21 + p[j + src_step + 1] * a22);
        return;
    }
    /* adjustRect: replicated border */
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
        for (int j = 0; j < rx; j++) dst[j] = cast8(p[rx] * b1 + p2[rx] * b2);
        for (int j = rx; j < rw; j++) dst[j] = cast8(p[j] * a11 + p[j + 1] * a12 + p2[j] * a21 + p2[j + 1] * a22);
        for (int j = rw > rx ? rw : rx; j < pw; j++) dst[j] = cast8(p[rw] * b1 + p2[rw] * b2);
        if (i < rh) p = p2;
    }
}

/* FeatureTracker::getLineMinSAD (feature_tracker.cpp:138-206).  Returns xprior (-1: none) and the minimal
 * mean absolute difference (255 when no candidate beat the initial `minsad = 255.`; the reference leaves
 * l1err uninitialised on its early returns -- the oracle writes 255 there too). */
void orc_line_min_sad(const uint8_t *iml, int lstride, const uint8_t *imr, int rstride, int w, int h,
                      float x, float y, int nwinsize, int go_left, float *xprior, float *l1err)
{
    *xprior = -1.f; *l1err = 255.f;
    if (nwinsize % 2 == 0) return;
    int halfwin = nwinsize / 2;
    /* int += float: the sum is formed in float and truncated toward zero (:154-161) */
    if (x - halfwin < 0) halfwin = (int)((float)halfwin + (x - halfwin));
    if (x + halfwin >= w) halfwin = (int)((float)halfwin + (x + halfwin - w - 1));
    if (y - halfwin < 0) halfwin = (int)((float)halfwin + (y - halfwin));
    if (y + halfwin >= h) halfwin = (int)((float)halfwin + (y + halfwin - h - 1));
    if (halfwin <= 0) return;
    const int ws = 2 * halfwin + 1, nbwinpx = ws * ws;
    uint8_t *patch = (uint8_t *)malloc((size_t)nbwinpx), *target = (uint8_t *)malloc((size_t)nbwinpx);
    float minsad = 255.f;
    orc_get_rect_subpix_8u(iml, lstride, w, h, patch, ws, ws, x, y);
    /* `c -= 1.` / `c += 1.`: float <- double(c) -+ 1.0, exact for pixel coordinates */
    for (float c = x; go_left ? (c >= halfwin) : (c < w - halfwin); c = (float)((double)c + (go_left ? -1. : 1.))) {
        orc_get_rect_subpix_8u(imr, rstride, w, h, target, ws, ws, c, y);
        int sad = 0;
        for (int i = 0; i < nbwinpx; i++) sad += abs((int)patch[i] - (int)target[i]);
        float e = (float)(double)sad;              /* l1err = cv::norm(...) (double -> float) */
        e /= nbwinpx;
        if (e < minsad) { minsad = e; *xprior = c; }
    }
    *l1err = minsad;
    free(patch); free(target);
}

void orc_line_min_sad_batch(const uint8_t *iml, int lstride, const uint8_t *imr, int rstride, int w, int h,
                            const float *xy, int n, int nwinsize, int go_left, float *xprior, float *l1err)
{
    for (int i = 0; i < n; i++)
        orc_line_min_sad(iml, lstride, imr, rstride, w, h, xy[2 * i], xy[2 * i + 1], nwinsize, go_left, &xprior[i], &l1err[i]);
}

/* MultiViewGeometry::computeSampsonDistance (multi_view_geometry.cpp:797-822): Eigen double products
 * ((a0 b0 + a1 b1) + a2 b2), results narrowed to float where the reference stores them in floats */
float orc_sampson_distance(const double F[9], float lx, float ly, float rx, float ry)
{
    const double l[3] = {(double)lx, (double)ly, 1.}, r[3] = {(double)rx, (double)ry, 1.};
    double rtF[3], Fl[3], Ftr[3];
    for (int j = 0; j < 3; j++) rtF[j] = (r[0] * F[j] + r[1] * F[3 + j]) + r[2] * F[6 + j];       /* r^T F */
    float num = (float)((rtF[0] * l[0] + rtF[1] * l[1]) + rtF[2] * l[2]);
    num *= num;
    for (int i = 0; i < 3; i++) Fl[i] = (F[3 * i] * l[0] + F[3 * i + 1] * l[1]) + F[3 * i + 2] * l[2];      /* F l */
    for (int j = 0; j < 3; j++) Ftr[j] = (F[j] * r[0] + F[3 + j] * r[1]) + F[6 + j] * r[2];               /* F^T r */
    const float x1 = (float)Ftr[0], x2 = (float)Fl[0], y1 = (float)Ftr[1], y2 = (float)Fl[1];
    const float den = x1 * x1 + y1 * y1 + x2 * x2 + y2 * y2;
    return sqrtf(num / den);
}

/* epipolar gate of MapManager::stereoMatching (:568-590) for n candidate pairs.
 * rkps is in/out: in rectified mode the accepted AND rejected right keypoints get y = lunpx.y (:578). */
void orc_stereo_epipolar_check(int rect, const double Frl[9], int model, const double K[4], const double *D, int nD,
                               const float *lunpx, float *rkps, int n, float *runpx, float *epi_err, uint8_t *ok)
{
    const double iK[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    orc_compute_keypoints(model, K, D, nD, iK, rkps, n, runpx, NULL);
    for (int i = 0; i < n; i++) {
        float e;
        if (rect) {
            e = fabsf(lunpx[2 * i + 1] - runpx[2 * i + 1]);
            rkps[2 * i + 1] = lunpx[2 * i + 1];
        } else e = orc_sampson_distance(Frl, lunpx[2 * i], lunpx[2 * i + 1], runpx[2 * i], runpx[2 * i + 1]);
        epi_err[i] = e;
        ok[i] = e <= 2.f;
    }
}
