/*
 * clahe.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * cv::CLAHE::apply for CV_8UC1 as the reference uses it:
 *   handle creation   src/ov2slam.cpp:85-89   cv::createCLAHE(fclahe_val, Size(w/50, h/50))
 *   application       src/visual_front_end.cpp:1159 (left image, every frame), src/mapper.cpp:76 (right image)
 * OpenCV is not under /root/reference; the public 3.4/4.x algorithm (imgproc/src/clahe.cpp:
 * CLAHE_CalcLut_Body, CLAHE_Interpolation_Body) is restated -- PARITY UNPINNED (no reference tests).
 */
#include "ov2_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

int orc_clahe(const uint8_t *src, int w, int h, int stride, double clip_limit, int tiles_x, int tiles_y,
              uint8_t *dst, int dst_stride)
{
    if (!src || !dst || w <= 0 || h <= 0 || tiles_x <= 0 || tiles_y <= 0) return -1;
    /* apply(): pad right/bottom with REFLECT_101 unless BOTH dimensions divide evenly */
    int ew = w, eh = h;
    if (!(w % tiles_x == 0 && h % tiles_y == 0)) {
        ew = w + (tiles_x - (w % tiles_x));
        eh = h + (tiles_y - (h % tiles_y));
    }
    const int tw = ew / tiles_x, th = eh / tiles_y;
    const int tile_total = tw * th;
    const float lut_scale = (float)(256 - 1) / (float)tile_total;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * tile_total / 256);
        if (clip < 1) clip = 1;
    }
    uint8_t *lut = (uint8_t *)malloc((size_t)tiles_x * tiles_y * 256);
    for (int t = 0; t < tiles_x * tiles_y; t++) {
        const int ty = t / tiles_x, tx = t % tiles_x;
        int hist[256];
        memset(hist, 0, sizeof(hist));
        for (int y = ty * th; y < (ty + 1) * th; y++) {
            const uint8_t *row = src + (size_t)reflect101(y, h) * stride;
            for (int x = tx * tw; x < (tx + 1) * tw; x++) hist[row[reflect101(x, w)]]++;
        }
        if (clip > 0) {
            int clipped = 0;
            for (int i = 0; i < 256; i++)
                if (hist[i] > clip) { clipped += hist[i] - clip; hist[i] = clip; }
            int batch = clipped / 256, residual = clipped - batch * 256;
            for (int i = 0; i < 256; i++) hist[i] += batch;
            if (residual != 0) {
                int step = 256 / residual; if (step < 1) step = 1;
                for (int i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++;
            }
        }
        int sum = 0;
        for (int i = 0; i < 256; i++) {
            sum += hist[i];
            long v = lrintf((float)sum * lut_scale);          /* saturate_cast<uchar>(float) = cvRound + clamp */
            lut[(size_t)t * 256 + i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
    for (int y = 0; y < h; y++) {
        const float tyf = (float)y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tiles_y - 1) ty2 = tiles_y - 1;
        for (int x = 0; x < w; x++) {
            const float txf = (float)x * inv_tw - 0.5f;
            int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
            const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tiles_x - 1) tx2 = tiles_x - 1;
            const int v = src[(size_t)y * stride + x];
            const float l11 = lut[((size_t)ty1 * tiles_x + tx1) * 256 + v], l12 = lut[((size_t)ty1 * tiles_x + tx2) * 256 + v];
            const float l21 = lut[((size_t)ty2 * tiles_x + tx1) * 256 + v], l22 = lut[((size_t)ty2 * tiles_x + tx2) * 256 + v];
            const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
            long r = lrintf(res);
            dst[(size_t)y * dst_stride + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
    free(lut);
    return 0;
}
