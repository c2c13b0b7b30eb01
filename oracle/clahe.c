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

typedef struct {
    const uint8_t *src; int w, h, stride, tiles_x, tiles_y, tw, th, clip; float lut_scale; uint8_t *lut;
    uint8_t *dst; int dst_stride; float inv_tw, inv_th;
    const int *ind1, *ind2; const float *xa, *xa1;      /* per-column tables, built once per apply() like OpenCV's
                                                           CLAHE_Interpolation_Body constructor (ind1_p, ind2_p, xa_p, xa1_p) */
} clahe_job;

/* CLAHE_CalcLut_Body: tiles in parallel */
static void clahe_lut_tiles(int t0, int t1, void *ctx)
{
    const clahe_job *jb = (const clahe_job *)ctx;
    const uint8_t *src = jb->src;
    const int w = jb->w, h = jb->h, stride = jb->stride, tiles_x = jb->tiles_x, tw = jb->tw, th = jb->th, clip = jb->clip;
    for (int t = t0; t < t1; t++) {
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
            long v = lrintf((float)sum * jb->lut_scale);      /* saturate_cast<uchar>(float) = cvRound + clamp */
            jb->lut[(size_t)t * 256 + i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

/* CLAHE_Interpolation_Body: rows in parallel */
static void clahe_interp_rows(int y0, int y1, void *ctx)
{
    const clahe_job *jb = (const clahe_job *)ctx;
    const uint8_t *src = jb->src, *lut = jb->lut; uint8_t *dst = jb->dst;
    const int w = jb->w, stride = jb->stride, tiles_x = jb->tiles_x, tiles_y = jb->tiles_y, dst_stride = jb->dst_stride;
    const float inv_th = jb->inv_th;
    for (int y = y0; y < y1; y++) {
        const float tyf = (float)y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tiles_y - 1) ty2 = tiles_y - 1;
        const uint8_t *lut1 = lut + (size_t)ty1 * tiles_x * 256, *lut2 = lut + (size_t)ty2 * tiles_x * 256;
        const uint8_t *srow = src + (size_t)y * stride;
        for (int x = 0; x < w; x++) {
            const int v = srow[x];
            const int i1 = jb->ind1[x] + v, i2 = jb->ind2[x] + v;
            const float xa = jb->xa[x], xa1 = jb->xa1[x];
            const float l11 = lut1[i1], l12 = lut1[i2], l21 = lut2[i1], l22 = lut2[i2];
            const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
            long r = lrintf(res);
            dst[(size_t)y * dst_stride + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
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
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * tile_total / 256);
        if (clip < 1) clip = 1;
    }
    clahe_job jb;
    jb.src = src; jb.w = w; jb.h = h; jb.stride = stride; jb.tiles_x = tiles_x; jb.tiles_y = tiles_y; jb.tw = tw; jb.th = th;
    jb.clip = clip; jb.lut_scale = (float)(256 - 1) / (float)tile_total;
    jb.lut = (uint8_t *)malloc((size_t)tiles_x * tiles_y * 256);
    jb.dst = dst; jb.dst_stride = dst_stride; jb.inv_tw = 1.0f / (float)tw; jb.inv_th = 1.0f / (float)th;
    int *ind = (int *)malloc(sizeof(int) * 2 * (size_t)w);
    float *xw = (float *)malloc(sizeof(float) * 2 * (size_t)w);
    for (int x = 0; x < w; x++) {
        const float txf = (float)x * jb.inv_tw - 0.5f;
        int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
        xw[x] = txf - (float)tx1; xw[w + x] = 1.0f - xw[x];
        if (tx1 < 0) tx1 = 0;
        if (tx2 > tiles_x - 1) tx2 = tiles_x - 1;
        ind[x] = tx1 * 256; ind[w + x] = tx2 * 256;
    }
    jb.ind1 = ind; jb.ind2 = ind + w; jb.xa = xw; jb.xa1 = xw + w;
    orc_parallel_for(tiles_x * tiles_y, clahe_lut_tiles, &jb, 1);
    orc_parallel_for(h, clahe_interp_rows, &jb, 8);
    free(ind); free(xw);
    free(jb.lut);
    return 0;
}
