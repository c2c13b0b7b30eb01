// stereo.hip -- data-parallel front half of MapManager::stereoMatching for gfx950
// (/root/reference/src/map_manager.cpp:367-611):
//   k_line_min_sad     FeatureTracker::getLineMinSAD (src/feature_tracker.cpp:138-206): for every left
//                      keypoint a 1-D scan along the same row of the right image (rectified pairs), cost =
//                      mean absolute difference of (2 halfwin + 1)^2 cv::getRectSubPix patches; one WAVEFRONT
//                      per keypoint, one candidate column per lane, the left patch staged in LDS;
//   k_epipolar_check   the gate at :568-590: right keypoint undistorted, |dy| (rectified) or Sampson
//                      distance (src/multi_view_geometry.cpp:797-822) <= 2, y snapped to the left row.
// The two fbKltTracking calls in between (:507, :546) are ov2_fb_klt on (left pyramid, right pyramid).
// Patch arithmetic is cv::getRectSubPix u8->u8: 16.16 fixed-point bilinear weights, (t + 2^15) >> 16,
// replicated border with the vertical-only weights b1/b2 outside the columns (adjustRect) -- integers,
// bit-exact against the oracle.
#include "keypoint_dev.hpp"

#pragma clang fp contract(off)

#define SAD_MAX_WS 21                 // halfwin can grow through the reference's `halfwin += (x + halfwin - cols - 1)` quirk

struct SadWeights { int a11, a12, a21, a22, b1, b2, ipx_off, ipy; };

__device__ __forceinline__ int sad_fixpt(float a) { return __float2int_rn(a * (float)(1 << 16)); }

// one pixel (i, j) of getRectSubPix(src, (ws, ws), centre) where ipx/ipy = floor(centre - (ws-1)/2)
__device__ __forceinline__ int sad_subpix(const uint8_t *__restrict__ img, int pitch, int w, int h, int ipx, int ipy, int i, int j,
                                          const SadWeights &W)
{
    int y0 = ipy + i, y1 = y0 + 1;
    y0 = min(max(y0, 0), h - 1); y1 = min(max(y1, 0), h - 1);
    const int x = ipx + j;
    const uint8_t *r0 = img + y0 * pitch, *r1 = img + y1 * pitch;
    int t;
    if (x < 0) t = (int)r0[0] * W.b1 + (int)r1[0] * W.b2;                       // left of the image: column 0, vertical blend only
    else if (x >= w - 1) t = (int)r0[w - 1] * W.b1 + (int)r1[w - 1] * W.b2;     // no right neighbour: column w-1, vertical blend only
    else t = (int)r0[x] * W.a11 + (int)r0[x + 1] * W.a12 + (int)r1[x] * W.a21 + (int)r1[x + 1] * W.a22;
    return ((t + (1 << 15)) >> 16) & 0xFF;
}

// blockIdx.y: batch item (ov2_stereo_match_batch: n point slots and one entry of n_dev per item; 0 for one keyframe)
__global__ __launch_bounds__(256) void k_line_min_sad(PyrDesc PL, PyrDesc PR, int level, int nwinsize, int go_left,
                                                      const float2 *__restrict__ pts, int n, float *__restrict__ xprior,
                                                      float *__restrict__ l1err, const int *__restrict__ n_dev)
{
    __shared__ uint8_t patch_all[4][SAD_MAX_WS * SAD_MAX_WS + 3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.y, il = blockIdx.x * 4 + wave;
    if (il >= (n_dev ? n_dev[item] : n)) return;                     // whole wavefront
    const int i = item * n + il;
    uint8_t *patch = patch_all[wave];
    const PyrLevelDesc L = PL.lv[level];
    const uint8_t *iml = PL.base + (long long)item * PL.item_stride + L.img_roi, *imr = PR.base + (long long)item * PR.item_stride + PR.lv[level].img_roi;
    const int pitch_l = L.img_pitch, pitch_r = PR.lv[level].img_pitch, w = L.w, h = L.h;
    const float x = pts[i].x, y = pts[i].y;
    float best_e = 255.f, best_c = -1.f;
    int halfwin = nwinsize / 2;
    // int += float: formed in float, truncated toward zero (feature_tracker.cpp:154-161)
    if (x - (float)halfwin < 0.f) halfwin = (int)((float)halfwin + (x - (float)halfwin));
    if (x + (float)halfwin >= (float)w) halfwin = (int)((float)halfwin + (x + (float)halfwin - (float)w - 1.f));
    if (y - (float)halfwin < 0.f) halfwin = (int)((float)halfwin + (y - (float)halfwin));
    if (y + (float)halfwin >= (float)h) halfwin = (int)((float)halfwin + (y + (float)halfwin - (float)h - 1.f));
    const int ws = 2 * halfwin + 1;
    if ((nwinsize & 1) && halfwin > 0 && ws <= SAD_MAX_WS) {
        const int npx = ws * ws;
        // the patch centre (x, y) and every candidate (x -+ k, y) share the fractional parts: one weight set
        const float cx = x - (float)(ws - 1) * 0.5f, cy = y - (float)(ws - 1) * 0.5f;
        const int ipx = (int)floorf(cx), ipy = (int)floorf(cy);
        const float a = cx - (float)ipx, b = cy - (float)ipy;
        SadWeights W;
        W.a11 = sad_fixpt((1.f - a) * (1.f - b)); W.a12 = sad_fixpt(a * (1.f - b)); W.a21 = sad_fixpt((1.f - a) * b); W.a22 = sad_fixpt(a * b);
        W.b1 = sad_fixpt(1.f - b); W.b2 = sad_fixpt(b);
        for (int e = lane; e < npx; e += 64) {
            const int pi = e / ws, pj = e - pi * ws;
            patch[e] = (uint8_t)sad_subpix(iml, pitch_l, w, h, ipx, ipy, pi, pj, W);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // candidates in scan order k = 0, 1, ...: c_k = x - k while c >= halfwin (left; exact in float, so every
        // candidate shares the patch's fractional offsets and weights), or c <- c + 1 while c < w - halfwin
        // (right; the running float sum may round when it crosses a power of two, so c is accumulated step by
        // step like the reference's `c += 1.` and the weights are recomputed from it)
        int best_k = 0x7fffffff;
        float cr = x;
        if (!go_left) for (int t = 0; t < lane; t++) cr += 1.f;
        for (int k = lane; ; k += 64) {
            const float c = go_left ? x - (float)k : cr;
            const bool valid = go_left ? (c >= (float)halfwin) : (c < (float)(w - halfwin));
            if (__builtin_amdgcn_ballot_w64(valid) == 0) break;        // candidates are monotone in k
            if (valid) {
                int cipx = ipx - k;                                     // floor(c - (ws-1)/2) = ipx - k going left
                SadWeights Wc = W;
                if (!go_left) {
                    const float ccx = c - (float)(ws - 1) * 0.5f;
                    cipx = (int)floorf(ccx);
                    const float ac = ccx - (float)cipx;
                    Wc.a11 = sad_fixpt((1.f - ac) * (1.f - b)); Wc.a12 = sad_fixpt(ac * (1.f - b));
                    Wc.a21 = sad_fixpt((1.f - ac) * b); Wc.a22 = sad_fixpt(ac * b);
                }
                int sad = 0;
                for (int pi = 0; pi < ws; pi++)
                    for (int pj = 0; pj < ws; pj++) {
                        const int t = sad_subpix(imr, pitch_r, w, h, cipx, ipy, pi, pj, Wc);
                        const int d = t - (int)patch[pi * ws + pj];
                        sad += d < 0 ? -d : d;
                    }
                float e = (float)sad;
                e /= (float)npx;
                if (e < best_e) { best_e = e; best_k = k; best_c = c; }   // k ascending per lane: first minimum kept
            }
            if (!go_left) for (int t = 0; t < 64; t++) cr += 1.f;
        }
        // first minimum in scan order over the whole wavefront: smallest (e, k)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float oe = __shfl_xor(best_e, off, 64);
            const int ok = __shfl_xor(best_k, off, 64);
            const float oc = __shfl_xor(best_c, off, 64);
            if (oe < best_e || (oe == best_e && ok < best_k)) { best_e = oe; best_k = ok; best_c = oc; }
        }
    }
    if (lane == 0) { xprior[i] = best_c; l1err[i] = best_e; }
}

// float Sampson distance exactly as MultiViewGeometry::computeSampsonDistance narrows its doubles
__device__ __forceinline__ float sampson(const double *F, float lx, float ly, float rx, float ry)
{
    const double l[3] = {(double)lx, (double)ly, 1.}, r[3] = {(double)rx, (double)ry, 1.};
    double rtF[3], Fl[3], Ftr[3];
#pragma unroll
    for (int j = 0; j < 3; j++) rtF[j] = (r[0] * F[j] + r[1] * F[3 + j]) + r[2] * F[6 + j];
    float num = (float)((rtF[0] * l[0] + rtF[1] * l[1]) + rtF[2] * l[2]);
    num *= num;
#pragma unroll
    for (int k = 0; k < 3; k++) Fl[k] = (F[3 * k] * l[0] + F[3 * k + 1] * l[1]) + F[3 * k + 2] * l[2];
#pragma unroll
    for (int j = 0; j < 3; j++) Ftr[j] = (F[j] * r[0] + F[3 + j] * r[1]) + F[6 + j] * r[2];
    const float x1 = (float)Ftr[0], x2 = (float)Fl[0], y1 = (float)Ftr[1], y2 = (float)Fl[1];
    const float den = x1 * x1 + y1 * y1 + x2 * x2 + y2 * y2;
    return sqrtf(num / den);
}

struct EpiParams { double F[9]; int rect; };

__global__ __launch_bounds__(256) void k_epipolar_check(EpiParams E, KpCalib c, const float2 *__restrict__ lunpx, float2 *__restrict__ rkps,
                                                        int n, float2 *__restrict__ runpx, float *__restrict__ epi_err, uint8_t *__restrict__ ok,
                                                        const int *__restrict__ n_dev)
{
    const int item = blockIdx.y, il = blockIdx.x * blockDim.x + threadIdx.x;
    if (il >= (n_dev ? min(n, n_dev[item]) : n)) return;
    const int i = item * n + il;
    const float2 l = lunpx[i];
    float2 rk = rkps[i];
    const float2 ru = kp_undistort_image_point(c, rk);
    float e;
    if (E.rect) {
        e = fabsf(l.y - ru.y);
        rk.y = l.y;                                      // map_manager.cpp:578
        rkps[i] = rk;
    } else e = sampson(E.F, l.x, l.y, ru.x, ru.y);
    runpx[i] = ru;
    epi_err[i] = e;
    ok[i] = e <= 2.f ? 1 : 0;
}

extern "C" {

int ov2_line_min_sad(ov2_ctx *ctx, const ov2_pyr *left, const ov2_pyr *right, int level, int nwinsize, int go_left,
                     const float *pts_xy_h, int n, float *xprior_h, float *l1err_h)
{
    OV2_REQUIRE(ctx && left && right, OV2_EINVAL, "NULL argument");
    if (n <= 0) return OV2_OK;
    OV2_REQUIRE(pts_xy_h && xprior_h && l1err_h, OV2_EINVAL, "NULL point buffer");
    OV2_REQUIRE(left->d.batch == 1 && right->d.batch == 1, OV2_EINVAL, "host-buffer entry point takes batch=1 pyramids");
    OV2_REQUIRE(level >= 0 && level < left->d.n_levels && level < right->d.n_levels, OV2_EINVAL, "level not in the pyramid");
    OV2_REQUIRE(left->d.lv[level].w == right->d.lv[level].w && left->d.lv[level].h == right->d.lv[level].h, OV2_EINVAL,
                "left/right level size differs");
    OV2_REQUIRE(nwinsize > 0 && nwinsize <= 9, OV2_EUNSUPPORTED, "getLineMinSAD window up to 9 (the reference uses 7)");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    if (int rcw = ov2_pyr_wait_ready(ctx, left)) return rcw;       // e.g. the front-end's left pyramid read by the mapper's context
    if (int rcw = ov2_pyr_wait_ready(ctx, right)) return rcw;
    // layout: [pts 8n][xprior 4n][l1err 4n]
    const size_t total = 16 * (size_t)n;
    int rc = ctx->reserve_device(total);  if (rc) return rc;
    rc = ctx->reserve_host(total);        if (rc) return rc;
    uint8_t *hs = (uint8_t *)ctx->h_scratch, *ds = (uint8_t *)ctx->d_scratch;
    memcpy(hs, pts_xy_h, 8 * (size_t)n);
    OV2_HIP_CHECK(hipMemcpyAsync(ds, hs, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_line_min_sad, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, left->d, right->d, level, nwinsize, go_left ? 1 : 0,
                       (const float2 *)ds, n, (float *)(ds + 8 * (size_t)n), (float *)(ds + 12 * (size_t)n), (const int *)nullptr);
    OV2_HIP_CHECK(hipGetLastError());
    OV2_HIP_CHECK(hipMemcpyAsync(hs + 8 * (size_t)n, ds + 8 * (size_t)n, 8 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    memcpy(xprior_h, hs + 8 * (size_t)n, 4 * (size_t)n);
    memcpy(l1err_h, hs + 12 * (size_t)n, 4 * (size_t)n);
    return OV2_OK;
}

int ov2_stereo_epipolar_check(ov2_ctx *ctx, int rect, const double Frl[9], int model, const double K[4], const double *D, int nD,
                              const float *lunpx_xy_h, float *rkps_xy_inout_h, int n, float *runpx_xy_h, float *epi_err_h, uint8_t *ok_h)
{
    OV2_REQUIRE(ctx, OV2_EINVAL, "ctx == NULL");
    if (n <= 0) return OV2_OK;
    OV2_REQUIRE(lunpx_xy_h && rkps_xy_inout_h && ok_h, OV2_EINVAL, "NULL point buffer");
    OV2_REQUIRE(rect || Frl, OV2_EINVAL, "Frl == NULL for a non-rectified pair");
    KpCalib c;
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    int rc = ov2_kp_calib(model, K, D, nD, I3, c);
    if (rc != OV2_OK) return rc;
    EpiParams E;
    for (int i = 0; i < 9; i++) E.F[i] = Frl ? Frl[i] : 0.;
    E.rect = rect ? 1 : 0;
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    // layout: [lunpx 8n][rkps 8n][runpx 8n][err 4n][ok n]
    const size_t N = (size_t)n, o_l = 0, o_r = 8 * N, o_u = 16 * N, o_e = 24 * N, o_k = 28 * N, total = 29 * N;
    rc = ctx->reserve_device(total);  if (rc) return rc;
    rc = ctx->reserve_host(total);    if (rc) return rc;
    uint8_t *hs = (uint8_t *)ctx->h_scratch, *ds = (uint8_t *)ctx->d_scratch;
    memcpy(hs + o_l, lunpx_xy_h, 8 * N);
    memcpy(hs + o_r, rkps_xy_inout_h, 8 * N);
    OV2_HIP_CHECK(hipMemcpyAsync(ds, hs, 16 * N, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_epipolar_check, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, E, c, (const float2 *)(ds + o_l),
                       (float2 *)(ds + o_r), n, (float2 *)(ds + o_u), (float *)(ds + o_e), ds + o_k, (const int *)nullptr);
    OV2_HIP_CHECK(hipGetLastError());
    OV2_HIP_CHECK(hipMemcpyAsync(hs + o_r, ds + o_r, total - o_r, hipMemcpyDeviceToHost, ctx->stream));
    OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    memcpy(rkps_xy_inout_h, hs + o_r, 8 * N);
    if (runpx_xy_h) memcpy(runpx_xy_h, hs + o_u, 8 * N);
    if (epi_err_h) memcpy(epi_err_h, hs + o_e, 4 * N);
    memcpy(ok_h, hs + o_k, N);
    return OV2_OK;
}

// MapManager::stereoMatching's data path (src/map_manager.cpp:367-611) in ONE enqueue and ONE synchronisation: getLineMinSAD
// priors on the coarsest level for every keypoint (rectified pairs, :421-439), both fbKltTracking calls and the retry of the
// failed 3-D-prior tracks in one k_track_klt launch (:497-565), the epipolar gate on the tracked right keypoints (:568-590).
int ov2_stereo_match(ov2_ctx *ctx, const ov2_pyr *left, const ov2_pyr *right, int nklt_win_size, int nklt_pyr_lvl, int max_iter,
                     float eps, float nklt_err, float fmax_fbklt_dist, int rect, const double Frl[9], int model, const double K[4],
                     const double *D, int nD, const float *kps_px_h, const float *kps_unpx_h, const float *priors3d_h,
                     const uint8_t *has_prior3d_h, int n, float *right_px_h, uint8_t *stereo_ok_h)
{
    OV2_REQUIRE(ctx && left && right, OV2_EINVAL, "NULL argument");
    if (n <= 0) return OV2_OK;
    OV2_REQUIRE(kps_px_h && kps_unpx_h && right_px_h && stereo_ok_h, OV2_EINVAL, "NULL point buffer");
    OV2_REQUIRE(has_prior3d_h == nullptr || priors3d_h != nullptr, OV2_EINVAL, "has_prior3d given without priors3d");
    OV2_REQUIRE(rect || Frl, OV2_EINVAL, "Frl == NULL for a non-rectified pair");
    OV2_REQUIRE(left->d.batch == 1 && right->d.batch == 1, OV2_EINVAL, "host-buffer entry point takes batch=1 pyramids");
    KpCalib c;
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    int rc = ov2_kp_calib(model, K, D, nD, I3, c);
    if (rc != OV2_OK) return rc;
    EpiParams E;
    for (int i = 0; i < 9; i++) E.F[i] = Frl ? Frl[i] : 0.;
    E.rect = rect ? 1 : 0;
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    if (int rcw = ov2_pyr_wait_ready(ctx, left)) return rcw;
    if (int rcw = ov2_pyr_wait_ready(ctx, right)) return rcw;
    const int lvl = nklt_pyr_lvl > left->d.n_levels - 1 ? left->d.n_levels - 1 : (nklt_pyr_lvl < 0 ? 0 : nklt_pyr_lvl);
    // layout: [kps 8n][unpx 8n][pri 8n][pts_top 8n][out 8n][runpx 8n][sadx 4n][l1 4n][err 4n][flags n][st n][ok n]
    const size_t N = (size_t)n, o_k = 0, o_u = 8 * N, o_p = 16 * N, o_t = 24 * N, o_o = 32 * N, o_r = 40 * N, o_s = 48 * N, o_l = 52 * N,
                 o_e = 56 * N, o_f = 60 * N, o_st = 61 * N, o_ok = 62 * N, total = 63 * N;
    rc = ctx->reserve_device(total);  if (rc) return rc;
    rc = ctx->reserve_host(total);    if (rc) return rc;
    uint8_t *hs = (uint8_t *)ctx->h_scratch, *ds = (uint8_t *)ctx->d_scratch;
    memcpy(hs + o_k, kps_px_h, 8 * N);
    memcpy(hs + o_u, kps_unpx_h, 8 * N);
    const float up = (float)(1 << lvl), down = 1.f / up;                  // pow(2, nklt_pyr_lvl) and its inverse (:417-418)
    for (int i = 0; i < n; i++) {
        const bool hp = has_prior3d_h && has_prior3d_h[i];
        hs[o_f + i] = hp ? 1 : 0;
        ((float *)(hs + o_p))[2 * i] = hp ? priors3d_h[2 * i] : kps_px_h[2 * i];
        ((float *)(hs + o_p))[2 * i + 1] = hp ? priors3d_h[2 * i + 1] : kps_px_h[2 * i + 1];
        ((float *)(hs + o_t))[2 * i] = kps_px_h[2 * i] * down;             // the keypoint on the coarsest level (:427)
        ((float *)(hs + o_t))[2 * i + 1] = kps_px_h[2 * i + 1] * down;
    }
    OV2_HIP_CHECK(hipMemcpyAsync(ds, hs, 32 * N, hipMemcpyHostToDevice, ctx->stream));
    OV2_HIP_CHECK(hipMemcpyAsync(ds + o_f, hs + o_f, N, hipMemcpyHostToDevice, ctx->stream));
    const float *sad_d = nullptr;
    if (rect) {
        hipLaunchKernelGGL(k_line_min_sad, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, left->d, right->d, lvl, 7, 1,
                           (const float2 *)(ds + o_t), n, (float *)(ds + o_s), (float *)(ds + o_l), (const int *)nullptr);
        sad_d = (const float *)(ds + o_s);
    } else {
        // not rectified: no line search, every keypoint without a 3-D prior starts from its own position (:440-466): a prior
        // of -1 is "nothing found" for k_track_klt
        OV2_HIP_CHECK(hipMemsetAsync(ds + o_s, 0xBF, 4 * N, ctx->stream));                     // 0xBFBFBFBF = -1.498 < 0
        sad_d = (const float *)(ds + o_s);
    }
    rc = ov2_launch_track_klt(ctx->stream, left, right, nklt_win_size, 1, lvl, max_iter, eps, nklt_err, fmax_fbklt_dist, n, nullptr,
                              (const float *)(ds + o_k), (const float *)(ds + o_p), ds + o_f, (float *)(ds + o_o), ds + o_st, nullptr, sad_d, up, ctx->track_impl, 1, ctx->lk_acc);
    if (rc != OV2_OK) return rc;
    hipLaunchKernelGGL(k_epipolar_check, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, E, c, (const float2 *)(ds + o_u),
                       (float2 *)(ds + o_o), n, (float2 *)(ds + o_r), (float *)(ds + o_e), ds + o_ok, (const int *)nullptr);
    OV2_HIP_CHECK(hipGetLastError());
    OV2_HIP_CHECK(hipMemcpyAsync(hs + o_o, ds + o_o, 8 * N, hipMemcpyDeviceToHost, ctx->stream));
    OV2_HIP_CHECK(hipMemcpyAsync(hs + o_st, ds + o_st, 2 * N, hipMemcpyDeviceToHost, ctx->stream));
    OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; i++) {
        const bool tracked = (hs[o_st + i] & 1) != 0;
        stereo_ok_h[i] = (tracked && hs[o_ok + i]) ? 1 : 0;
        right_px_h[2 * i] = tracked ? ((const float *)(hs + o_o))[2 * i] : 0.f;
        right_px_h[2 * i + 1] = tracked ? ((const float *)(hs + o_o))[2 * i + 1] : 0.f;
    }
    return OV2_OK;
}

// ov2_stereo_match for the keyframes of a lock-step batch (all sequences of a rank reach their keyframes together): items [0, n_items)
// of two batch pyramids, n_max point slots per item, ONE enqueue and ONE synchronisation for all of them -- the same three kernels with
// the grid extended by the item.  Per item the results are those of ov2_stereo_match on that item (tests/test_gpu_stereo.py).
int ov2_stereo_match_batch(ov2_ctx *ctx, const ov2_pyr *left, const ov2_pyr *right, int n_items, int nklt_win_size, int nklt_pyr_lvl, int max_iter,
                           float eps, float nklt_err, float fmax_fbklt_dist, int rect, const double Frl[9], int model, const double K[4],
                           const double *D, int nD, int n_max, const float *kps_px_h, const float *kps_unpx_h, const float *priors3d_h,
                           const uint8_t *has_prior3d_h, const int *n_h, float *right_px_h, uint8_t *stereo_ok_h)
{
    OV2_REQUIRE(ctx && left && right && n_h, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(n_items >= 1 && n_items <= left->d.batch && n_items <= right->d.batch && n_max >= 1, OV2_EINVAL, "n_items / n_max out of range");
    OV2_REQUIRE(kps_px_h && kps_unpx_h && right_px_h && stereo_ok_h, OV2_EINVAL, "NULL point buffer");
    OV2_REQUIRE(has_prior3d_h == nullptr || priors3d_h != nullptr, OV2_EINVAL, "has_prior3d given without priors3d");
    OV2_REQUIRE(rect || Frl, OV2_EINVAL, "Frl == NULL for a non-rectified pair");
    int n_total = 0;
    for (int b = 0; b < n_items; b++) { OV2_REQUIRE(n_h[b] >= 0 && n_h[b] <= n_max, OV2_EINVAL, "an item carries more keypoints than n_max slots"); n_total += n_h[b]; }
    if (n_total == 0) return OV2_OK;
    KpCalib c;
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    int rc = ov2_kp_calib(model, K, D, nD, I3, c);
    if (rc != OV2_OK) return rc;
    EpiParams E;
    for (int i = 0; i < 9; i++) E.F[i] = Frl ? Frl[i] : 0.;
    E.rect = rect ? 1 : 0;
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    if (int rcw = ov2_pyr_wait_ready(ctx, left)) return rcw;
    if (int rcw = ov2_pyr_wait_ready(ctx, right)) return rcw;
    const int lvl = nklt_pyr_lvl > left->d.n_levels - 1 ? left->d.n_levels - 1 : (nklt_pyr_lvl < 0 ? 0 : nklt_pyr_lvl);
    // layout (N = n_items * n_max slots): [kps 8N][unpx 8N][pri 8N][pts_top 8N][flags N][counts 4*items, 256-aligned] | [out 8N][runpx 8N][sadx 4N][l1 4N][err 4N][st N][ok N]
    const size_t N = (size_t)n_items * n_max, o_k = 0, o_u = 8 * N, o_p = 16 * N, o_t = 24 * N, o_f = 32 * N, o_n = (33 * N + 255) & ~(size_t)255;
    const size_t in_bytes = (o_n + 4 * (size_t)n_items + 255) & ~(size_t)255;
    const size_t o_o = in_bytes, o_r = o_o + 8 * N, o_s = o_r + 8 * N, o_l = o_s + 4 * N, o_e = o_l + 4 * N, o_st = o_e + 4 * N, o_ok = o_st + N, total = o_ok + N;
    rc = ctx->reserve_device(total);  if (rc) return rc;
    rc = ctx->reserve_host(total);    if (rc) return rc;
    uint8_t *hs = (uint8_t *)ctx->h_scratch, *ds = (uint8_t *)ctx->d_scratch;
    const float up = (float)(1 << lvl), down = 1.f / up;                  // pow(2, nklt_pyr_lvl) and its inverse (:417-418)
    for (int b = 0; b < n_items; b++) {
        const size_t o = (size_t)b * n_max, n = (size_t)n_h[b];
        ((int *)(hs + o_n))[b] = n_h[b];
        memcpy(hs + o_k + 8 * o, kps_px_h + 2 * o, 8 * n);
        memcpy(hs + o_u + 8 * o, kps_unpx_h + 2 * o, 8 * n);
        for (size_t i = o; i < o + n; i++) {
            const bool hp = has_prior3d_h && has_prior3d_h[i];
            hs[o_f + i] = hp ? 1 : 0;
            ((float *)(hs + o_p))[2 * i] = hp ? priors3d_h[2 * i] : kps_px_h[2 * i];
            ((float *)(hs + o_p))[2 * i + 1] = hp ? priors3d_h[2 * i + 1] : kps_px_h[2 * i + 1];
            ((float *)(hs + o_t))[2 * i] = kps_px_h[2 * i] * down;             // the keypoint on the coarsest level (:427)
            ((float *)(hs + o_t))[2 * i + 1] = kps_px_h[2 * i + 1] * down;
        }
    }
    OV2_HIP_CHECK(hipMemcpyAsync(ds, hs, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    const int *n_d = (const int *)(ds + o_n);
    if (rect) hipLaunchKernelGGL(k_line_min_sad, dim3((n_max + 3) / 4, n_items), dim3(256), 0, ctx->stream, left->d, right->d, lvl, 7, 1,
                                 (const float2 *)(ds + o_t), n_max, (float *)(ds + o_s), (float *)(ds + o_l), n_d);
    else OV2_HIP_CHECK(hipMemsetAsync(ds + o_s, 0xBF, 4 * N, ctx->stream));                       // "nothing found" (see ov2_stereo_match)
    rc = ov2_launch_track_klt(ctx->stream, left, right, nklt_win_size, 1, lvl, max_iter, eps, nklt_err, fmax_fbklt_dist, n_max, n_d,
                              (const float *)(ds + o_k), (const float *)(ds + o_p), ds + o_f, (float *)(ds + o_o), ds + o_st, nullptr,
                              (const float *)(ds + o_s), up, ctx->track_impl, n_items, ctx->lk_acc);
    if (rc != OV2_OK) return rc;
    hipLaunchKernelGGL(k_epipolar_check, dim3((n_max + 255) / 256, n_items), dim3(256), 0, ctx->stream, E, c, (const float2 *)(ds + o_u),
                       (float2 *)(ds + o_o), n_max, (float2 *)(ds + o_r), (float *)(ds + o_e), ds + o_ok, n_d);
    OV2_HIP_CHECK(hipGetLastError());
    OV2_HIP_CHECK(hipMemcpyAsync(hs + o_o, ds + o_o, 8 * N, hipMemcpyDeviceToHost, ctx->stream));
    OV2_HIP_CHECK(hipMemcpyAsync(hs + o_st, ds + o_st, 2 * N, hipMemcpyDeviceToHost, ctx->stream));
    OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < n_items; b++)
        for (size_t i = (size_t)b * n_max; i < (size_t)b * n_max + (size_t)n_h[b]; i++) {
            const bool tracked = (hs[o_st + i] & 1) != 0;
            stereo_ok_h[i] = (tracked && hs[o_ok + i]) ? 1 : 0;
            right_px_h[2 * i] = tracked ? ((const float *)(hs + o_o))[2 * i] : 0.f;
            right_px_h[2 * i + 1] = tracked ? ((const float *)(hs + o_o))[2 * i + 1] : 0.f;
        }
    return OV2_OK;
}

} // extern "C"
