// track.hip -- single-sequence per-frame tracker: VisualFrontEnd::preprocessImage + VisualFrontEnd::kltTracking
// (/root/reference/src/visual_front_end.cpp:1143-1177 and :132-275) as ONE enqueue on the context's stream.
//
// The drop-in case has one camera stream, ~300 keypoints per frame and a 50 ms frame budget: nothing here is
// bandwidth-bound, everything is latency -- PCIe round trips, kernel launches, host synchronisations.  Per frame:
//   H2D frame (pinned, one copy) -> k_clahe_lut -> k_clahe_apply (writes pyramid level 0 + border) -> k_pyr_level x3
//   H2D keypoint block (one copy: n, keypoints, priors, flags) -> k_track_klt (both fbKltTracking calls of the
//   reference and the retry of lost prior tracks in one launch, lk.hip) -> D2H result block (one copy) -> ONE sync.
// With use_graph the whole sequence is captured once per pyramid parity and replayed with hipGraphLaunch.
#include "common.hpp"
#include "keypoint_dev.hpp"
#include <new>
#include <vector>

int ov2_launch_compute_keypoints(hipStream_t s, const KpCalib &c, const float *px_d, int n_max, const int *n_dev, float *unpx_d, double *bv_d, int items = 1);

struct ov2_tracker {
    ov2_ctx *ctx = nullptr;
    ov2_tracker_config cfg;
    ov2_pyr *pyr[2] = {nullptr, nullptr};
    int cur = 0;                   // index of cur_pyr_; prev_pyr_ = pyr[cur ^ 1]
    int frames = 0;
    // pinned host / device mirrors
    uint8_t *himg = nullptr, *dimg = nullptr; size_t img_pitch = 0, img_bytes = 0;
    uint8_t *lut = nullptr;
    uint8_t *hblk = nullptr, *dblk = nullptr;
    uint8_t *kblk = nullptr;       // what the LK kernel dereferences: dblk, or the device alias of the pinned block (zero_copy)
    bool zero_copy = false;        // the kernel reads the 9 KB keypoint block from / writes its results to pinned host memory
                                   // itself: two blit kernels (5.5 us each + their launch gaps) cost more than the PCIe round trip
    size_t o_n, o_kps, o_pri, o_flg, in_bytes, o_out, o_st, o_it, o_unpx, o_bv, blk_bytes;
    // optional Frame::computeKeypoint (undistorted pixel + bearing, src/frame.cpp:246-254) of every output position in the same enqueue
    bool has_calib = false;
    KpCalib calib;
    std::vector<float> last_unpx;      // 2 per keypoint of the last klt / track_frame call
    std::vector<double> last_bv;       // 3 per keypoint
    int last_n = 0;                    // keypoints of the LAST klt / track_frame call that completed (0 after a call that tracked nothing or failed)
    hipGraphExec_t gexec[2] = {nullptr, nullptr};
    bool graph_ok = false;
};

static inline size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }

// preprocessImage body for pyramid `dst` (no swap, no event): frame H2D + CLAHE / level-0 copy + coarser levels
static int enqueue_preprocess(ov2_tracker *t, ov2_pyr *dst)
{
    ov2_ctx *ctx = t->ctx;
    const ov2_tracker_config &c = t->cfg;
    OV2_HIP_CHECK(hipMemcpyAsync(t->dimg, t->himg, t->img_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (c.use_clahe) {
        const PyrLevelDesc &L0 = dst->d.lv[0];
        int rc = ov2_launch_clahe(ctx, t->dimg, c.w, c.h, (int)t->img_pitch, 0, 1, c.clahe_clip, c.tiles_x, c.tiles_y,
                                  dst->d.base + L0.img_roi, L0.img_pitch, (size_t)dst->d.item_stride, t->lut, dst->d.win);
        if (rc != OV2_OK) return rc;
        return ov2_launch_pyr_build(ctx, dst, nullptr, 0, 0);
    }
    return ov2_launch_pyr_build(ctx, dst, t->dimg, (int)t->img_pitch, 0);
}

// kltTracking body: keypoint block H2D, the fused LK launch, result block D2H
static int enqueue_klt(ov2_tracker *t, const ov2_pyr *prev, const ov2_pyr *cur)
{
    ov2_ctx *ctx = t->ctx;
    const ov2_tracker_config &c = t->cfg;
    if (!t->zero_copy) OV2_HIP_CHECK(hipMemcpyAsync(t->dblk, t->hblk, t->in_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint8_t *k = t->kblk;
    int rc = ov2_launch_track_klt(ctx->stream, prev, cur, c.win, c.prior_pyr_lvl, c.nklt_pyr_lvl, c.max_iter, c.eps, c.err_th,
                                  c.fb_dist, c.n_max, (const int *)(k + t->o_n), (const float *)(k + t->o_kps),
                                  (const float *)(k + t->o_pri), k + t->o_flg, (float *)(k + t->o_out),
                                  k + t->o_st, (int *)(k + t->o_it), nullptr, 0.f, ctx->track_impl, 1, ctx->lk_acc);
    if (rc != OV2_OK) return rc;
    if (t->has_calib) {
        rc = ov2_launch_compute_keypoints(ctx->stream, t->calib, (const float *)(k + t->o_out), c.n_max, (const int *)(k + t->o_n),
                                          (float *)(k + t->o_unpx), (double *)(k + t->o_bv));
        if (rc != OV2_OK) return rc;
    }
    if (!t->zero_copy)
        OV2_HIP_CHECK(hipMemcpyAsync(t->hblk + t->o_out, t->dblk + t->o_out, t->blk_bytes - t->o_out, hipMemcpyDeviceToHost, ctx->stream));
    return OV2_OK;
}

static void stage_image(ov2_tracker *t, const uint8_t *img_h, int stride)
{
    if (img_h == t->himg && (size_t)stride == t->img_pitch) return;      // already in the pinned buffer
    const int w = t->cfg.w, h = t->cfg.h;
    if ((size_t)stride == t->img_pitch) { memcpy(t->himg, img_h, (size_t)stride * h); return; }
    for (int y = 0; y < h; y++) memcpy(t->himg + (size_t)y * t->img_pitch, img_h + (size_t)y * stride, (size_t)w);
}

static void stage_points(ov2_tracker *t, const float *kps, const float *pri, const uint8_t *has_prior, int n, int use_prior)
{
    *(int *)(t->hblk + t->o_n) = n;
    memcpy(t->hblk + t->o_kps, kps, 8 * (size_t)n);
    memcpy(t->hblk + t->o_pri, pri, 8 * (size_t)n);
    uint8_t *f = t->hblk + t->o_flg;
    if (use_prior && has_prior) for (int i = 0; i < n; i++) f[i] = has_prior[i] ? 1 : 0;
    else memset(f, 0, (size_t)n);
}

// after a synchronisation: copy the m results of the staged chunk out
static void collect_chunk(ov2_tracker *t, int m, float *out_xy, uint8_t *status, int off = 0)
{
    memcpy(out_xy, t->hblk + t->o_out, 8 * (size_t)m);
    memcpy(status, t->hblk + t->o_st, (size_t)m);
    if (t->has_calib) {
        if (t->last_unpx.size() < 2 * (size_t)(off + m)) { t->last_unpx.resize(2 * (size_t)(off + m)); t->last_bv.resize(3 * (size_t)(off + m)); }
        memcpy(t->last_unpx.data() + 2 * (size_t)off, t->hblk + t->o_unpx, 8 * (size_t)m);
        memcpy(t->last_bv.data() + 3 * (size_t)off, t->hblk + t->o_bv, 24 * (size_t)m);
    }
}

// A frame may carry more keypoints than the tracker's capacity (the reference only prunes at the next keyframe,
// src/map_manager.cpp:74, while extractKeypoints tops cells up that several tracked keypoints share): keypoints
// [off, n) are run n_max at a time through the same fused launch (keypoints are independent inside fbKltTracking).
static int klt_overflow_chunks(ov2_tracker *t, const float *kps, const float *pri, const uint8_t *has_prior, int off, int n,
                               int use_prior, float *out_xy, uint8_t *status)
{
    for (; off < n; off += t->cfg.n_max) {
        const int m = n - off < t->cfg.n_max ? n - off : t->cfg.n_max;
        stage_points(t, kps + 2 * (size_t)off, pri + 2 * (size_t)off, has_prior ? has_prior + off : nullptr, m, use_prior);
        const int rc = enqueue_klt(t, t->pyr[t->cur ^ 1], t->pyr[t->cur]);
        if (rc != OV2_OK) return rc;
        OV2_HIP_CHECK(hipStreamSynchronize(t->ctx->stream));
        collect_chunk(t, m, out_xy + 2 * (size_t)off, status + off, off);
    }
    return OV2_OK;
}

// the cross-keypoint rule of visual_front_end.cpp:225-230 over all n results
static int apply_p3p_rule(ov2_tracker *t, const float *kps, const uint8_t *has_prior, int use_prior, int n, float *out_xy,
                          uint8_t *status, int *p3p_req)
{
    size_t nbkps = 0, nbgood = 0;
    if (use_prior && has_prior)
        for (int i = 0; i < n; i++) if (has_prior[i]) { nbkps++; if ((status[i] & 3) == 1) nbgood++; }
    int p3p = 0;
    if (nbkps > 0 && (double)nbgood < 0.33 * (double)nbkps) {
        // "Motion model might be quite wrong": vpriors = vkps for the second call (:229) -- only the lost prior
        // tracks had a prior different from their keypoint, so only they are re-run, from the keypoints themselves
        p3p = 1;
        std::vector<int> idx;
        for (int i = 0; i < n; i++) if (status[i] & 2) idx.push_back(i);
        if (!idx.empty()) {
            const int m = (int)idx.size();
            std::vector<float> k2(2 * (size_t)m), p2(2 * (size_t)m);
            std::vector<uint8_t> s2((size_t)m);
            for (int j = 0; j < m; j++) { k2[2 * j] = p2[2 * j] = kps[2 * idx[j]]; k2[2 * j + 1] = p2[2 * j + 1] = kps[2 * idx[j] + 1]; }
            const ov2_tracker_config &c = t->cfg;
            const int rc = ov2_fb_klt(t->ctx, t->pyr[t->cur ^ 1], t->pyr[t->cur], c.win, c.nklt_pyr_lvl, c.max_iter, c.eps, c.err_th,
                                      c.fb_dist, k2.data(), p2.data(), m, s2.data(), nullptr);
            if (rc != OV2_OK) return rc;
            for (int j = 0; j < m; j++) {
                out_xy[2 * idx[j]] = p2[2 * j]; out_xy[2 * idx[j] + 1] = p2[2 * j + 1];
                status[idx[j]] = (uint8_t)(2 | (s2[j] ? 1 : 0));
            }
            if (t->has_calib) {                                        // their undistorted pixels / bearings follow the new positions
                std::vector<float> u2(2 * (size_t)m);
                std::vector<double> b2(3 * (size_t)m);
                const KpCalib &kc = t->calib;
                const double Kk[4] = {kc.fx, kc.fy, kc.cx, kc.cy};
                const int rck = ov2_compute_keypoints(t->ctx, kc.model, Kk, kc.nD ? kc.k : nullptr, kc.nD, kc.iK, p2.data(), m, u2.data(), b2.data());
                if (rck != OV2_OK) return rck;
                for (int j = 0; j < m; j++) {
                    memcpy(&t->last_unpx[2 * (size_t)idx[j]], &u2[2 * (size_t)j], 8);
                    memcpy(&t->last_bv[3 * (size_t)idx[j]], &b2[3 * (size_t)j], 24);
                }
            }
        }
    }
    if (p3p_req) *p3p_req = p3p;
    return OV2_OK;
}

static void tracker_free(ov2_tracker *t)
{
    if (!t) return;
    if (t->ctx) { (void)hipSetDevice(t->ctx->device); (void)hipStreamSynchronize(t->ctx->stream); }
    for (int i = 0; i < 2; i++) { if (t->gexec[i]) (void)hipGraphExecDestroy(t->gexec[i]); ov2_pyr_destroy(t->pyr[i]); }
    if (t->himg) (void)hipHostFree(t->himg);
    if (t->hblk) (void)hipHostFree(t->hblk);
    if (t->dimg) (void)hipFree(t->dimg);
    if (t->dblk) (void)hipFree(t->dblk);
    if (t->lut) (void)hipFree(t->lut);
    delete t;
}

// capture preprocess(dst = pyr[parity]) + klt(prev = pyr[parity ^ 1], cur = pyr[parity]) into an executable graph
static int capture_graph(ov2_tracker *t, int parity)
{
    ov2_ctx *ctx = t->ctx;
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return OV2_EUNSUPPORTED; }
    int rc = enqueue_preprocess(t, t->pyr[parity]);
    if (rc == OV2_OK) rc = enqueue_klt(t, t->pyr[parity ^ 1], t->pyr[parity]);
    const hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    if (rc != OV2_OK || e != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); (void)hipGetLastError(); return rc != OV2_OK ? rc : OV2_EUNSUPPORTED; }
    const hipError_t ie = hipGraphInstantiate(&t->gexec[parity], g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ie != hipSuccess) { t->gexec[parity] = nullptr; (void)hipGetLastError(); return OV2_EUNSUPPORTED; }
    return OV2_OK;
}

extern "C" {

int ov2_tracker_create(ov2_ctx *ctx, const ov2_tracker_config *cfg, ov2_tracker **out)
{
    OV2_REQUIRE(ctx && cfg && out, OV2_EINVAL, "NULL argument");
    *out = nullptr;
    OV2_REQUIRE(cfg->w > 0 && cfg->h > 0 && cfg->n_max > 0 && cfg->nklt_pyr_lvl >= 0 && cfg->prior_pyr_lvl >= 0, OV2_EINVAL, "bad tracker geometry");
    OV2_REQUIRE(!cfg->use_clahe || (cfg->tiles_x > 0 && cfg->tiles_y > 0 && cfg->tiles_x <= cfg->w && cfg->tiles_y <= cfg->h), OV2_EINVAL, "bad CLAHE tiles");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    ov2_tracker *t = new (std::nothrow) ov2_tracker();
    OV2_REQUIRE(t != nullptr, OV2_ENOMEM, "out of host memory");
    t->ctx = ctx; t->cfg = *cfg;
    int rc = OV2_OK;
    for (int i = 0; i < 2 && rc == OV2_OK; i++) rc = ov2_pyr_create(ctx, cfg->w, cfg->h, cfg->win, cfg->nklt_pyr_lvl, 1, &t->pyr[i]);
    if (rc != OV2_OK) { tracker_free(t); return rc; }
    t->img_pitch = up16((size_t)cfg->w);
    t->img_bytes = t->img_pitch * (size_t)cfg->h;
    const size_t nm = (size_t)cfg->n_max;
    t->o_n = 0; t->o_kps = 16; t->o_pri = t->o_kps + 8 * nm; t->o_flg = t->o_pri + 8 * nm; t->in_bytes = up16(t->o_flg + nm);
    t->o_out = t->in_bytes; t->o_st = t->o_out + 8 * nm; t->o_it = up16(t->o_st + nm);
    t->o_unpx = up16(t->o_it + 4 * nm); t->o_bv = up16(t->o_unpx + 8 * nm); t->blk_bytes = t->o_bv + 24 * nm;
    const size_t lut_bytes = cfg->use_clahe ? (size_t)cfg->tiles_x * cfg->tiles_y * 256 : 256;
    hipError_t e = hipHostMalloc((void **)&t->himg, t->img_bytes + 256, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&t->hblk, t->blk_bytes, hipHostMallocMapped);
    if (e == hipSuccess) e = hipMalloc((void **)&t->dimg, t->img_bytes + 256);
    if (e == hipSuccess) e = hipMalloc((void **)&t->dblk, t->blk_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&t->lut, lut_bytes);
    if (e == hipSuccess) e = hipMemsetAsync(t->dblk, 0, t->blk_bytes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(t->dimg, 0, t->img_bytes + 256, ctx->stream);
    if (e != hipSuccess) { ov2_set_error("ov2_tracker_create: %s", hipGetErrorString(e)); tracker_free(t); return OV2_ENOMEM; }
    memset(t->himg, 0, t->img_bytes + 256);
    memset(t->hblk, 0, t->blk_bytes);
    t->kblk = t->dblk;
    {   // kernels read / write the pinned block through its device alias (no staging copies); the device block is the fallback
        void *alias = nullptr;
        if (hipHostGetDevicePointer(&alias, t->hblk, 0) == hipSuccess && alias) { t->kblk = (uint8_t *)alias; t->zero_copy = true; }
        else (void)hipGetLastError();
    }
    t->graph_ok = cfg->use_graph != 0;
    if (t->graph_ok) {
        // Both per-parity graphs are captured HERE, before the caller's other threads exist (the reference constructs everything in
        // SlamManager's constructor, ov2slam.cpp:91-113, before its threads start).  Capturing lazily inside the first two
        // track_frame calls raced with the mapper thread: while this stream was in capture, hipStreamWaitEvent on another context
        // for a pyramid event last recorded on this stream failed with "dependency created on uncaptured work in another stream".
        OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));                    // the memsets above
        for (int parity = 0; parity < 2 && t->graph_ok; parity++) {
            const int rcg = capture_graph(t, parity);
            if (rcg == OV2_EUNSUPPORTED) t->graph_ok = false;                    // stream cannot be captured: plain enqueue per frame
            else if (rcg != OV2_OK) { tracker_free(t); return rcg; }
        }
    }
    *out = t;
    return OV2_OK;
}

void ov2_tracker_destroy(ov2_tracker *t) { tracker_free(t); }

int ov2_tracker_set_calibration(ov2_tracker *t, int model, const double K[4], const double *D, int nD, const double iK[9])
{
    OV2_REQUIRE(t, OV2_EINVAL, "NULL tracker");
    KpCalib c;
    const int rc = ov2_kp_calib(model, K, D, nD, iK, c);
    if (rc != OV2_OK) return rc;
    OV2_HIP_CHECK(hipSetDevice(t->ctx->device));
    OV2_HIP_CHECK(hipStreamSynchronize(t->ctx->stream));
    t->calib = c; t->has_calib = true;
    // the per-frame enqueue gained a kernel: the captured graphs are stale (call this right after ov2_tracker_create, before the
    // caller's other threads run: see the note on graph capture there)
    for (int i = 0; i < 2; i++) if (t->gexec[i]) { (void)hipGraphExecDestroy(t->gexec[i]); t->gexec[i] = nullptr; }
    for (int parity = 0; parity < 2 && t->graph_ok; parity++) {
        const int rcg = capture_graph(t, parity);
        if (rcg == OV2_EUNSUPPORTED) t->graph_ok = false;
        else if (rcg != OV2_OK) return rcg;
    }
    return OV2_OK;
}

int ov2_tracker_last_keypoints(const ov2_tracker *t, int n, float *unpx_xy_h, double *bv_xyz_h)
{
    OV2_REQUIRE(t && t->has_calib, OV2_EINVAL, "no calibration set on this tracker");
    OV2_REQUIRE(n >= 0 && n <= t->last_n, OV2_EINVAL, "more keypoints than the last tracking call returned");
    if (unpx_xy_h) memcpy(unpx_xy_h, t->last_unpx.data(), 8 * (size_t)n);
    if (bv_xyz_h) memcpy(bv_xyz_h, t->last_bv.data(), 24 * (size_t)n);
    return OV2_OK;
}

uint8_t *ov2_tracker_image_buffer(ov2_tracker *t, int *stride)
{
    if (!t) return nullptr;
    if (stride) *stride = (int)t->img_pitch;
    return t->himg;
}

const ov2_pyr *ov2_tracker_cur_pyr(const ov2_tracker *t) { return t ? t->pyr[t->cur] : nullptr; }
const ov2_pyr *ov2_tracker_prev_pyr(const ov2_tracker *t) { return t ? t->pyr[t->cur ^ 1] : nullptr; }
int ov2_tracker_frames(const ov2_tracker *t) { return t ? t->frames : 0; }
int ov2_tracker_uses_graph(const ov2_tracker *t) { return t && t->graph_ok ? 1 : 0; }

int ov2_tracker_preprocess(ov2_tracker *t, const uint8_t *img_h, int stride)
{
    OV2_REQUIRE(t && img_h, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(stride >= t->cfg.w, OV2_EINVAL, "stride < width");
    OV2_HIP_CHECK(hipSetDevice(t->ctx->device));
    // the previous frame's H2D must have left the pinned buffer (it has: every klt / track_frame call synchronises;
    // two preprocess calls in a row are ordered explicitly)
    if (t->frames > 0) OV2_HIP_CHECK(hipEventSynchronize(t->pyr[t->cur]->ready));
    stage_image(t, img_h, stride);
    if (t->frames > 0) t->cur ^= 1;                                   // prev_pyr_.swap(cur_pyr_)  (:1169)
    const int rc = enqueue_preprocess(t, t->pyr[t->cur]);
    if (rc != OV2_OK) { if (t->frames > 0) t->cur ^= 1; return rc; }  // nothing was swapped as far as the caller is concerned
    t->frames++;
    return ov2_pyr_mark_ready(t->ctx, t->pyr[t->cur]);
}

int ov2_tracker_klt(ov2_tracker *t, const float *kps_xy_h, const float *prior_xy_h, const uint8_t *has_prior_h, int n,
                    int klt_use_prior, float *out_xy_h, uint8_t *status_h, int *p3p_req)
{
    OV2_REQUIRE(t, OV2_EINVAL, "NULL tracker");
    if (p3p_req) *p3p_req = 0;
    t->last_n = 0;                                                     // ov2_tracker_last_keypoints describes THIS call from here on
    if (n <= 0) return OV2_OK;
    OV2_REQUIRE(kps_xy_h && prior_xy_h && out_xy_h && status_h, OV2_EINVAL, "NULL point buffer");
    OV2_REQUIRE(t->frames >= 2, OV2_EINVAL, "kltTracking needs two preprocessed frames");
    OV2_HIP_CHECK(hipSetDevice(t->ctx->device));
    int rc = klt_overflow_chunks(t, kps_xy_h, prior_xy_h, has_prior_h, 0, n, klt_use_prior, out_xy_h, status_h);
    if (rc != OV2_OK) return rc;
    rc = apply_p3p_rule(t, kps_xy_h, has_prior_h, klt_use_prior, n, out_xy_h, status_h, p3p_req);
    if (rc == OV2_OK && t->has_calib) t->last_n = n;
    return rc;
}

int ov2_tracker_track_frame(ov2_tracker *t, const uint8_t *img_h, int stride, const float *kps_xy_h,
                            const float *prior_xy_h, const uint8_t *has_prior_h, int n, int klt_use_prior,
                            float *out_xy_h, uint8_t *status_h, int *p3p_req)
{
    OV2_REQUIRE(t && img_h, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(stride >= t->cfg.w, OV2_EINVAL, "stride < width");
    OV2_REQUIRE(n >= 0, OV2_EINVAL, "negative keypoint count");
    OV2_REQUIRE(n == 0 || (kps_xy_h && prior_xy_h && out_xy_h && status_h), OV2_EINVAL, "NULL point buffer");
    if (p3p_req) *p3p_req = 0;
    t->last_n = 0;
    if (t->frames == 0 || n == 0) {
        // first frame (trackMono returns right after preprocessImage) or nothing to track
        const int rc = ov2_tracker_preprocess(t, img_h, stride);
        if (rc != OV2_OK) return rc;
        if (n > 0) memset(status_h, 0, (size_t)n);
        return ov2_ctx_sync(t->ctx);
    }
    OV2_HIP_CHECK(hipSetDevice(t->ctx->device));
    OV2_HIP_CHECK(hipEventSynchronize(t->pyr[t->cur]->ready));         // an asynchronous preprocess may still read the pinned frame
    stage_image(t, img_h, stride);
    const int n0 = n < t->cfg.n_max ? n : t->cfg.n_max;                // the rest: klt_overflow_chunks
    stage_points(t, kps_xy_h, prior_xy_h, has_prior_h, n0, klt_use_prior);
    t->cur ^= 1;                                                       // prev_pyr_.swap(cur_pyr_)  (:1169)
    int rc = OV2_OK;
    bool launched = false;
    if (t->graph_ok) {
        if (!t->gexec[t->cur]) {
            rc = capture_graph(t, t->cur);
            if (rc == OV2_EUNSUPPORTED) { t->graph_ok = false; rc = OV2_OK; }      // stream cannot be captured: plain enqueue
            else if (rc != OV2_OK) { t->cur ^= 1; return rc; }
        }
        if (t->graph_ok) {
            OV2_HIP_CHECK(hipGraphLaunch(t->gexec[t->cur], t->ctx->stream));
            launched = true;
        }
    }
    if (!launched) {
        rc = enqueue_preprocess(t, t->pyr[t->cur]);
        if (rc == OV2_OK) rc = enqueue_klt(t, t->pyr[t->cur ^ 1], t->pyr[t->cur]);
        if (rc != OV2_OK) { t->cur ^= 1; return rc; }
    }
    t->frames++;
    rc = ov2_pyr_mark_ready(t->ctx, t->pyr[t->cur]);
    if (rc != OV2_OK) return rc;
    OV2_HIP_CHECK(hipStreamSynchronize(t->ctx->stream));
    collect_chunk(t, n0, out_xy_h, status_h);
    if (n > n0) {
        rc = klt_overflow_chunks(t, kps_xy_h, prior_xy_h, has_prior_h, n0, n, klt_use_prior, out_xy_h, status_h);
        if (rc != OV2_OK) return rc;
    }
    rc = apply_p3p_rule(t, kps_xy_h, has_prior_h, klt_use_prior, n, out_xy_h, status_h, p3p_req);
    if (rc == OV2_OK && t->has_calib) t->last_n = n;
    return rc;
}

} // extern "C"
