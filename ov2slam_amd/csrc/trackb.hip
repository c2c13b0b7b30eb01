// trackb.hip -- lock-step tracker: `batch` camera streams advance one frame per call (ov2_btracker_*).
//
// The offline / batch mode of the reference's benchmark protocol (/root/reference/benchmark_scripts/euroc_bench.sh:3-27 plays whole
// sequences; BASELINE.json configs[4] shards 11 of them over the GPUs of a node).  A rank that owns several sequences can push
// each through its own ov2_tracker (track.hip) -- every stream is then a chain of ~10 small dependent launches per frame and the
// streams together saturate the command processor's launch rate with the CUs ~5 % busy (profiles/archive/r4_stream_concurrency.txt).
// Here the per-frame enqueue of VisualFrontEnd::preprocessImage + VisualFrontEnd::kltTracking (+ Frame::computeKeypoint)
// (src/visual_front_end.cpp:1143-1177, :132-275; src/frame.cpp:246-254) is issued ONCE for all streams: the same kernels as
// track.hip -- CLAHE, pyramid levels, the fused kltTracking kernel (lkw.hip / lk.hip), k_compute_keypoints -- with the grid
// extended by the batch item.  Per step: one H2D of the frames, the kernels, one synchronisation.  Items [0, n_active) take part
// (sequences of different length drop out from the tail).
// An offline host knows frame f + 1 while frame f is tracked: the step is a three-stage pipeline on three streams --
//   copy stream      H2D of frame f + 2                       (ov2_btracker_upload)
//   prep stream      CLAHE + pyramid of frame f + 1           (ov2_btracker_prepare; into the NEXT pyramid set)
//   context stream   kltTracking + computeKeypoint of frame f (ov2_btracker_track_frame: waits for its pyramids, one sync)
// with three pinned staging sets and eight pyramid sets in rotation; events order producer -> consumer and consumer -> the producer
// that reuses a buffer.  A caller that uses neither look-ahead call gets the same results from the plain in-order enqueue.
// Results per item are bit-identical to an ov2_tracker fed the same frames and keypoints (tests/test_gpu_lockstep.py).
#include "common.hpp"
#include "keypoint_dev.hpp"
#include <deque>
#include <new>
#include <vector>

int ov2_launch_compute_keypoints(hipStream_t s, const KpCalib &c, const float *px_d, int n_max, const int *n_dev, float *unpx_d, double *bv_d, int items = 1);

#define BT_SETS 8       // pyramid sets in rotation: cur, prev, the one being prepared for the next frame, five of slack for the mapper
                        // context (a keyframe's pyramids outlive it by BT_SETS - 1 steps: the keyframe period of the shipped parameter files)
#define BT_IMG_SETS 3   // staging sets: frame f (read by its pre-processing), f + 1 (arriving), f + 2 (being filled by the host)

struct ov2_btracker {
    ov2_ctx *ctx = nullptr;
    ov2_tracker_config cfg;
    int batch = 0;
    // pyramid sets in rotation (cur_pyr_, prev_pyr_, ...): a keyframe's pyramids stay valid for BT_SETS - 1 more steps (BT_SETS - 2
    // when the next frame is prepared ahead), so the mapper contexts that stereo-match it rarely hold the SLAM thread up (with two
    // sets 12 % of the lock-step wall clock was that wait)
    ov2_pyr *pyr[BT_SETS] = {};
    std::vector<ov2_pyr *> view[BT_SETS];    // batch-1 aliases of the items (stereo matching, the p3p retry, single-item detection)
    int cur = 0;                       // index of cur_pyr_ (= (frames - 1) % BT_SETS); prev_pyr_ = pyr[(cur + BT_SETS - 1) % BT_SETS]
    int prev() const { return (cur + BT_SETS - 1) % BT_SETS; }
    long frames = 0;
    // frames: pinned staging sets + their device mirrors; a copy stream for uploads and a stream for pre-processing started ahead
    uint8_t *himg[BT_IMG_SETS] = {nullptr, nullptr, nullptr}, *dimg[BT_IMG_SETS] = {nullptr, nullptr, nullptr};
    size_t img_pitch = 0, img_bytes = 0;          // per item (img_bytes a multiple of 256)
    hipStream_t cs = nullptr, ps = nullptr;
    ov2_ctx side;                                 // what the launchers of clahe.hip / pyramid.hip see when they enqueue on `ps`
    uint8_t *lut = nullptr;                       // CLAHE tables of a step (the context's scratch is busy with the detector on the main stream)
    hipEvent_t up_ev[BT_IMG_SETS] = {nullptr, nullptr, nullptr};     // set `which` has arrived in dimg[which] (copy stream)
    hipEvent_t used_ev[BT_IMG_SETS] = {nullptr, nullptr, nullptr};   // the kernels that read dimg[which] are done (main or prep stream)
    int up_n[BT_IMG_SETS] = {0, 0, 0};            // items covered by the pending upload of the set (0: none pending)
    // Frame k lives in pyramid set k % BT_SETS.  pre_count = frames whose preprocessImage has been enqueued (ahead or in order);
    // frames prepared ahead wait in prepq (at most two: the one about to be tracked and the one after it)
    struct Prep { int which, n, set; };
    std::deque<Prep> prepq;
    long pre_count = 0;
    // keypoint block: pinned + mapped (the kernels read / write it through its device alias) or mirrored on the device
    uint8_t *hblk = nullptr, *dblk = nullptr, *kblk = nullptr;
    bool zero_copy = false;
    size_t o_n = 0, o_kps = 0, o_pri = 0, o_flg = 0, in_bytes = 0, o_out = 0, o_st = 0, o_unpx = 0, o_bv = 0, blk_bytes = 0;
    bool has_calib = false;
    KpCalib calib;
    std::vector<int> last_n;           // per item: keypoints of the last track_frame (0 before / after a call that tracked nothing)
    // keyframe detection: device mirrors of the items' current keypoints and of the detector's output, pinned staging
    uint8_t *d_det = nullptr, *h_det = nullptr; size_t det_in_bytes = 0;
    float *d_out = nullptr; uint8_t *h_out = nullptr; int out_cap = 0;
    // a step between ov2_btracker_track_frame_begin and _end: what _end needs to finish it
    struct Pending { bool on = false, tracked = false; int n_active = 0, use_prior = 0; const float *kps = nullptr; const uint8_t *has_prior = nullptr; std::vector<int> n; } pend;
};

static inline size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }
static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

static void btracker_free(ov2_btracker *t)
{
    if (!t) return;
    if (t->ctx) { (void)hipSetDevice(t->ctx->device); (void)hipStreamSynchronize(t->ctx->stream); }
    if (t->cs) { (void)hipStreamSynchronize(t->cs); (void)hipStreamDestroy(t->cs); }
    if (t->ps) { (void)hipStreamSynchronize(t->ps); (void)hipStreamDestroy(t->ps); }
    if (t->lut) (void)hipFree(t->lut);
    for (int i = 0; i < BT_SETS; i++) {
        for (ov2_pyr *v : t->view[i]) ov2_pyr_destroy(v);
        ov2_pyr_destroy(t->pyr[i]);
    }
    for (int i = 0; i < BT_IMG_SETS; i++) {
        if (t->up_ev[i]) (void)hipEventDestroy(t->up_ev[i]);
        if (t->used_ev[i]) (void)hipEventDestroy(t->used_ev[i]);
        if (t->himg[i]) (void)hipHostFree(t->himg[i]);
        if (t->dimg[i]) (void)hipFree(t->dimg[i]);
    }
    if (t->hblk) (void)hipHostFree(t->hblk);
    if (t->dblk) (void)hipFree(t->dblk);
    if (t->h_det) (void)hipHostFree(t->h_det);
    if (t->d_det) (void)hipFree(t->d_det);
    if (t->h_out) (void)hipHostFree(t->h_out);
    if (t->d_out) (void)hipFree(t->d_out);
    delete t;
}

// a pyramid handle that covers items [0, n) of `p` (host-side descriptor copy: the launchers size their grids from d.batch)
static inline ov2_pyr prefix_of(const ov2_pyr *p, int n) { ov2_pyr q = *p; q.d.batch = n; return q; }

// preprocessImage of items [0, n) into pyramid `dst` from dimg[which], enqueued on `on` (the caller's context, or the tracker's
// prep-stream context when the frame is prepared ahead)
static int enqueue_preprocess(ov2_btracker *t, ov2_ctx *on, ov2_pyr *dst, int which, int n)
{
    const ov2_tracker_config &c = t->cfg;
    ov2_pyr q = prefix_of(dst, n);
    int rc;
    if (c.use_clahe) {
        const PyrLevelDesc &L0 = q.d.lv[0];
        int l1_done = 0;
        rc = ov2_launch_clahe(on, t->dimg[which], c.w, c.h, (int)t->img_pitch, t->img_bytes, n, c.clahe_clip, c.tiles_x, c.tiles_y,
                              q.d.base + L0.img_roi, L0.img_pitch, (size_t)q.d.item_stride, t->lut, q.d.win, &q.d, &l1_done);
        if (rc != OV2_OK) return rc;
        rc = ov2_launch_pyr_build(on, &q, nullptr, 0, 0, l1_done);
    } else
        rc = ov2_launch_pyr_build(on, &q, t->dimg[which], (int)t->img_pitch, t->img_bytes);
    if (rc != OV2_OK) return rc;
    OV2_HIP_CHECK(hipEventRecord(t->used_ev[which], on->stream));
    return OV2_OK;
}

// kltTracking of items [0, n): keypoint block in, the fused LK launch, computeKeypoint, result block out
static int enqueue_klt(ov2_btracker *t, const ov2_pyr *prev, const ov2_pyr *cur, int n)
{
    ov2_ctx *ctx = t->ctx;
    const ov2_tracker_config &c = t->cfg;
    if (!t->zero_copy) OV2_HIP_CHECK(hipMemcpyAsync(t->dblk, t->hblk, t->in_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint8_t *k = t->kblk;
    int rc = ov2_launch_track_klt(ctx->stream, prev, cur, c.win, c.prior_pyr_lvl, c.nklt_pyr_lvl, c.max_iter, c.eps, c.err_th, c.fb_dist,
                                  c.n_max, (const int *)(k + t->o_n), (const float *)(k + t->o_kps), (const float *)(k + t->o_pri),
                                  k + t->o_flg, (float *)(k + t->o_out), k + t->o_st, nullptr, nullptr, 0.f, ctx->track_impl, n, ctx->lk_acc);
    if (rc != OV2_OK) return rc;
    if (t->has_calib) {
        rc = ov2_launch_compute_keypoints(ctx->stream, t->calib, (const float *)(k + t->o_out), c.n_max, (const int *)(k + t->o_n),
                                          (float *)(k + t->o_unpx), (double *)(k + t->o_bv), n);
        if (rc != OV2_OK) return rc;
    }
    if (!t->zero_copy)
        OV2_HIP_CHECK(hipMemcpyAsync(t->hblk + t->o_out, t->dblk + t->o_out, t->blk_bytes - t->o_out, hipMemcpyDeviceToHost, ctx->stream));
    return OV2_OK;
}

// Frames of items [0, n) -> dimg[which] on the main stream, `which` = the staging set the step reads.  Frames that already sit in
// a pinned slot are not copied on the host; a set uploaded ahead (ov2_btracker_upload) is only waited for; *prepared: the set was
// pre-processed ahead as well (ov2_btracker_prepare) -- nothing is enqueued for it here.
static int stage_and_upload(ov2_btracker *t, int n, const uint8_t *const *img_h, int stride, int *which_out, bool *prepared)
{
    *prepared = false;
    const int w = t->cfg.w, h = t->cfg.h;
    int which = -1;
    for (int s = 0; s < BT_IMG_SETS && which < 0; s++) {
        bool all = (size_t)stride == t->img_pitch;
        for (int b = 0; b < n && all; b++) all = img_h[b] == t->himg[s] + (size_t)b * t->img_bytes;
        if (all) which = s;
    }
    const bool in_place = which >= 0;
    if (!in_place) which = (int)(t->frames % BT_IMG_SETS);
    // the set's previous H2D (an upload started ahead, or the inline copy of an earlier step) must have left the pinned slots before
    // they are rewritten, and a pending look-ahead upload of a set that is now filled differently is void
    if (!t->prepq.empty()) {
        // frames were prepared ahead: the step must consume the oldest of them, exactly
        const ov2_btracker::Prep &pf = t->prepq.front();
        OV2_REQUIRE(in_place && pf.which == which && pf.n >= n && pf.set == (int)(t->frames % BT_SETS), OV2_EINVAL,
                    "ov2_btracker_track_frame: the frames passed are not the ones ov2_btracker_prepare was called for");
        *prepared = true; *which_out = which;
        return OV2_OK;
    }
    if (!in_place) {
        if (t->up_n[which]) { OV2_HIP_CHECK(hipEventSynchronize(t->up_ev[which])); t->up_n[which] = 0; }
        for (int b = 0; b < n; b++) {
            uint8_t *dst = t->himg[which] + (size_t)b * t->img_bytes;
            if ((size_t)stride == t->img_pitch) memcpy(dst, img_h[b], (size_t)stride * (h - 1) + (size_t)w);   // the last row holds w valid bytes only
            else for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * t->img_pitch, img_h[b] + (size_t)y * stride, (size_t)w);
        }
    }
    if (in_place && t->up_n[which] >= n) OV2_HIP_CHECK(hipStreamWaitEvent(t->ctx->stream, t->up_ev[which], 0));
    else {
        if (t->up_n[which]) OV2_HIP_CHECK(hipStreamWaitEvent(t->ctx->stream, t->up_ev[which], 0));     // a shorter look-ahead copy: order after it
        OV2_HIP_CHECK(hipMemcpyAsync(t->dimg[which], t->himg[which], (size_t)n * t->img_bytes, hipMemcpyHostToDevice, t->ctx->stream));
    }
    t->up_n[which] = 0;
    *which_out = which;
    return OV2_OK;
}

static void stage_points(ov2_btracker *t, int n_active, const float *kps, const float *pri, const uint8_t *has_prior, const int *n_h, int use_prior)
{
    const size_t nm = (size_t)t->cfg.n_max;
    int *nd = (int *)(t->hblk + t->o_n);
    for (int b = 0; b < t->batch; b++) nd[b] = b < n_active ? n_h[b] : 0;
    for (int b = 0; b < n_active; b++) {
        const size_t n = (size_t)n_h[b], o = (size_t)b * nm;
        if (!n) continue;
        memcpy(t->hblk + t->o_kps + 8 * o, kps + 2 * o, 8 * n);
        memcpy(t->hblk + t->o_pri + 8 * o, pri + 2 * o, 8 * n);
        uint8_t *f = t->hblk + t->o_flg + o;
        if (use_prior && has_prior) for (size_t i = 0; i < n; i++) f[i] = has_prior[o + i] ? 1 : 0;
        else memset(f, 0, n);
    }
}

// the cross-keypoint rule of visual_front_end.cpp:225-230 for item b (same as track.hip: apply_p3p_rule), on the item's views
static int apply_p3p_rule(ov2_btracker *t, int b, const float *kps, const uint8_t *has_prior, int use_prior, int n, float *out_xy,
                          uint8_t *status, int *p3p_req)
{
    size_t nbkps = 0, nbgood = 0;
    if (use_prior && has_prior)
        for (int i = 0; i < n; i++) if (has_prior[i]) { nbkps++; if ((status[i] & 3) == 1) nbgood++; }
    int p3p = 0;
    if (nbkps > 0 && (double)nbgood < 0.33 * (double)nbkps) {
        p3p = 1;                                                        // vpriors = vkps for the second call (:229)
        std::vector<int> idx;
        for (int i = 0; i < n; i++) if (status[i] & 2) idx.push_back(i);
        if (!idx.empty()) {
            const int m = (int)idx.size();
            std::vector<float> k2(2 * (size_t)m), p2(2 * (size_t)m);
            std::vector<uint8_t> s2((size_t)m);
            for (int j = 0; j < m; j++) { k2[2 * j] = p2[2 * j] = kps[2 * idx[j]]; k2[2 * j + 1] = p2[2 * j + 1] = kps[2 * idx[j] + 1]; }
            const ov2_tracker_config &c = t->cfg;
            const int rc = ov2_fb_klt(t->ctx, t->view[t->prev()][b], t->view[t->cur][b], c.win, c.nklt_pyr_lvl, c.max_iter, c.eps, c.err_th,
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
                float *un = (float *)(t->hblk + t->o_unpx) + 2 * (size_t)b * t->cfg.n_max;
                double *bv = (double *)(t->hblk + t->o_bv) + 3 * (size_t)b * t->cfg.n_max;
                for (int j = 0; j < m; j++) {
                    memcpy(un + 2 * (size_t)idx[j], &u2[2 * (size_t)j], 8);
                    memcpy(bv + 3 * (size_t)idx[j], &b2[3 * (size_t)j], 24);
                }
            }
        }
    }
    if (p3p_req) *p3p_req = p3p;
    return OV2_OK;
}

static int detect_common(ov2_btracker *t, int mode, int n_active, int cell, const float *cur_xy_h, const int *ncur_h, const int roi[4],
                         double *quality_inout, int *fast_th_inout, int mask_mode, int do_subpix, float *out_xy_h, int out_cap, int *out_n_h)
{
    OV2_REQUIRE(t && out_xy_h && out_n_h, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(n_active >= 1 && n_active <= t->batch, OV2_EINVAL, "n_active out of range");
    OV2_REQUIRE(t->frames > 0, OV2_EINVAL, "no frame has been preprocessed yet");
    OV2_REQUIRE(cell >= 8 && out_cap >= (mode == 0 ? 1 : 2) * (t->cfg.w / cell) * (t->cfg.h / cell), OV2_EINVAL, "out_cap too small for this cell size");
    ov2_ctx *ctx = t->ctx;
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t nm = (size_t)t->cfg.n_max, o_cur = up256(4 * (size_t)t->batch);
    int *nh = (int *)t->h_det;
    for (int b = 0; b < n_active; b++) {
        const int n = ncur_h ? ncur_h[b] : 0;
        OV2_REQUIRE(n >= 0 && (size_t)n <= nm && (n == 0 || cur_xy_h), OV2_EINVAL, "bad current keypoints");
        nh[b] = n;
        if (n) memcpy(t->h_det + o_cur + 8 * nm * b, cur_xy_h + 2 * nm * b, 8 * (size_t)n);
    }
    if (out_cap > t->out_cap) {                                          // grow-only result mirrors
        OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (t->d_out) (void)hipFree(t->d_out);
        if (t->h_out) (void)hipHostFree(t->h_out);
        t->d_out = nullptr; t->h_out = nullptr; t->out_cap = 0;
        OV2_HIP_CHECK(hipMalloc((void **)&t->d_out, 8 * (size_t)out_cap * t->batch));
        OV2_HIP_CHECK(hipHostMalloc((void **)&t->h_out, 8 * (size_t)out_cap * t->batch, hipHostMallocDefault));
        t->out_cap = out_cap;
    }
    OV2_HIP_CHECK(hipMemcpyAsync(t->d_det, t->h_det, o_cur + 8 * nm * (size_t)n_active, hipMemcpyHostToDevice, ctx->stream));
    const ov2_pyr q = prefix_of(t->pyr[t->cur], n_active);
    const float *cur_d = (const float *)(t->d_det + o_cur);
    const int *ncur_d = (const int *)t->d_det;
    int rc;
    if (mode == 0) rc = ov2_detect_grid_fast_batch_d(ctx, &q, cell, cur_d, (int)nm, ncur_d, fast_th_inout, mask_mode, do_subpix, t->d_out, t->out_cap, out_n_h);
    else rc = ov2_detect_singlescale_batch_d(ctx, &q, cell, cur_d, (int)nm, ncur_d, roi, quality_inout, do_subpix, t->d_out, t->out_cap, out_n_h);
    if (rc != OV2_OK) return rc;
    OV2_HIP_CHECK(hipMemcpyAsync(t->h_out, t->d_out, 8 * (size_t)t->out_cap * n_active, hipMemcpyDeviceToHost, ctx->stream));
    OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < n_active; b++)
        if (out_n_h[b] > 0) memcpy(out_xy_h + 2 * (size_t)out_cap * b, t->h_out + 8 * (size_t)t->out_cap * b, 8 * (size_t)out_n_h[b]);
    return OV2_OK;
}

extern "C" {

int ov2_btracker_create(ov2_ctx *ctx, const ov2_tracker_config *cfg, int batch, ov2_btracker **out)
{
    OV2_REQUIRE(ctx && cfg && out, OV2_EINVAL, "NULL argument");
    *out = nullptr;
    OV2_REQUIRE(batch >= 1 && batch <= 65535, OV2_EINVAL, "batch out of range");
    OV2_REQUIRE(cfg->w > 0 && cfg->h > 0 && cfg->n_max > 0 && cfg->nklt_pyr_lvl >= 0 && cfg->prior_pyr_lvl >= 0, OV2_EINVAL, "bad tracker geometry");
    OV2_REQUIRE(!cfg->use_clahe || (cfg->tiles_x > 0 && cfg->tiles_y > 0 && cfg->tiles_x <= cfg->w && cfg->tiles_y <= cfg->h), OV2_EINVAL, "bad CLAHE tiles");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    ov2_btracker *t = new (std::nothrow) ov2_btracker();
    OV2_REQUIRE(t != nullptr, OV2_ENOMEM, "out of host memory");
    t->ctx = ctx; t->cfg = *cfg; t->batch = batch;
    t->last_n.assign((size_t)batch, 0);
    int rc = OV2_OK;
    for (int i = 0; i < BT_SETS && rc == OV2_OK; i++) {
        rc = ov2_pyr_create(ctx, cfg->w, cfg->h, cfg->win, cfg->nklt_pyr_lvl, batch, &t->pyr[i]);
        for (int b = 0; b < batch && rc == OV2_OK; b++) {
            ov2_pyr *v = nullptr;
            rc = ov2_pyr_item_view(t->pyr[i], b, &v);
            if (rc == OV2_OK) t->view[i].push_back(v);
        }
    }
    if (rc != OV2_OK) { btracker_free(t); return rc; }
    t->img_pitch = up16((size_t)cfg->w);
    t->img_bytes = up256(t->img_pitch * (size_t)cfg->h);
    const size_t nm = (size_t)cfg->n_max * (size_t)batch;
    t->o_n = 0; t->o_kps = up256(4 * (size_t)batch); t->o_pri = t->o_kps + 8 * nm; t->o_flg = t->o_pri + 8 * nm; t->in_bytes = up256(t->o_flg + nm);
    t->o_out = t->in_bytes; t->o_st = t->o_out + 8 * nm; t->o_unpx = up256(t->o_st + nm); t->o_bv = up256(t->o_unpx + 8 * nm); t->blk_bytes = t->o_bv + 24 * nm;
    t->det_in_bytes = up256(4 * (size_t)batch) + 8 * nm;
    hipError_t e = hipSuccess;
    for (int i = 0; i < BT_IMG_SETS && e == hipSuccess; i++) {
        e = hipHostMalloc((void **)&t->himg[i], t->img_bytes * batch + 256, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&t->dimg[i], t->img_bytes * batch + 256);
        if (e == hipSuccess) e = hipMemsetAsync(t->dimg[i], 0, t->img_bytes * batch + 256, ctx->stream);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&t->up_ev[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&t->used_ev[i], hipEventDisableTiming);
        if (e == hipSuccess) memset(t->himg[i], 0, t->img_bytes * batch + 256);
    }
    {   // the look-ahead streams inherit the priority of the context's stream
        int prio = 0;
        if (e == hipSuccess) e = hipStreamGetPriority(ctx->stream, &prio);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&t->cs, hipStreamNonBlocking, prio);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&t->ps, hipStreamNonBlocking, prio);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&t->lut, (size_t)batch * (cfg->use_clahe ? (size_t)cfg->tiles_x * cfg->tiles_y : 1) * 256 + 256);
    if (e == hipSuccess) e = hipHostMalloc((void **)&t->hblk, t->blk_bytes, hipHostMallocMapped);
    if (e == hipSuccess) e = hipMalloc((void **)&t->dblk, t->blk_bytes);
    if (e == hipSuccess) e = hipMemsetAsync(t->dblk, 0, t->blk_bytes, ctx->stream);
    if (e == hipSuccess) e = hipHostMalloc((void **)&t->h_det, t->det_in_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&t->d_det, t->det_in_bytes);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ov2_set_error("ov2_btracker_create: %s", hipGetErrorString(e)); btracker_free(t); return OV2_ENOMEM; }
    memset(t->hblk, 0, t->blk_bytes);
    memset(t->h_det, 0, t->det_in_bytes);
    t->side.device = ctx->device; t->side.stream = t->ps; t->side.owns_stream = false;
    t->kblk = t->dblk;
    {   // the kernels read / write the pinned block through its device alias (no staging copies); the device block is the fallback
        void *alias = nullptr;
        if (hipHostGetDevicePointer(&alias, t->hblk, 0) == hipSuccess && alias) { t->kblk = (uint8_t *)alias; t->zero_copy = true; }
        else (void)hipGetLastError();
    }
    *out = t;
    return OV2_OK;
}

void ov2_btracker_destroy(ov2_btracker *t) { btracker_free(t); }
int ov2_btracker_batch(const ov2_btracker *t) { return t ? t->batch : 0; }
int ov2_btracker_frames(const ov2_btracker *t) { return t ? (int)t->frames : 0; }

uint8_t *ov2_btracker_image_buffer(ov2_btracker *t, int which, int item, int *stride)
{
    if (!t || which < 0 || which >= BT_IMG_SETS || item < 0 || item >= t->batch) return nullptr;
    if (stride) *stride = (int)t->img_pitch;
    return t->himg[which] + (size_t)item * t->img_bytes;
}

int ov2_btracker_upload(ov2_btracker *t, int which, int n_active)
{
    OV2_REQUIRE(t && which >= 0 && which < BT_IMG_SETS, OV2_EINVAL, "bad staging set");
    OV2_REQUIRE(n_active >= 1 && n_active <= t->batch, OV2_EINVAL, "n_active out of range");
    OV2_HIP_CHECK(hipSetDevice(t->ctx->device));
    OV2_HIP_CHECK(hipStreamWaitEvent(t->cs, t->used_ev[which], 0));      // the step that read dimg[which] last (no-op before the first record)
    OV2_HIP_CHECK(hipMemcpyAsync(t->dimg[which], t->himg[which], (size_t)n_active * t->img_bytes, hipMemcpyHostToDevice, t->cs));
    OV2_HIP_CHECK(hipEventRecord(t->up_ev[which], t->cs));
    t->up_n[which] = n_active;
    return OV2_OK;
}

int ov2_btracker_prepare(ov2_btracker *t, int which, int n_active)
{
    OV2_REQUIRE(t && which >= 0 && which < BT_IMG_SETS, OV2_EINVAL, "bad staging set");
    OV2_REQUIRE(n_active >= 1 && n_active <= t->batch, OV2_EINVAL, "n_active out of range");
    OV2_REQUIRE(t->prepq.size() < 2, OV2_EINVAL, "two prepared frames are already waiting for their ov2_btracker_track_frame");
    OV2_HIP_CHECK(hipSetDevice(t->ctx->device));
    // the frames: uploaded ahead (wait for the copy stream) or copied here, on the prep stream
    if (t->up_n[which] >= n_active) OV2_HIP_CHECK(hipStreamWaitEvent(t->ps, t->up_ev[which], 0));
    else {
        if (t->up_n[which]) OV2_HIP_CHECK(hipStreamWaitEvent(t->ps, t->up_ev[which], 0));
        OV2_HIP_CHECK(hipStreamWaitEvent(t->ps, t->used_ev[which], 0));
        OV2_HIP_CHECK(hipMemcpyAsync(t->dimg[which], t->himg[which], (size_t)n_active * t->img_bytes, hipMemcpyHostToDevice, t->ps));
    }
    t->up_n[which] = 0;
    // target: the pyramid set of frame pre_count.  Its last readers on the context's stream -- the tracking kernels of the step that
    // had it as prev_pyr_ -- finished before that step's synchronisation (at most two frames are ever ahead of the tracked one);
    // consumers on other contexts are the caller's to wait for (ov2_btracker_pyramid_sets)
    const int set = (int)(t->pre_count % BT_SETS);
    t->side.clahe_strips = t->ctx->clahe_strips;
    int rc = enqueue_preprocess(t, &t->side, t->pyr[set], which, n_active);
    if (rc != OV2_OK) return rc;
    rc = ov2_pyr_mark_ready(&t->side, t->pyr[set]);
    if (rc != OV2_OK) return rc;
    t->prepq.push_back({which, n_active, set});
    t->pre_count++;
    return OV2_OK;
}

int ov2_btracker_set_calibration(ov2_btracker *t, int model, const double K[4], const double *D, int nD, const double iK[9])
{
    OV2_REQUIRE(t, OV2_EINVAL, "NULL tracker");
    KpCalib c;
    const int rc = ov2_kp_calib(model, K, D, nD, iK, c);
    if (rc != OV2_OK) return rc;
    t->calib = c; t->has_calib = true;
    return OV2_OK;
}

int ov2_btracker_track_frame_begin(ov2_btracker *t, int n_active, const uint8_t *const *img_h, int stride, const float *kps_xy_h,
                                   const float *prior_xy_h, const uint8_t *has_prior_h, const int *n_h, int klt_use_prior)
{
    OV2_REQUIRE(t && img_h, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(!t->pend.on, OV2_EINVAL, "ov2_btracker_track_frame_begin: the previous step has not been finished (ov2_btracker_track_frame_end)");
    OV2_REQUIRE(n_active >= 1 && n_active <= t->batch, OV2_EINVAL, "n_active out of range");
    OV2_REQUIRE(stride >= t->cfg.w, OV2_EINVAL, "stride < width");
    const size_t nm = (size_t)t->cfg.n_max;
    int n_total = 0;
    for (int b = 0; b < n_active; b++) {
        OV2_REQUIRE(img_h[b] != nullptr, OV2_EINVAL, "NULL frame of an active item");
        const int n = n_h ? n_h[b] : 0;
        OV2_REQUIRE(n >= 0 && (size_t)n <= nm, OV2_EINVAL, "an item carries more keypoints than cfg.n_max slots");
        n_total += n;
    }
    OV2_REQUIRE(n_total == 0 || (kps_xy_h && prior_xy_h && n_h), OV2_EINVAL, "NULL point buffer");
    for (int b = 0; b < t->batch; b++) t->last_n[(size_t)b] = 0;
    ov2_ctx *ctx = t->ctx;
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    int which = 0;
    bool prepared = false;
    int rc = stage_and_upload(t, n_active, img_h, stride, &which, &prepared);
    if (rc != OV2_OK) return rc;
    // preprocessImage: already under way on the prep stream (the context's stream waits for the pyramids' event), or in order here
    const int old_cur = t->cur;
    // the bookkeeping (prepq / pre_count) is committed only when the whole step has been enqueued: a failure further down leaves
    // the tracker where it was, and the same frames can be passed again
    auto preprocess = [&]() -> int {
        if (prepared) return ov2_pyr_wait_ready(ctx, t->pyr[t->cur]);
        const int rcp = enqueue_preprocess(t, ctx, t->pyr[t->cur], which, n_active);
        if (rcp != OV2_OK) return rcp;
        return ov2_pyr_mark_ready(ctx, t->pyr[t->cur]);
    };
    auto commit = [&]() { if (prepared) t->prepq.pop_front(); else t->pre_count++; t->frames++; };
    t->cur = (int)(t->frames % BT_SETS);                                 // prev_pyr_.swap(cur_pyr_)  (:1169): frame k lives in set k % BT_SETS
    ov2_btracker::Pending &P = t->pend;
    P.n_active = n_active; P.use_prior = klt_use_prior; P.kps = kps_xy_h; P.has_prior = has_prior_h;
    P.n.assign((size_t)n_active, 0);
    if (n_h) for (int b = 0; b < n_active; b++) P.n[(size_t)b] = n_h[b];
    if (t->frames == 0 || n_total == 0) {
        // first frame (trackMono returns right after preprocessImage) or nothing to track anywhere
        rc = preprocess();
        if (rc != OV2_OK) { t->cur = old_cur; return rc; }
        commit();
        P.tracked = false; P.on = true;
        return OV2_OK;
    }
    stage_points(t, n_active, kps_xy_h, prior_xy_h, has_prior_h, n_h, klt_use_prior);
    rc = preprocess();
    if (rc == OV2_OK) rc = enqueue_klt(t, t->pyr[t->prev()], t->pyr[t->cur], n_active);
    if (rc != OV2_OK) { t->cur = old_cur; return rc; }
    commit();
    P.tracked = true; P.on = true;
    return OV2_OK;
}

int ov2_btracker_track_frame_end(ov2_btracker *t, float *out_xy_h, uint8_t *status_h, int *p3p_req)
{
    OV2_REQUIRE(t, OV2_EINVAL, "NULL tracker");
    ov2_btracker::Pending &P = t->pend;
    OV2_REQUIRE(P.on, OV2_EINVAL, "ov2_btracker_track_frame_end without ov2_btracker_track_frame_begin");
    P.on = false;
    const size_t nm = (size_t)t->cfg.n_max;
    ov2_ctx *ctx = t->ctx;
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    if (p3p_req) for (int b = 0; b < P.n_active; b++) p3p_req[b] = 0;
    if (!P.tracked) {
        if (status_h) for (int b = 0; b < P.n_active; b++) if (P.n[(size_t)b] > 0) memset(status_h + nm * b, 0, (size_t)P.n[(size_t)b]);
        return ov2_ctx_sync(ctx);
    }
    OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    OV2_REQUIRE(out_xy_h && status_h, OV2_EINVAL, "NULL result buffer");
    for (int b = 0; b < P.n_active; b++) {
        const size_t n = (size_t)P.n[(size_t)b], o = nm * (size_t)b;
        if (!n) continue;
        memcpy(out_xy_h + 2 * o, t->hblk + t->o_out + 8 * o, 8 * n);
        memcpy(status_h + o, t->hblk + t->o_st + o, n);
        const int rc = apply_p3p_rule(t, b, P.kps + 2 * o, P.has_prior ? P.has_prior + o : nullptr, P.use_prior, (int)n, out_xy_h + 2 * o,
                                      status_h + o, p3p_req ? p3p_req + b : nullptr);
        if (rc != OV2_OK) return rc;
        if (t->has_calib) t->last_n[(size_t)b] = (int)n;
    }
    return OV2_OK;
}

int ov2_btracker_track_frame(ov2_btracker *t, int n_active, const uint8_t *const *img_h, int stride, const float *kps_xy_h,
                             const float *prior_xy_h, const uint8_t *has_prior_h, const int *n_h, int klt_use_prior,
                             float *out_xy_h, uint8_t *status_h, int *p3p_req)
{
    if (t && n_active >= 1 && n_active <= t->batch && n_h) {              // (the one-call form checks its result buffers before anything is enqueued)
        int n_total = 0;
        for (int b = 0; b < n_active; b++) n_total += n_h[b] > 0 ? n_h[b] : 0;
        OV2_REQUIRE(n_total == 0 || (out_xy_h && status_h), OV2_EINVAL, "NULL point buffer");
    }
    const int rc = ov2_btracker_track_frame_begin(t, n_active, img_h, stride, kps_xy_h, prior_xy_h, has_prior_h, n_h, klt_use_prior);
    if (rc != OV2_OK) return rc;
    return ov2_btracker_track_frame_end(t, out_xy_h, status_h, p3p_req);
}

int ov2_btracker_last_keypoints(const ov2_btracker *t, int item, int n, float *unpx_xy_h, double *bv_xyz_h)
{
    OV2_REQUIRE(t && t->has_calib, OV2_EINVAL, "no calibration set on this tracker");
    OV2_REQUIRE(item >= 0 && item < t->batch, OV2_EINVAL, "batch item out of range");
    OV2_REQUIRE(n >= 0 && n <= t->last_n[(size_t)item], OV2_EINVAL, "more keypoints than the last tracking call returned for this item");
    const size_t o = (size_t)item * t->cfg.n_max;
    if (unpx_xy_h) memcpy(unpx_xy_h, t->hblk + t->o_unpx + 8 * o, 8 * (size_t)n);
    if (bv_xyz_h) memcpy(bv_xyz_h, t->hblk + t->o_bv + 24 * o, 24 * (size_t)n);
    return OV2_OK;
}

int ov2_btracker_detect_singlescale(ov2_btracker *t, int n_active, int cell, const float *cur_xy_h, const int *ncur_h, const int roi[4],
                                    double *quality_inout, int do_subpix, float *out_xy_h, int out_cap, int *out_n_h)
{
    OV2_REQUIRE(quality_inout != nullptr && roi != nullptr, OV2_EINVAL, "quality_inout / roi == NULL");
    return detect_common(t, 1, n_active, cell, cur_xy_h, ncur_h, roi, quality_inout, nullptr, 0, do_subpix, out_xy_h, out_cap, out_n_h);
}

int ov2_btracker_detect_grid_fast(ov2_btracker *t, int n_active, int cell, const float *cur_xy_h, const int *ncur_h, int *fast_th_inout,
                                  int mask_mode, int do_subpix, float *out_xy_h, int out_cap, int *out_n_h)
{
    OV2_REQUIRE(fast_th_inout != nullptr, OV2_EINVAL, "fast_th_inout == NULL");
    return detect_common(t, 0, n_active, cell, cur_xy_h, ncur_h, nullptr, nullptr, fast_th_inout, mask_mode, do_subpix, out_xy_h, out_cap, out_n_h);
}

const ov2_pyr *ov2_btracker_cur_pyr(const ov2_btracker *t) { return t ? t->pyr[t->cur] : nullptr; }
const ov2_pyr *ov2_btracker_prev_pyr(const ov2_btracker *t) { return t ? t->pyr[t->prev()] : nullptr; }
int ov2_btracker_pyramid_sets(const ov2_btracker *t) { return t ? BT_SETS : 0; }
const ov2_pyr *ov2_btracker_cur_item(const ov2_btracker *t, int item) { return t && item >= 0 && item < t->batch ? t->view[t->cur][(size_t)item] : nullptr; }
const ov2_pyr *ov2_btracker_prev_item(const ov2_btracker *t, int item) { return t && item >= 0 && item < t->batch ? t->view[t->prev()][(size_t)item] : nullptr; }

} // extern "C"
