// visual_front_end.hpp -- C++ adapter for the GPU half of VisualFrontEnd (/root/reference/src/visual_front_end.cpp):
// preprocessImage (:1143-1177) + kltTracking (:132-275) on ov2_tracker_* -- ONE enqueue, ONE synchronisation per frame.
// The reference's members prev_pyr_ / cur_pyr_ live inside the tracker object; kltTracking's two fbKltTracking calls, the
// retry of lost prior tracks and the bp3preq_ rule are reproduced by the library (include/ov2slam_hip.h).
#pragma once
#include "ov2_types.hpp"

namespace ov2 {

class FrameTracker {
public:
    // SlamParams fields of the same names (slam_params.hpp): nklt_win_size_, nklt_pyr_lvl_, nmax_iter_, fmax_px_precision_,
    // nklt_err_, fmax_fbklt_dist_, use_clahe_, fclahe_val_, nbmaxkps_
    FrameTracker(Context &ctx, int img_w, int img_h, int nklt_win_size, int nklt_pyr_lvl, int nmax_iter, float fmax_px_precision,
                 float nklt_err, float fmax_fbklt_dist, bool use_clahe, double fclahe_val, int nbmaxkps, bool use_graph = true)
    {
        ov2_tracker_config c{};
        c.w = img_w; c.h = img_h; c.win = nklt_win_size; c.nklt_pyr_lvl = nklt_pyr_lvl; c.prior_pyr_lvl = 1;
        c.max_iter = nmax_iter; c.eps = fmax_px_precision; c.err_th = nklt_err; c.fb_dist = fmax_fbklt_dist;
        c.use_clahe = use_clahe ? 1 : 0; c.clahe_clip = fclahe_val; c.tiles_x = img_w / 50; c.tiles_y = img_h / 50;   // ov2slam.cpp:85-89
        // headroom: a frame can carry more than nbmaxkps_ keypoints between keyframes (extractKeypoints tops up cells that several
        // tracked keypoints share, map_manager.cpp:74 prunes at the next keyframe only); beyond 2 x the library chunks (same results)
        c.n_max = 2 * nbmaxkps; c.use_graph = use_graph ? 1 : 0;
        if (ov2_tracker_create(ctx.get(), &c, &t_) != OV2_OK) throw std::runtime_error(std::string("ov2_tracker_create: ") + ov2_last_error());
    }
    ~FrameTracker() { ov2_tracker_destroy(t_); }              // destroy before the Context it was created on
    FrameTracker(const FrameTracker &) = delete;
    FrameTracker &operator=(const FrameTracker &) = delete;

    // pinned staging image: let the image source (ROS callback / decoder) write here to skip the host-side copy
    uint8_t *imageBuffer(int *stride) { return ov2_tracker_image_buffer(t_, stride); }

    // VisualFrontEnd::preprocessImage(img_raw): asynchronous
    bool preprocessImage(const Image8 &img_raw) { return !img_raw.empty() && ov2_tracker_preprocess(t_, img_raw.data, img_raw.step) == OV2_OK; }

    // VisualFrontEnd::kltTracking on the keypoints of pcurframe_: vkps[i] = kp.px_, vpriors[i] = projected map point for
    // keypoints with a usable 3-D prior (vhasprior[i] = 1) and kp.px_ otherwise (:160-182).  On return vpriors[i] is the
    // tracked pixel (what the reference passes to updateKeypoint), vkpstatus[i] whether the observation survives (false ->
    // removeObsFromCurFrameById, :260), bp3preq mirrors bp3preq_ (:225-230).  The reference has no error channel, so an error
    // degrades to "nothing tracked" -- and is kept: lastError() / lastErrorMessage() say why (never silent).
    void kltTracking(const std::vector<Point2f> &vkps, std::vector<Point2f> &vpriors, const std::vector<uint8_t> &vhasprior,
                     bool klt_use_prior, std::vector<bool> &vkpstatus, bool &bp3preq)
    {
        const size_t n = vkps.size();
        vkpstatus.assign(n, false);
        bp3preq = false;
        last_rc_ = OV2_OK; last_msg_.clear();
        if (n == 0) return;
        if (vpriors.size() != n || (!vhasprior.empty() && vhasprior.size() != n)) {
            last_rc_ = OV2_EINVAL; last_msg_ = "kltTracking: vkps / vpriors / vhasprior differ in length"; return;
        }
        std::vector<Point2f> out(n);
        std::vector<uint8_t> st(n, 0);
        int p3p = 0;
        const int rc = ov2_tracker_klt(t_, &vkps[0].x, &vpriors[0].x, vhasprior.empty() ? nullptr : vhasprior.data(), (int)n,
                                       klt_use_prior ? 1 : 0, &out[0].x, st.data(), &p3p);
        if (rc != OV2_OK) { last_rc_ = rc; last_msg_ = ov2_last_error(); return; }
        vpriors.swap(out);
        for (size_t i = 0; i < n; i++) vkpstatus[i] = (st[i] & 1) != 0;
        bp3preq = p3p != 0;
    }

    // preprocessImage + kltTracking in one call (one graph launch): for callers whose priors do not depend on the new image
    // (the constant-velocity motion model of the reference does not: it uses the frame time and the previous poses)
    bool trackFrame(const Image8 &img_raw, const std::vector<Point2f> &vkps, std::vector<Point2f> &vpriors, const std::vector<uint8_t> &vhasprior,
                    bool klt_use_prior, std::vector<bool> &vkpstatus, bool &bp3preq)
    {
        const size_t n = vkps.size();
        vkpstatus.assign(n, false);
        bp3preq = false;
        last_rc_ = OV2_OK; last_msg_.clear();
        if (img_raw.empty()) return false;
        if (vpriors.size() != n || (!vhasprior.empty() && vhasprior.size() != n)) {
            last_rc_ = OV2_EINVAL; last_msg_ = "trackFrame: vkps / vpriors / vhasprior differ in length"; return false;
        }
        std::vector<Point2f> out(n);
        std::vector<uint8_t> st(n, 0);
        int p3p = 0;
        const int rc = ov2_tracker_track_frame(t_, img_raw.data, img_raw.step, n ? &vkps[0].x : nullptr, n ? &vpriors[0].x : nullptr,
                                               vhasprior.empty() ? nullptr : vhasprior.data(), (int)n, klt_use_prior ? 1 : 0,
                                               n ? &out[0].x : nullptr, n ? st.data() : nullptr, &p3p);
        if (rc != OV2_OK) { last_rc_ = rc; last_msg_ = ov2_last_error(); return false; }
        if (n) vpriors.swap(out);
        for (size_t i = 0; i < n; i++) vkpstatus[i] = (st[i] & 1) != 0;
        bp3preq = p3p != 0;
        return true;
    }

    // Frame::computeKeypoint (undistorted pixel + bearing, src/frame.cpp:246-254) for every tracked position inside the per-frame
    // enqueue: call once after construction with the LEFT camera's model (CameraCalibration: K_, Dcv_, iK_), then read the results
    // of the last kltTracking / trackFrame with lastKeypoints (row-major: unpx 2 floats, bv 3 doubles per keypoint)
    bool setCalibration(int model, const double K[4], const double *D, int nD, const double iK[9])
    {
        return ov2_tracker_set_calibration(t_, model, K, D, nD, iK) == OV2_OK;
    }
    bool lastKeypoints(size_t n, std::vector<Point2f> &vunpx, std::vector<double> &vbv) const
    {
        vunpx.resize(n); vbv.resize(3 * n);
        return n == 0 || ov2_tracker_last_keypoints(t_, (int)n, &vunpx[0].x, vbv.data()) == OV2_OK;
    }

    // cur_pyr_ / prev_pyr_ for createKeyframe, stereo matching and the device-resident detectors (valid until the next frame)
    const ov2_pyr *curPyr() const { return ov2_tracker_cur_pyr(t_); }
    const ov2_pyr *prevPyr() const { return ov2_tracker_prev_pyr(t_); }

    // why the last kltTracking / trackFrame tracked nothing (OV2_OK when it did not fail)
    int lastError() const { return last_rc_; }
    const std::string &lastErrorMessage() const { return last_msg_; }

private:
    ov2_tracker *t_ = nullptr;
    int last_rc_ = OV2_OK;
    std::string last_msg_;
};

}  // namespace ov2
