// feature_tracker.hpp -- C++ adapter with the signature of the reference's FeatureTracker
// (/root/reference/include/feature_tracker.hpp:36-56, src/feature_tracker.cpp:35-137), forwarding to
// the C ABI.  Same argument meaning, same in/out conventions, same failure behaviour: any library
// error degrades to "nothing tracked" (all status false), never an exception.
#pragma once
#include "ov2_types.hpp"

namespace ov2 {

class FeatureTracker {
public:
    // reference: FeatureTracker(int nmax_iter, float fmax_px_precision, cv::Ptr<cv::CLAHE> pclahe)
    FeatureTracker(int nmax_iter, float fmax_px_precision) : nmax_iter_(nmax_iter), fmax_px_precision_(fmax_px_precision) {}

    // reference: void fbKltTracking(const std::vector<cv::Mat> &vprevpyr, const std::vector<cv::Mat> &vcurpyr,
    //                int nwinsize, int nbpyrlvl, float ferr, float fmax_fbklt_dist, std::vector<cv::Point2f> &vkps,
    //                std::vector<cv::Point2f> &vpriorkps, std::vector<bool> &vkpstatus) const
    // `ctx` is the calling thread's context (the method is const and is called concurrently from the
    // SLAM thread and the mapper thread: src/visual_front_end.cpp:196 / src/map_manager.cpp:510).
    void fbKltTracking(Context &ctx, const Pyramid &vprevpyr, const Pyramid &vcurpyr, int nwinsize, int nbpyrlvl,
                       float ferr, float fmax_fbklt_dist, std::vector<Point2f> &vkps, std::vector<Point2f> &vpriorkps,
                       std::vector<bool> &vkpstatus) const
    {
        fbKltTracking(ctx, vprevpyr.get(), vcurpyr.get(), nwinsize, nbpyrlvl, ferr, fmax_fbklt_dist, vkps, vpriorkps, vkpstatus);
    }
    // the same on raw handles (e.g. prev = SlamGpu::kf_front, cur = FrameTracker::curPyr(): VisualFrontEnd::kltTrackingFromKF,
    // src/visual_front_end.cpp:363, :409)
    void fbKltTracking(Context &ctx, const ov2_pyr *vprevpyr, const ov2_pyr *vcurpyr, int nwinsize, int nbpyrlvl,
                       float ferr, float fmax_fbklt_dist, std::vector<Point2f> &vkps, std::vector<Point2f> &vpriorkps,
                       std::vector<bool> &vkpstatus) const
    {
        if (vkps.empty()) return;                                   // :43-46
        const size_t n = vkps.size();
        vkpstatus.reserve(vkpstatus.size() + n);
        std::vector<uint8_t> st(n, 0);
        std::vector<Point2f> priors(vpriorkps);
        const int rc = ov2_fb_klt(ctx.get(), vprevpyr, vcurpyr, nwinsize, nbpyrlvl, nmax_iter_, fmax_px_precision_,
                                  ferr, fmax_fbklt_dist, &vkps[0].x, &priors[0].x, (int)n, st.data(), nullptr);
        if (rc == OV2_OK) vpriorkps.swap(priors);
        for (size_t i = 0; i < n; i++) vkpstatus.push_back(rc == OV2_OK && st[i] != 0);
    }

    // reference: void getLineMinSAD(const cv::Mat &iml, const cv::Mat &imr, const cv::Point2f &pt, const int nwinsize,
    //                float &xprior, float &l1err, bool bgoleft) const                              (:138-206)
    // -- here for a whole vector of points on pyramid level `level` in one launch (the reference loops over the
    // keypoints and calls it per point, src/map_manager.cpp:421-439).  xprior[i] = -1 when nothing qualified.
    void getLineMinSAD(Context &ctx, const Pyramid &leftpyr, const Pyramid &rightpyr, int level, const std::vector<Point2f> &vpts,
                       int nwinsize, std::vector<float> &vxprior, std::vector<float> &vl1err, bool bgoleft) const
    {
        vxprior.assign(vpts.size(), -1.f);
        vl1err.assign(vpts.size(), 255.f);
        if (vpts.empty()) return;
        const int rc = ov2_line_min_sad(ctx.get(), leftpyr.get(), rightpyr.get(), level, nwinsize, bgoleft ? 1 : 0, &vpts[0].x,
                                        (int)vpts.size(), vxprior.data(), vl1err.data());
        if (rc != OV2_OK) vxprior.assign(vpts.size(), -1.f);       // degrade to "no prior", like an early return
    }

    // MapManager::stereoMatching's data path (src/map_manager.cpp:367-611) for the keypoints of a keyframe in ONE enqueue and ONE
    // synchronisation (ov2_stereo_match): SAD priors on the coarsest level (rectified pairs), both fbKltTracking calls with the
    // retry of the failed 3-D-prior tracks, the epipolar gate.  vpriors3d[i] / vhasprior3d[i]: the projected map point of
    // keypoint i where one exists (:402-413, :468-480).  On return vstereo_ok[i] decides updateKeypointStereo(id, vrightkps[i]).
    // rmodel / rK / rD: the RIGHT camera (OV2_CAM_PINHOLE / OV2_CAM_FISHEYE, fx fy cx cy, distortion).
    void stereoMatching(Context &ctx, const ov2_pyr *leftpyr, const ov2_pyr *rightpyr, int nklt_win_size, int nklt_pyr_lvl, float nklt_err,
                        float fmax_fbklt_dist, bool rect, const double *Frl, int rmodel, const double rK[4], const std::vector<double> &rD,
                        const std::vector<Point2f> &vleftkps, const std::vector<Point2f> &vleftunpx, const std::vector<Point2f> &vpriors3d,
                        const std::vector<uint8_t> &vhasprior3d, std::vector<Point2f> &vrightkps, std::vector<bool> &vstereo_ok) const
    {
        const size_t n = vleftkps.size();
        vrightkps.assign(n, Point2f());
        vstereo_ok.assign(n, false);
        if (n == 0) return;
        std::vector<uint8_t> ok(n, 0);
        const bool pri = vpriors3d.size() == n && vhasprior3d.size() == n;
        const int rc = ov2_stereo_match(ctx.get(), leftpyr, rightpyr, nklt_win_size, nklt_pyr_lvl, nmax_iter_, fmax_px_precision_, nklt_err,
                                        fmax_fbklt_dist, rect ? 1 : 0, Frl, rmodel, rK, rD.empty() ? nullptr : rD.data(), (int)rD.size(),
                                        &vleftkps[0].x, &vleftunpx[0].x, pri ? &vpriors3d[0].x : nullptr, pri ? vhasprior3d.data() : nullptr,
                                        (int)n, &vrightkps[0].x, ok.data());
        if (rc != OV2_OK) return;                                   // degrade to "no stereo observation"
        for (size_t i = 0; i < n; i++) vstereo_ok[i] = ok[i] != 0;
    }

    // reference: bool inBorder(const cv::Point2f &pt, const cv::Mat &im) const  (:216-221)
    bool inBorder(const Point2f &pt, int cols, int rows) const
    {
        const float BORDER_SIZE = 1.f;
        return BORDER_SIZE <= pt.x && pt.x < cols - BORDER_SIZE && BORDER_SIZE <= pt.y && pt.y < rows - BORDER_SIZE;
    }

private:
    int nmax_iter_;
    float fmax_px_precision_;
};

}  // namespace ov2
