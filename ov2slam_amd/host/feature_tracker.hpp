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
        if (vkps.empty()) return;                                   // :43-46
        const size_t n = vkps.size();
        vkpstatus.reserve(vkpstatus.size() + n);
        std::vector<uint8_t> st(n, 0);
        std::vector<Point2f> priors(vpriorkps);
        const int rc = ov2_fb_klt(ctx.get(), vprevpyr.get(), vcurpyr.get(), nwinsize, nbpyrlvl, nmax_iter_, fmax_px_precision_,
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
