// feature_extractor.hpp -- C++ adapter with the signatures of the reference's FeatureExtractor grid
// detectors (/root/reference/include/feature_extractor.hpp:40-46, src/feature_extractor.cpp:288-570).
// Holds the same adaptive state (nfast_th_, dmaxquality_).  Errors degrade to an empty vector, like
// the reference's empty-image path (:291-294, :446-449).
#pragma once
#include "ov2_types.hpp"

namespace ov2 {

class FeatureExtractor {
public:
    // reference: FeatureExtractor(size_t nmaxpts, size_t nmaxdist, double dmaxquality, int nfast_th)
    FeatureExtractor(size_t nmaxpts, size_t nmaxdist, double dmaxquality, int nfast_th, int mask_mode = OV2_MASK_AS_EXECUTED)
        : nmaxpts_(nmaxpts), nmaxdist_(nmaxdist), dmaxquality_(dmaxquality), nfast_th_(nfast_th), mask_mode_(mask_mode) {}

    // reference: std::vector<cv::Point2f> detectGridFAST(const cv::Mat &im, const int ncellsize,
    //                const std::vector<cv::Point2f> &vcurkps, const cv::Rect &roi)
    std::vector<Point2f> detectGridFAST(Context &ctx, const Image8 &im, const int ncellsize,
                                        const std::vector<Point2f> &vcurkps, const Rect & /*roi: unused there too*/)
    {
        if (im.empty()) return std::vector<Point2f>();
        std::vector<Point2f> out((size_t)(im.cols / ncellsize) * (im.rows / ncellsize) + 1);
        int n = 0;
        const int rc = ov2_detect_grid_fast(ctx.get(), im.data, im.cols, im.rows, im.step, ncellsize,
                                            vcurkps.empty() ? nullptr : &vcurkps[0].x, (int)vcurkps.size(),
                                            &nfast_th_, mask_mode_, 1, &out[0].x, &n);
        out.resize(rc == OV2_OK ? (size_t)n : 0);
        return out;
    }

    // reference: std::vector<cv::Point2f> detectSingleScale(const cv::Mat &im, const int ncellsize,
    //                const std::vector<cv::Point2f> &vcurkps, const cv::Rect &roi)
    std::vector<Point2f> detectSingleScale(Context &ctx, const Image8 &im, const int ncellsize,
                                           const std::vector<Point2f> &vcurkps, const Rect &roi)
    {
        if (im.empty()) return std::vector<Point2f>();
        std::vector<Point2f> out(2 * (size_t)(im.cols / ncellsize) * (im.rows / ncellsize) + 1);
        int n = 0;
        const int r[4] = {roi.x, roi.y, roi.width, roi.height};
        const int rc = ov2_detect_singlescale(ctx.get(), im.data, im.cols, im.rows, im.step, ncellsize,
                                              vcurkps.empty() ? nullptr : &vcurkps[0].x, (int)vcurkps.size(),
                                              r, &dmaxquality_, 1, &out[0].x, &n);
        out.resize(rc == OV2_OK ? (size_t)n : 0);
        return out;
    }

    // the same detectors on level 0 of a device-resident pyramid (FrameTracker::curPyr(): the CLAHE'd cur_img_ of the keyframe,
    // src/map_manager.cpp:312-320) -- no image upload
    std::vector<Point2f> detectSingleScale(Context &ctx, const ov2_pyr *pyr, const int ncellsize, const std::vector<Point2f> &vcurkps, const Rect &roi)
    {
        int w = 0, h = 0;
        if (!pyr || ov2_pyr_level_size(pyr, 0, &w, &h) != OV2_OK) return std::vector<Point2f>();
        std::vector<Point2f> out(2 * (size_t)(w / ncellsize) * (h / ncellsize) + 1);
        int n = 0;
        const int r[4] = {roi.x, roi.y, roi.width, roi.height};
        const int rc = ov2_detect_singlescale_d(ctx.get(), pyr, 0, ncellsize, vcurkps.empty() ? nullptr : &vcurkps[0].x, (int)vcurkps.size(),
                                                r, &dmaxquality_, 1, &out[0].x, &n);
        out.resize(rc == OV2_OK ? (size_t)n : 0);
        return out;
    }
    std::vector<Point2f> detectGridFAST(Context &ctx, const ov2_pyr *pyr, const int ncellsize, const std::vector<Point2f> &vcurkps, const Rect &)
    {
        int w = 0, h = 0;
        if (!pyr || ov2_pyr_level_size(pyr, 0, &w, &h) != OV2_OK) return std::vector<Point2f>();
        std::vector<Point2f> out((size_t)(w / ncellsize) * (h / ncellsize) + 1);
        int n = 0;
        const int rc = ov2_detect_grid_fast_d(ctx.get(), pyr, 0, ncellsize, vcurkps.empty() ? nullptr : &vcurkps[0].x, (int)vcurkps.size(),
                                              &nfast_th_, mask_mode_, 1, &out[0].x, &n);
        out.resize(rc == OV2_OK ? (size_t)n : 0);
        return out;
    }

    // Offline batch mode: detectSingleScale / detectGridFAST on EVERY batch item of a pyramid in one call
    // (ov2_detect_singlescale_batch_d / ov2_detect_grid_fast_batch_d).  cur_xy_d / ncur_d / out_xy_d are device buffers
    // (batch x cur_cap points, batch counts, batch x out_cap points); vquality / vfast_th hold one dmaxquality_ / nfast_th_ per
    // sequence and are updated like the members; vout_n receives the number of points written per item.  Returns the C ABI code.
    static int detectSingleScaleBatch(Context &ctx, const ov2_pyr *pyr, int ncellsize, const float *cur_xy_d, int cur_cap, const int *ncur_d,
                                      const Rect &roi, std::vector<double> &vquality, float *out_xy_d, int out_cap, std::vector<int> &vout_n)
    {
        const int r[4] = {roi.x, roi.y, roi.width, roi.height};
        vout_n.assign(vquality.size(), 0);
        if (vquality.empty()) return OV2_EINVAL;
        return ov2_detect_singlescale_batch_d(ctx.get(), pyr, ncellsize, cur_xy_d, cur_cap, ncur_d, r, vquality.data(), 1, out_xy_d, out_cap, vout_n.data());
    }
    static int detectGridFASTBatch(Context &ctx, const ov2_pyr *pyr, int ncellsize, const float *cur_xy_d, int cur_cap, const int *ncur_d,
                                   std::vector<int> &vfast_th, int mask_mode, float *out_xy_d, int out_cap, std::vector<int> &vout_n)
    {
        vout_n.assign(vfast_th.size(), 0);
        if (vfast_th.empty()) return OV2_EINVAL;
        return ov2_detect_grid_fast_batch_d(ctx.get(), pyr, ncellsize, cur_xy_d, cur_cap, ncur_d, vfast_th.data(), mask_mode, 1, out_xy_d, out_cap, vout_n.data());
    }

    size_t nmaxpts_, nmaxdist_;
    double dmaxquality_;       // feature_extractor.hpp:50
    int nfast_th_;             // feature_extractor.hpp:52
    int mask_mode_;
};

}  // namespace ov2
