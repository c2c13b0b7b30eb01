// fe_capi.cpp -- C entry points over the REFERENCE'S OWN FeatureExtractor (/root/reference/src/feature_extractor.cpp, compiled from where it
// lies against the stand-in OpenCV of oracle/ref/standin_cv, whose algorithms are the oracle's restatements): detectSingleScale and
// detectGridFAST as the reference wrote them -- the grid walk, the occupancy table, the in-image and roi tests, the two masked arg-max passes
// and the secondary top-up, the FAST response sort and its `>= 20` gate, both threshold adaptations.  tests/test_reference_factors.py compares
// the keypoint lists and the adapted thresholds with oracle/detect.c.  TEST INFRASTRUCTURE ONLY.
#include "feature_extractor.hpp"

#include <opencv2/imgproc.hpp>

extern cv::Ptr<cv::FastFeatureDetector> pfast_;              // file-scope state of feature_extractor.cpp (:66): the detector outlives the object

extern "C" {

static std::vector<cv::Point2f> to_points(const float *xy, int n)
{
    std::vector<cv::Point2f> v((size_t)n);
    for (int i = 0; i < n; i++) v[(size_t)i] = cv::Point2f(xy[2 * i], xy[2 * i + 1]);
    return v;
}

// FeatureExtractor::detectSingleScale; quality_inout = dmaxquality_ before / after.  Returns the number of points (<= cap are written).
int ref_detect_singlescale(const uint8_t *img, int w, int h, int stride, int cell, const float *cur_xy, int ncur, const int roi[4],
                           double *quality_inout, float *out_xy, int cap)
{
    FeatureExtractor fe;
    fe.dmaxquality_ = *quality_inout;
    const std::vector<cv::Point2f> r = fe.detectSingleScale(cv::Mat::wrap(img, h, w, (size_t)stride), cell, to_points(cur_xy, ncur), cv::Rect(roi[0], roi[1], roi[2], roi[3]));
    *quality_inout = fe.dmaxquality_;
    for (size_t i = 0; i < r.size() && (int)i < cap; i++) { out_xy[2 * i] = r[i].x; out_xy[2 * i + 1] = r[i].y; }
    return (int)r.size();
}

// FeatureExtractor::detectGridFAST with a fresh FAST detector at *fast_th_inout (the reference creates it on first use and keeps it)
int ref_detect_grid_fast(const uint8_t *img, int w, int h, int stride, int cell, const float *cur_xy, int ncur, int *fast_th_inout, float *out_xy, int cap)
{
    pfast_ = nullptr;
    FeatureExtractor fe;
    fe.nfast_th_ = *fast_th_inout;
    const std::vector<cv::Point2f> r = fe.detectGridFAST(cv::Mat::wrap(img, h, w, (size_t)stride), cell, to_points(cur_xy, ncur), cv::Rect(0, 0, w, h));
    *fast_th_inout = fe.nfast_th_;
    for (size_t i = 0; i < r.size() && (int)i < cap; i++) { out_xy[2 * i] = r[i].x; out_xy[2 * i + 1] = r[i].y; }
    return (int)r.size();
}
}
