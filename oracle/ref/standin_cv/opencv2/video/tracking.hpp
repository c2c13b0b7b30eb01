// Stand-in for <opencv2/video/tracking.hpp> (see ../core.hpp): cv::calcOpticalFlowPyrLK on pyramids = the oracle's orc_lk_track
#pragma once
#include "../core.hpp"
extern "C" int orc_lk_track(const void *prev, const void *next, const float *prev_xy, float *next_xy, int n, uint8_t *status, float *err,
                            int win, int max_level, int max_count, double epsilon, int flags, double min_eig_threshold, int *iters_out, int nthreads);
namespace cv {
enum { OPTFLOW_USE_INITIAL_FLOW = 4, OPTFLOW_LK_GET_MIN_EIGENVALS = 8 };
inline void calcOpticalFlowPyrLK(const std::vector<Mat> &prevPyr, const std::vector<Mat> &nextPyr, const std::vector<Point2f> &prevPts,
                                 std::vector<Point2f> &nextPts, std::vector<uchar> &status, std::vector<float> &err, Size winSize, int maxLevel,
                                 TermCriteria criteria, int flags, double minEigThreshold = 1e-4)
{
    const int n = (int)prevPts.size();
    status.assign((size_t)n, 0); err.assign((size_t)n, 0.f);
    if (!(flags & OPTFLOW_USE_INITIAL_FLOW)) nextPts = prevPts;
    nextPts.resize((size_t)n);
    static_assert(sizeof(Point2f) == 8, "points are float pairs");
    orc_lk_track(prevPyr.at(0).orc_pyr_handle, nextPyr.at(0).orc_pyr_handle, n ? &prevPts[0].x : nullptr, n ? &nextPts[0].x : nullptr, n,
                 status.data(), err.data(), winSize.width, maxLevel, criteria.maxCount, criteria.epsilon, flags, minEigThreshold, nullptr, 1);
}
}   // namespace cv
