// Stand-in for <opencv2/core.hpp>: exactly what /root/reference/src/feature_tracker.cpp (+ include/feature_tracker.hpp) uses, so that the
// reference's own FeatureTracker compiles from where it lies (oracle/ref/Makefile).  The OpenCV ALGORITHMS behind it -- calcOpticalFlowPyrLK,
// getRectSubPix, norm -- are the oracle's restatements (oracle/frontend.c, oracle/stereo.c): what this pins is the reference's FIRST-PARTY
// code around them (level clamp, status / error / border filters, backward pass, forward-backward distance; the window shrinking, scan
// bounds and arg-min of getLineMinSAD), not OpenCV.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <vector>

typedef unsigned char uchar;                                 // (core/hal/interface.h puts it in the global namespace)
namespace cv {
using ::uchar;
struct Point2f {
    float x, y;
    Point2f() : x(0), y(0) {}
    Point2f(float a, float b) : x(a), y(b) {}
};
inline Point2f operator-(const Point2f &a, const Point2f &b) { return Point2f(a.x - b.x, a.y - b.y); }
inline double norm(const Point2f &p) { return std::sqrt((double)p.x * p.x + (double)p.y * p.y); }      // (core/types.hpp: norm(Point_<_Tp>))
struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
// an 8-bit single-channel image: a view of caller memory, or (getRectSubPix's output) its own buffer; pyramid levels also carry the
// oracle pyramid they belong to
struct Mat {
    int rows = 0, cols = 0;
    const uint8_t *data = nullptr; size_t step = 0;
    std::shared_ptr<std::vector<uint8_t>> own;
    const void *orc_pyr_handle = nullptr;
};
struct TermCriteria {
    enum { COUNT = 1, MAX_ITER = COUNT, EPS = 2 };
    int type, maxCount; double epsilon;
    TermCriteria(int t, int c, double e) : type(t), maxCount(c), epsilon(e) {}
};
template <class T> using Ptr = std::shared_ptr<T>;
class CLAHE;
enum { NORM_L1 = 2 };
inline double norm(const Mat &a, const Mat &b, int)                                                      // NORM_L1 of two u8 patches
{
    double s = 0;
    for (int i = 0; i < a.rows; i++) for (int j = 0; j < a.cols; j++) s += std::abs((int)a.data[i * a.step + j] - (int)b.data[i * b.step + j]);
    return s;
}
}   // namespace cv
