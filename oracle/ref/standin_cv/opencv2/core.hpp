// Stand-in for <opencv2/core.hpp>: exactly what /root/reference/src/feature_tracker.cpp and src/feature_extractor.cpp (+ their headers) use, so
// that the reference's own FeatureTracker / FeatureExtractor compile from where they lie (oracle/ref/Makefile).  The OpenCV ALGORITHMS behind
// them -- calcOpticalFlowPyrLK, getRectSubPix, norm, GaussianBlur + cornerMinEigenVal, FAST, circle, cornerSubPix, the keypoint mask filter --
// are the oracle's restatements (oracle/frontend.c, stereo.c, detect.c): what this pins is the reference's FIRST-PARTY code around them
// (grid walk, occupancy, region tests, thresholds and their adaptation, the sort, the top-up; level clamp, status / error / border filters,
// backward pass; window shrinking, scan bounds, arg-min), not OpenCV.  cv::Mat keeps OpenCV's sharing semantics (header copies and ROI views
// alias the buffer, clone() copies).  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <memory>
#include <vector>

typedef unsigned char uchar;                                 // (core/hal/interface.h puts it in the global namespace)
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_32FC1 5

namespace cv {
using ::uchar;
inline int cvRound(double v) { return (int)lrint(v); }
struct Point2f;
struct Point {
    int x, y;
    Point() : x(0), y(0) {}
    Point(int a, int b) : x(a), y(b) {}
    Point(const Point2f &p);                                  // saturate_cast<int>(float) = cvRound
};
struct Point2f {
    float x, y;
    Point2f() : x(0), y(0) {}
    Point2f(float a, float b) : x(a), y(b) {}
    Point2f(const Point &p) : x((float)p.x), y((float)p.y) {}
};
inline Point::Point(const Point2f &p) : x(cvRound(p.x)), y(cvRound(p.y)) {}
inline Point2f operator-(const Point2f &a, const Point2f &b) { return Point2f(a.x - b.x, a.y - b.y); }
inline bool operator==(const Point2f &a, const Point2f &b) { return a.x == b.x && a.y == b.y; }
inline double norm(const Point2f &p) { return std::sqrt((double)p.x * p.x + (double)p.y * p.y); }      // (core/types.hpp: norm(Point_<_Tp>))
struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
struct Rect { int x, y, width, height; Rect() : x(0), y(0), width(0), height(0) {} Rect(int a, int b, int w, int h) : x(a), y(b), width(w), height(h) {} };
struct Scalar { double v; Scalar(double a = 0) : v(a) {} };
struct Range { int start, end; Range(int a, int b) : start(a), end(b) {} };
inline void parallel_for_(const Range &r, const std::function<void(const Range &)> &f, double = -1.) { f(r); }      // (one stripe: the order of a single thread)

// 8-bit or 32-bit-float single-channel image.  A header copy or a ROI view shares the buffer; clone() does not.  Pyramid levels (the
// tracker's stand-in) also carry the oracle pyramid they belong to.
struct Mat {
    int rows = 0, cols = 0, type_ = CV_8U;
    uint8_t *data = nullptr; size_t step = 0;                 // (data of a view: first byte of the ROI)
    std::shared_ptr<std::vector<uint8_t>> own;                // the shared buffer (null: caller memory)
    uint8_t *whole = nullptr; int whole_rows = 0, whole_cols = 0;       // the parent image (Mat::locateROI)
    const void *orc_pyr_handle = nullptr;
    Mat() {}
    Mat(int r, int c, int type, const Scalar &s = Scalar(0)) { create(r, c, type); fill(s.v); }
    int elem() const { return type_ == CV_32F ? 4 : 1; }
    void create(int r, int c, int type)
    {
        rows = r; cols = c; type_ = type; step = (size_t)c * elem();
        own = std::make_shared<std::vector<uint8_t>>((size_t)r * step);
        data = own->data(); whole = data; whole_rows = r; whole_cols = c;
    }
    void fill(double v)
    {
        for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) { if (type_ == CV_32F) ((float *)(data + i * step))[j] = (float)v; else data[i * step + j] = (uint8_t)v; }
    }
    static Mat ones(int r, int c, int type) { return Mat(r, c, type, Scalar(1)); }
    static Mat wrap(const uint8_t *p, int r, int c, size_t stride)       // a u8 image in caller memory
    {
        Mat m; m.rows = r; m.cols = c; m.type_ = CV_8U; m.data = const_cast<uint8_t *>(p); m.step = stride; m.whole = m.data; m.whole_rows = r; m.whole_cols = c;
        return m;
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    void release() { *this = Mat(); }
    Mat clone() const
    {
        Mat m; m.create(rows, cols, type_);
        for (int i = 0; i < rows; i++) memcpy(m.data + i * m.step, data + i * step, (size_t)cols * elem());
        return m;
    }
    Mat operator()(const Rect &r) const
    {
        Mat v = *this;
        v.rows = r.height; v.cols = r.width; v.data = data + (size_t)r.y * step + (size_t)r.x * elem();
        return v;
    }
    Mat row(int k) const { return (*this)(Rect(0, k, cols, 1)); }
    void roi_origin(int &x0, int &y0) const { const size_t off = (size_t)(data - whole); y0 = (int)(off / step); x0 = (int)((off - (size_t)y0 * step) / elem()); }
    float &f(int i, int j) { return ((float *)(data + i * step))[j]; }
    float f(int i, int j) const { return ((const float *)(data + i * step))[j]; }
    Mat mul(const Mat &o) const                                 // per-element product of two CV_32F images
    {
        assert(type_ == CV_32F && o.type_ == CV_32F && rows == o.rows && cols == o.cols);
        Mat m; m.create(rows, cols, CV_32F);
        for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) m.f(i, j) = f(i, j) * o.f(i, j);
        return m;
    }
};
// first minimum / maximum in row-major order (core/src/minmax.cpp: strict comparisons, a running index)
inline void minMaxLoc(const Mat &m, double *minv, double *maxv, Point *minl, Point *maxl)
{
    assert(m.type_ == CV_32F && m.rows > 0 && m.cols > 0);
    float mn = m.f(0, 0), mx = mn; Point pn(0, 0), px(0, 0);
    for (int i = 0; i < m.rows; i++) for (int j = 0; j < m.cols; j++) { const float v = m.f(i, j); if (v < mn) { mn = v; pn = Point(j, i); } if (v > mx) { mx = v; px = Point(j, i); } }
    if (minv) *minv = mn; if (maxv) *maxv = mx; if (minl) *minl = pn; if (maxl) *maxl = px;
}
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
