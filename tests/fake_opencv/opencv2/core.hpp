// Stand-in for <opencv2/core.hpp> used ONLY by tests/test_golden.py to syntax-check the -DOV2_WITH_OPENCV branch of
// ov2slam_amd/host/*.hpp in an image that has no OpenCV: the members the adapters touch, with OpenCV's names and types
// (core/types.hpp: Point_<float>, Rect_<int>; core/mat.hpp: Mat::data / cols / rows / step).  Not a substitute for a real build.
#pragma once
#include <cstddef>
namespace cv {
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Rect { int x, y, width, height; };
struct MatStep { size_t v; operator size_t() const { return v; } };
struct Mat { unsigned char *data; int cols, rows; MatStep step; bool empty() const { return data == nullptr; } };
}  // namespace cv
