// Stand-in <opencv2/core.hpp> for the SYNTAX CHECK of the patched reference sources: declarations only, OpenCV's names and shapes.
#pragma once
#include <cstddef>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>
namespace cv {
template <class T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T x_, T y_) : x(x_), y(y_) {} };
typedef Point_<float> Point2f; typedef Point_<int> Point; typedef Point_<double> Point2d;
template <class T> struct Point3_ { T x, y, z; };
typedef Point3_<float> Point3f;
struct Rect { int x, y, width, height; Rect() : x(0), y(0), width(0), height(0) {} Rect(int a, int b, int c, int d) : x(a), y(b), width(c), height(d) {} };
struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
struct Scalar { double v[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {} };
struct MatStep { size_t v; operator size_t() const { return v; } };
struct Mat { unsigned char *data = nullptr; int cols = 0, rows = 0; MatStep step{0}; bool empty() const { return data == nullptr; } Mat clone() const { return *this; } };
template <class T> struct Ptr { T *p = nullptr; T *operator->() const { return p; } };
struct CLAHE { void apply(const Mat &, Mat &) {} };
struct TermCriteria { enum { COUNT = 1, MAX_ITER = 1, EPS = 2 }; TermCriteria() {} TermCriteria(int, int, double) {} };
struct FileNode { bool empty() const { return true; } template <class T> operator T() const { return T(); } FileNode operator[](const char *) const { return FileNode(); } };
struct FileStorage { enum { READ = 0 }; FileStorage() {} FileStorage(const std::string &, int) {} FileNode operator[](const char *) const { return FileNode(); } bool isOpened() const { return true; } void release() {} };
struct KeyPoint { Point2f pt; };
struct DMatch {};
template <class T> inline void swap(T &a, T &b) { T t = a; a = b; b = t; }
}  // namespace cv
