// ov2_types.hpp -- minimal stand-ins for the OpenCV value types that appear in the reference's
// hot-path signatures (OpenCV is not available in the build image).  In a real OV2SLAM tree these
// aliases are replaced by `using Point2f = cv::Point2f;` etc.: the layouts
// are identical (two packed floats / four ints), which is all the C ABI relies on.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/ov2slam_hip.h"

namespace ov2 {

struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Rect { int x, y, width, height; };

// non-owning 8-bit single-channel image view (cv::Mat CV_8UC1: data, cols, rows, step)
struct Image8 {
    const uint8_t *data = nullptr; int cols = 0, rows = 0, step = 0;
    bool empty() const { return data == nullptr || cols <= 0 || rows <= 0; }
};

// one ov2_ctx per thread that calls into the library (SLAM thread, mapper thread, estimator thread)
class Context {
public:
    explicit Context(int device = 0) {
        if (ov2_ctx_create(device, &ctx_) != OV2_OK) throw std::runtime_error(std::string("ov2_ctx_create: ") + ov2_last_error());
    }
    ~Context() { ov2_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    ov2_ctx *get() const { return ctx_; }
private:
    ov2_ctx *ctx_ = nullptr;
};

// Device-resident replacement of the std::vector<cv::Mat> that cv::buildOpticalFlowPyramid fills
// (src/visual_front_end.cpp:1172, :53, src/mapper.cpp:81).  Movable like the cv::Mat vector that is
// handed from the front-end to the mapper thread inside Keyframe (src/ov2slam.cpp:175-180).
class Pyramid {
public:
    Pyramid() = default;
    Pyramid(Pyramid &&o) noexcept : p_(o.p_) { o.p_ = nullptr; }
    Pyramid &operator=(Pyramid &&o) noexcept { if (this != &o) { ov2_pyr_destroy(p_); p_ = o.p_; o.p_ = nullptr; } return *this; }
    Pyramid(const Pyramid &) = delete;
    Pyramid &operator=(const Pyramid &) = delete;
    ~Pyramid() { ov2_pyr_destroy(p_); }
    bool empty() const { return p_ == nullptr; }
    size_t size() const { return p_ ? 2 * (size_t)ov2_pyr_levels(p_) : 0; }     // 2 Mats per level in the reference
    void swap(Pyramid &o) { ov2_pyr *t = p_; p_ = o.p_; o.p_ = t; }
    ov2_pyr *get() const { return p_; }
    // cv::buildOpticalFlowPyramid(img, *this, Size(win,win), max_level)
    int build(Context &ctx, const Image8 &img, int win, int max_level) {
        if (img.empty()) return OV2_EINVAL;
        int w = 0, h = 0;
        if (p_ && (ov2_pyr_level_size(p_, 0, &w, &h) != OV2_OK || w != img.cols || h != img.rows)) { ov2_pyr_destroy(p_); p_ = nullptr; }
        if (!p_) { const int rc = ov2_pyr_create(ctx.get(), img.cols, img.rows, win, max_level, 1, &p_); if (rc != OV2_OK) return rc; }
        return ov2_pyr_build_h(ctx.get(), p_, img.data, img.step, 0);
    }
private:
    ov2_pyr *p_ = nullptr;
};

}  // namespace ov2
