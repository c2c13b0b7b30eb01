// ov2_types.hpp -- minimal stand-ins for the OpenCV value types that appear in the reference's
// hot-path signatures (OpenCV is not available in the build image).  In a real OV2SLAM tree these
// aliases are replaced by `using Point2f = cv::Point2f;` etc.: the layouts
// are identical (two packed floats / four ints), which is all the C ABI relies on.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "../../include/ov2slam_hip.h"

#ifdef OV2_WITH_OPENCV
// In an OV2SLAM tree (compile with -DOV2_WITH_OPENCV): the adapters below take the reference's own value types, so the
// method bodies in INTEGRATION.md are these headers verbatim.  tests/test_golden.py syntax-checks this branch against a
// minimal stand-in for <opencv2/core.hpp> (tests/fake_opencv: the members the adapters touch, nothing else), because the
// build image has no OpenCV.
#include <opencv2/core.hpp>
#endif

namespace ov2 {

#ifdef OV2_WITH_OPENCV
using Point2f = cv::Point2f;
using Rect = cv::Rect;
// non-owning view of a CV_8UC1 cv::Mat
struct Image8 {
    const uint8_t *data = nullptr; int cols = 0, rows = 0, step = 0;
    Image8() = default;
    Image8(const cv::Mat &m) : data(m.data), cols(m.cols), rows(m.rows), step((int)m.step) {}      // implicit on purpose: cv::Mat call sites stay unchanged
    Image8(const uint8_t *d, int c, int r, int s) : data(d), cols(c), rows(r), step(s) {}
    bool empty() const { return data == nullptr || cols <= 0 || rows <= 0; }
};
#else
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Rect { int x, y, width, height; };

// non-owning 8-bit single-channel image view (cv::Mat CV_8UC1: data, cols, rows, step)
struct Image8 {
    const uint8_t *data = nullptr; int cols = 0, rows = 0, step = 0;
    Image8() = default;
    Image8(const uint8_t *d, int c, int r, int s) : data(d), cols(c), rows(r), step(s) {}
    bool empty() const { return data == nullptr || cols <= 0 || rows <= 0; }
};
#endif

// one ov2_ctx per thread that calls into the library (SLAM thread, mapper thread, estimator thread)
class Context {
public:
    explicit Context(int device = 0) {
        // header / library mismatch: option structs may differ in size (include/ov2slam_hip.h, OV2_ABI_VERSION)
        if (ov2_version() != OV2_ABI_VERSION) throw std::runtime_error("libov2slam_hip.so ABI version differs from ov2slam_hip.h");
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
    Pyramid(Pyramid &&o) noexcept : p_(o.p_), win_(o.win_), max_level_(o.max_level_) { o.p_ = nullptr; }
    Pyramid &operator=(Pyramid &&o) noexcept { if (this != &o) { ov2_pyr_destroy(p_); p_ = o.p_; win_ = o.win_; max_level_ = o.max_level_; o.p_ = nullptr; } return *this; }
    Pyramid(const Pyramid &) = delete;
    Pyramid &operator=(const Pyramid &) = delete;
    ~Pyramid() { ov2_pyr_destroy(p_); }
    bool empty() const { return p_ == nullptr; }
    size_t size() const { return p_ ? 2 * (size_t)ov2_pyr_levels(p_) : 0; }     // 2 Mats per level in the reference
    void swap(Pyramid &o) { std::swap(p_, o.p_); std::swap(win_, o.win_); std::swap(max_level_, o.max_level_); }
    ov2_pyr *get() const { return p_; }
    // cv::buildOpticalFlowPyramid(img, *this, Size(win,win), max_level)
    int build(Context &ctx, const Image8 &img, int win, int max_level) {
        if (img.empty()) return OV2_EINVAL;
        int w = 0, h = 0;
        // a different image size, window or level count needs a new device layout (the old geometry would make every
        // later ov2_fb_klt fail with "LK window differs" / return fewer levels than asked for)
        if (p_ && (ov2_pyr_level_size(p_, 0, &w, &h) != OV2_OK || w != img.cols || h != img.rows || win != win_ || max_level != max_level_)) { ov2_pyr_destroy(p_); p_ = nullptr; }
        if (!p_) { const int rc = ov2_pyr_create(ctx.get(), img.cols, img.rows, win, max_level, 1, &p_); if (rc != OV2_OK) return rc; win_ = win; max_level_ = max_level; }
        // asynchronous on ctx's stream; the image is staged in the context's pinned buffer before this returns (img.data may be
        // released at once).  Consumers on ANOTHER context wait on the pyramid's ready event themselves.
        return ov2_pyr_build_h(ctx.get(), p_, img.data, img.step, 0);
    }
    // pclahe_->apply(img_raw, img) followed by cv::buildOpticalFlowPyramid(img, *this, ...)  (src/visual_front_end.cpp:1159-1172,
    // src/mapper.cpp:75-81) in one enqueue: the equalised image is level 0 of the pyramid and never leaves the device.
    // Tile grid like src/ov2slam.cpp:85-89: img.cols / 50 x img.rows / 50.
    int buildClahe(Context &ctx, const Image8 &img, int win, int max_level, double clip_limit) {
        if (img.empty()) return OV2_EINVAL;
        int w = 0, h = 0;
        if (p_ && (ov2_pyr_level_size(p_, 0, &w, &h) != OV2_OK || w != img.cols || h != img.rows || win != win_ || max_level != max_level_)) { ov2_pyr_destroy(p_); p_ = nullptr; }
        if (!p_) { const int rc = ov2_pyr_create(ctx.get(), img.cols, img.rows, win, max_level, 1, &p_); if (rc != OV2_OK) return rc; win_ = win; max_level_ = max_level; }
        return ov2_pyr_build_clahe_h(ctx.get(), p_, img.data, img.step, clip_limit, img.cols / 50, img.rows / 50);
    }
private:
    ov2_pyr *p_ = nullptr;
    int win_ = 0, max_level_ = -1;
};

}  // namespace ov2
