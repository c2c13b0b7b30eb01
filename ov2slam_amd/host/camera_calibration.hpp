// camera_calibration.hpp -- host-side adapter mirroring the reference's per-keypoint
// CameraCalibration::undistortImagePoint (/root/reference/src/camera_calibration.cpp:313-333) and
// Frame::computeKeypoint (src/frame.cpp:246-254) on top of ov2_compute_keypoints (one launch for a
// whole vector of keypoints instead of one cv::undistortPoints call per point).
#pragma once
#include <vector>
#include <array>
#include "ov2_types.hpp"

namespace ov2 {

struct CameraCalibration {
    enum Model { Pinhole = OV2_CAM_PINHOLE, Fisheye = OV2_CAM_FISHEYE };
    Model model_ = Pinhole;
    double K_[4] = {1., 1., 0., 0.};          // fx, fy, cx, cy
    std::vector<double> D_;                   // empty after rectification (Dcv_.release(), camera_calibration.cpp:107)
    double iK_[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};   // row-major K_.inverse(), filled by the caller from its Eigen matrix

    // Frame::computeKeypoint for a vector of raw pixel positions: unpx_ and bv_ of every keypoint
    int computeKeypoints(ov2_ctx *ctx, const std::vector<Point2f> &vpx, std::vector<Point2f> &vunpx,
                         std::vector<std::array<double, 3>> *vbv = nullptr) const
    {
        vunpx.resize(vpx.size());
        if (vbv) vbv->resize(vpx.size());
        if (vpx.empty()) return OV2_OK;
        return ov2_compute_keypoints(ctx, (int)model_, K_, D_.empty() ? nullptr : D_.data(), (int)D_.size(), iK_,
                                     &vpx[0].x, (int)vpx.size(), &vunpx[0].x, vbv ? (*vbv)[0].data() : nullptr);
    }

    // single-point form with the reference's signature
    Point2f undistortImagePoint(ov2_ctx *ctx, const Point2f &pt) const
    {
        std::vector<Point2f> in{pt}, out;
        return computeKeypoints(ctx, in, out) == OV2_OK ? out[0] : pt;
    }
};

} // namespace ov2
