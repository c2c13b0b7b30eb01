// keypoint_dev.hpp -- device functions shared by keypoint.hip and stereo.hip: the reference's
// CameraCalibration::undistortImagePoint (src/camera_calibration.cpp:313-333) restated in fp64
// (cv::undistortPoints / cv::fisheye::undistortPoints with P = K), no FMA contraction.
#pragma once
#include "common.hpp"
#include <math.h>

#pragma clang fp contract(off)

struct KpCalib {
    double fx, fy, cx, cy;
    double k[14];
    double iK[9];
    int nD, model;
};

__device__ __forceinline__ void kp_undistort_pinhole(const KpCalib &c, double u, double v, double &ox, double &oy)
{
    const double ifx = 1. / c.fx, ify = 1. / c.fy;
    double x = (u - c.cx) * ifx, y = (v - c.cy) * ify;
    const double x0 = x, y0 = y;
    const double *k = c.k;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) { x = (u - c.cx) * ifx; y = (v - c.cy) * ify; break; }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = c.fx * x + 0. * y + c.cx, yy = 0. * x + c.fy * y + c.cy, ww = 1. / (0. * x + 0. * y + 1.);
    ox = xx * ww; oy = yy * ww;
}

__device__ __forceinline__ bool kp_undistort_fisheye(const KpCalib &c, double u, double v, double &ox, double &oy)
{
    const double EPS = 1e-8, PI_2 = 3.1415926535897932384626433832795 / 2.;
    const double pwx = (u - c.cx) / c.fx, pwy = (v - c.cy) / c.fy;
    double theta_d = sqrt(pwx * pwx + pwy * pwy);
    theta_d = fmin(fmax(-PI_2, theta_d), PI_2);
    bool converged = false;
    double theta = theta_d, scale = 0.0;
    if (fabs(theta_d) > EPS) {
        for (int j = 0; j < 10; j++) {
            const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta6 * theta2;
            const double k0_theta2 = c.k[0] * theta2, k1_theta4 = c.k[1] * theta4, k2_theta6 = c.k[2] * theta6, k3_theta8 = c.k[3] * theta8;
            const double theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                     (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
            theta = theta - theta_fix;
            if (fabs(theta_fix) < EPS) { converged = true; break; }
        }
        scale = tan(theta) / theta_d;
    } else converged = true;
    const bool flipped = (theta_d < 0 && theta > 0) || (theta_d > 0 && theta < 0);
    if (!converged || flipped) return false;
    const double pux = pwx * scale, puy = pwy * scale;
    const double prx = (0. + c.fx * pux) + 0. * puy + c.cx * 1.0;
    const double pry = (0. + 0. * pux) + c.fy * puy + c.cy * 1.0;
    const double prz = (0. + 0. * pux) + 0. * puy + 1. * 1.0;
    ox = prx / prz; oy = pry / prz;
    return true;
}


// host: validate and pack the calibration (defined in keypoint.hip)
int ov2_kp_calib(int model, const double K[4], const double *D, int nD, const double iK[9], KpCalib &c);

// undistortImagePoint on a float pixel: `return pt` when there is no distortion vector
__device__ __forceinline__ float2 kp_undistort_image_point(const KpCalib &c, float2 p)
{
    if (c.nD <= 0) return p;
    double ox, oy;
    if (c.model == OV2_CAM_FISHEYE) {
        if (kp_undistort_fisheye(c, (double)p.x, (double)p.y, ox, oy)) return make_float2((float)ox, (float)oy);
        return make_float2(-1000000.0f, -1000000.0f);
    }
    kp_undistort_pinhole(c, (double)p.x, (double)p.y, ox, oy);
    return make_float2((float)ox, (float)oy);
}
