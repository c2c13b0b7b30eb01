/*
 * undistort.c -- CPU ORACLE (test infrastructure) for per-keypoint undistortion + bearing vectors.
 *
 * Follows Frame::computeKeypoint (/root/reference/src/frame.cpp:246-254):
 *     kp.unpx_ = pcalib_leftcam_->undistortImagePoint(pt);            camera_calibration.cpp:313-333
 *     kp.bv_   = (iK_ * (unpx.x, unpx.y, 1)).normalized();
 * undistortImagePoint delegates to cv::undistortPoints(src, dst, K, D, noArray(), K) (pinhole) or
 * cv::fisheye::undistortPoints(src, dst, K, D, Mat(), K) (fisheye).  OpenCV is not vendored: the two
 * functions below restate the public OpenCV 4.x implementations
 *   modules/calib3d/src/undistort.dispatch.cpp  cvUndistortPointsInternal, criteria = (MAX_ITER, 5, 0.01)
 *   modules/calib3d/src/fisheye.cpp             cv::fisheye::undistortPoints, criteria = (COUNT+EPS, 10, 1e-8)
 * operation by operation, in double, without FMA contraction ("parity unpinned", see ov2_oracle.h).
 * The bearing product follows Eigen's coefficient order for a 3x3 * 3x1 product ((a0 b0 + a1 b1) + a2 b2)
 * and normalize() = division of every component by sqrt((x^2 + y^2) + z^2).
 */
#include "ov2_oracle.h"
#include <math.h>
#include <float.h>


void orc_undistort_pinhole(const double K[4], const double *D, int nD, const float *px, int n, float *out)
{
    double k[14] = {0};
    for (int i = 0; i < nD && i < 14; i++) k[i] = D[i];
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const double ifx = 1. / fx, ify = 1. / fy;
    /* RR = P * I with P = K: exact copies of K's entries */
    for (int i = 0; i < n; i++) {
        double x = (double)px[2 * i], y = (double)px[2 * i + 1];
        const double u = x, v = y;
        x = (x - cx) * ifx;
        y = (y - cy) * ify;
        if (nD > 0) {
            /* tilt model off (k[12] = k[13] = 0): invMatTilt is the exact identity */
            const double x0 = x, y0 = y;
            for (int j = 0; j < 5; j++) {
                const double r2 = x * x + y * y;
                const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
                if (icdist < 0) {                 /* test: undistortPoints with r > 1 */
                    x = (u - cx) * ifx;
                    y = (v - cy) * ify;
                    break;
                }
                const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
                const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
                x = (x0 - deltaX) * icdist;
                y = (y0 - deltaY) * icdist;
            }
        }
        const double xx = fx * x + 0. * y + cx;
        const double yy = 0. * x + fy * y + cy;
        const double ww = 1. / (0. * x + 0. * y + 1.);
        out[2 * i] = (float)(xx * ww);
        out[2 * i + 1] = (float)(yy * ww);
    }
}

void orc_undistort_fisheye(const double K[4], const double D[4], const float *px, int n, float *out)
{
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const double EPS = 1e-8, PI_2 = 3.1415926535897932384626433832795 / 2.;
    for (int i = 0; i < n; i++) {
        const double pix = (double)px[2 * i], piy = (double)px[2 * i + 1];
        const double pwx = (pix - cx) / fx, pwy = (piy - cy) / fy;
        double theta_d = sqrt(pwx * pwx + pwy * pwy);
        theta_d = fmin(fmax(-PI_2, theta_d), PI_2);           /* the model is only valid up to 180 degrees FOV */
        int converged = 0;
        double theta = theta_d, scale = 0.0;
        if (fabs(theta_d) > EPS) {
            for (int j = 0; j < 10; j++) {
                const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta6 * theta2;
                const double k0_theta2 = D[0] * theta2, k1_theta4 = D[1] * theta4, k2_theta6 = D[2] * theta6, k3_theta8 = D[3] * theta8;
                const double theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                         (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
                theta = theta - theta_fix;
                if (fabs(theta_fix) < EPS) { converged = 1; break; }
            }
            scale = tan(theta) / theta_d;
        } else converged = 1;
        const int theta_flipped = (theta_d < 0 && theta > 0) || (theta_d > 0 && theta < 0);
        if (converged && !theta_flipped) {
            const double pux = pwx * scale, puy = pwy * scale;
            const double prx = (0. + fx * pux) + 0. * puy + cx * 1.0;
            const double pry = (0. + 0. * pux) + fy * puy + cy * 1.0;
            const double prz = (0. + 0. * pux) + 0. * puy + 1. * 1.0;
            out[2 * i] = (float)(prx / prz);
            out[2 * i + 1] = (float)(pry / prz);
        } else {
            out[2 * i] = -1000000.0f;
            out[2 * i + 1] = -1000000.0f;
        }
    }
}

void orc_compute_keypoints(int model, const double K[4], const double *D, int nD, const double iK[9],
                           const float *px, int n, float *unpx, double *bv)
{
    if (nD <= 0) for (int i = 0; i < 2 * n; i++) unpx[i] = px[i];          /* Dcv_.empty(): return pt (camera_calibration.cpp:317-319) */
    else if (model == ORC_CAM_FISHEYE) orc_undistort_fisheye(K, D, px, n, unpx);
    else orc_undistort_pinhole(K, D, nD, px, n, unpx);
    if (!bv) return;
    for (int i = 0; i < n; i++) {
        const double x = (double)unpx[2 * i], y = (double)unpx[2 * i + 1];
        double b[3];
        for (int r = 0; r < 3; r++) b[r] = (iK[3 * r] * x + iK[3 * r + 1] * y) + iK[3 * r + 2] * 1.;
        const double nrm = sqrt((b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]);
        bv[3 * i] = b[0] / nrm; bv[3 * i + 1] = b[1] / nrm; bv[3 * i + 2] = b[2] / nrm;
    }
}
