// lk_params.hpp -- parameter block shared by the single-frame LK kernels (lk.hip: k_fb_klt / k_track_klt; lkw.hip: k_track_klt_w).
#pragma once
#include "common.hpp"

struct LKParams {
    int win, max_level, max_iter;
    double eps2;          // clamp(eps,0,10)^2 (double, like cv::TermCriteria::epsilon)
    float min_eig_th;     // 1e-4f
    int flags;
    float err_th, fb_dist;
    int do_fb;            // 1: fbKltTracking, 0: single calcOpticalFlowPyrLK
    int n_max;            // points per batch item (stride)
};

// lkw.hip: a whole wavefront per keypoint (window 9): the single-frame kernel of record
int ov2_launch_track_klt_w(hipStream_t s, const PyrDesc &P, const PyrDesc &C, const LKParams &prm, int lp, int lf, int n_max, const int *n_dev,
                           const float *kps, const float *priors, const uint8_t *flags, float *out_xy, uint8_t *status, int *iters,
                           const float *sad_x, float sad_up, int items = 1, int lk_acc = OV2_LK_ACC_INT64);
