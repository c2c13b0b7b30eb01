// multi_view_geometry.hpp -- C++ adapter for MultiViewGeometry::ceresPnP (/root/reference/src/multi_view_geometry.cpp:492-586), the
// motion-only BA the front end runs per frame (src/visual_front_end.cpp:788-801) and the loop closer per candidate
// (src/loop_closer.cpp:882): ONE pose, ReprojectionErrorSE3 factors with fixed world points (OV2_RES_PNP), DENSE_QR in the
// reference, the same two-pass protocol:
//   pass 1    Huber(sqrt(chi2th)) if buse_robust, nmaxiter iterations, function_tolerance 1e-3, 5 ms time limit    (:519-544)
//   outliers  chi2err_ > chi2th or depth <= 0 on the values cached by the last Evaluate; their residual blocks are
//             removed when bapply_l2_after_robust; false when every observation is bad                            (:547-565)
//   pass 2    loss reset to L2, same options, only if bapply_l2_after_robust and outliers were found               (:567-570)
// The reference function is static and is called from two threads: the caller passes the context of ITS thread
// (ov2::SlamGpu::threadContext()).
#pragma once
#include <cmath>
#include "ov2_types.hpp"

namespace ov2 {

// vunkps: n x (u, v) undistorted pixels; vwpts: n x (x, y, z) world points; vscales: n pyramid scales (sigma = 2^scale);
// Twc: [tx ty tz qx qy qz qw], in / out (left unchanged when the function returns false before pass 2, like the reference's early return).
// max_solver_time_s: the reference's 0.005; <= 0 = no limit (results then do not depend on machine load).
// *error (may be NULL) receives the library's message when a solve could not run (the caller then falls back to Ceres).
inline bool ceresPnP(Context &ctx, const double *vunkps, const double *vwpts, const int *vscales, size_t n, double Twc[7], int nmaxiter,
                     float chi2th, bool buse_robust, bool bapply_l2_after_robust, float fx, float fy, float cx, float cy,
                     std::vector<int> &voutliersidx, double max_solver_time_s = 0.005, bool *library_ok = nullptr, std::string *error = nullptr)
{
    if (library_ok) *library_ok = true;
    if (n == 0) return false;
    std::vector<uint8_t> rtype(n, (uint8_t)OV2_RES_PNP), active(n, 1), dpos(n, 1), kfc(1, 0);
    std::vector<int> rkf(n, 0), rlm(n, -1);
    std::vector<double> sigma(n), chi2(n, 0.0);
    for (size_t i = 0; i < n; i++) sigma[i] = std::pow(2., vscales ? vscales[i] : 0);
    ov2_ba_problem p{};
    double pose[7];
    for (int i = 0; i < 7; i++) pose[i] = Twc[i];
    p.n_kf = 1; p.poses = pose; p.kf_const = kfc.data();
    p.n_lm = 0;
    p.n_res = (int)n; p.res_type = rtype.data(); p.res_kf = rkf.data(); p.res_lm = rlm.data(); p.res_uv = vunkps; p.res_sigma = sigma.data();
    p.res_xyz = vwpts; p.res_active = nullptr;
    p.calib_l[0] = fx; p.calib_l[1] = fy; p.calib_l[2] = cx; p.calib_l[3] = cy;
    for (int i = 0; i < 4; i++) p.calib_r[i] = p.calib_l[i];
    p.T_rl[6] = 1.0;
    ov2_ba_options o; ov2_ba_default_options(&o);
    o.max_iter = nmaxiter; o.function_tolerance = 1e-3; o.huber_delta = buse_robust ? std::sqrt((double)chi2th) : -1.0;
    o.max_solver_time_s = max_solver_time_s;
    double pose_out[7];
    ov2_ba_result r{};
    r.poses_out = pose_out; r.chi2_last_eval = chi2.data(); r.depthpos_last_eval = dpos.data();
    if (ov2_ba_solve(ctx.get(), &p, &o, &r) != OV2_OK) {
        if (library_ok) *library_ok = false;
        if (error) *error = ov2_last_error();
        return false;
    }
    size_t nbbad = 0;
    for (size_t i = 0; i < n; i++)
        if (chi2[i] > (double)chi2th || !dpos[i]) {                                            // :547-560
            if (bapply_l2_after_robust) active[i] = 0;
            voutliersidx.push_back((int)i);
            nbbad++;
        }
    if (nbbad == n) return false;                                                              // :562-564 (Twc untouched)
    int termination = r.termination;
    if (bapply_l2_after_robust && !voutliersidx.empty()) {                                     // :567-570
        for (int i = 0; i < 7; i++) pose[i] = pose_out[i];
        p.res_active = active.data();
        o.huber_delta = -1.0;
        if (ov2_ba_solve(ctx.get(), &p, &o, &r) != OV2_OK) {
            if (library_ok) *library_ok = false;
            if (error) *error = ov2_last_error();
            return false;
        }
        termination = r.termination;
    }
    for (int i = 0; i < 7; i++) Twc[i] = pose_out[i];                                           // :572
    return termination != OV2_TERM_FAILURE;                                                    // Summary::IsSolutionUsable (:574)
}

}  // namespace ov2
