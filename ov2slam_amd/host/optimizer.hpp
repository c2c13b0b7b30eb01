// optimizer.hpp -- C++ adapter for the solve stage of Optimizer::localBA
// (/root/reference/src/optimizer.cpp:436-627).  The map walk (:43-430) fills a FlatProblem instead of a
// ceres::Problem (one push per AddParameterBlock / AddResidualBlock call); solve()
// then replaces the two ceres::Solve calls and the outlier bookkeeping between them, and the caller's
// write-back (:741-883) reads poses / invdepth / bad_obs from the result.
#pragma once
#include <cmath>
#include "ov2_types.hpp"

namespace ov2 {

struct FlatProblem {
    std::vector<double> poses;          // 7 per keyframe  [tx ty tz qx qy qz qw]   (PoseParametersBlock)
    std::vector<uint8_t> kf_const;      // SetParameterBlockConstant
    std::vector<double> invdepth;       // InvDepthParametersBlock
    std::vector<int> lm_anchor_kf;
    std::vector<double> lm_anchor_uv;   // 2 per landmark
    std::vector<uint8_t> res_type;      // OV2_RES_*
    std::vector<int> res_kf, res_lm;
    std::vector<double> res_uv, res_sigma, res_xyz;   // res_xyz: 3 per residual, used by OV2_RES_PNP blocks only
    double calib_l[4] = {0, 0, 0, 0}, calib_r[4] = {0, 0, 0, 0}, T_rl[7] = {0, 0, 0, 0, 0, 0, 1};

    int addKeyframe(const double pose[7], bool constant) { poses.insert(poses.end(), pose, pose + 7); kf_const.push_back(constant); return (int)kf_const.size() - 1; }
    int addLandmark(double inv_depth, int anchor_kf, double u, double v) {
        invdepth.push_back(inv_depth); lm_anchor_kf.push_back(anchor_kf); lm_anchor_uv.push_back(u); lm_anchor_uv.push_back(v);
        return (int)invdepth.size() - 1;
    }
    int addResidual(int type, int kf, int lm, double u, double v, double sigma) {
        res_type.push_back((uint8_t)type); res_kf.push_back(kf); res_lm.push_back(lm); res_uv.push_back(u); res_uv.push_back(v); res_sigma.push_back(sigma);
        res_xyz.push_back(0); res_xyz.push_back(0); res_xyz.push_back(0);
        return (int)res_type.size() - 1;
    }
    int addPnPResidual(int kf, const double xyz[3], double u, double v, double sigma) {
        const int i = addResidual(OV2_RES_PNP, kf, -1, u, v, sigma);
        res_xyz[3 * i] = xyz[0]; res_xyz[3 * i + 1] = xyz[1]; res_xyz[3 * i + 2] = xyz[2];
        return i;
    }
    ov2_ba_problem view(const uint8_t *res_active) const {
        ov2_ba_problem p;
        p.n_kf = (int)kf_const.size(); p.poses = poses.data(); p.kf_const = kf_const.data();
        p.n_lm = (int)invdepth.size(); p.invdepth = invdepth.data(); p.lm_anchor_kf = lm_anchor_kf.data(); p.lm_anchor_uv = lm_anchor_uv.data();
        p.n_res = (int)res_type.size(); p.res_type = res_type.data(); p.res_kf = res_kf.data(); p.res_lm = res_lm.data();
        p.res_uv = res_uv.data(); p.res_sigma = res_sigma.data(); p.res_active = res_active; p.res_xyz = res_xyz.data();
        for (int i = 0; i < 4; i++) { p.calib_l[i] = calib_l[i]; p.calib_r[i] = calib_r[i]; }
        for (int i = 0; i < 7; i++) p.T_rl[i] = T_rl[i];
        return p;
    }
};

// Flat form of the ceres::Problem Optimizer::structureOnlyBA builds (src/optimizer.cpp:2594-2781): constant
// keyframes, 3-D points (PointXYZParametersBlock), ReprojectionErrorKSE3XYZ / RightCamKSE3XYZ blocks.
struct FlatStructureProblem {
    std::vector<double> poses, xyz;     // 7 per keyframe, 3 per point
    std::vector<uint8_t> res_type;      // OV2_XYZ_*
    std::vector<int> res_kf, res_pt;
    std::vector<double> res_uv, res_sigma;
    double calib_l[4] = {0, 0, 0, 0}, calib_r[4] = {0, 0, 0, 0}, T_rl[7] = {0, 0, 0, 0, 0, 0, 1};

    int addKeyframe(const double pose[7]) { poses.insert(poses.end(), pose, pose + 7); return (int)poses.size() / 7 - 1; }          // :2660-2670
    int addPoint(const double p[3]) { xyz.insert(xyz.end(), p, p + 3); return (int)xyz.size() / 3 - 1; }                              // :2641-2644
    void addResidual(int type, int kf, int pt, double u, double v, double sigma) {                                                      // :2675-2727
        res_type.push_back((uint8_t)type); res_kf.push_back(kf); res_pt.push_back(pt); res_uv.push_back(u); res_uv.push_back(v); res_sigma.push_back(sigma);
    }
};

// Flat form of the buse_inv_depth: 0 problem (src/optimizer.cpp:207-209 PointXYZParametersBlock, :333-384 XYZ residual blocks
// with variable poses): FlatStructureProblem + SetParameterBlockConstant flags.
struct FlatXYZProblem : FlatStructureProblem {
    std::vector<uint8_t> kf_const;
    int addKeyframe(const double pose[7], bool constant) { kf_const.push_back(constant); return FlatStructureProblem::addKeyframe(pose); }
    ov2_xyzba_problem view(const uint8_t *res_active) const {
        ov2_xyzba_problem p{};
        p.n_kf = (int)poses.size() / 7; p.poses = poses.data(); p.kf_const = kf_const.data();
        p.n_pts = (int)xyz.size() / 3; p.xyz = xyz.data();
        p.n_res = (int)res_type.size(); p.res_type = res_type.data(); p.res_kf = res_kf.data(); p.res_pt = res_pt.data();
        p.res_uv = res_uv.data(); p.res_sigma = res_sigma.data(); p.res_active = res_active;
        for (int i = 0; i < 4; i++) { p.calib_l[i] = calib_l[i]; p.calib_r[i] = calib_r[i]; }
        for (int i = 0; i < 7; i++) p.T_rl[i] = T_rl[i];
        return p;
    }
};

struct LocalBAResult {
    bool ok = false, l2_done = false;
    // why a solve was skipped (ok == false): the library's return code and message, e.g. OV2_EUNSUPPORTED "reduced system too
    // large ..." when more than 1024 keyframes are optimised (~450 for 3-D point landmarks) -- log it, do not drop it
    int error_code = OV2_OK;
    std::string error;
    std::vector<double> poses, invdepth, chi2;
    std::vector<uint8_t> depthpos, bad_obs;     // bad_obs[i] = 1: observation i is an outlier (:500-592, :637-735)
    int iterations[2] = {0, 0};
    double solve_ms[2] = {0, 0};
};

class Optimizer {
public:
    Optimizer(double robust_mono_th, bool apply_l2_after_robust) : robust_mono_th_(robust_mono_th), apply_l2_after_robust_(apply_l2_after_robust) {}
    void signalStopLocalBA() { bstop_localba_ = 1; }        // optimizer.hpp:48 (called by the estimator thread while localBA runs)
    // The reference clears bstop_localba_ at the END of localBA (src/optimizer.cpp:896), after the map write-back: a signal raised
    // during the write-back (Optimizer::signalStopLocalBA fires whenever blocalba_is_on_) is dropped there and must not reach the
    // next localBA.  The caller of solveLocalBA clears this copy at the same place (integration/ov2slam_hip.patch does).
    void clearStopLocalBA() { bstop_localba_ = 0; }
    bool stopLocalBA() const { return bstop_localba_ != 0; } // optimizer.hpp:49
    // Ceres' max_solver_time_in_seconds of the first pass: the reference sets 0.2 s, doubled unless force_realtime
    // (src/optimizer.cpp:463-467), and halves it for the L2 pass (:612).  0 (default) = no limit.
    void setMaxSolverTime(double pass1_seconds) { max_solver_time_s_ = pass1_seconds; }

    // Optimizer::localBA's solve stage (src/optimizer.cpp:436-735) as ONE library call: the problem stays in HBM between the
    // robust and the L2 pass, outlier tests and block removal run on the device (ov2_local_ba).  want_chi2: also download the
    // per-block chi2err_ / isdepthpositive_ values (the reference's write-back only needs bad_obs).
    // The stop flag is handed over LIVE (ov2_local_ba_options::stop_flag): the library reads it after pass 1, where the
    // reference evaluates !stopLocalBA() (:603-604).  It is NOT cleared here: see clearStopLocalBA().
    LocalBAResult solveLocalBA(Context &ctx, FlatProblem &fp, bool buse_robust_cost, bool want_chi2 = false)
    {
        LocalBAResult R;
        const size_t n_res = fp.res_type.size();
        R.poses.resize(fp.poses.size()); R.invdepth.resize(fp.invdepth.size());
        R.bad_obs.assign(n_res, 0);
        if (want_chi2) { R.chi2.assign(n_res, 0.0); R.depthpos.assign(n_res, 1); }
        ov2_local_ba_options opt; ov2_local_ba_default_options(&opt);
        opt.robust_mono_th = robust_mono_th_; opt.use_robust_cost = buse_robust_cost ? 1 : 0;
        opt.apply_l2_after_robust = apply_l2_after_robust_ ? 1 : 0; opt.stop_requested = 0; opt.stop_flag = &bstop_localba_;
        opt.pass1.max_solver_time_s = max_solver_time_s_; opt.pass2.max_solver_time_s = 0.5 * max_solver_time_s_;
        ov2_local_ba_result res{};
        res.poses_out = R.poses.data(); res.invdepth_out = R.invdepth.data(); res.bad_obs = R.bad_obs.data();
        if (want_chi2) { res.chi2_last_eval = R.chi2.data(); res.depthpos_last_eval = R.depthpos.data(); }
        ov2_ba_problem p = fp.view(nullptr);
        R.error_code = ov2_local_ba(ctx.get(), &p, &opt, &res);
        if (R.error_code != OV2_OK) { R.error = ov2_last_error(); return R; }       // BA skipped: caller logs R.error
        R.ok = true; R.l2_done = res.l2_done != 0;
        if (res.pass2_error != OV2_OK) { R.error_code = res.pass2_error; R.error = ov2_last_error(); }     // pass 1's result is valid and kept
        for (int i = 0; i < 2; i++) { R.iterations[i] = res.iterations[i]; R.solve_ms[i] = res.solve_ms[i]; }
        return R;
    }

    // Optimizer::localBA with buse_inv_depth: 0 (src/optimizer.cpp:207-209, :333-384): the same two-pass protocol on 3-D point
    // landmarks; R.invdepth holds the optimised points (3 per landmark).
    LocalBAResult solveLocalBAXYZ(Context &ctx, FlatXYZProblem &fp, bool buse_robust_cost) const
    {
        LocalBAResult R;
        const size_t n_res = fp.res_type.size();
        R.poses.resize(fp.poses.size()); R.invdepth.resize(fp.xyz.size());
        R.chi2.assign(n_res, 0.0); R.depthpos.assign(n_res, 1); R.bad_obs.assign(n_res, 0);
        ov2_ba_options opt; ov2_ba_default_options(&opt);
        opt.max_iter = 5; opt.function_tolerance = 1e-3;
        opt.huber_delta = buse_robust_cost ? std::sqrt(robust_mono_th_) : -1.0;
        ov2_xyzba_result res{};
        res.poses_out = R.poses.data(); res.xyz_out = R.invdepth.data(); res.chi2_last_eval = R.chi2.data(); res.depthpos_last_eval = R.depthpos.data();
        ov2_xyzba_problem p = fp.view(nullptr);
        if ((R.error_code = ov2_xyz_ba_solve(ctx.get(), &p, &opt, &res)) != OV2_OK) { R.error = ov2_last_error(); return R; }
        R.ok = true; R.iterations[0] = res.iterations; R.solve_ms[0] = res.solve_ms;
        std::vector<uint8_t> active(n_res, 1);
        size_t nbbad = 0; bool left_rem = false, right_rem = false;
        for (size_t i = 0; i < n_res; i++) {
            const bool bad = R.chi2[i] > robust_mono_th_ || !R.depthpos[i];
            R.bad_obs[i] = bad; nbbad += bad;
            if (bad && apply_l2_after_robust_) active[i] = 0;
            if (!bad && fp.res_type[i] == OV2_XYZ_LEFT) left_rem = true;
            if (!bad && fp.res_type[i] == OV2_XYZ_RIGHT) right_rem = true;
        }
        if (apply_l2_after_robust_ && buse_robust_cost && !stopLocalBA() && nbbad > 0) {
            if (left_rem && right_rem) opt.huber_delta = -1.0;                   // :606-608
            opt.max_iter = 10;
            fp.poses = R.poses; fp.xyz = R.invdepth;
            p = fp.view(active.data());
            if (ov2_xyz_ba_solve(ctx.get(), &p, &opt, &res) == OV2_OK) {
                R.l2_done = true; R.iterations[1] = res.iterations; R.solve_ms[1] = res.solve_ms;
                for (size_t i = 0; i < n_res; i++)
                    if (active[i] && (R.chi2[i] > robust_mono_th_ || !R.depthpos[i])) R.bad_obs[i] = 1;
            }
        }
        return R;
    }

    // Optimizer::structureOnlyBA (src/optimizer.cpp:2594-2781): Huber(sqrt(robust_mono_th)), 10 iterations, function_tolerance
    // 1e-3 (:2742-2758).  On success `xyz_out` holds what the reference writes back through updateMapPoint (:2768-2779).
    bool solveStructureOnlyBA(Context &ctx, const FlatStructureProblem &sp, std::vector<double> &xyz_out) const
    {
        ov2_sba_problem p{};
        p.n_kf = (int)sp.poses.size() / 7; p.poses = sp.poses.data();
        p.n_pts = (int)sp.xyz.size() / 3; p.xyz = sp.xyz.data();
        p.n_res = (int)sp.res_type.size(); p.res_type = sp.res_type.data(); p.res_kf = sp.res_kf.data(); p.res_pt = sp.res_pt.data();
        p.res_uv = sp.res_uv.data(); p.res_sigma = sp.res_sigma.data(); p.res_active = nullptr;
        for (int i = 0; i < 4; i++) { p.calib_l[i] = sp.calib_l[i]; p.calib_r[i] = sp.calib_r[i]; }
        for (int i = 0; i < 7; i++) p.T_rl[i] = sp.T_rl[i];
        ov2_ba_options opt; ov2_ba_default_options(&opt);
        opt.max_iter = 10; opt.function_tolerance = 1e-3; opt.huber_delta = std::sqrt(robust_mono_th_);
        xyz_out.assign(sp.xyz.size(), 0.0);
        ov2_sba_result res{};
        res.xyz_out = xyz_out.data();
        return ov2_structure_ba(ctx.get(), &p, &opt, &res) == OV2_OK;
    }

    // Optimizer::looseBA (src/optimizer.cpp:900-1672): same residual blocks, ONE solve (5 it, function_tolerance
    // 1e-4, :1297-1310), then the chi2 / depth test over all three residual lists (:1327-1430).
    LocalBAResult solveLooseBA(Context &ctx, FlatProblem &fp, bool buse_robust_cost) const
    {
        return solveOnce(ctx, fp, buse_robust_cost, 5, 1e-4, /*test_anchor_right*/ true);
    }

    // Optimizer::fullBA (src/optimizer.cpp:1674-2332): pass 1 with max 100 iterations and Ceres' default
    // tolerances (:2055-2061); outlier test over the left / right lists (:2067-2138); if apply_l2_after_robust and
    // outliers were found, pass 2 with the loss reset to L2 when the left list is non-empty (:2143-2149); test again.
    LocalBAResult solveFullBA(Context &ctx, FlatProblem &fp, bool buse_robust_cost) const
    {
        LocalBAResult R = solveOnce(ctx, fp, buse_robust_cost, 100, 1e-6, /*test_anchor_right*/ false);
        if (!R.ok) return R;
        const size_t n_res = fp.res_type.size();
        std::vector<uint8_t> active(n_res, 1);
        size_t nbbad = 0; bool left_rem = false;
        for (size_t i = 0; i < n_res; i++) {
            nbbad += R.bad_obs[i];
            if (R.bad_obs[i] && apply_l2_after_robust_) active[i] = 0;
            if (!R.bad_obs[i] && fp.res_type[i] == OV2_RES_LEFT) left_rem = true;
        }
        if (apply_l2_after_robust_ && nbbad > 0) {
            ov2_ba_options opt; ov2_ba_default_options(&opt);
            opt.max_iter = 100; opt.function_tolerance = 1e-6;
            opt.huber_delta = (buse_robust_cost && !left_rem) ? std::sqrt(robust_mono_th_) : -1.0;
            ov2_ba_result res{};
            res.poses_out = R.poses.data(); res.invdepth_out = R.invdepth.data(); res.chi2_last_eval = R.chi2.data(); res.depthpos_last_eval = R.depthpos.data();
            fp.poses = R.poses; fp.invdepth = R.invdepth;
            ov2_ba_problem p = fp.view(active.data());
            if (ov2_ba_solve(ctx.get(), &p, &opt, &res) == OV2_OK) { R.l2_done = true; R.iterations[1] = res.iterations; R.solve_ms[1] = res.solve_ms; }
        }
        for (size_t i = 0; i < n_res; i++)
            if (active[i] && (fp.res_type[i] == OV2_RES_LEFT || fp.res_type[i] == OV2_RES_RIGHT) && (R.chi2[i] > robust_mono_th_ || !R.depthpos[i]))
                R.bad_obs[i] = 1;
        return R;
    }

private:
    LocalBAResult solveOnce(Context &ctx, FlatProblem &fp, bool buse_robust_cost, int max_iter, double ftol, bool test_anchor_right) const
    {
        LocalBAResult R;
        const size_t n_res = fp.res_type.size();
        R.poses.resize(fp.poses.size()); R.invdepth.resize(fp.invdepth.size());
        R.chi2.assign(n_res, 0.0); R.depthpos.assign(n_res, 1); R.bad_obs.assign(n_res, 0);
        ov2_ba_options opt; ov2_ba_default_options(&opt);
        opt.max_iter = max_iter; opt.function_tolerance = ftol;
        opt.huber_delta = buse_robust_cost ? std::sqrt(robust_mono_th_) : -1.0;
        ov2_ba_result res{};
        res.poses_out = R.poses.data(); res.invdepth_out = R.invdepth.data(); res.chi2_last_eval = R.chi2.data(); res.depthpos_last_eval = R.depthpos.data();
        ov2_ba_problem p = fp.view(nullptr);
        if ((R.error_code = ov2_ba_solve(ctx.get(), &p, &opt, &res)) != OV2_OK) { R.error = ov2_last_error(); return R; }
        R.ok = true; R.iterations[0] = res.iterations; R.solve_ms[0] = res.solve_ms;
        for (size_t i = 0; i < n_res; i++) {
            const bool tested = fp.res_type[i] == OV2_RES_LEFT || fp.res_type[i] == OV2_RES_RIGHT || (test_anchor_right && fp.res_type[i] == OV2_RES_RIGHT_ANCH);
            R.bad_obs[i] = tested && (R.chi2[i] > robust_mono_th_ || !R.depthpos[i]);
        }
        return R;
    }

    double robust_mono_th_;
    bool apply_l2_after_robust_;
    volatile int bstop_localba_ = 0;
    double max_solver_time_s_ = 0.0;
};

}  // namespace ov2
