// capture_ba.cpp -- the REAL reference solve stage of Optimizer::localBA on fixed problems: the reference's own factors
// (/root/reference/src/ceres_parametrization.cpp, compiled from where it lies) and parameter blocks, the vendored Ceres' TR-LM with
// DENSE_SCHUR, the exact option values and the two-pass outlier protocol of /root/reference/src/optimizer.cpp:436-735.  Not built in
// this repo's image (no Eigen / Ceres / Sophus build there); see CMakeLists.txt.  Input: tests/golden/ref_inputs/ba_<tag>.bin (the flat
// problem layout of ov2slam_amd/stream.py: write_ba_problem, written by make_inputs.py).  Output, per problem, tests/golden/ref/:
//   ba_<tag>_pass1_{poses,invdepth,chi2,depthpos,summary}.npy   after the first ceres::Solve (5 it, ftol 1e-3, Huber sqrt(5.9915))
//   ba_<tag>_final_{poses,invdepth,bad_obs,summary}.npy         after the outlier removal + the L2 pass (10 it) + the second test
// summary = [iterations (summary.iterations.size() - 1), num_successful_steps, termination_type, initial_cost, final_cost, l2_done].
// tests/test_reference_fixtures.py compares the oracle (always) and the HIP solver (-m gpu) with them.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <ceres/ceres.h>

#include "ceres_parametrization.hpp"

static void npy_save(const std::string &path, const char *descr, const std::vector<size_t> &shape, const void *data, size_t bytes)
{
    std::string sh = "(";
    for (size_t i = 0; i < shape.size(); i++) sh += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
    sh += ")";
    std::string hdr = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': " + sh + ", }";
    while ((10 + hdr.size() + 1) % 64) hdr += ' ';
    hdr += '\n';
    std::ofstream f(path, std::ios::binary);
    const char magic[8] = {'\x93', 'N', 'U', 'M', 'P', 'Y', 1, 0};
    f.write(magic, 8);
    const uint16_t hl = (uint16_t)hdr.size();
    f.write((const char *)&hl, 2);
    f.write(hdr.data(), (std::streamsize)hdr.size());
    f.write((const char *)data, (std::streamsize)bytes);
}

struct Flat {
    int n_kf = 0, n_lm = 0, n_res = 0;
    std::vector<double> poses, invdepth, lm_anchor_uv, res_uv, res_sigma, calib_l, calib_r, T_rl;
    std::vector<uint8_t> kf_const, res_type;
    std::vector<int32_t> lm_anchor_kf, res_kf, res_lm;
};

template <class T> static void rd(std::ifstream &f, std::vector<T> &v, size_t n) { v.resize(n); f.read((char *)v.data(), (std::streamsize)(n * sizeof(T))); }

static bool load(const std::string &path, Flat &p)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    int32_t hdr[3];
    f.read((char *)hdr, 12);
    p.n_kf = hdr[0]; p.n_lm = hdr[1]; p.n_res = hdr[2];
    rd(f, p.poses, 7 * (size_t)p.n_kf); rd(f, p.kf_const, (size_t)p.n_kf); rd(f, p.invdepth, (size_t)p.n_lm); rd(f, p.lm_anchor_kf, (size_t)p.n_lm);
    rd(f, p.lm_anchor_uv, 2 * (size_t)p.n_lm); rd(f, p.res_type, (size_t)p.n_res); rd(f, p.res_kf, (size_t)p.n_res); rd(f, p.res_lm, (size_t)p.n_res);
    rd(f, p.res_uv, 2 * (size_t)p.n_res); rd(f, p.res_sigma, (size_t)p.n_res); rd(f, p.calib_l, 4); rd(f, p.calib_r, 4); rd(f, p.T_rl, 7);
    return (bool)f;
}

// chi2err_ / isdepthpositive_ of a residual block of type t (OV2_RES_LEFT 0, RIGHT 1, RIGHT_ANCH 2)
static void cached(ceres::CostFunction *f, int t, double &chi2, bool &dpos)
{
    if (t == 0) { auto *e = static_cast<DirectLeftSE3::ReprojectionErrorKSE3AnchInvDepth *>(f); chi2 = e->chi2err_; dpos = e->isdepthpositive_; }
    else if (t == 1) { auto *e = static_cast<DirectLeftSE3::ReprojectionErrorRightCamKSE3AnchInvDepth *>(f); chi2 = e->chi2err_; dpos = e->isdepthpositive_; }
    else { auto *e = static_cast<DirectLeftSE3::ReprojectionErrorRightAnchCamKSE3AnchInvDepth *>(f); chi2 = e->chi2err_; dpos = e->isdepthpositive_; }
}

static int run(const std::string &in, const std::string &outdir, const std::string &tag)
{
    Flat p;
    if (!load(in, p)) { std::cerr << "cannot read " << in << "\n"; return 1; }
    const double mono_th = 5.9915;                                   // robust_mono_th_ of every shipped parameter file
    ceres::Problem problem;
    auto *loss_function = new ceres::LossFunctionWrapper(new ceres::HuberLoss(std::sqrt(mono_th)), ceres::TAKE_OWNERSHIP);   // optimizer.cpp:49
    auto ordering = new ceres::ParameterBlockOrdering;
    CalibParametersBlock calibpar(0, p.calib_l[0], p.calib_l[1], p.calib_l[2], p.calib_l[3]), rightcalibpar(0, p.calib_r[0], p.calib_r[1], p.calib_r[2], p.calib_r[3]);
    problem.AddParameterBlock(calibpar.values(), 4); ordering->AddElementToGroup(calibpar.values(), 1); problem.SetParameterBlockConstant(calibpar.values());
    problem.AddParameterBlock(rightcalibpar.values(), 4); ordering->AddElementToGroup(rightcalibpar.values(), 1); problem.SetParameterBlockConstant(rightcalibpar.values());
    PoseParametersBlock rlextrinpose;
    for (int i = 0; i < 7; i++) rlextrinpose.values()[i] = p.T_rl[i];
    problem.AddParameterBlock(rlextrinpose.values(), 7, new SE3LeftParameterization()); ordering->AddElementToGroup(rlextrinpose.values(), 1);
    problem.SetParameterBlockConstant(rlextrinpose.values());
    std::vector<PoseParametersBlock, Eigen::aligned_allocator<PoseParametersBlock>> poses((size_t)p.n_kf);
    for (int k = 0; k < p.n_kf; k++) {
        for (int i = 0; i < 7; i++) poses[k].values()[i] = p.poses[7 * (size_t)k + i];
        problem.AddParameterBlock(poses[k].values(), 7, new SE3LeftParameterization());          // :172
        ordering->AddElementToGroup(poses[k].values(), 1);
        if (p.kf_const[k]) problem.SetParameterBlockConstant(poses[k].values());                 // :176-185
    }
    std::vector<InvDepthParametersBlock, Eigen::aligned_allocator<InvDepthParametersBlock>> lms((size_t)p.n_lm);
    for (int l = 0; l < p.n_lm; l++) {
        lms[l] = InvDepthParametersBlock(l, p.lm_anchor_kf[l], 1. / p.invdepth[l]);
        problem.AddParameterBlock(lms[l].values(), 1); ordering->AddElementToGroup(lms[l].values(), 0);   // :266-267
    }
    std::vector<ceres::CostFunction *> fs((size_t)p.n_res);
    std::vector<ceres::ResidualBlockId> rids((size_t)p.n_res);
    for (int i = 0; i < p.n_res; i++) {
        const int l = p.res_lm[i], k = p.res_kf[i], ka = p.lm_anchor_kf[l];
        const double u = p.res_uv[2 * (size_t)i], v = p.res_uv[2 * (size_t)i + 1], ua = p.lm_anchor_uv[2 * (size_t)l], va = p.lm_anchor_uv[2 * (size_t)l + 1], sg = p.res_sigma[i];
        if (p.res_type[i] == 0) {                                                                   // :299-312, :365-376
            fs[i] = new DirectLeftSE3::ReprojectionErrorKSE3AnchInvDepth(u, v, ua, va, sg);
            rids[i] = problem.AddResidualBlock(fs[i], loss_function, calibpar.values(), poses[ka].values(), poses[k].values(), lms[l].values());
        } else if (p.res_type[i] == 1) {                                                            // :314-328
            fs[i] = new DirectLeftSE3::ReprojectionErrorRightCamKSE3AnchInvDepth(u, v, ua, va, sg);
            rids[i] = problem.AddResidualBlock(fs[i], loss_function, calibpar.values(), rightcalibpar.values(), poses[ka].values(), poses[k].values(),
                                               rlextrinpose.values(), lms[l].values());
        } else {                                                                                    // :269-289
            fs[i] = new DirectLeftSE3::ReprojectionErrorRightAnchCamKSE3AnchInvDepth(u, v, ua, va, sg);
            rids[i] = problem.AddResidualBlock(fs[i], loss_function, calibpar.values(), rightcalibpar.values(), rlextrinpose.values(), lms[l].values());
        }
    }
    ceres::Solver::Options options;                                                                 // :436-470
    options.linear_solver_ordering.reset(ordering);
    options.linear_solver_type = ceres::DENSE_SCHUR;
    options.trust_region_strategy_type = ceres::LEVENBERG_MARQUARDT;
    options.num_threads = 1;
    options.max_num_iterations = 5;
    options.function_tolerance = 1.e-3;
    options.max_solver_time_in_seconds = 1e6;                        // (0.2 / 0.4 s in the reference: off here so that the capture does not depend on the box)
    ceres::Solver::Summary summary;
    ceres::Solve(options, &problem, &summary);                                                     // :479

    auto dump_state = [&](const std::string &stage) {
        std::vector<double> po(7 * (size_t)p.n_kf), lam((size_t)p.n_lm);
        for (int k = 0; k < p.n_kf; k++) for (int i = 0; i < 7; i++) po[7 * (size_t)k + i] = poses[k].values()[i];
        for (int l = 0; l < p.n_lm; l++) lam[l] = lms[l].getInvDepth();
        npy_save(outdir + "/ba_" + tag + "_" + stage + "_poses.npy", "<f8", {(size_t)p.n_kf, 7}, po.data(), 8 * po.size());
        npy_save(outdir + "/ba_" + tag + "_" + stage + "_invdepth.npy", "<f8", {(size_t)p.n_lm}, lam.data(), 8 * lam.size());
    };
    auto dump_summary = [&](const std::string &stage, const ceres::Solver::Summary &s, double l2_done) {
        const double v[6] = {(double)s.iterations.size() - 1, (double)s.num_successful_steps, (double)s.termination_type, s.initial_cost, s.final_cost, l2_done};
        npy_save(outdir + "/ba_" + tag + "_" + stage + "_summary.npy", "<f8", {6}, v, 48);
    };
    dump_state("pass1");
    std::vector<double> chi2((size_t)p.n_res);
    std::vector<uint8_t> dpos((size_t)p.n_res), bad((size_t)p.n_res, 0);
    size_t nbbad = 0, left_rem = 0, right_rem = 0;
    for (int i = 0; i < p.n_res; i++) {                                                            // :492-594
        bool d;
        cached(fs[i], p.res_type[i], chi2[i], d);
        dpos[i] = d;
        if (chi2[i] > mono_th || !d) { bad[i] = 1; nbbad++; problem.RemoveResidualBlock(rids[i]); }   // apply_l2_after_robust_ = 1 in every shipped file
        else if (p.res_type[i] == 0) left_rem++;
        else if (p.res_type[i] == 1) right_rem++;
    }
    npy_save(outdir + "/ba_" + tag + "_pass1_chi2.npy", "<f8", {(size_t)p.n_res}, chi2.data(), 8 * chi2.size());
    npy_save(outdir + "/ba_" + tag + "_pass1_depthpos.npy", "|u1", {(size_t)p.n_res}, dpos.data(), dpos.size());
    dump_summary("pass1", summary, 0);
    bool l2 = false;
    ceres::Solver::Summary summary2;
    if (nbbad > 0) {                                                                               // :603-627
        if (left_rem > 0 && right_rem > 0) loss_function->Reset(nullptr, ceres::TAKE_OWNERSHIP);
        options.max_num_iterations = 10;
        options.function_tolerance = 1.e-3;
        ceres::Solve(options, &problem, &summary2);
        l2 = true;
        for (int i = 0; i < p.n_res; i++) {                                                        // :637-735
            if (bad[i]) continue;
            double c; bool d;
            cached(fs[i], p.res_type[i], c, d);
            if (c > mono_th || !d) bad[i] = 1;
        }
    }
    dump_state("final");
    npy_save(outdir + "/ba_" + tag + "_final_bad_obs.npy", "|u1", {(size_t)p.n_res}, bad.data(), bad.size());
    dump_summary("final", l2 ? summary2 : summary, l2 ? 1 : 0);
    std::printf("%s: pass 1 %d iterations (%s), %zu outliers, pass 2 %s\n", tag.c_str(), (int)summary.iterations.size() - 1,
                ceres::TerminationTypeToString(summary.termination_type), nbbad, l2 ? "run" : "skipped");
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::cerr << "usage: ov2_ref_capture_ba <ref_inputs dir> <out dir>\n"; return 2; }
    int rc = 0;
    for (const char *tag : {"kf8_mono", "kf8_stereo", "kf12_stereo", "kf50_mono", "kf50_stereo"})
        rc |= run(std::string(argv[1]) + "/ba_" + tag + ".bin", argv[2], tag);
    return rc;
}
