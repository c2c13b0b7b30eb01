// patch_usage_check.cpp -- the adapter calls integration/ov2slam_hip.patch inserts into the reference, with the same argument
// types (cv::Mat, std::vector<cv::Point2f>, cv::Rect, SlamParams-typed scalars), so that a change of an adapter's signature
// breaks HERE (g++ -fsyntax-only -DOV2_WITH_OPENCV against tests/fake_opencv, tests/test_integration_patch.py) and not in a
// reference tree nobody can build in this image.  One function per patched site; nothing is executed.
#include <iostream>
#include <unordered_map>
#include "slam_gpu.hpp"

// src/ov2slam.cpp: SlamManager constructor
std::shared_ptr<ov2::SlamGpu> site_ov2slam(double img_w, double img_h, int nbmaxkps, int nmaxdist, double dmaxquality, int nfast_th, int nmax_iter,
                                          float fmax_px_precision, int nklt_win_size, int nklt_pyr_lvl, int nklt_err, float fmax_fbklt_dist,
                                          bool use_clahe, float fclahe_val, float robust_mono_th, bool apply_l2_after_robust)
{
    std::shared_ptr<ov2::SlamGpu> pgpu_;
    pgpu_.reset(new ov2::SlamGpu(0, (int)img_w, (int)img_h, nbmaxkps, nmaxdist, dmaxquality, nfast_th, nmax_iter, fmax_px_precision, nklt_win_size,
                                 nklt_pyr_lvl, nklt_err, fmax_fbklt_dist, use_clahe, fclahe_val, robust_mono_th, apply_l2_after_robust));
    return pgpu_;
}

// src/visual_front_end.cpp: preprocessImage / kltTracking
bool site_front_end(ov2::SlamGpu &gpu, cv::Mat &img_raw, bool klt_use_prior)
{
    if (!gpu.trk->preprocessImage(img_raw)) std::cerr << ov2_last_error();
    std::vector<cv::Point2f> vpx, vpri;
    std::vector<uint8_t> vhasprior;
    std::vector<bool> vkpstatus;
    bool bp3preq = false;
    auto &trk = *gpu.trk;
    trk.kltTracking(vpx, vpri, vhasprior, klt_use_prior, vkpstatus, bp3preq);
    if (trk.lastError() != OV2_OK) std::cerr << trk.lastErrorMessage();
    return bp3preq;
}

// src/map_manager.cpp: extractKeypoints / stereoMatching
size_t site_map_manager(ov2::SlamGpu &gpu, int nmaxdist, const std::vector<cv::Point2f> &vpts, const cv::Rect &roi_rect, int nklt_win_size,
                        int nklt_pyr_lvl, int nklt_err, float fmax_fbklt_dist, bool bdo_stereo_rect, const double *Frl, bool pinhole,
                        const double rK[4], const std::vector<double> &rD)
{
    std::vector<cv::Point2f> vnewpts = gpu.extract.detectGridFAST(gpu.frontend, gpu.trk->curPyr(), nmaxdist, vpts, roi_rect);
    vnewpts = gpu.extract.detectSingleScale(gpu.frontend, gpu.trk->curPyr(), nmaxdist, vpts, roi_rect);
    std::vector<cv::Point2f> vlkps, vlunpx, vpri3d, vrkps;
    std::vector<uint8_t> vhasprior;
    std::vector<bool> vstereo_ok;
    gpu.track.stereoMatching(gpu.mapper, gpu.kf_left.get(), gpu.kf_right.get(), nklt_win_size, nklt_pyr_lvl, nklt_err, fmax_fbklt_dist,
                             bdo_stereo_rect, Frl, pinhole ? OV2_CAM_PINHOLE : OV2_CAM_FISHEYE, rK, rD, vlkps, vlunpx, vpri3d, vhasprior, vrkps,
                             vstereo_ok);
    return vnewpts.size() + vrkps.size();
}

// src/mapper.cpp: the keyframe's pyramids
int site_mapper(ov2::SlamGpu &gpu, const cv::Mat &imleftraw, const cv::Mat &imrightraw, int nklt_win_size, int nklt_pyr_lvl, float fclahe_val, bool use_clahe)
{
    int rcl, rcr;
    if (use_clahe) {
        rcl = gpu.kf_left.buildClahe(gpu.mapper, imleftraw, nklt_win_size, nklt_pyr_lvl, fclahe_val);
        rcr = gpu.kf_right.buildClahe(gpu.mapper, imrightraw, nklt_win_size, nklt_pyr_lvl, fclahe_val);
    } else {
        rcl = gpu.kf_left.build(gpu.mapper, imleftraw, nklt_win_size, nklt_pyr_lvl);
        rcr = gpu.kf_right.build(gpu.mapper, imrightraw, nklt_win_size, nklt_pyr_lvl);
    }
    return rcl != OV2_OK || rcr != OV2_OK;
}

// src/optimizer.cpp: localBA / signalStopLocalBA
bool site_optimizer(ov2::SlamGpu &gpu, double *pose_values, double invdepth, double u, double v, double scale, double max_solver_time,
                    bool buse_robust_cost)
{
    ov2::FlatProblem fp;
    std::unordered_map<int, int> map_kfid_fpidx, map_lmid_fpidx;
    fp.calib_l[0] = 1; fp.calib_r[3] = 1;
    std::copy(pose_values, pose_values + 7, fp.T_rl);
    map_kfid_fpidx[3] = fp.addKeyframe(pose_values, false);
    fp.kf_const[map_kfid_fpidx.at(3)] = 1;
    map_kfid_fpidx[4] = fp.addKeyframe(pose_values, true);
    map_lmid_fpidx[7] = fp.addLandmark(invdepth, map_kfid_fpidx.at(3), u, v);
    fp.addResidual(OV2_RES_RIGHT_ANCH, map_kfid_fpidx.at(3), map_lmid_fpidx.at(7), u, v, std::pow(2., scale));
    fp.addResidual(OV2_RES_LEFT, map_kfid_fpidx.at(4), map_lmid_fpidx.at(7), u, v, std::pow(2., scale));
    fp.addResidual(OV2_RES_RIGHT, map_kfid_fpidx.at(4), map_lmid_fpidx.at(7), u, v, std::pow(2., scale));
    gpu.opt.setMaxSolverTime(max_solver_time);
    ov2::LocalBAResult hipres;
    hipres = gpu.opt.solveLocalBA(gpu.estimator, fp, buse_robust_cost);
    if (hipres.ok) {
        for (const auto &id_idx : map_kfid_fpidx) std::copy(hipres.poses.begin() + 7 * id_idx.second, hipres.poses.begin() + 7 * id_idx.second + 7, pose_values);
        invdepth = hipres.invdepth[map_lmid_fpidx.at(7)];
        for (size_t i = 0; i < hipres.bad_obs.size(); i++) if (hipres.bad_obs[i]) return false;
    } else {
        std::cerr << hipres.error;
    }
    gpu.opt.signalStopLocalBA();
    gpu.opt.clearStopLocalBA();
    return hipres.ok && invdepth > 0;
}

// src/optimizer.cpp: the buse_inv_depth: 0 form of localBA, looseBA / fullBA / structureOnlyBA, the per-thread fallback flag
bool site_optimizer_other(ov2::SlamGpu &gpu, double *pose_values, double *point_values, double u, double v, double scale, bool buse_robust_cost)
{
    const bool bhipwalk = !ov2::SlamGpu::forceCeres();
    ov2::FlatXYZProblem fpx;
    ov2::FlatProblem fp;
    std::unordered_map<int, int> map_kfid_fpidx, map_lmid_fpidx;
    std::copy(fp.calib_l, fp.calib_l + 4, fpx.calib_l);
    std::copy(pose_values, pose_values + 7, fpx.T_rl);
    map_kfid_fpidx[3] = fp.addKeyframe(pose_values, false);
    fpx.addKeyframe(pose_values, false);
    fp.kf_const[map_kfid_fpidx.at(3)] = 1; fpx.kf_const[map_kfid_fpidx.at(3)] = 1;
    map_lmid_fpidx[7] = fpx.addPoint(point_values);
    fpx.addResidual(OV2_XYZ_LEFT, map_kfid_fpidx.at(3), map_lmid_fpidx.at(7), u, v, std::pow(2., scale));
    fpx.addResidual(OV2_XYZ_RIGHT, map_kfid_fpidx.at(3), map_lmid_fpidx.at(7), u, v, std::pow(2., scale));
    ov2::LocalBAResult hipres = gpu.opt.solveLocalBAXYZ(gpu.estimator, fpx, buse_robust_cost);
    if (hipres.ok) std::copy(hipres.invdepth.begin() + 3 * map_lmid_fpidx.at(7), hipres.invdepth.begin() + 3 * map_lmid_fpidx.at(7) + 3, point_values);
    hipres = gpu.lc_opt.solveLooseBA(gpu.threadContext(), fp, buse_robust_cost);
    hipres = gpu.lc_opt.solveFullBA(gpu.threadContext(), fp, buse_robust_cost);
    ov2::FlatStructureProblem sp;
    map_kfid_fpidx[5] = sp.addKeyframe(pose_values);
    map_lmid_fpidx[9] = sp.addPoint(point_values);
    sp.addResidual(OV2_XYZ_LEFT, map_kfid_fpidx.at(5), map_lmid_fpidx.at(9), u, v, std::pow(2., scale));
    std::vector<double> vhipxyz;
    const bool bhipdone = gpu.lc_opt.solveStructureOnlyBA(gpu.threadContext(), sp, vhipxyz);
    ov2::SlamGpu::forceCeres() = true;
    ov2::SlamGpu::forceCeres() = false;
    gpu.setDeterministicBA(true);
    return bhipwalk && bhipdone && hipres.ok;
}

// src/multi_view_geometry.cpp: ceresPnP (the reference passes std::vector<Eigen::Vector2d / Vector3d>: packed doubles)
bool site_ceres_pnp(const std::vector<double> &vunkps, const std::vector<double> &vwpts, const std::vector<int> &vscales, double *pose_values, int nmaxiter,
                    float chi2th, bool buse_robust, bool bapply_l2_after_robust, float fx, float fy, float cx, float cy, std::vector<int> &voutliersidx)
{
    if (ov2::SlamGpu::global() == nullptr) return false;
    std::vector<int> vhipoutliers;
    bool bhiplibok = true;
    std::string hiperr;
    const bool bhipsuccess = ov2::ceresPnP(ov2::SlamGpu::global()->threadContext(), vunkps.data(), vwpts.data(), vscales.data(), vscales.size(), pose_values,
                                           nmaxiter, chi2th, buse_robust, bapply_l2_after_robust, fx, fy, cx, cy, vhipoutliers, 0.005, &bhiplibok, &hiperr);
    if (bhiplibok) voutliersidx.insert(voutliersidx.end(), vhipoutliers.begin(), vhipoutliers.end());
    else std::cerr << hiperr;
    return bhipsuccess;
}

// src/visual_front_end.cpp: btrack_keyframetoframe (the keyframe's pyramid on the device, both fbKltTracking calls of kltTrackingFromKF)
size_t site_track_from_kf(ov2::SlamGpu &gpu, const cv::Mat &cur_img, bool use_clahe, int nklt_win_size, int nklt_pyr_lvl, float fclahe_val, int nklt_err,
                          float fmax_fbklt_dist, std::vector<cv::Point2f> &vkps, std::vector<cv::Point2f> &vpriors)
{
    const int rckf = use_clahe ? gpu.kf_front.buildClahe(gpu.frontend, cur_img, nklt_win_size, nklt_pyr_lvl, fclahe_val)
                               : gpu.kf_front.build(gpu.frontend, cur_img, nklt_win_size, nklt_pyr_lvl);
    if (rckf != OV2_OK) std::cerr << ov2_last_error();
    std::vector<bool> vkpstatus;
    gpu.track.fbKltTracking(gpu.frontend, gpu.kf_front.get(), gpu.trk->curPyr(), nklt_win_size, nklt_pyr_lvl, nklt_err, fmax_fbklt_dist, vkps, vpriors, vkpstatus);
    return vkpstatus.size();
}

// src/map_manager.cpp: detectors on a host image (do_klt: 0)
size_t site_detect_host(ov2::SlamGpu &gpu, const cv::Mat &im, int nmaxdist, const std::vector<cv::Point2f> &vpts, const cv::Rect &roi_rect)
{
    std::vector<cv::Point2f> vnewpts = gpu.extract.detectGridFAST(gpu.frontend, im, nmaxdist, vpts, roi_rect);
    vnewpts = gpu.extract.detectSingleScale(gpu.frontend, im, nmaxdist, vpts, roi_rect);
    return vnewpts.size();
}
