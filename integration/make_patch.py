#!/usr/bin/env python3
"""Generates integration/ov2slam_hip.patch: the reference-side binding of libov2slam_hip.so as text that `git apply` accepts.

Every edit is an insertion (or a small wrap) under `#ifdef OV2SLAM_HIP`, anchored on a string of the reference that must occur
exactly once; the default build of the reference is unchanged.  Run from anywhere:  python integration/make_patch.py
[reference root, default /root/reference].  tests/test_integration_patch.py regenerates the patch and checks that the committed
file is what this script produces and that `git apply --check` accepts it on a scratch copy of the reference tree.

What the patch wires (INTEGRATION.md has the reasoning):
  CMakeLists.txt                -DOV2SLAM_AMD_ROOT=<checkout of this repository> links the library and defines OV2SLAM_HIP
  include/slam_params.hpp       std::shared_ptr<ov2::SlamGpu> pgpu_ : contexts, tracker and adapters, reachable from every class
  src/ov2slam.cpp               constructs it next to FeatureExtractor / FeatureTracker (before the worker threads start)
  src/visual_front_end.cpp      preprocessImage (:1143-1177) and kltTracking (:132-275) -> ov2::FrameTracker
  src/map_manager.cpp           extractKeypoints' detectors (:312-320) on the tracker's pyramid; stereoMatching (:367-611): the
                                map walk stays, SAD priors + both fbKltTracking calls + the epipolar gate -> ov2_stereo_match
  src/mapper.cpp                the keyframe's two pyramids (:75-81) on the mapper thread's context
  src/optimizer.cpp             localBA (:43-897): the map walk also fills an ov2::FlatProblem, the two ceres::Solve calls and
                                the outlier loops between them -> ov2_local_ba; signalStopLocalBA reaches the library's live flag
Not wired (they keep OpenCV / Ceres): btrack_keyframetoframe, buse_inv_depth: 0, looseBA / fullBA / ceresPnP (entry points exist).
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = ["CMakeLists.txt", "include/slam_params.hpp", "src/ov2slam.cpp", "src/visual_front_end.cpp", "src/map_manager.cpp",
         "src/mapper.cpp", "src/optimizer.cpp"]


def span(s, anchor):
    """(start, end) of the one place where `anchor` occurs in s -- trailing blanks of the reference's lines are not part of
    an anchor's spelling (the reference has many), everything else is matched literally"""
    import re
    pat = r"[ \t]*\n".join(re.escape(line.rstrip()) for line in anchor.split("\n"))
    m = list(re.finditer(pat, s))
    assert len(m) == 1, "anchor must occur exactly once (%d): %r" % (len(m), anchor[:80])
    return m[0].start(), m[0].end()


def once(s, anchor):
    return span(s, anchor)[0]


def after(s, anchor, text):
    i = span(s, anchor)[1]
    return s[:i] + text + s[i:]


def before(s, anchor, text):
    i = span(s, anchor)[0]
    return s[:i] + text + s[i:]


def replace(s, anchor, text):
    """the matched text is kept verbatim wherever `text` repeats a line of the anchor: only the lines that differ are edits"""
    i, j = span(s, anchor)
    old = s[i:j].split("\n")
    new = text.split("\n")
    by_stripped = {}
    for line in old:
        by_stripped.setdefault(line.rstrip(), line)
    new = [by_stripped.get(line.rstrip(), line) if line.rstrip() in by_stripped else line for line in new]
    return s[:i] + "\n".join(new) + s[j:]


def edit_cmake(s):
    return after(s, """    PRIVATE
      ${OpenCV_LIBS}
      Ceres::ceres
)
""", """
# MI355X front-end + local-BA hot path: cmake -DOV2SLAM_AMD_ROOT=<checkout of ov2slam_amd, library built with make -C ov2slam_amd/csrc>
if (OV2SLAM_AMD_ROOT)
  find_library(OV2SLAM_HIP_LIB ov2slam_hip PATHS ${OV2SLAM_AMD_ROOT}/ov2slam_amd NO_DEFAULT_PATH)
  if (NOT OV2SLAM_HIP_LIB)
    message(FATAL_ERROR "libov2slam_hip.so not found under ${OV2SLAM_AMD_ROOT}/ov2slam_amd")
  endif ()
  target_include_directories(${PROJECT_NAME} PUBLIC ${OV2SLAM_AMD_ROOT}/include ${OV2SLAM_AMD_ROOT}/ov2slam_amd/host)
  target_compile_definitions(${PROJECT_NAME} PUBLIC OV2SLAM_HIP OV2_WITH_OPENCV)
  target_link_libraries(${PROJECT_NAME} PUBLIC ${OV2SLAM_HIP_LIB})
endif (OV2SLAM_AMD_ROOT)
""")


def edit_slam_params(s):
    s = after(s, """#include "profiler.hpp"
""", """
#ifdef OV2SLAM_HIP
#include <memory>
namespace ov2 { struct SlamGpu; }      // slam_gpu.hpp of ov2slam_amd/host
#endif
""")
    return after(s, """    void reset();
""", """
#ifdef OV2SLAM_HIP
    // libov2slam_hip.so: one context per thread, the per-frame tracker, the detector / tracker / optimizer adapters
    std::shared_ptr<ov2::SlamGpu> pgpu_;
#endif
""")


def edit_ov2slam(s):
    s = after(s, """#include "ov2slam.hpp"
""", """
#ifdef OV2SLAM_HIP
#include "slam_gpu.hpp"
#endif
""")
    return before(s, """    // Map Manager will handle Keyframes / MapPoints
    pmap_.reset( new MapManager(""", """#ifdef OV2SLAM_HIP
    // Before any worker thread exists: the tracker captures its per-frame hipGraphs here
    pslamstate_->pgpu_.reset( new ov2::SlamGpu(0, (int)pcalib_model_left_->img_w_, (int)pcalib_model_left_->img_h_,
                                pslamstate_->nbmaxkps_, pslamstate_->nmaxdist_, pslamstate_->dmaxquality_, pslamstate_->nfast_th_,
                                pslamstate_->nmax_iter_, pslamstate_->fmax_px_precision_,
                                pslamstate_->nklt_win_size_, pslamstate_->nklt_pyr_lvl_,
                                pslamstate_->nklt_err_, pslamstate_->fmax_fbklt_dist_,
                                pslamstate_->use_clahe_, pslamstate_->fclahe_val_,
                                pslamstate_->robust_mono_th_, pslamstate_->apply_l2_after_robust_) );
#endif

""")


def edit_front_end(s):
    s = after(s, """#include "visual_front_end.hpp"
#include "multi_view_geometry.hpp"
""", """
#ifdef OV2SLAM_HIP
#include "slam_gpu.hpp"
#endif
""")
    s = after(s, """void VisualFrontEnd::kltTracking()
{
    if( pslamstate_->debug_ || pslamstate_->log_timings_ )
        Profiler::Start("2.FE_TM_KLT-Tracking");
""", """
#ifdef OV2SLAM_HIP
    if( !pslamstate_->btrack_keyframetoframe_ )
    {
        // ONE list: keypoints with a usable 3-D prior carry it (flag 1), the others start from their own position.  Both
        // fbKltTracking calls below, the hand-over of the prior tracks the first one lost and the bp3preq_ rule run inside
        // ov2_tracker_klt, in one enqueue on the pyramids preprocessImage() left on the device.
        std::vector<int> vids;
        std::vector<cv::Point2f> vpx, vpri;
        std::vector<uint8_t> vhasprior;
        vids.reserve(pcurframe_->nbkps_); vpx.reserve(pcurframe_->nbkps_);
        vpri.reserve(pcurframe_->nbkps_); vhasprior.reserve(pcurframe_->nbkps_);

        for( const auto &it : pcurframe_->mapkps_ )
        {
            auto &kp = it.second;
            cv::Point2f prior = kp.px_;
            uint8_t hasprior = 0;
            if( pslamstate_->klt_use_prior_ && kp.is3d_ ) {
                cv::Point2f projpx = pcurframe_->projWorldToImageDist(pmap_->map_plms_.at(kp.lmid_)->getPoint());
                if( pcurframe_->isInImage(projpx) ) {
                    prior = projpx;
                    hasprior = 1;
                }
            }
            vids.push_back(kp.lmid_); vpx.push_back(kp.px_);
            vpri.push_back(prior); vhasprior.push_back(hasprior);
        }

        std::vector<bool> vkpstatus;
        bool bp3preq = false;
        auto &trk = *pslamstate_->pgpu_->trk;
        trk.kltTracking(vpx, vpri, vhasprior, pslamstate_->klt_use_prior_, vkpstatus, bp3preq);
        if( trk.lastError() != OV2_OK )
            std::cerr << "\\n [ov2slam_hip] kltTracking : " << trk.lastErrorMessage();
        if( bp3preq )
            bp3preq_ = true;

        for( size_t i = 0 ; i < vids.size() ; i++ ) {
            if( vkpstatus.at(i) ) {
                pcurframe_->updateKeypoint(vids.at(i), vpri.at(i));
            } else {
                pmap_->removeObsFromCurFrameById(vids.at(i));
            }
        }

        if( pslamstate_->debug_ || pslamstate_->log_timings_ )
            Profiler::StopAndDisplay(pslamstate_->debug_, "2.FE_TM_KLT-Tracking");
        return;
    }
#endif
""")
    s = after(s, """void VisualFrontEnd::preprocessImage(cv::Mat &img_raw)
{
    if( pslamstate_->debug_ || pslamstate_->log_timings_ )
        Profiler::Start("2.FE_TM_preprocessImage");
""", """
#ifdef OV2SLAM_HIP
    if( pslamstate_->do_klt_ && !pslamstate_->btrack_keyframetoframe_ )
    {
        // H2D + CLAHE + pyramid of the new frame in one asynchronous enqueue; the tracker swaps its prev / cur pyramids
        // itself.  The equalised image stays on the device (level 0 of the pyramid): cur_img_ keeps the raw frame for the
        // visualiser and the descriptor code.
        if( !pslamstate_->pgpu_->trk->preprocessImage(img_raw) )
            std::cerr << "\\n [ov2slam_hip] preprocessImage : " << ov2_last_error();
        cv::swap(cur_img_, prev_img_);
        cur_img_ = img_raw;

        if( pslamstate_->debug_ || pslamstate_->log_timings_ )
            Profiler::StopAndDisplay(pslamstate_->debug_, "2.FE_TM_preprocessImage");
        return;
    }
#endif
""")
    return s


def edit_map_manager(s):
    s = after(s, """#include "map_manager.hpp"
""", """
#ifdef OV2SLAM_HIP
#include "slam_gpu.hpp"
#endif
""")
    # detectors: on level 0 of the pyramid the front end has just built (no upload)
    s = replace(s, """            vnewpts = pfeatextract_->detectGridFAST(im, pslamstate_->nmaxdist_, vpts, pcurframe_->pcalib_leftcam_->roi_rect_);
""", """#ifdef OV2SLAM_HIP
            vnewpts = pslamstate_->pgpu_->extract.detectGridFAST(pslamstate_->pgpu_->frontend, pslamstate_->pgpu_->trk->curPyr(),
                            pslamstate_->nmaxdist_, vpts, pcurframe_->pcalib_leftcam_->roi_rect_);
#else
            vnewpts = pfeatextract_->detectGridFAST(im, pslamstate_->nmaxdist_, vpts, pcurframe_->pcalib_leftcam_->roi_rect_);
#endif
""")
    s = replace(s, """            vnewpts = pfeatextract_->detectSingleScale(im, pslamstate_->nmaxdist_, vpts, pcurframe_->pcalib_leftcam_->roi_rect_);
""", """#ifdef OV2SLAM_HIP
            vnewpts = pslamstate_->pgpu_->extract.detectSingleScale(pslamstate_->pgpu_->frontend, pslamstate_->pgpu_->trk->curPyr(),
                            pslamstate_->nmaxdist_, vpts, pcurframe_->pcalib_leftcam_->roi_rect_);
#else
            vnewpts = pfeatextract_->detectSingleScale(im, pslamstate_->nmaxdist_, vpts, pcurframe_->pcalib_leftcam_->roi_rect_);
#endif
""")
    # stereoMatching: the per-keypoint SAD scan reads host pyramid levels -- ov2_stereo_match does it for every keypoint without a 3-D prior
    s = before(s, """            float xprior = -1.;
            float l1err;
""", """#ifndef OV2SLAM_HIP         // (rectified pairs: the SAD prior of a keypoint without 3-D prior is found inside ov2_stereo_match)
""")
    s = after(s, """            if( xprior >= 0 && xprior <= kp.px_.x ) {
                priorpt.x = xprior;
            }
""", """#endif
""")
    s = before(s, """    // Storing good tracks
    std::vector<cv::Point2f> vgoodrkps;""", """#ifdef OV2SLAM_HIP
    {
        // The lists above in ONE call on the mapper thread's context: SAD priors on the coarsest level (rectified pairs), both
        // fbKltTracking calls (the 3-D-prior tracks that fail on two levels retry on the full pyramid from the first call's
        // forward result), the epipolar gate.  Mapper::run has just built both pyramids (kf_left / kf_right).
        auto &gpu = *pslamstate_->pgpu_;
        std::vector<int> vids(v3dkpids);
        std::vector<cv::Point2f> vlkps(v3dkps), vpri3d(v3dpriors), vlunpx, vrkps;
        std::vector<uint8_t> vhasprior(v3dkps.size(), 1);
        vids.insert(vids.end(), vkpids.begin(), vkpids.end());
        vlkps.insert(vlkps.end(), vkps.begin(), vkps.end());
        vpri3d.insert(vpri3d.end(), vkps.begin(), vkps.end());
        vhasprior.resize(vlkps.size(), 0);
        vlunpx.reserve(vids.size());
        for( const auto &id : vids )
            vlunpx.push_back(frame.getKeypointById(id).unpx_);

        const auto &pcalr = frame.pcalib_rightcam_;
        const double rK[4] = {pcalr->fx_, pcalr->fy_, pcalr->cx_, pcalr->cy_};
        const std::vector<double> rD(pcalr->D_.data(), pcalr->D_.data() + 4);
        const Eigen::Matrix<double,3,3,Eigen::RowMajor> Frl = frame.Frl_;
        std::vector<bool> vstereo_ok;

        gpu.track.stereoMatching(gpu.mapper, gpu.kf_left.get(), gpu.kf_right.get(),
                    pslamstate_->nklt_win_size_, pslamstate_->nklt_pyr_lvl_,
                    pslamstate_->nklt_err_, pslamstate_->fmax_fbklt_dist_,
                    pslamstate_->bdo_stereo_rect_, Frl.data(),
                    pcalr->model_ == CameraCalibration::Pinhole ? OV2_CAM_PINHOLE : OV2_CAM_FISHEYE, rK, rD,
                    vlkps, vlunpx, vpri3d, vhasprior, vrkps, vstereo_ok);

        size_t nbstereo = 0;
        for( size_t i = 0 ; i < vids.size() ; i++ ) {
            if( vstereo_ok.at(i) ) {
                frame.updateKeypointStereo(vids.at(i), vrkps.at(i));
                nbstereo++;
            }
        }

        if( pslamstate_->debug_ )
            std::cout << "\\n \\t>>> Nb of stereo tracks: " << nbstereo << " out of " << vids.size() << "\\n";

        if( pslamstate_->debug_ || pslamstate_->log_timings_ )
            Profiler::StopAndDisplay(pslamstate_->debug_, "1.KF_stereoMatching");
        return;
    }
#endif

""")
    return s


def edit_mapper(s):
    s = after(s, """#include "mapper.hpp"
#include "opencv2/video/tracking.hpp"
""", """
#ifdef OV2SLAM_HIP
#include "slam_gpu.hpp"
#endif
""")
    s = before(s, """                cv::Mat imright;
                if( pslamstate_->use_clahe_ ) {
                    pmap_->ptracker_->pclahe_->apply(kf.imrightraw_, imright);""", """#ifdef OV2SLAM_HIP
                {
                    // both pyramids of the keyframe on THIS thread's context, from the raw images the keyframe queued: the front
                    // end's pyramid pair has moved on by the time a queued keyframe is processed
                    auto &gpu = *pslamstate_->pgpu_;
                    int rcl, rcr;
                    if( pslamstate_->use_clahe_ ) {
                        rcl = gpu.kf_left.buildClahe(gpu.mapper, kf.imleftraw_, pslamstate_->nklt_win_size_, pslamstate_->nklt_pyr_lvl_, pslamstate_->fclahe_val_);
                        rcr = gpu.kf_right.buildClahe(gpu.mapper, kf.imrightraw_, pslamstate_->nklt_win_size_, pslamstate_->nklt_pyr_lvl_, pslamstate_->fclahe_val_);
                    } else {
                        rcl = gpu.kf_left.build(gpu.mapper, kf.imleftraw_, pslamstate_->nklt_win_size_, pslamstate_->nklt_pyr_lvl_);
                        rcr = gpu.kf_right.build(gpu.mapper, kf.imrightraw_, pslamstate_->nklt_win_size_, pslamstate_->nklt_pyr_lvl_);
                    }
                    if( rcl != OV2_OK || rcr != OV2_OK )
                        std::cerr << "\\n [ov2slam_hip] keyframe pyramids : " << ov2_last_error();
                    else
                        pmap_->stereoMatching(*pnewkf, kf.vpyr_imleft_, kf.vpyr_imright_);
                }
#else
""")
    s = after(s, """                pmap_->stereoMatching(*pnewkf, kf.vpyr_imleft_, vpyr_imright);
""", """#endif
""")
    return s


def edit_optimizer(s):
    """localBA (the first function of the file: the anchors are searched inside its text only) + signalStopLocalBA"""
    s = after(s, """#include "ceres_parametrization.hpp"
""", """
#ifdef OV2SLAM_HIP
#include "slam_gpu.hpp"
#endif
""")
    lo = once(s, "void Optimizer::localBA(Frame &newframe, const bool buse_robust_cost)")
    hi = once(s, "void Optimizer::looseBA(int inikfid, const int nkfid, const bool buse_robust_cost)")
    head, f, tail = s[:lo], s[lo:hi], s[hi:]

    f = after(f, """    auto ordering = new ceres::ParameterBlockOrdering;
""", """
#ifdef OV2SLAM_HIP
    // The same problem in the flat form libov2slam_hip.so takes: one push per AddParameterBlock / AddResidualBlock below
    // (inverse-depth form; buse_inv_depth: 0 keeps Ceres).
    ov2::FlatProblem fp;
    std::unordered_map<int,int> map_kfid_fpidx, map_lmid_fpidx;
    std::vector<std::pair<int,int>> vfp_kfid_lmid;      // per flat residual block: (kfid, lmid) ...
    std::vector<int> vfp_list;                           // ... and its list: 0 left, 1 right, 2 right of the anchor
    bool bhipdone = false;
    ov2::LocalBAResult hipres;
#endif
""")
    f = after(f, """    problem.SetParameterBlockConstant(calibpar.values());
""", """
#ifdef OV2SLAM_HIP
    fp.calib_l[0] = pcalibleft->fx_; fp.calib_l[1] = pcalibleft->fy_; fp.calib_l[2] = pcalibleft->cx_; fp.calib_l[3] = pcalibleft->cy_;
#endif
""")
    f = after(f, """        problem.SetParameterBlockConstant(rlextrinpose.values());
""", """
#ifdef OV2SLAM_HIP
        fp.calib_r[0] = pcalibright->fx_; fp.calib_r[1] = pcalibright->fy_; fp.calib_r[2] = pcalibright->cx_; fp.calib_r[3] = pcalibright->cy_;
        std::copy(rlextrinpose.values(), rlextrinpose.values() + 7, fp.T_rl);
#endif
""")
    # keyframes of the covisibility walk (free or constant, decided right below)
    f = before(f, """        // For those to optimize, get their 3D MPs
        // for the others, set them as constant""", """#ifdef OV2SLAM_HIP
        map_kfid_fpidx[kfid] = fp.addKeyframe(map_id_posespar_.at(kfid).values(), false);
#endif

""")
    f = after(f, """            set_cstkfids.insert(kfid);
            problem.SetParameterBlockConstant(map_id_posespar_.at(kfid).values());
            all_cst = true;
""", """#ifdef OV2SLAM_HIP
            fp.kf_const[map_kfid_fpidx.at(kfid)] = 1;
#endif
""")
    # observing keyframes outside the covisibility window: constant
    f = after(f, """                set_cstkfids.insert(kfid);
                problem.SetParameterBlockConstant(map_id_posespar_.at(kfid).values());
""", """#ifdef OV2SLAM_HIP
                map_kfid_fpidx[kfid] = fp.addKeyframe(map_id_posespar_.at(kfid).values(), true);
#endif
""")
    # anchored inverse depth + the anchor's right-camera block
    f = after(f, """                    problem.AddParameterBlock(map_id_invptspar_.at(lmid).values(), 1);
                    ordering->AddElementToGroup(map_id_invptspar_.at(lmid).values(), 0);
""", """#ifdef OV2SLAM_HIP
                    map_lmid_fpidx[lmid] = fp.addLandmark(map_id_invptspar_.at(lmid).getInvDepth(), map_kfid_fpidx.at(kfanchid), unanch_u, unanch_v);
#endif
""")
    f = after(f, """                        vanchright_reprojerr_kfid_lmid.push_back(std::make_pair(f, std::make_pair(rid, std::make_pair(kfid,lmid))));
""", """#ifdef OV2SLAM_HIP
                        fp.addResidual(OV2_RES_RIGHT_ANCH, map_kfid_fpidx.at(kfid), map_lmid_fpidx.at(lmid), kp.runpx_.x, kp.runpx_.y, std::pow(2.,kp.scale_));
                        vfp_kfid_lmid.push_back(std::make_pair(kfid,lmid)); vfp_list.push_back(2);
#endif
""")
    # stereo observation: left + right block
    f = after(f, """                    vreprojerr_kfid_lmid.push_back(std::make_pair(f, std::make_pair(rid, std::make_pair(kfid, lmid))));
""", """#ifdef OV2SLAM_HIP
                    fp.addResidual(OV2_RES_LEFT, map_kfid_fpidx.at(kfid), map_lmid_fpidx.at(lmid), kp.unpx_.x, kp.unpx_.y, std::pow(2.,kp.scale_));
                    vfp_kfid_lmid.push_back(std::make_pair(kfid,lmid)); vfp_list.push_back(0);
                    fp.addResidual(OV2_RES_RIGHT, map_kfid_fpidx.at(kfid), map_lmid_fpidx.at(lmid), kp.runpx_.x, kp.runpx_.y, std::pow(2.,kp.scale_));
                    vfp_kfid_lmid.push_back(std::make_pair(kfid,lmid)); vfp_list.push_back(1);
#endif
""")
    # mono observation
    f = after(f, """                vreprojerr_kfid_lmid.push_back(std::make_pair(f, std::make_pair(rid, std::make_pair(kfid,kp.lmid_))));
""", """#ifdef OV2SLAM_HIP
                if( pslamstate_->buse_inv_depth_ ) {
                    fp.addResidual(OV2_RES_LEFT, map_kfid_fpidx.at(kfid), map_lmid_fpidx.at(lmid), kp.unpx_.x, kp.unpx_.y, std::pow(2.,kp.scale_));
                    vfp_kfid_lmid.push_back(std::make_pair(kfid,lmid)); vfp_list.push_back(0);
                }
#endif
""")
    # gauge fix
    f = after(f, """            problem.SetParameterBlockConstant(map_id_posespar_.at(it->first).values());
            set_cstkfids.insert(it->first);
""", """#ifdef OV2SLAM_HIP
            fp.kf_const[map_kfid_fpidx.at(it->first)] = 1;
#endif
""")
    # the solve stage: both ceres::Solve calls + the outlier logic between them in ONE library call on the estimator thread's context
    f = replace(f, """    ceres::Solver::Summary summary;
    ceres::Solve(options, &problem, &summary);

    if( pslamstate_->debug_ )
        std::cout << summary.FullReport() << std::endl;

    if( pslamstate_->debug_ || pslamstate_->log_timings_ )
        Profiler::StopAndDisplay(pslamstate_->debug_, "2.BA_Optimize");
""", """    ceres::Solver::Summary summary;
#ifdef OV2SLAM_HIP
    if( pslamstate_->buse_inv_depth_ )
    {
        auto &gpu = *pslamstate_->pgpu_;
        // (same wall-clock budget as the two solves below: t for the robust pass, t / 2 for the L2 pass)
        gpu.opt.setMaxSolverTime(options.max_solver_time_in_seconds);
        hipres = gpu.opt.solveLocalBA(gpu.estimator, fp, buse_robust_cost);
        bhipdone = hipres.ok;
        if( bhipdone ) {
            // poses / inverse depths into the parameter blocks the write-back at the end of this function reads
            for( const auto &id_idx : map_kfid_fpidx )
                std::copy(hipres.poses.begin() + 7 * id_idx.second, hipres.poses.begin() + 7 * id_idx.second + 7, map_id_posespar_.at(id_idx.first).values());
            for( const auto &id_idx : map_lmid_fpidx )
                map_id_invptspar_.at(id_idx.first).values()[0] = hipres.invdepth[id_idx.second];
        } else {
            std::cerr << "\\n [ov2slam_hip] localBA falls back to Ceres : " << hipres.error;
        }
    }
    if( !bhipdone )
#endif
    ceres::Solve(options, &problem, &summary);

    if( pslamstate_->debug_ )
        std::cout << summary.FullReport() << std::endl;

    if( pslamstate_->debug_ || pslamstate_->log_timings_ )
        Profiler::StopAndDisplay(pslamstate_->debug_, "2.BA_Optimize");
""")
    f = after(f, """    vbadkflmids.reserve(vreprojerr_kfid_lmid.size() / 10);
    vbadstereokflmids.reserve(vright_reprojerr_kfid_lmid.size() / 10);
""", """
#ifdef OV2SLAM_HIP
    if( bhipdone )
    {
        // One flag per residual block says what both outlier tests (below, and after the L2 pass) decided; the three factor
        // lists are emptied so that the loops that read chi2err_ / isdepthpositive_ from the Ceres factors have nothing to do.
        for( size_t i = 0 ; i < hipres.bad_obs.size() ; i++ ) {
            if( !hipres.bad_obs[i] )
                continue;
            set_badlmids.insert(vfp_kfid_lmid[i].second);
            if( vfp_list[i] == 0 ) {
                vbadkflmids.push_back(vfp_kfid_lmid[i]);
                nbbadobsmono++;
            } else {
                vbadstereokflmids.push_back(vfp_kfid_lmid[i]);
                nbbadobsrightcam++;
            }
        }
        vreprojerr_kfid_lmid.clear();
        vright_reprojerr_kfid_lmid.clear();
        vanchright_reprojerr_kfid_lmid.clear();
    }
#endif
""")
    f = replace(f, """    if( pslamstate_->apply_l2_after_robust_ && buse_robust_cost
        && !stopLocalBA() && nbbadobs > 0 )
    {
        if( !vreprojerr_kfid_lmid.empty() && !vright_reprojerr_kfid_lmid.empty() ) {""", """#ifdef OV2SLAM_HIP
    if( !bhipdone )         // (the L2 pass ran inside ov2_local_ba, on the same conditions)
#endif
    if( pslamstate_->apply_l2_after_robust_ && buse_robust_cost
        && !stopLocalBA() && nbbadobs > 0 )
    {
        if( !vreprojerr_kfid_lmid.empty() && !vright_reprojerr_kfid_lmid.empty() ) {""")
    s = head + f + tail
    # Estimator::addNewKf raises the stop flag while a localBA runs: the library polls the adapter's flag after its first pass
    s = replace(s, """    std::lock_guard<std::mutex> lock(localba_mutex_);
    bstop_localba_ = true;
""", """    std::lock_guard<std::mutex> lock(localba_mutex_);
    bstop_localba_ = true;
#ifdef OV2SLAM_HIP
    if( pslamstate_->pgpu_ )
        pslamstate_->pgpu_->opt.signalStopLocalBA();
#endif
""")
    return s


EDITS = {"CMakeLists.txt": edit_cmake, "include/slam_params.hpp": edit_slam_params, "src/ov2slam.cpp": edit_ov2slam,
         "src/visual_front_end.cpp": edit_front_end, "src/map_manager.cpp": edit_map_manager, "src/mapper.cpp": edit_mapper,
         "src/optimizer.cpp": edit_optimizer}


def generate(ref_root):
    """returns the patch text (paths a/<file> b/<file>, relative to the reference root)"""
    with tempfile.TemporaryDirectory() as td:
        for side in ("a", "b"):
            for f in FILES:
                os.makedirs(os.path.dirname(os.path.join(td, side, f)), exist_ok=True)
                shutil.copy(os.path.join(ref_root, f), os.path.join(td, side, f))
        for f in FILES:
            p = os.path.join(td, "b", f)
            src = open(p, encoding="utf-8").read()
            out = EDITS[f](src)
            assert out != src, f
            assert out.count("#ifdef OV2SLAM_HIP") + out.count("#ifndef OV2SLAM_HIP") == out.count("#endif") - src.count("#endif") or f == "CMakeLists.txt", f
            assert out.count("{") - src.count("{") == out.count("}") - src.count("}"), f       # inserted braces balance
            open(p, "w", encoding="utf-8").write(out)
        r = subprocess.run(["git", "diff", "--no-index", "--no-color", "-U3", "a", "b"], cwd=td, capture_output=True, text=True)
        assert r.returncode in (0, 1), r.stderr
        # `git diff --no-index a b` spells the paths a/a/<file> b/b/<file>
        return r.stdout.replace(" a/a/", " a/").replace(" b/b/", " b/")


HEADER = """# ov2slam_hip.patch -- reference-side binding of libov2slam_hip.so (generated by integration/make_patch.py; do not edit).
# Apply from the root of an OV2SLAM checkout:   git apply ov2slam_hip.patch
# Build:  cmake -DOV2SLAM_AMD_ROOT=<checkout of this repository> ...   (without it nothing changes: every edit is under
# #ifdef OV2SLAM_HIP).  What is wired and what is not: the docstring of integration/make_patch.py, INTEGRATION.md.
"""


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    text = HEADER + generate(ref)
    out = os.path.join(HERE, "ov2slam_hip.patch")
    open(out, "w", encoding="utf-8").write(text)
    print("%s: %d lines, %d files" % (out, text.count("\n"), len(FILES)))


if __name__ == "__main__":
    main()
