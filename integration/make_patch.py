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
  src/optimizer.cpp             localBA (:34-897), looseBA (:900-1672), fullBA (:1674-2332), structureOnlyBA (:2594-2781): the map walk
                                fills an ov2::FlatProblem / FlatXYZProblem / FlatStructureProblem INSTEAD of the Ceres problem (no factor is
                                allocated), the ceres::Solve calls and the outlier loops between them -> ov2_local_ba / ov2_ba_solve /
                                ov2_xyz_ba_solve / ov2_structure_ba; if the library refuses a problem the function re-enters itself once with
                                the Ceres problem built.  Both landmark forms (buse_inv_depth 1 / 0) in localBA; looseBA / fullBA take the
                                library for the inverse-depth form (what every shipped parameter file selects) and Ceres otherwise.
                                signalStopLocalBA reaches the library's live flag, which is cleared where the reference clears its own (:896)
  src/multi_view_geometry.cpp   ceresPnP (:492-586) -> ov2::ceresPnP (ov2_ba_solve on OV2_RES_PNP blocks) on the calling thread's context
  src/visual_front_end.cpp      also kltTrackingFromKF (btrack_keyframetoframe: 1): both fbKltTracking calls on (kf_front, cur) device pyramids
  include/slam_params.hpp, src/slam_params.cpp   `hip_deterministic_ba` (optional key): OV2_OPT_BA_DETERMINISTIC on the estimator context
Not wired: nothing of SURVEY section 8 -- do_klt: 0 (descriptor matching instead of KLT, not a shipped mode) still detects through the
library's host-image entry points; the pose-graph solvers and OpenGV RANSAC are outside the hot path.
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = ["CMakeLists.txt", "include/slam_params.hpp", "src/slam_params.cpp", "src/ov2slam.cpp", "src/visual_front_end.cpp", "src/map_manager.cpp",
         "src/mapper.cpp", "src/optimizer.cpp", "src/multi_view_geometry.cpp"]


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
    // yaml key hip_deterministic_ba (optional, default 0): bundle adjustments bit-identical from run to run, like the reference's
    // single-threaded Ceres (the library's default accumulates with fp64 atomics in arrival order: 1.7x faster, spread 1e-13)
    bool bhip_deterministic_ba_ = false;
#endif
""")


def edit_slam_params_cpp(s):
    return after(s, """    apply_l2_after_robust_ = static_cast<int>(fsSettings["apply_l2_after_robust"]);
""", """#ifdef OV2SLAM_HIP
    if( !fsSettings["hip_deterministic_ba"].empty() )
        bhip_deterministic_ba_ = static_cast<int>(fsSettings["hip_deterministic_ba"]);
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
    pslamstate_->pgpu_->setDeterministicBA(pslamstate_->bhip_deterministic_ba_);
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
    if( pslamstate_->do_klt_ )
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
    # btrack_keyframetoframe: 1 -- the keyframe's pyramid (kf_pyr_, :52) lives on the device next to the tracker's pair
    s = replace(s, """        if( pslamstate_->btrack_keyframetoframe_ ) {
            cv::buildOpticalFlowPyramid(cur_img_, kf_pyr_, pslamstate_->klt_win_size_, pslamstate_->nklt_pyr_lvl_);
        }
""", """        if( pslamstate_->btrack_keyframetoframe_ ) {
#ifdef OV2SLAM_HIP
            // (cur_img_ holds the RAW frame under OV2SLAM_HIP: the equalised image never leaves the device)
            auto &gpu = *pslamstate_->pgpu_;
            const int rckf = pslamstate_->use_clahe_
                    ? gpu.kf_front.buildClahe(gpu.frontend, cur_img_, pslamstate_->nklt_win_size_, pslamstate_->nklt_pyr_lvl_, pslamstate_->fclahe_val_)
                    : gpu.kf_front.build(gpu.frontend, cur_img_, pslamstate_->nklt_win_size_, pslamstate_->nklt_pyr_lvl_);
            if( rckf != OV2_OK )
                std::cerr << "\\n [ov2slam_hip] keyframe pyramid : " << ov2_last_error();
#else
            cv::buildOpticalFlowPyramid(cur_img_, kf_pyr_, pslamstate_->klt_win_size_, pslamstate_->nklt_pyr_lvl_);
#endif
        }
""")
    # kltTrackingFromKF (:277-480): its two fbKltTracking calls on (keyframe pyramid, current pyramid), both on the device
    lo = once(s, "void VisualFrontEnd::kltTrackingFromKF()")
    hi = once(s, "void VisualFrontEnd::epipolar2d2dFiltering()")
    head, f, tail = s[:lo], s[lo:hi], s[hi:]
    f = replace(f, """        ptracker_->fbKltTracking(
                    kf_pyr_, 
                    cur_pyr_, 
                    pslamstate_->nklt_win_size_, 
                    nbpyrlvl, 
                    pslamstate_->nklt_err_, 
                    pslamstate_->fmax_fbklt_dist_, 
                    v3dkps, 
                    v3dpriors, 
                    vkpstatus);
""", """#ifdef OV2SLAM_HIP
        pslamstate_->pgpu_->track.fbKltTracking(pslamstate_->pgpu_->frontend, pslamstate_->pgpu_->kf_front.get(), pslamstate_->pgpu_->trk->curPyr(),
                    pslamstate_->nklt_win_size_, nbpyrlvl, pslamstate_->nklt_err_, pslamstate_->fmax_fbklt_dist_,
                    v3dkps, v3dpriors, vkpstatus);
#else
        ptracker_->fbKltTracking(
                    kf_pyr_, 
                    cur_pyr_, 
                    pslamstate_->nklt_win_size_, 
                    nbpyrlvl, 
                    pslamstate_->nklt_err_, 
                    pslamstate_->fmax_fbklt_dist_, 
                    v3dkps, 
                    v3dpriors, 
                    vkpstatus);
#endif
""")
    f = replace(f, """        ptracker_->fbKltTracking(
                    kf_pyr_, 
                    cur_pyr_, 
                    pslamstate_->nklt_win_size_, 
                    pslamstate_->nklt_pyr_lvl_, 
                    pslamstate_->nklt_err_, 
                    pslamstate_->fmax_fbklt_dist_, 
                    vkps, 
                    vpriors, 
                    vkpstatus);
""", """#ifdef OV2SLAM_HIP
        pslamstate_->pgpu_->track.fbKltTracking(pslamstate_->pgpu_->frontend, pslamstate_->pgpu_->kf_front.get(), pslamstate_->pgpu_->trk->curPyr(),
                    pslamstate_->nklt_win_size_, pslamstate_->nklt_pyr_lvl_, pslamstate_->nklt_err_, pslamstate_->fmax_fbklt_dist_,
                    vkps, vpriors, vkpstatus);
#else
        ptracker_->fbKltTracking(
                    kf_pyr_, 
                    cur_pyr_, 
                    pslamstate_->nklt_win_size_, 
                    pslamstate_->nklt_pyr_lvl_, 
                    pslamstate_->nklt_err_, 
                    pslamstate_->fmax_fbklt_dist_, 
                    vkps, 
                    vpriors, 
                    vkpstatus);
#endif
""")
    s = head + f + tail
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
            // level 0 of the pyramid preprocessImage has just built (do_klt: the equalised frame, no upload); without KLT tracking the
            // front end kept the OpenCV path and `im` is the image to upload
            if( pslamstate_->do_klt_ )
                vnewpts = pslamstate_->pgpu_->extract.detectGridFAST(pslamstate_->pgpu_->frontend, pslamstate_->pgpu_->trk->curPyr(),
                            pslamstate_->nmaxdist_, vpts, pcurframe_->pcalib_leftcam_->roi_rect_);
            else
                vnewpts = pslamstate_->pgpu_->extract.detectGridFAST(pslamstate_->pgpu_->frontend, im,
                            pslamstate_->nmaxdist_, vpts, pcurframe_->pcalib_leftcam_->roi_rect_);
#else
            vnewpts = pfeatextract_->detectGridFAST(im, pslamstate_->nmaxdist_, vpts, pcurframe_->pcalib_leftcam_->roi_rect_);
#endif
""")
    s = replace(s, """            vnewpts = pfeatextract_->detectSingleScale(im, pslamstate_->nmaxdist_, vpts, pcurframe_->pcalib_leftcam_->roi_rect_);
""", """#ifdef OV2SLAM_HIP
            if( pslamstate_->do_klt_ )
                vnewpts = pslamstate_->pgpu_->extract.detectSingleScale(pslamstate_->pgpu_->frontend, pslamstate_->pgpu_->trk->curPyr(),
                            pslamstate_->nmaxdist_, vpts, pcurframe_->pcalib_leftcam_->roi_rect_);
            else
                vnewpts = pslamstate_->pgpu_->extract.detectSingleScale(pslamstate_->pgpu_->frontend, im,
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


HIP_DECL = """
#ifdef OV2SLAM_HIP
    // The same problem in the flat form libov2slam_hip.so takes.  bhipwalk: the library solves it, and the map walk below then fills ONLY
    // the flat problem -- no Ceres factor or parameter block is allocated for the observations (their construction is a third of this
    // function on the CPU).  If the library refuses the problem the function re-enters itself ONCE with the Ceres problem built.
    auto &hipgpu = *pslamstate_->pgpu_;
    const bool bhipwalk = !ov2::SlamGpu::forceCeres()%(extra_cond)s;
    std::unique_ptr<ceres::LossFunctionWrapper> hiploss_guard(bhipwalk ? loss_function : nullptr);   // (no residual block will own it)
    ov2::FlatProblem fp;                                 // buse_inv_depth: 1 -- anchored inverse depths
    ov2::FlatXYZProblem fpx;                             // buse_inv_depth: 0 -- 3-D points
    std::unordered_map<int,int> map_kfid_fpidx, map_lmid_fpidx;
    std::vector<std::pair<int,int>> vfp_kfid_lmid;      // per flat residual block: (kfid, lmid) ...
    std::vector<int> vfp_list;                           // ... and the list its Ceres twin would be in: 0 left, 1 right, 2 right of the anchor
    bool bhipdone = false;
    ov2::LocalBAResult hipres;
#endif
"""

HIP_OBS = """#ifdef OV2SLAM_HIP
            if( bhipwalk )
            {
                // this observation as flat residual block(s); `continue` skips the Ceres factor construction below
                const double hipsig = std::pow(2.,kp.scale_);
                const int hipkf = map_kfid_fpidx.at(kfid);
                if( pslamstate_->buse_inv_depth_ ) {
                    if( kfanchid < 0 ) {
                        kfanchid = kfid;
                        unanch_u = kp.unpx_.x;
                        unanch_v = kp.unpx_.y;
                        double zanch = (pkf->getTcw() * plm->getPoint()).z();
                        map_id_invptspar_.emplace(lmid, InvDepthParametersBlock(lmid, kfanchid, zanch));
                        map_lmid_fpidx[lmid] = fp.addLandmark(map_id_invptspar_.at(lmid).getInvDepth(), hipkf, unanch_u, unanch_v);
                        if( kp.is_stereo_ ) {
                            fp.addResidual(OV2_RES_RIGHT_ANCH, hipkf, map_lmid_fpidx.at(lmid), kp.runpx_.x, kp.runpx_.y, hipsig);
                            vfp_kfid_lmid.push_back(std::make_pair(kfid,lmid)); vfp_list.push_back(%(anch_list)d);
                        }
%(count_anchor)s                        continue;
                    }
                    fp.addResidual(OV2_RES_LEFT, hipkf, map_lmid_fpidx.at(lmid), kp.unpx_.x, kp.unpx_.y, hipsig);
                    vfp_kfid_lmid.push_back(std::make_pair(kfid,lmid)); vfp_list.push_back(0);
                    if( kp.is_stereo_ ) {
                        fp.addResidual(OV2_RES_RIGHT, hipkf, map_lmid_fpidx.at(lmid), kp.runpx_.x, kp.runpx_.y, hipsig);
                        vfp_kfid_lmid.push_back(std::make_pair(kfid,lmid)); vfp_list.push_back(1);
                    }
                } else {
                    fpx.addResidual(OV2_XYZ_LEFT, hipkf, map_lmid_fpidx.at(lmid), kp.unpx_.x, kp.unpx_.y, hipsig);
                    vfp_kfid_lmid.push_back(std::make_pair(kfid,lmid)); vfp_list.push_back(0);
                    if( kp.is_stereo_ ) {
                        fpx.addResidual(OV2_XYZ_RIGHT, hipkf, map_lmid_fpidx.at(lmid), kp.runpx_.x, kp.runpx_.y, hipsig);
                        vfp_kfid_lmid.push_back(std::make_pair(kfid,lmid)); vfp_list.push_back(1);
                    }
                }
%(count_obs)s                continue;
            }
#endif
"""

HIP_COPY_BACK = """            // poses / landmarks into the parameter blocks the write-back at the end of this function reads
            for( const auto &id_idx : map_kfid_fpidx )
                std::copy(hipres.poses.begin() + 7 * id_idx.second, hipres.poses.begin() + 7 * id_idx.second + 7, map_id_posespar_.at(id_idx.first).values());
            if( pslamstate_->buse_inv_depth_ ) {
                for( const auto &id_idx : map_lmid_fpidx )
                    map_id_invptspar_.at(id_idx.first).values()[0] = hipres.invdepth[id_idx.second];
            } else {
                for( const auto &id_idx : map_lmid_fpidx )
                    std::copy(hipres.invdepth.begin() + 3 * id_idx.second, hipres.invdepth.begin() + 3 * id_idx.second + 3, map_id_pointspar_.at(id_idx.first).values());
            }
"""


def wire_ba(f, kind):
    """the common edits of localBA / looseBA / fullBA (same code shape in the reference); kind: "local" | "loose" | "full" """
    count = kind != "full"                      # fullBA keeps no nbmono / nbstereo tally in this part of the walk
    f = after(f, """    auto ordering = new ceres::ParameterBlockOrdering;
""", HIP_DECL % {"extra_cond": "" if kind == "local" else " && pslamstate_->buse_inv_depth_"})
    f = after(f, """    problem.SetParameterBlockConstant(calibpar.values());
""", """
#ifdef OV2SLAM_HIP
    fp.calib_l[0] = pcalibleft->fx_; fp.calib_l[1] = pcalibleft->fy_; fp.calib_l[2] = pcalibleft->cx_; fp.calib_l[3] = pcalibleft->cy_;
    std::copy(fp.calib_l, fp.calib_l + 4, fpx.calib_l);
#endif
""")
    f = after(f, """        problem.SetParameterBlockConstant(rlextrinpose.values());
""", """
#ifdef OV2SLAM_HIP
        fp.calib_r[0] = pcalibright->fx_; fp.calib_r[1] = pcalibright->fy_; fp.calib_r[2] = pcalibright->cx_; fp.calib_r[3] = pcalibright->cy_;
        std::copy(fp.calib_r, fp.calib_r + 4, fpx.calib_r);
        std::copy(rlextrinpose.values(), rlextrinpose.values() + 7, fp.T_rl);
        std::copy(rlextrinpose.values(), rlextrinpose.values() + 7, fpx.T_rl);
#endif
""")
    # keyframes of the window (free or constant, decided right below)
    if kind == "local":
        f = before(f, """        // For those to optimize, get their 3D MPs
        // for the others, set them as constant""", """#ifdef OV2SLAM_HIP
        map_kfid_fpidx[kfid] = fp.addKeyframe(map_id_posespar_.at(kfid).values(), false);
        fpx.addKeyframe(map_id_posespar_.at(kfid).values(), false);
#endif

""")
        f = after(f, """            set_cstkfids.insert(kfid);
            problem.SetParameterBlockConstant(map_id_posespar_.at(kfid).values());
            all_cst = true;
""", """#ifdef OV2SLAM_HIP
            fp.kf_const[map_kfid_fpidx.at(kfid)] = 1; fpx.kf_const[map_kfid_fpidx.at(kfid)] = 1;
#endif
""")
    else:
        f = after(f, """        ordering->AddElementToGroup(map_id_posespar_.at(pkf->kfid_).values(), 1);
""", """#ifdef OV2SLAM_HIP
        map_kfid_fpidx[pkf->kfid_] = fp.addKeyframe(map_id_posespar_.at(pkf->kfid_).values(), false);
        fpx.addKeyframe(map_id_posespar_.at(pkf->kfid_).values(), false);
#endif
""")
        f = after(f, """            set_cstkfids.insert(pkf->kfid_);
            problem.SetParameterBlockConstant(map_id_posespar_.at(pkf->kfid_).values());
""", """#ifdef OV2SLAM_HIP
            fp.kf_const[map_kfid_fpidx.at(pkf->kfid_)] = 1; fpx.kf_const[map_kfid_fpidx.at(pkf->kfid_)] = 1;
#endif
""")
    # observing keyframes outside the window: constant
    f = after(f, """                set_cstkfids.insert(kfid);
                problem.SetParameterBlockConstant(map_id_posespar_.at(kfid).values());
""", """#ifdef OV2SLAM_HIP
                map_kfid_fpidx[kfid] = fp.addKeyframe(map_id_posespar_.at(kfid).values(), true);
                fpx.addKeyframe(map_id_posespar_.at(kfid).values(), true);
#endif
""")
    # 3-D point landmarks: the parameter block is the flat problem's under bhipwalk
    f = before(f, """        if( !pslamstate_->buse_inv_depth_ )
        {
            map_id_pointspar_.emplace(lmid, PointXYZParametersBlock(lmid, plm->getPoint()));
""", """#ifdef OV2SLAM_HIP
        if( bhipwalk ) {
            if( !pslamstate_->buse_inv_depth_ ) {
                map_id_pointspar_.emplace(lmid, PointXYZParametersBlock(lmid, plm->getPoint()));
                map_lmid_fpidx[lmid] = fpx.addPoint(map_id_pointspar_.at(lmid).values());
            }
        } else
#endif
""")
    # every observation
    obs = HIP_OBS % {"anch_list": 1 if kind == "full" else 2,        # fullBA files the anchor's right-camera block under the right list (:1892)
                     "count_anchor": ("""                        if( kp.is_stereo_ ) {
                            nbstereo++;
                        } else {
                            nbmono++;
                        }
""" if count else ""),
                     "count_obs": ("""                if( kp.is_stereo_ ) {
                    nbstereo++;
                } else {
                    nbmono++;
                }
""" if count else "")}
    if kind == "full":
        f = after(f, """            if( kp.lmid_ < 0 ) {
                pmap_->removeMapPointObs(lmid, kfid);
                continue;
            }
""", obs)
    else:
        f = after(f, """            if( kp.lmid_ != lmid ) {
                pmap_->removeMapPointObs(lmid, kfid);
                continue;
            }
""", obs)
    # gauge fix
    f = after(f, """            problem.SetParameterBlockConstant(map_id_posespar_.at(it->first).values());
            set_cstkfids.insert(it->first);
""", """#ifdef OV2SLAM_HIP
            fp.kf_const[map_kfid_fpidx.at(it->first)] = 1; fpx.kf_const[map_kfid_fpidx.at(it->first)] = 1;
#endif
""")
    return f


def edit_optimizer(s):
    """localBA, looseBA, fullBA, structureOnlyBA (anchors are searched inside each function's text only) + the stop flag"""
    s = after(s, """#include "ceres_parametrization.hpp"
""", """
#ifdef OV2SLAM_HIP
#include "slam_gpu.hpp"
#endif
""")
    cut = [once(s, "void Optimizer::localBA(Frame &newframe, const bool buse_robust_cost)"),
           once(s, "void Optimizer::looseBA(int inikfid, const int nkfid, const bool buse_robust_cost)"),
           once(s, "void Optimizer::fullBA(const bool buse_robust_cost)"),
           once(s, "void Optimizer::signalStopLocalBA()"),
           once(s, "void Optimizer::structureOnlyBA(const std::vector<int> &vlm2optids)"),
           once(s, "bool Optimizer::fullPoseGraph(")]
    head, f_local, f_loose, f_full, mid, f_struct, tail = s[:cut[0]], s[cut[0]:cut[1]], s[cut[1]:cut[2]], s[cut[2]:cut[3]], s[cut[3]:cut[4]], s[cut[4]:cut[5]], s[cut[5]:]

    # ------------------------------------------------------------------------------------------------ localBA
    f = wire_ba(f_local, "local")
    # the solve stage: both ceres::Solve calls + the outlier logic between them in ONE library call on the estimator thread's context
    f = replace(f, """    ceres::Solver::Summary summary;
    ceres::Solve(options, &problem, &summary);

    if( pslamstate_->debug_ )
        std::cout << summary.FullReport() << std::endl;

    if( pslamstate_->debug_ || pslamstate_->log_timings_ )
        Profiler::StopAndDisplay(pslamstate_->debug_, "2.BA_Optimize");
""", """    ceres::Solver::Summary summary;
#ifdef OV2SLAM_HIP
    if( bhipwalk )
    {
        // (same wall-clock budget as the two solves below: t for the robust pass, t / 2 for the L2 pass)
        hipgpu.opt.setMaxSolverTime(options.max_solver_time_in_seconds);
        if( pslamstate_->buse_inv_depth_ )
            hipres = hipgpu.opt.solveLocalBA(hipgpu.estimator, fp, buse_robust_cost);
        else
            hipres = hipgpu.opt.solveLocalBAXYZ(hipgpu.estimator, fpx, buse_robust_cost);
        bhipdone = hipres.ok;
        if( bhipdone ) {
""" + HIP_COPY_BACK + """        } else {
            std::cerr << "\\n [ov2slam_hip] localBA falls back to Ceres : " << hipres.error;
            if( pslamstate_->debug_ || pslamstate_->log_timings_ )
                Profiler::StopAndDisplay(pslamstate_->debug_, "2.BA_Optimize");
            ov2::SlamGpu::forceCeres() = true;          // (this thread only)
            localBA(newframe, buse_robust_cost);
            ov2::SlamGpu::forceCeres() = false;
            return;
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
    bad_lists = """
#ifdef OV2SLAM_HIP
    if( bhipdone )
    {
        // One flag per residual block says what the outlier tests decided; the three factor lists are empty (bhipwalk created no
        // Ceres factor), so the loops below that read chi2err_ / isdepthpositive_ from the factors have nothing to do.
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
    }
#endif
"""
    f = after(f, """    vbadkflmids.reserve(vreprojerr_kfid_lmid.size() / 10);
    vbadstereokflmids.reserve(vright_reprojerr_kfid_lmid.size() / 10);
""", bad_lists)
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
    # the adapter's copy of the stop flag lives as long as the reference's: cleared HERE, after the map write-back (a signal raised
    # during the write-back is dropped, like the reference's own)
    f = replace(f, """    bstop_localba_ = false;
}
""", """    bstop_localba_ = false;
#ifdef OV2SLAM_HIP
    pslamstate_->pgpu_->opt.clearStopLocalBA();
#endif
}
""")
    f_local = f

    # ------------------------------------------------------------------------------------------------ looseBA (loop closer's thread)
    f = wire_ba(f_loose, "loose")
    f = replace(f, """    ceres::Solver::Summary summary;
    ceres::Solve(options, &problem, &summary);
""", """    ceres::Solver::Summary summary;
#ifdef OV2SLAM_HIP
    if( bhipwalk )
    {
        hipres = hipgpu.lc_opt.solveLooseBA(hipgpu.threadContext(), fp, buse_robust_cost);
        bhipdone = hipres.ok;
        if( bhipdone ) {
""" + HIP_COPY_BACK + """        } else {
            std::cerr << "\\n [ov2slam_hip] looseBA falls back to Ceres : " << hipres.error;
            if( pslamstate_->debug_ || pslamstate_->log_timings_ )
                Profiler::StopAndDisplay(pslamstate_->debug_, "2.LC_LooseBA_Optimize");
            ov2::SlamGpu::forceCeres() = true;          // (this thread only)
            looseBA(inikfid, nkfid, buse_robust_cost);
            ov2::SlamGpu::forceCeres() = false;
            return;
        }
    }
    if( !bhipdone )
#endif
    ceres::Solve(options, &problem, &summary);
""")
    f = after(f, """    vbadkflmids.reserve(vreprojerr_kfid_lmid.size() / 10);
    vbadstereokflmids.reserve(vright_reprojerr_kfid_lmid.size() / 10);
""", bad_lists)
    f_loose = f

    # ------------------------------------------------------------------------------------------------ fullBA (mapper thread, end of the run)
    f = wire_ba(f_full, "full")
    f = replace(f, """    ceres::Solver::Summary summary;
    ceres::Solve(options, &problem, &summary);
""", """    ceres::Solver::Summary summary;
#ifdef OV2SLAM_HIP
    if( bhipwalk )
    {
        // both passes and both outlier tests (left / right lists) in the adapter: Optimizer::fullBA's protocol, :2055-2262
        hipres = hipgpu.lc_opt.solveFullBA(hipgpu.threadContext(), fp, buse_robust_cost);
        bhipdone = hipres.ok;
        if( bhipdone ) {
""" + HIP_COPY_BACK + """        } else {
            std::cerr << "\\n [ov2slam_hip] fullBA falls back to Ceres : " << hipres.error;
            ov2::SlamGpu::forceCeres() = true;          // (this thread only)
            fullBA(buse_robust_cost);
            ov2::SlamGpu::forceCeres() = false;
            return;
        }
    }
    if( !bhipdone )
#endif
    ceres::Solve(options, &problem, &summary);
""")
    f = after(f, """    size_t nbbadobsmono = 0;
    size_t nbbadobsrightcam = 0;
""", """
#ifdef OV2SLAM_HIP
    if( bhipdone )
    {
        // what the two outlier loops below do for a bad left / right observation (their lists are empty under bhipwalk)
        for( size_t i = 0 ; i < hipres.bad_obs.size() ; i++ ) {
            if( !hipres.bad_obs[i] )
                continue;
            const int kfid = vfp_kfid_lmid[i].first, lmid = vfp_kfid_lmid[i].second;
            if( vfp_list[i] == 0 ) {
                pmap_->removeMapPointObs(lmid,kfid);
                nbbadobsmono++;
            } else {
                map_local_pkfs.at(kfid)->removeStereoKeypointById(lmid);
                nbbadobsrightcam++;
            }
            set_badlmids.insert(lmid);
        }
    }
#endif
""")
    f = replace(f, """    if( pslamstate_->apply_l2_after_robust_ && nbbadobs > 0 ) 
    {""", """#ifdef OV2SLAM_HIP
    if( !bhipdone )         // (the L2 pass ran inside solveFullBA, on the same conditions)
#endif
    if( pslamstate_->apply_l2_after_robust_ && nbbadobs > 0 ) 
    {""")
    f_full = f

    # ------------------------------------------------------------------------------------------------ structureOnlyBA (loop closer's thread)
    f = f_struct
    f = after(f, """    auto ordering = new ceres::ParameterBlockOrdering;
""", """
#ifdef OV2SLAM_HIP
    // 3-D points against constant keyframes: the flat problem of ov2_structure_ba (see localBA above for bhipwalk)
    auto &hipgpu = *pslamstate_->pgpu_;
    const bool bhipwalk = !ov2::SlamGpu::forceCeres();
    std::unique_ptr<ceres::LossFunctionWrapper> hiploss_guard(bhipwalk ? loss_function : nullptr);
    ov2::FlatStructureProblem sp;
    std::unordered_map<int,int> map_kfid_fpidx, map_lmid_fpidx;
#endif
""")
    f = after(f, """    problem.SetParameterBlockConstant(calibpar.values());
""", """
#ifdef OV2SLAM_HIP
    sp.calib_l[0] = pcalibleft->fx_; sp.calib_l[1] = pcalibleft->fy_; sp.calib_l[2] = pcalibleft->cx_; sp.calib_l[3] = pcalibleft->cy_;
#endif
""")
    f = after(f, """        problem.SetParameterBlockConstant(rlextrinpose.values());
""", """
#ifdef OV2SLAM_HIP
        sp.calib_r[0] = pcalibright->fx_; sp.calib_r[1] = pcalibright->fy_; sp.calib_r[2] = pcalibright->cx_; sp.calib_r[3] = pcalibright->cy_;
        std::copy(rlextrinpose.values(), rlextrinpose.values() + 7, sp.T_rl);
#endif
""")
    f = replace(f, """        problem.AddParameterBlock(map_id_pointspar_.at(lmid).values(), 3);
        ordering->AddElementToGroup(map_id_pointspar_.at(lmid).values(), 0);
""", """#ifdef OV2SLAM_HIP
        if( bhipwalk ) {
            map_lmid_fpidx[lmid] = sp.addPoint(map_id_pointspar_.at(lmid).values());
        } else {
#endif
        problem.AddParameterBlock(map_id_pointspar_.at(lmid).values(), 3);
        ordering->AddElementToGroup(map_id_pointspar_.at(lmid).values(), 0);
#ifdef OV2SLAM_HIP
        }
#endif
""")
    f = after(f, """            if( kp.lmid_ != lmid ) {
                continue;
            }
""", """#ifdef OV2SLAM_HIP
            if( bhipwalk )
            {
                if( !map_id_posespar_.count(kfid) ) {
                    map_id_posespar_.emplace(kfid, PoseParametersBlock(kfid, pkf->getTwc()));
                    map_kfid_fpidx[kfid] = sp.addKeyframe(map_id_posespar_.at(kfid).values());
                }
                sp.addResidual(OV2_XYZ_LEFT, map_kfid_fpidx.at(kfid), map_lmid_fpidx.at(lmid), kp.unpx_.x, kp.unpx_.y, std::pow(2.,kp.scale_));
                if( kp.is_stereo_ )
                    sp.addResidual(OV2_XYZ_RIGHT, map_kfid_fpidx.at(kfid), map_lmid_fpidx.at(lmid), kp.runpx_.x, kp.runpx_.y, std::pow(2.,kp.scale_));
                continue;
            }
#endif
""")
    f = replace(f, """    ceres::Solver::Summary summary;
    ceres::Solve(options, &problem, &summary);
""", """    ceres::Solver::Summary summary;
#ifdef OV2SLAM_HIP
    bool bhipdone = false;
    if( bhipwalk )
    {
        std::vector<double> vhipxyz;
        bhipdone = hipgpu.lc_opt.solveStructureOnlyBA(hipgpu.threadContext(), sp, vhipxyz);
        if( bhipdone ) {
            for( const auto &id_idx : map_lmid_fpidx )
                std::copy(vhipxyz.begin() + 3 * id_idx.second, vhipxyz.begin() + 3 * id_idx.second + 3, map_id_pointspar_.at(id_idx.first).values());
        } else {
            std::cerr << "\\n [ov2slam_hip] structureOnlyBA falls back to Ceres : " << ov2_last_error();
            if( pslamstate_->debug_ || pslamstate_->log_timings_ )
                Profiler::StopAndDisplay(pslamstate_->debug_, "2.LC_StructBA_Optimize");
            ov2::SlamGpu::forceCeres() = true;          // (this thread only)
            structureOnlyBA(vlm2optids);
            ov2::SlamGpu::forceCeres() = false;
            return;
        }
    }
    if( !bhipdone )
#endif
    ceres::Solve(options, &problem, &summary);
""")
    f_struct = f

    s = head + f_local + f_loose + f_full + mid + f_struct + tail
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


def edit_multi_view_geometry(s):
    s = after(s, """#include "multi_view_geometry.hpp"
""", """
#ifdef OV2SLAM_HIP
#include <iostream>
#include "ceres_parametrization.hpp"
#include "slam_gpu.hpp"
#endif
""")
    return after(s, """    const float fx, const float fy, const float cx, const float cy, 
    std::vector<int> &voutliersidx)
{
    assert( vunkps.size() == vwpts.size() );
""", """
#ifdef OV2SLAM_HIP
    if( ov2::SlamGpu::global() != nullptr && !vunkps.empty() )
    {
        // Motion-only BA on the calling thread's context (this function is static and runs on the SLAM thread and on the loop closer's).
        // Eigen::Vector2d / Vector3d are two / three packed doubles: the vectors ARE the flat arrays the library takes.
        PoseParametersBlock hippose(0, Twc);
        std::vector<int> vhipoutliers;
        bool bhiplibok = true;
        std::string hiperr;
        const bool bhipsuccess = ov2::ceresPnP(ov2::SlamGpu::global()->threadContext(), vunkps[0].data(), vwpts[0].data(), vscales.data(),
                                               vunkps.size(), hippose.values(), nmaxiter, chi2th, buse_robust, bapply_l2_after_robust,
                                               fx, fy, cx, cy, vhipoutliers, 0.005, &bhiplibok, &hiperr);
        if( bhiplibok ) {
            voutliersidx.insert(voutliersidx.end(), vhipoutliers.begin(), vhipoutliers.end());
            if( vhipoutliers.size() != vunkps.size() )      // (every observation bad: the reference returns before touching Twc, :562-564)
                Twc = hippose.getPose();
            return bhipsuccess;
        }
        std::cerr << "\\n [ov2slam_hip] ceresPnP falls back to Ceres : " << hiperr;
    }
#endif
""")


EDITS = {"CMakeLists.txt": edit_cmake, "include/slam_params.hpp": edit_slam_params, "src/slam_params.cpp": edit_slam_params_cpp, "src/ov2slam.cpp": edit_ov2slam,
         "src/visual_front_end.cpp": edit_front_end, "src/map_manager.cpp": edit_map_manager, "src/mapper.cpp": edit_mapper,
         "src/optimizer.cpp": edit_optimizer, "src/multi_view_geometry.cpp": edit_multi_view_geometry}


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
