// slam_gpu.hpp -- the one object a patched OV2SLAM tree shares between its threads (integration/ov2slam_hip.patch adds
// `std::shared_ptr<ov2::SlamGpu> pgpu_` to SlamParams, which every class of the reference already holds as pslamstate_).
// One context per calling thread (SLAM thread / mapper thread / estimator thread), the per-frame tracker of the
// SLAM thread, the adapters that carry the reference's adaptive detector thresholds, and the two pyramids the mapper thread
// builds per keyframe.  Construct it where the reference constructs its FeatureExtractor / FeatureTracker, BEFORE the mapper
// and estimator threads start (src/ov2slam.cpp:91-113): the tracker captures its hipGraphs at construction.
#pragma once
#include <atomic>
#include <memory>
#include "ov2_types.hpp"
#include "feature_extractor.hpp"
#include "feature_tracker.hpp"
#include "visual_front_end.hpp"
#include "optimizer.hpp"
#include "multi_view_geometry.hpp"

namespace ov2 {

struct SlamGpu {
    Context frontend, mapper, estimator;          // SLAM thread (visualTracking, createKeyframe) / Mapper::run / Estimator::run
    FeatureExtractor extract;                     // detectGridFAST / detectSingleScale with nfast_th_ / dmaxquality_ adapting as in the reference
    FeatureTracker track;                         // fbKltTracking / stereoMatching data path
    Optimizer opt;                                // solveLocalBA (ov2_local_ba)
    std::unique_ptr<FrameTracker> trk;            // preprocessImage + kltTracking of every frame (SLAM thread)
    Pyramid kf_left, kf_right;                    // mapper thread: the keyframe's pyramids, rebuilt from the raw images it queued
    Pyramid kf_front;                             // SLAM thread, btrack_keyframetoframe: the last keyframe's pyramid (kf_pyr_, visual_front_end.cpp:52)
    Optimizer lc_opt;                             // the loop closer's own Optimizer instance (include/loop_closer.hpp:78): looseBA / structureOnlyBA
    int device;

    // arguments: the SlamParams fields of the same names (include/slam_params.hpp) and the left image size
    SlamGpu(int device, int img_w, int img_h, int nbmaxkps, int nmaxdist, double dmaxquality, int nfast_th, int nmax_iter,
            float fmax_px_precision, int nklt_win_size, int nklt_pyr_lvl, float nklt_err, float fmax_fbklt_dist, bool use_clahe,
            double fclahe_val, double robust_mono_th, bool apply_l2_after_robust)
        : frontend(device), mapper(device), estimator(device),
          extract((size_t)nbmaxkps, (size_t)nmaxdist, dmaxquality, nfast_th), track(nmax_iter, fmax_px_precision),
          opt(robust_mono_th, apply_l2_after_robust),
          trk(new FrameTracker(frontend, img_w, img_h, nklt_win_size, nklt_pyr_lvl, nmax_iter, fmax_px_precision, nklt_err,
                               fmax_fbklt_dist, use_clahe, fclahe_val, nbmaxkps)),
          lc_opt(robust_mono_th, apply_l2_after_robust), device(device)
    { global() = this; }
    ~SlamGpu() { trk.reset(); if (global() == this) global() = nullptr; }   // the tracker goes before the context it was created on

    // the one instance, for the reference's static functions that have no SlamParams at hand (MultiViewGeometry::ceresPnP)
    static SlamGpu *&global() { static SlamGpu *p = nullptr; return p; }
    // Per thread: a bundle adjustment whose library call failed re-enters the reference function once with this flag up, and that pass
    // builds and solves the Ceres problem as the unpatched code would (localBA: estimator thread; looseBA / structureOnlyBA: loop
    // closer; fullBA: mapper thread)
    static bool &forceCeres() { static thread_local bool f = false; return f; }

    // One context per calling thread for the reference's STATIC entry points (MultiViewGeometry::ceresPnP runs on the SLAM thread
    // and on the loop closer's, src/visual_front_end.cpp:791, src/loop_closer.cpp:882) and for the threads that have no member
    // context above (LoopCloser::run, the final fullBA on the mapper thread uses `mapper`): created on first use, destroyed with the thread.
    // Keyed by instance (uid: an address can be reused) and device; the deterministic option is (re)applied when it changes.
    Context &threadContext() const
    {
        struct Tls { unsigned long uid = 0; int device = -1; int det = -1; std::unique_ptr<Context> ctx; };
        static thread_local Tls tls;
        if (!tls.ctx || tls.uid != uid || tls.device != device) {
            tls.ctx.reset(new Context(device));
            tls.uid = uid; tls.device = device; tls.det = -1;
        }
        const int det = deterministic_ba.load(std::memory_order_relaxed) ? 1 : 0;
        if (tls.det != det) { ov2_ctx_set_option(tls.ctx->get(), OV2_OPT_BA_DETERMINISTIC, det); tls.det = det; }
        return *tls.ctx;
    }
    // OV2_OPT_BA_DETERMINISTIC on every context that runs bundle adjustments (SlamParams::bhip_deterministic_ba_): localBA on
    // `estimator`, the final fullBA on `mapper`, looseBA / structureOnlyBA / ceresPnP on the calling thread's context.  The reference
    // solves with num_threads = 1 and is reproducible from run to run; the library's default accumulates with fp64 atomics (1.7x faster)
    void setDeterministicBA(bool on)
    {
        ov2_ctx_set_option(estimator.get(), OV2_OPT_BA_DETERMINISTIC, on ? 1 : 0);
        ov2_ctx_set_option(mapper.get(), OV2_OPT_BA_DETERMINISTIC, on ? 1 : 0);
        deterministic_ba.store(on, std::memory_order_relaxed);
    }
    std::atomic<bool> deterministic_ba{false};
    const unsigned long uid = nextUid();
    static unsigned long nextUid() { static std::atomic<unsigned long> n{0}; return ++n; }
};

}  // namespace ov2
