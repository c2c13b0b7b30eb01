// adapter_run.cpp -- test driver for the C++ adapters in ov2slam_amd/host/*.hpp (the classes a maintainer of the reference
// would call): runs FrameTracker, FeatureExtractor, FeatureTracker + Pyramid and Optimizer on the case file written by
// tests/test_gpu_host_adapters.py and dumps what they return.  File format (both ways): a sequence of arrays, each an int64
// byte count followed by the raw bytes.
#include <cstdio>
#include <cstring>
#include "../../ov2slam_amd/host/feature_tracker.hpp"
#include "../../ov2slam_amd/host/feature_extractor.hpp"
#include "../../ov2slam_amd/host/optimizer.hpp"
#include "../../ov2slam_amd/host/visual_front_end.hpp"

template <class T> static std::vector<T> rd(FILE *f)
{
    long long nb = 0;
    if (fread(&nb, 8, 1, f) != 1) throw std::runtime_error("short case file");
    std::vector<T> v((size_t)nb / sizeof(T));
    if (nb && fread(v.data(), 1, (size_t)nb, f) != (size_t)nb) throw std::runtime_error("short case file");
    return v;
}
template <class T> static void wr(FILE *f, const T *p, size_t n)
{
    const long long nb = (long long)(n * sizeof(T));
    fwrite(&nb, 8, 1, f);
    if (nb) fwrite(p, 1, (size_t)nb, f);
}
static void wr_pts(FILE *f, const std::vector<ov2::Point2f> &v) { wr(f, v.empty() ? nullptr : &v[0].x, 2 * v.size()); }
static void wr_bools(FILE *f, const std::vector<bool> &v) { std::vector<uint8_t> b(v.begin(), v.end()); wr(f, b.data(), b.size()); }
static std::vector<ov2::Point2f> pts(const std::vector<float> &v) { std::vector<ov2::Point2f> p(v.size() / 2); for (size_t i = 0; i < p.size(); i++) p[i] = ov2::Point2f(v[2 * i], v[2 * i + 1]); return p; }

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: adapter_run <case> <result>\n"); return 2; }
    try {
        FILE *fi = fopen(argv[1], "rb"), *fo = fopen(argv[2], "wb");
        if (!fi || !fo) throw std::runtime_error("cannot open files");
        const std::vector<int> dims = rd<int>(fi);                       // w, h, cell
        const int w = dims[0], h = dims[1], cell = dims[2];
        const std::vector<uint8_t> img0 = rd<uint8_t>(fi), img1 = rd<uint8_t>(fi);
        const std::vector<ov2::Point2f> kps = pts(rd<float>(fi)), pri = pts(rd<float>(fi));
        const std::vector<uint8_t> hasprior = rd<uint8_t>(fi);
        ov2::Context ctx(0);
        const ov2::Image8 I0(img0.data(), w, h, w), I1(img1.data(), w, h, w);

        // ---- VisualFrontEnd: preprocessImage + kltTracking, then createKeyframe's detector on cur_pyr_ ----
        {
            ov2::FrameTracker ft(ctx, w, h, 9, 3, 30, 0.01f, 30.f, 0.5f, true, 3.0, 512);
            std::vector<ov2::Point2f> none, nonep;
            std::vector<bool> st;
            bool p3p = false;
            if (!ft.trackFrame(I0, none, nonep, std::vector<uint8_t>(), true, st, p3p)) throw std::runtime_error("trackFrame(0) failed");
            std::vector<ov2::Point2f> out = pri;
            if (!ft.trackFrame(I1, kps, out, hasprior, true, st, p3p)) throw std::runtime_error("trackFrame(1) failed");
            wr_pts(fo, out); wr_bools(fo, st);
            const int p3 = p3p; wr(fo, &p3, 1);
            std::vector<ov2::Point2f> cur;
            for (size_t i = 0; i < out.size(); i++) if (st[i] && cur.size() < 40) cur.push_back(out[i]);
            ov2::FeatureExtractor fx(0, 0, 0.001, 10);
            const ov2::Rect roi{5, 5, w - 10, h - 10};
            const std::vector<ov2::Point2f> det = fx.detectSingleScale(ctx, ft.curPyr(), cell, cur, roi);
            wr_pts(fo, det); wr(fo, &fx.dmaxquality_, 1);
            // split API: preprocessImage is asynchronous, kltTracking synchronises
            if (!ft.preprocessImage(I0)) throw std::runtime_error("preprocessImage failed");
            std::vector<ov2::Point2f> back = kps;
            ft.kltTracking(kps, back, hasprior, false, st, p3p);
            wr_pts(fo, back); wr_bools(fo, st);
        }
        // ---- FeatureExtractor on a host image ----
        {
            ov2::FeatureExtractor fx(0, 0, 0.001, 10);
            const std::vector<ov2::Point2f> none;
            const ov2::Rect roi{5, 5, w - 10, h - 10};
            const std::vector<ov2::Point2f> fast = fx.detectGridFAST(ctx, I1, cell, none, roi);
            wr_pts(fo, fast); wr(fo, &fx.nfast_th_, 1);
            const std::vector<ov2::Point2f> ss = fx.detectSingleScale(ctx, I1, cell, none, roi);
            wr_pts(fo, ss);
        }
        // ---- FeatureTracker::fbKltTracking on two Pyramids (the mapper thread's stereo / the front-end's calls) ----
        {
            ov2::Pyramid p0, p1;
            if (p0.build(ctx, I0, 9, 3) != OV2_OK || p1.build(ctx, I1, 9, 3) != OV2_OK) throw std::runtime_error("Pyramid::build failed");
            ov2::FeatureTracker trk(30, 0.01f);
            std::vector<ov2::Point2f> k = kps, p = pri;
            std::vector<bool> st;
            trk.fbKltTracking(ctx, p0, p1, 9, 3, 30.f, 0.5f, k, p, st);
            wr_pts(fo, p); wr_bools(fo, st);
            // MapManager::stereoMatching's data path on the same two pyramids (left = frame 0, right = frame 1)
            const double rK[4] = {458.654, 457.296, 367.215, 248.375};
            std::vector<ov2::Point2f> right;
            std::vector<bool> sok;
            trk.stereoMatching(ctx, p0.get(), p1.get(), 9, 3, 30.f, 0.5f, true, nullptr, OV2_CAM_PINHOLE, rK, std::vector<double>(), kps, kps, pri,
                               hasprior, right, sok);
            wr_pts(fo, right); wr_bools(fo, sok);
        }
        // ---- Optimizer::localBA solve stage ----
        {
            const std::vector<double> poses = rd<double>(fi);
            const std::vector<uint8_t> kf_const = rd<uint8_t>(fi);
            const std::vector<double> invdepth = rd<double>(fi);
            const std::vector<int> anchor = rd<int>(fi);
            const std::vector<double> auv = rd<double>(fi);
            const std::vector<uint8_t> rtype = rd<uint8_t>(fi);
            const std::vector<int> rkf = rd<int>(fi), rlm = rd<int>(fi);
            const std::vector<double> ruv = rd<double>(fi), rsig = rd<double>(fi), calib = rd<double>(fi);   // calib: l[4] r[4] T_rl[7]
            ov2::FlatProblem fp;
            for (size_t k = 0; k < kf_const.size(); k++) fp.addKeyframe(&poses[7 * k], kf_const[k] != 0);
            for (size_t l = 0; l < invdepth.size(); l++) fp.addLandmark(invdepth[l], anchor[l], auv[2 * l], auv[2 * l + 1]);
            for (size_t i = 0; i < rtype.size(); i++) fp.addResidual(rtype[i], rkf[i], rlm[i], ruv[2 * i], ruv[2 * i + 1], rsig[i]);
            for (int i = 0; i < 4; i++) { fp.calib_l[i] = calib[i]; fp.calib_r[i] = calib[4 + i]; }
            for (int i = 0; i < 7; i++) fp.T_rl[i] = calib[8 + i];
            ov2::Optimizer opt(5.9915, true);
            const ov2::LocalBAResult R = opt.solveLocalBA(ctx, fp, true);
            const int flags[4] = {R.ok, R.l2_done, R.iterations[0], R.iterations[1]};
            wr(fo, flags, 4); wr(fo, R.poses.data(), R.poses.size()); wr(fo, R.invdepth.data(), R.invdepth.size());
            wr(fo, R.bad_obs.data(), R.bad_obs.size());
            // The stop flag's lifetime is the reference's (src/optimizer.cpp:896): a signal raised AFTER the solve (the write-back
            // window) stays set until the caller clears it where the reference clears its own -- it must not leak into the next
            // localBA once cleared, and it must still be visible before that.
            opt.signalStopLocalBA();
            const int still_set = opt.stopLocalBA() ? 1 : 0;
            ov2::FlatProblem fp2 = fp;
            const ov2::LocalBAResult Rs = opt.solveLocalBA(ctx, fp2, true);            // flag still up: no L2 pass
            opt.clearStopLocalBA();
            ov2::FlatProblem fp3 = fp;
            const ov2::LocalBAResult Rc = opt.solveLocalBA(ctx, fp3, true);            // cleared: same as the first call
            const int stop_flags[4] = {still_set, Rs.ok && !Rs.l2_done, Rc.ok && Rc.l2_done == R.l2_done, Rc.iterations[1] == R.iterations[1]};
            wr(fo, stop_flags, 4);
        }
        fclose(fi); fclose(fo);
    } catch (const std::exception &e) {
        fprintf(stderr, "adapter_run: %s (%s)\n", e.what(), ov2_last_error());
        return 1;
    }
    return 0;
}
