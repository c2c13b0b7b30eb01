// stream_driver.cpp -- the keyframe cycle of ov2slam_amd/stream.py as a native host program: one camera, three
// std::threads on three contexts of one GPU, nothing but the C ABI of include/ov2slam_hip.h.
//
// The reference's host side is C++ (src/ov2slam.cpp:116-237 SLAM thread, src/mapper.cpp:62-95 mapper thread,
// src/estimator.cpp:33-98 estimator thread); the Python driver of bench.py spends more time in numpy bookkeeping than in
// the library (0.24 ms per frame, 0.16 of it library calls).  This program runs the same schedule with the bookkeeping a C++
// front-end would have: per frame ov2_tracker_track_frame (preprocessImage + kltTracking + computeKeypoint), per keyframe
// ov2_detect_singlescale_d + ov2_compute_keypoints, the mapper's ov2_pyr_build_clahe_h + ov2_stereo_match (FIFO), the
// estimator's ov2_local_ba (newest keyframe only, estimator.cpp:195-205, or every keyframe with policy "all").
// What the reference computes on the CPU around these calls (pose estimation, triangulation, the map walk that builds the BA
// problem) is stood in for by synthetic ground truth: priors come from the true flow, BA windows are pre-generated.
//
//   stream_driver <case file> [policy: newest|all]         -> one JSON line on stdout
// Case file (written by bench.py: little-endian, see read_case): image size, views, flow offsets, BA problems.
// Build: g++ -O2 -std=c++17 -pthread tools/stream_driver.cpp -I. -Lov2slam_amd -lov2slam_hip -Wl,-rpath,<dir>
#define OV2_DRIVER_NAME "stream_driver"
#include "tools/driver_common.hpp"

// one sequence: its SLAM thread is the caller, its mapper and estimator threads are started here; the result line goes to `json`.
// `ready` / `go`: several sequences of one process (the batch mode of config 5: a rank's sequences share its GPU) start streaming
// together once every one of them has initialised.
static int run_sequence(const char *case_path, bool ba_all, int device, std::atomic<int> *ready, std::atomic<int> *go, std::string &json)
{
    const Case C = read_case(case_path);
    const int w = C.w, h = C.h;
    const double K[4] = {458.654, 457.296, 367.215, 248.375};
    const double iK[9] = {1 / K[0], 0, -K[2] / K[0], 0, 1 / K[1], -K[3] / K[1], 0, 0, 1};
    ov2_ctx *ctxA, *ctxB, *ctxC;
    CK(ov2_ctx_create(device, &ctxA)); CK(ov2_ctx_create(device, &ctxB)); CK(ov2_ctx_create(device, &ctxC));
    ov2_tracker_config tc{};
    tc.w = w; tc.h = h; tc.win = 9; tc.nklt_pyr_lvl = 3; tc.prior_pyr_lvl = 1; tc.max_iter = 30; tc.eps = 0.01f; tc.err_th = 30.f; tc.fb_dist = 0.5f;
    tc.use_clahe = 1; tc.clahe_clip = 3.0; tc.tiles_x = w / 50; tc.tiles_y = h / 50; tc.n_max = 2 * C.nbmaxkps; tc.use_graph = 1;
    ov2_tracker *trk;
    CK(ov2_tracker_create(ctxA, &tc, &trk));
    CK(ov2_tracker_set_calibration(trk, OV2_CAM_PINHOLE, K, nullptr, 0, iK));
    ov2_pyr *pyrR;
    CK(ov2_pyr_create(ctxB, w, h, 9, 3, 1, &pyrR));

    // ---- counters -------------------------------------------------------------------------------------------------
    long frames = 0, tracked = 0, attempted = 0, err_n = 0, keyframes = 0, stereo_kfs = 0, stereo_ok = 0, stereo_kps = 0;
    long ba_solves = 0, ba_skipped = 0, ba_iterations = 0;
    double err_sq = 0, mapper_busy = 0, ba_busy = 0, ba_device_ms = 0, slam_wait = 0, slam_lib = 0;
    Fnv tdig, ddig, sdig;       // what the library returned (tracking + computeKeypoint / detection / stereo matching), for tests

    Queue<std::unique_ptr<KfJob>> map_q;
    Queue<int> ba_q;
    std::mutex done_m; std::condition_variable done_cv; int mapper_done_kf = -1;       // last keyframe frame index the mapper has consumed

    std::thread mapper([&] {
        std::unique_ptr<KfJob> j;
        while (map_q.pop(j)) {
            const double t0 = now();
            const int n = (int)j->hp.size();
            CK(ov2_pyr_build_clahe_h(ctxB, pyrR, j->right_img, w, 3.0, w / 50, h / 50));                      // asynchronous
            std::vector<float> right(2 * (size_t)n); std::vector<uint8_t> ok(n);
            CK(ov2_stereo_match(ctxB, j->left, pyrR, 9, 3, 30, 0.01f, 30.f, 0.5f, 1, nullptr, OV2_CAM_PINHOLE, K, nullptr, 0, j->kps.data(),
                                j->unpx.data(), j->p3.data(), j->hp.data(), n, right.data(), ok.data()));
            mapper_busy += now() - t0;
            sdig.val(j->f); sdig.val(n); sdig.add(right.data(), 8 * (size_t)n); sdig.add(ok.data(), (size_t)n);
            { std::lock_guard<std::mutex> l(done_m); mapper_done_kf = j->f; }
            done_cv.notify_all();
            stereo_kfs++; stereo_kps += n;
            for (int i = 0; i < n; i++) stereo_ok += ok[i];
            if (!C.ba.empty()) ba_q.push(j->f);
        }
        ba_q.close();
    });
    std::thread estimator([&] {
        int f, nsolve = 0;
        while (ba_q.pop(f)) {
            if (!ba_all) { int g; while (ba_q.try_pop(g)) { ba_skipped++; f = g; } }      // only the last received keyframe (estimator.cpp:195-205)
            const BAProb &p = C.ba[nsolve++ % C.ba.size()];
            ov2_ba_problem P; fill_ba_problem(p, P);
            ov2_local_ba_options O; ov2_local_ba_default_options(&O);
            std::vector<double> poses(7 * (size_t)p.n_kf), lam(p.n_lm);
            std::vector<uint8_t> bad(p.n_res);
            ov2_local_ba_result R{};
            R.poses_out = poses.data(); R.invdepth_out = lam.data(); R.bad_obs = bad.data();
            const double t0 = now();
            CK(ov2_local_ba(ctxC, &P, &O, &R));
            ba_busy += now() - t0;
            ba_solves++; ba_iterations += R.iterations[0] + R.iterations[1]; ba_device_ms += R.solve_ms[0] + R.solve_ms[1];
        }
    });

    // ---- SLAM thread ----------------------------------------------------------------------------------------------
    std::mt19937 rng(12345);
    std::normal_distribution<float> gauss(0.f, 1.f);
    std::vector<float> kps, pri, out, unpx, nk; std::vector<double> bv, gt; std::vector<uint8_t> hp, st; std::vector<int> age, na;
    const int roi[4] = {5, 5, w - 10, h - 10};
    double quality = 0.001;
    auto keyframe = [&](int f) {
        const int ncur = (int)age.size(), cap = 2 * (w / C.cell) * (h / C.cell);
        std::vector<float> nw(2 * (size_t)cap); int nn = 0;
        double tl = now();
        CK(ov2_detect_singlescale_d(ctxA, ov2_tracker_cur_pyr(trk), 0, C.cell, kps.data(), ncur, roi, &quality, 1, nw.data(), &nn));
        slam_lib += now() - tl;
        keyframes++;
        ddig.val(f); ddig.val(nn); ddig.add(nw.data(), 8 * (size_t)nn); ddig.val(quality);
        nn = std::max(0, std::min(nn, C.nbmaxkps - ncur));
        kps.insert(kps.end(), nw.begin(), nw.begin() + 2 * (size_t)nn);
        age.insert(age.end(), nn, 0);
        const int n = (int)age.size();
        auto j = std::make_unique<KfJob>();
        j->f = f; j->left = ov2_tracker_cur_pyr(trk); j->right_img = C.right[view_index(C, f)].data();
        j->kps = kps; j->unpx.resize(2 * (size_t)n); j->p3.resize(2 * (size_t)n); j->hp.resize(n);
        tl = now();
        if (n) CK(ov2_compute_keypoints(ctxA, OV2_CAM_PINHOLE, K, nullptr, 0, iK, kps.data(), n, j->unpx.data(), nullptr));   // createKeyframe: new keypoints
        slam_lib += now() - tl;
        for (int i = 0; i < n; i++) {
            j->hp[i] = age[i] > 0;
            j->p3[2 * i] = kps[2 * i] - (float)C.disparity + gauss(rng); j->p3[2 * i + 1] = kps[2 * i + 1] + gauss(rng);
        }
        map_q.push(std::move(j));
    };
    if (ready) { ready->fetch_add(1); while (!go->load()) std::this_thread::sleep_for(std::chrono::microseconds(100)); }
    const double t_begin = wall();
    const double t0 = now();
    CK(ov2_tracker_track_frame(trk, C.left[0].data(), w, nullptr, nullptr, nullptr, 0, 1, nullptr, nullptr, nullptr));
    frames = 1;
    keyframe(0);
    for (int f = 1; f < C.n_frames; f++) {
        const int n = (int)age.size();
        pri.resize(2 * (size_t)n); hp.resize(n); out.resize(2 * (size_t)n); st.resize(n); gt.resize(2 * (size_t)n);
        const Flow flow(C, f - 1, f);
        for (int i = 0; i < n; i++) {
            flow(kps[2 * i], kps[2 * i + 1], gt[2 * i], gt[2 * i + 1]);
            hp[i] = age[i] > 0;
            pri[2 * i] = hp[i] ? (float)(gt[2 * i] + C.prior_sigma * gauss(rng)) : kps[2 * i];
            pri[2 * i + 1] = hp[i] ? (float)(gt[2 * i + 1] + C.prior_sigma * gauss(rng)) : kps[2 * i + 1];
        }
        if (f >= 2 && (f - 2) % C.kf_every == 0) {   // this frame overwrites the pyramid of frame f - 2 (two pyramids alternate): if that
            const int kf2 = f - 2;                  // frame was a keyframe the mapper may still be reading it (whatever kf_every is)
            const double tw = now();
            std::unique_lock<std::mutex> l(done_m);
            done_cv.wait(l, [&] { return mapper_done_kf >= kf2; });
            slam_wait += now() - tw;
        }
        int p3p = 0;
        double tl = now();
        CK(ov2_tracker_track_frame(trk, C.left[view_index(C, f)].data(), w, kps.data(), pri.data(), hp.data(), n, 1, out.data(), st.data(), &p3p));
        unpx.resize(2 * (size_t)n); bv.resize(3 * (size_t)n);
        if (n) CK(ov2_tracker_last_keypoints(trk, n, unpx.data(), bv.data()));      // Frame::computeKeypoint of the tracked positions (same enqueue)
        slam_lib += now() - tl;
        tdig.val(f); tdig.val(n); tdig.val(p3p); tdig.add(out.data(), 8 * (size_t)n); tdig.add(st.data(), (size_t)n);
        tdig.add(unpx.data(), 8 * (size_t)n); tdig.add(bv.data(), 24 * (size_t)n);
        frames++; attempted += n;
        nk.clear(); na.clear();
        for (int i = 0; i < n; i++) {
            if (!(st[i] & 1)) continue;
            tracked++;
            const double ex = out[2 * i] - gt[2 * i], ey = out[2 * i + 1] - gt[2 * i + 1];
            err_sq += ex * ex + ey * ey; err_n++;
            const float x = out[2 * i], y = out[2 * i + 1];
            if (x > 8 && x < w - 9 && y > 8 && y < h - 9) { nk.push_back(x); nk.push_back(y); na.push_back(age[i] + 1); }
        }
        kps.swap(nk); age.swap(na);
        if (f % C.kf_every == 0) keyframe(f);
    }
    CK(ov2_ctx_sync(ctxA));
    const double slam_s = now() - t0;
    map_q.close();
    mapper.join(); estimator.join();
    const double total_s = now() - t0;
    const double t_end = wall();
    char line[2048];
    snprintf(line, sizeof(line), "{\"frames\": %ld, \"seconds\": %.6f, \"slam_thread_seconds\": %.6f, \"slam_library_s\": %.6f, \"tracked\": %ld, \"attempted\": %ld, "
           "\"err_sq_sum\": %.6f, \"err_n\": %ld, \"keyframes\": %ld, \"stereo_kfs\": %ld, \"stereo_ok\": %ld, \"stereo_kps\": %ld, \"mapper_busy_s\": %.6f, "
           "\"ba_solves\": %ld, \"ba_skipped_kfs\": %ld, \"ba_iterations\": %ld, \"ba_busy_s\": %.6f, \"ba_device_ms\": %.4f, \"slam_wait_for_mapper_s\": %.6f, "
           "\"ba_policy\": \"%s\", \"device\": %d, \"t_begin\": %.6f, \"t_end\": %.6f, \"mode\": \"stream\", "
           "\"track_digest\": \"%016llx\", \"detect_digest\": \"%016llx\", \"stereo_digest\": \"%016llx\"}",
           frames, total_s, slam_s, slam_lib, tracked, attempted, err_sq, err_n, keyframes, stereo_kfs, stereo_ok, stereo_kps, mapper_busy,
           ba_solves, ba_skipped, ba_iterations, ba_busy, ba_device_ms, slam_wait, ba_all ? "all" : "newest", device, t_begin, t_end,
           (unsigned long long)tdig.h, (unsigned long long)ddig.h, (unsigned long long)sdig.h);
    json = line;
    ov2_tracker_destroy(trk); ov2_pyr_destroy(pyrR);
    ov2_ctx_destroy(ctxA); ov2_ctx_destroy(ctxB); ov2_ctx_destroy(ctxC);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: stream_driver <case>[,<case>...] [newest|all] [device]\n"); return 2; }
    if (ov2_version() != OV2_ABI_VERSION) { fprintf(stderr, "stream_driver: header / library ABI mismatch\n"); return 2; }
    const bool ba_all = argc > 2 && !strcmp(argv[2], "all");
    // one process per GPU: the rank's device index comes from the launcher (argv[3], else OV2_DEVICE, else 0)
    const char *dev_s = argc > 3 ? argv[3] : getenv("OV2_DEVICE");
    const int device = dev_s ? atoi(dev_s) : 0;
    // several comma-separated cases: the sequences run CONCURRENTLY in this process, three threads and three contexts each
    std::vector<std::string> cases;
    for (std::string rest = argv[1]; !rest.empty();) {
        const size_t c = rest.find(',');
        cases.push_back(rest.substr(0, c));
        rest = c == std::string::npos ? "" : rest.substr(c + 1);
    }
    std::vector<std::string> out(cases.size());
    std::vector<int> rc(cases.size(), 0);
    if (cases.size() == 1) rc[0] = run_sequence(cases[0].c_str(), ba_all, device, nullptr, nullptr, out[0]);
    else {
        std::atomic<int> ready{0}, go{0};
        std::vector<std::thread> th;
        for (size_t i = 0; i < cases.size(); i++)
            th.emplace_back([&, i] { rc[i] = run_sequence(cases[i].c_str(), ba_all, device, &ready, &go, out[i]); if (rc[i]) go.store(1); });
        while (ready.load() < (int)cases.size() && !go.load()) std::this_thread::sleep_for(std::chrono::microseconds(200));
        go.store(1);
        for (auto &t : th) t.join();
    }
    int bad = 0;
    for (size_t i = 0; i < cases.size(); i++) { if (rc[i]) bad = rc[i]; else printf("%s\n", out[i].c_str()); }
    return bad;
}
