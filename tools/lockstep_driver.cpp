// lockstep_driver.cpp -- a rank's sequences of BASELINE.json configs[4] advanced IN LOCK-STEP: the offline / batch form of the
// keyframe cycle tools/stream_driver.cpp runs per camera stream.
//
// The reference's benchmark protocol plays whole sequences (benchmark_scripts/euroc_bench.sh:3-27); a rank that owns several does
// not need their frames one stream at a time.  tools/stream_driver.cpp gives every sequence its own SLAM thread and tracker: each
// stream is a chain of ~10 small dependent launches per frame, and a rank's streams together saturate the launch rate with the
// CUs ~5 % busy (profiles/archive/r4_stream_concurrency.txt).  Here ONE SLAM thread steps all sequences of the rank through the lock-step
// tracker (ov2_btracker_*, csrc/trackb.hip): per step one frame upload, one CLAHE + pyramid enqueue, one fused kltTracking launch
// and one computeKeypoint launch cover every sequence; at the common keyframes one batched detectSingleScale call; the keyframes
// then go to the rank's mapper thread -- ONE batched right-image CLAHE + pyramid (ov2_pyr_build_clahe_hb) and ONE
// ov2_stereo_match_batch on the tracker's pyramids for all sequences -- and from there to the rank's estimator thread, which solves
// the windows of the sequences that have a keyframe waiting in ONE ov2_local_ba_batch call (per sequence the newest keyframe only,
// like src/estimator.cpp:195-205, or every keyframe with policy "all").  [estimator 0: the round-4 form, an estimator thread and
// context per sequence calling ov2_local_ba: ~100 launches per solve from eleven threads queue behind each other and behind the
// front end's in the command processor.]  Sequences that end drop out of the batch (they are ordered longest first, so the
// active ones are always items [0, n_active)).
// Image "decoding" (here: a copy of the synthetic view into the tracker's pinned slot) runs on loader threads one step ahead,
// the frames' H2D on the tracker's copy stream beside the previous step's kernels.  A step is ov2_btracker_track_frame_begin (enqueue)
// -> upload of frame f + 2, pre-processing of frame f + 1 (their enqueue cost runs beside the tracking kernels) -> _end (wait,
// results); the per-sequence host work around it (priors from the motion model's stand-in, bookkeeping, digests -- what each sequence's
// own SLAM thread does in stream_driver) runs on a small worker pool (argv[7], default 3 threads + the SLAM thread).
// Per-sequence inputs (frames, keypoints, priors, random streams) are those of stream_driver, so the per-sequence results must be
// bit-identical: both programs print FNV-1a digests of everything the library returned (tests/test_gpu_stream.py compares them).
//
//   lockstep_driver <case>[,<case>...] [newest|all] [device] [loader threads] [stream priorities 0|1] [estimator groups, 0 = a thread per sequence] [host workers]
//                                                                     -> one JSON line per sequence (input order) + one summary line
// Build: g++ -O2 -std=c++17 -pthread tools/lockstep_driver.cpp -I. -Lov2slam_amd -lov2slam_hip -Wl,-rpath,<dir>
#define OV2_DRIVER_NAME "lockstep_driver"
#include "tools/driver_common.hpp"

#include <algorithm>
#include <functional>
#include <atomic>
#include <numeric>

struct Seq {
    int id = 0;                                    // position on the command line
    Case C;
    // SLAM-side state of the sequence
    std::mt19937 rng{12345};
    std::normal_distribution<float> gauss{0.f, 1.f};
    std::vector<float> kps; std::vector<int> age;
    double quality = 0.001;
    long frames = 0, tracked = 0, attempted = 0, err_n = 0, keyframes = 0;
    double err_sq = 0;
    Fnv tdig, ddig, sdig;
    std::vector<float> tmp_unpx, tmp_kps; std::vector<double> tmp_bv; std::vector<int> tmp_age;     // (per-sequence scratch of the pool's workers)
    // estimator thread of the sequence (its own context)
    ov2_ctx *ctxC = nullptr;
    Queue<int> ba_q;
    long stereo_kfs = 0, stereo_ok = 0, stereo_kps = 0, ba_solves = 0, ba_skipped = 0, ba_iterations = 0;
    double mapper_busy = 0, ba_busy = 0, ba_device_ms = 0;
    std::thread estimator;
    double t_last_frame = 0, t_drained = 0;
};

// a keyframe step of the whole batch for the rank's mapper thread
struct KfBatch {
    int f, na;
    const ov2_pyr *left;                             // the tracker's pyramids of frame f (all items)
    std::vector<const uint8_t *> right_img;
    std::vector<float> kps, unpx, p3; std::vector<uint8_t> hp; std::vector<int> n;     // n_max slots per item
};

// loader threads: "decode" frame f of every active sequence into the tracker's pinned slots of set f % 3
struct Loader {
    std::vector<std::thread> th; std::mutex m; std::condition_variable cv_go, cv_done;
    int want = -1, done_count = 0, n_threads = 0; bool quit = false;
    std::vector<int> seen;
};

// worker threads for the per-sequence host work of a step (priors, bookkeeping, digests: what each sequence's own SLAM thread does in
// stream_driver -- one thread doing it for eleven sequences was 40 % of the step).  run(n, fn): fn(0..n-1), the caller takes part.
struct Pool {
    std::vector<std::thread> th; std::mutex m; std::condition_variable cv_go, cv_done;
    std::function<void(int)> fn; int n = 0, gen = 0, busy = 0; std::atomic<int> next{0}; bool quit = false;
    void start(int nt)
    {
        for (int t = 0; t < nt; t++)
            th.emplace_back([this] {
                int seen = 0;
                for (;;) {
                    { std::unique_lock<std::mutex> l(m); cv_go.wait(l, [&] { return quit || gen != seen; }); if (quit) return; seen = gen; }
                    for (int i; (i = next.fetch_add(1)) < n;) fn(i);
                    { std::lock_guard<std::mutex> l(m); busy--; }
                    cv_done.notify_one();
                }
            });
    }
    void run(int count, const std::function<void(int)> &f)
    {
        if (th.empty() || count <= 1) { for (int i = 0; i < count; i++) f(i); return; }
        { std::lock_guard<std::mutex> l(m); fn = f; n = count; next.store(0); busy = (int)th.size(); gen++; }
        cv_go.notify_all();
        for (int i; (i = next.fetch_add(1)) < count;) f(i);
        std::unique_lock<std::mutex> l(m); cv_done.wait(l, [&] { return busy == 0; });
    }
    void stop() { { std::lock_guard<std::mutex> l(m); quit = true; } cv_go.notify_all(); for (auto &t : th) t.join(); th.clear(); }
};

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: lockstep_driver <case>[,<case>...] [newest|all] [device] [loader threads]\n"); return 2; }
    if (ov2_version() != OV2_ABI_VERSION) { fprintf(stderr, "lockstep_driver: header / library ABI mismatch\n"); return 2; }
    const bool ba_all = argc > 2 && !strcmp(argv[2], "all");
    const char *dev_s = argc > 3 ? argv[3] : getenv("OV2_DEVICE");      // one process per GPU: the rank's device from the launcher
    const int device = dev_s ? atoi(dev_s) : 0;
    const int n_load = argc > 4 ? std::max(1, atoi(argv[4])) : 4;
    // stream priorities (argv[5]): tracking context high, localBA contexts low -- the SLAM thread is the real-time one
    // (the reference's estimator takes whatever keyframe is newest when it is free, src/estimator.cpp:195-205)
    // ("a,b,e": the priorities of the tracking, mapper and estimator contexts one by one, for experiments)
    // Default: ON with per-sequence estimator threads (their ~1000 launches per step then yield to the tracker's: 13.5k -> 19.5k fps), OFF with
    // the batched estimator -- any non-default priority on any of the three contexts then costs the SLAM thread a third of its
    // speed (profiles/r5_lockstep_priorities.json: 30.0k fps at 0,0,0, 21.3k-23.3k with every other combination).
    const bool est_batch_dflt = argc > 6 ? atoi(argv[6]) != 0 : true;
    const bool use_prio = argc > 5 ? atoi(argv[5]) != 0 || strchr(argv[5], ',') : !est_batch_dflt;
    int prioA = use_prio ? 1 : 0, prioB = use_prio ? 1 : 0, prioE = use_prio ? -1 : 0;
    if (argc > 5 && strchr(argv[5], ',')) sscanf(argv[5], "%d,%d,%d", &prioA, &prioB, &prioE);
    const bool est_batch = argc > 6 ? atoi(argv[6]) != 0 : true;
    // (argv[6] > 1: that many estimator threads, each batching the sequences b with b % groups == its number -- while one group's batch
    // is on the GPU the next one's problems are sorted and staged on the host)
    const int n_work = argc > 7 ? std::max(0, atoi(argv[7])) : 3;           // worker threads for the per-sequence host work of a step (+ the SLAM thread)
    const int est_groups = est_batch ? std::max(1, argc > 6 ? atoi(argv[6]) : 1) : 0;
    std::vector<std::string> paths;
    for (std::string rest = argv[1]; !rest.empty();) {
        const size_t c = rest.find(',');
        paths.push_back(rest.substr(0, c));
        rest = c == std::string::npos ? "" : rest.substr(c + 1);
    }
    const int N = (int)paths.size();
    std::vector<std::unique_ptr<Seq>> S;
    for (int i = 0; i < N; i++) { S.emplace_back(new Seq()); S.back()->id = i; S.back()->C = read_case(paths[(size_t)i].c_str()); }
    // longest first: the sequences still running are always items [0, n_active) of the batch
    std::stable_sort(S.begin(), S.end(), [](const std::unique_ptr<Seq> &a, const std::unique_ptr<Seq> &b) { return a->C.n_frames > b->C.n_frames; });
    const Case &C0 = S[0]->C;
    const int w = C0.w, h = C0.h, kf_every = C0.kf_every, cell = C0.cell, nbmaxkps = C0.nbmaxkps;
    for (auto &s : S)
        if (s->C.w != w || s->C.h != h || s->C.kf_every != kf_every || s->C.cell != cell || s->C.nbmaxkps != nbmaxkps) {
            fprintf(stderr, "lockstep_driver: the sequences of a batch must share image size, keyframe cadence and detector geometry\n"); return 2;
        }
    const double K[4] = {458.654, 457.296, 367.215, 248.375};
    const double iK[9] = {1 / K[0], 0, -K[2] / K[0], 0, 1 / K[1], -K[3] / K[1], 0, 0, 1};
    ov2_ctx *ctxA;
    CK(ov2_ctx_create_with_priority(device, prioA, &ctxA));
    ov2_tracker_config tc{};
    tc.w = w; tc.h = h; tc.win = 9; tc.nklt_pyr_lvl = 3; tc.prior_pyr_lvl = 1; tc.max_iter = 30; tc.eps = 0.01f; tc.err_th = 30.f; tc.fb_dist = 0.5f;
    tc.use_clahe = 1; tc.clahe_clip = 3.0; tc.tiles_x = w / 50; tc.tiles_y = h / 50; tc.n_max = 2 * nbmaxkps; tc.use_graph = 0;
    const int NM = tc.n_max;
    ov2_btracker *trk;
    CK(ov2_btracker_create(ctxA, &tc, N, &trk));
    CK(ov2_btracker_set_calibration(trk, OV2_CAM_PINHOLE, K, nullptr, 0, iK));
    int pitch = 0;
    (void)ov2_btracker_image_buffer(trk, 0, 0, &pitch);
    const int sets = ov2_btracker_pyramid_sets(trk);            // step f overwrites the pyramids of frame f - sets

    // ---- the rank's mapper thread (one batched job per keyframe step) and the per-sequence estimator threads -------------
    ov2_ctx *ctxB;
    CK(ov2_ctx_create_with_priority(device, prioB, &ctxB));
    ov2_pyr *pyrR;
    CK(ov2_pyr_create(ctxB, w, h, 9, 3, N, &pyrR));
    Queue<std::unique_ptr<KfBatch>> map_q;
    std::vector<std::unique_ptr<Queue<std::pair<int, int>>>> est_q;      // (batch item, keyframe) for the rank's estimator thread(s)
    for (int g = 0; g < std::max(1, est_groups); g++) est_q.emplace_back(new Queue<std::pair<int, int>>());
    std::mutex done_m; std::condition_variable done_cv; int mapper_done_kf = -1;
    double mapper_busy_total = 0;
    std::thread mapper([&] {
        std::unique_ptr<KfBatch> j;
        std::vector<float> right(2 * (size_t)N * NM); std::vector<uint8_t> ok((size_t)N * NM);
        while (map_q.pop(j)) {
            const double t0 = now();
            CK(ov2_pyr_build_clahe_hb(ctxB, pyrR, j->na, j->right_img.data(), w, 3.0, w / 50, h / 50));                  // asynchronous
            CK(ov2_stereo_match_batch(ctxB, j->left, pyrR, j->na, 9, 3, 30, 0.01f, 30.f, 0.5f, 1, nullptr, OV2_CAM_PINHOLE, K, nullptr, 0, NM,
                                      j->kps.data(), j->unpx.data(), j->p3.data(), j->hp.data(), j->n.data(), right.data(), ok.data()));
            const double dt = now() - t0;
            mapper_busy_total += dt;
            { std::lock_guard<std::mutex> l(done_m); mapper_done_kf = j->f; }
            done_cv.notify_all();
            for (int b = 0; b < j->na; b++) {
                Seq &s = *S[(size_t)b];
                const int n = j->n[(size_t)b];
                const size_t o = (size_t)b * NM;
                s.mapper_busy += dt / j->na;
                s.sdig.val(j->f); s.sdig.val(n); s.sdig.add(&right[2 * o], 8 * (size_t)n); s.sdig.add(&ok[o], (size_t)n);
                s.stereo_kfs++; s.stereo_kps += n;
                for (int i = 0; i < n; i++) s.stereo_ok += ok[o + i];
                if (!s.C.ba.empty()) { if (est_batch) est_q[(size_t)(b % est_groups)]->push({b, j->f}); else s.ba_q.push(j->f); }
                if (j->f + kf_every > s.C.n_frames - 1) s.ba_q.close();                // the sequence's last keyframe
            }
        }
        for (auto &s : S) s->ba_q.close();
        for (auto &q : est_q) q->close();
    });
    auto est_q_ptr = [&](int g) { return est_q[(size_t)g].get(); };
    std::vector<ov2_ctx *> ctxE((size_t)est_groups, nullptr);
    std::vector<std::thread> estimator_b;
    std::atomic<long> est_batches{0}, est_problems{0};
    for (int g = 0; g < est_groups; g++) {
        CK(ov2_ctx_create_with_priority(device, prioE, &ctxE[(size_t)g]));
        estimator_b.emplace_back([&, g] {
            Queue<std::pair<int, int>> &est_q = *est_q_ptr(g);
            std::vector<std::deque<int>> pending((size_t)N);
            std::vector<int> nsolve((size_t)N, 0), who;
            std::vector<ov2_ba_problem> P; std::vector<ov2_local_ba_options> O; std::vector<ov2_local_ba_result> R;
            std::vector<std::vector<double>> poses((size_t)N), lam((size_t)N);
            std::vector<std::vector<uint8_t>> bad((size_t)N);
            std::pair<int, int> it;
            for (;;) {
                bool any = false;
                for (auto &q : pending) any = any || !q.empty();
                if (!any) { if (!est_q.pop(it)) break; pending[(size_t)it.first].push_back(it.second); }
                while (est_q.try_pop(it)) pending[(size_t)it.first].push_back(it.second);
                who.clear(); P.clear(); O.clear(); R.clear();
                for (int b = 0; b < N; b++) {
                    std::deque<int> &q = pending[(size_t)b];
                    if (q.empty()) continue;
                    Seq &s = *S[(size_t)b];
                    if (!ba_all) { s.ba_skipped += (long)q.size() - 1; q.clear(); }      // only the last received keyframe (estimator.cpp:195-205)
                    else q.pop_front();
                    const BAProb &p = s.C.ba[(size_t)nsolve[(size_t)b]++ % s.C.ba.size()];
                    ov2_ba_problem Pb; fill_ba_problem(p, Pb);
                    ov2_local_ba_options Ob; ov2_local_ba_default_options(&Ob);
                    poses[(size_t)b].resize(7 * (size_t)p.n_kf); lam[(size_t)b].resize((size_t)p.n_lm); bad[(size_t)b].resize((size_t)p.n_res);
                    ov2_local_ba_result Rb{};
                    Rb.poses_out = poses[(size_t)b].data(); Rb.invdepth_out = lam[(size_t)b].data(); Rb.bad_obs = bad[(size_t)b].data();
                    who.push_back(b); P.push_back(Pb); O.push_back(Ob); R.push_back(Rb);
                }
                const double t0 = now();
                int nb = 0;
                CK(ov2_local_ba_batch(ctxE[(size_t)g], (int)who.size(), P.data(), O.data(), R.data(), &nb));
                const double dt = now() - t0, t1 = now();
                est_batches++; est_problems += (long)who.size();
                for (size_t k = 0; k < who.size(); k++) {
                    Seq &s = *S[(size_t)who[k]];
                    s.ba_busy += dt / who.size();
                    s.ba_solves++; s.ba_iterations += R[k].iterations[0] + R[k].iterations[1]; s.ba_device_ms += (R[k].solve_ms[0] + R[k].solve_ms[1]) / who.size();
                    s.t_drained = t1;
                }
            }
        });
    }
    for (auto &sp : S) {
        if (est_batch) break;
        Seq *s = sp.get();
        CK(ov2_ctx_create_with_priority(device, prioE, &s->ctxC));
        s->estimator = std::thread([s, ba_all] {
            int f, nsolve = 0;
            while (s->ba_q.pop(f)) {
                if (!ba_all) { int g; while (s->ba_q.try_pop(g)) { s->ba_skipped++; f = g; } }      // only the last received keyframe (estimator.cpp:195-205)
                const BAProb &p = s->C.ba[(size_t)nsolve++ % s->C.ba.size()];
                ov2_ba_problem P; fill_ba_problem(p, P);
                ov2_local_ba_options O; ov2_local_ba_default_options(&O);
                std::vector<double> poses(7 * (size_t)p.n_kf), lam(p.n_lm);
                std::vector<uint8_t> bad(p.n_res);
                ov2_local_ba_result R{};
                R.poses_out = poses.data(); R.invdepth_out = lam.data(); R.bad_obs = bad.data();
                const double t0 = now();
                CK(ov2_local_ba(s->ctxC, &P, &O, &R));
                s->ba_busy += now() - t0;
                s->ba_solves++; s->ba_iterations += R.iterations[0] + R.iterations[1]; s->ba_device_ms += R.solve_ms[0] + R.solve_ms[1];
            }
            s->t_drained = now();
        });
    }

    auto n_active_at = [&](int f) { int n = 0; while (n < N && S[(size_t)n]->C.n_frames > f) n++; return n; };
    const int F = S[0]->C.n_frames;

    // ---- loader threads -------------------------------------------------------------------------------------------------
    Loader LD; LD.n_threads = n_load; LD.seen.assign((size_t)n_load, -1);
    for (int tid = 0; tid < n_load; tid++)
        LD.th.emplace_back([&, tid] {
            for (;;) {
                int f;
                { std::unique_lock<std::mutex> l(LD.m); LD.cv_go.wait(l, [&] { return LD.quit || LD.want > LD.seen[(size_t)tid]; }); if (LD.quit) return; f = LD.want; }
                const int na = n_active_at(f);
                for (int b = tid; b < na; b += n_load) {
                    const Case &C = S[(size_t)b]->C;
                    int st = 0;
                    uint8_t *dst = ov2_btracker_image_buffer(trk, f % 3, b, &st);
                    const uint8_t *src = C.left[(size_t)view_index(C, f)].data();
                    if (st == w) memcpy(dst, src, (size_t)w * h);
                    else for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * st, src + (size_t)y * w, (size_t)w);
                }
                { std::lock_guard<std::mutex> l(LD.m); LD.seen[(size_t)tid] = f; LD.done_count++; }
                LD.cv_done.notify_all();
            }
        });
    auto load_kick = [&](int f) { { std::lock_guard<std::mutex> l(LD.m); LD.want = f; LD.done_count = 0; } LD.cv_go.notify_all(); };
    auto load_wait = [&] { std::unique_lock<std::mutex> l(LD.m); LD.cv_done.wait(l, [&] { return LD.done_count == LD.n_threads; }); };

    // ---- the SLAM thread of the rank ------------------------------------------------------------------------------------------
    const size_t BN = (size_t)N * NM;
    std::vector<float> kps(2 * BN), pri(2 * BN), out(2 * BN);
    std::vector<uint8_t> hp(BN), st(BN);
    std::vector<int> nper((size_t)N), p3p((size_t)N), ncur((size_t)N), out_n((size_t)N);
    std::vector<double> gt(2 * BN), quality((size_t)N, 0.001);
    std::vector<const uint8_t *> imgs((size_t)N);
    const int cap = 2 * (w / cell) * (h / cell);
    std::vector<float> det((size_t)N * 2 * cap);
    const int roi[4] = {5, 5, w - 10, h - 10};
    double lib_s = 0, wait_loader = 0, wait_mapper = 0;
    double t_begin_s = 0, t_upload_s = 0, t_prepare_s = 0, t_end_s = 0, t_detect_s = 0, t_host_s = 0;      // where the SLAM thread's step goes

    auto keyframe = [&](int f, int na) {
        for (int b = 0; b < na; b++) {
            Seq &s = *S[(size_t)b];
            ncur[(size_t)b] = (int)s.age.size();
            if (!s.kps.empty()) memcpy(&kps[2 * (size_t)b * NM], s.kps.data(), 4 * s.kps.size());
        }
        double tl = now();
        CK(ov2_btracker_detect_singlescale(trk, na, cell, kps.data(), ncur.data(), roi, quality.data(), 1, det.data(), cap, out_n.data()));
        lib_s += now() - tl;
        size_t total = 0;
        for (int b = 0; b < na; b++) {
            Seq &s = *S[(size_t)b];
            int nn = out_n[(size_t)b];
            const float *nw = &det[(size_t)b * 2 * cap];
            s.keyframes++;
            s.ddig.val(f); s.ddig.val(nn); s.ddig.add(nw, 8 * (size_t)nn); s.ddig.val(quality[(size_t)b]);
            nn = std::max(0, std::min(nn, nbmaxkps - ncur[(size_t)b]));
            s.kps.insert(s.kps.end(), nw, nw + 2 * (size_t)nn);
            s.age.insert(s.age.end(), nn, 0);
            total += s.age.size();
        }
        // createKeyframe: undistorted pixels of every keypoint of every sequence, one call
        std::vector<float> all(2 * total), un(2 * total);
        size_t o = 0;
        for (int b = 0; b < na; b++) { const Seq &s = *S[(size_t)b]; if (!s.kps.empty()) memcpy(&all[2 * o], s.kps.data(), 4 * s.kps.size()); o += s.age.size(); }
        tl = now();
        if (total) CK(ov2_compute_keypoints(ctxA, OV2_CAM_PINHOLE, K, nullptr, 0, iK, all.data(), (int)total, un.data(), nullptr));
        lib_s += now() - tl;
        auto j = std::make_unique<KfBatch>();
        j->f = f; j->na = na; j->left = ov2_btracker_cur_pyr(trk);
        j->right_img.resize((size_t)na); j->n.resize((size_t)na);
        j->kps.resize(2 * (size_t)na * NM); j->unpx.resize(2 * (size_t)na * NM); j->p3.resize(2 * (size_t)na * NM); j->hp.resize((size_t)na * NM);
        o = 0;
        for (int b = 0; b < na; b++) {
            Seq &s = *S[(size_t)b];
            const int n = (int)s.age.size();
            const size_t ob = (size_t)b * NM;
            j->n[(size_t)b] = n;
            j->right_img[(size_t)b] = s.C.right[(size_t)view_index(s.C, f)].data();
            if (n) { memcpy(&j->kps[2 * ob], s.kps.data(), 8 * (size_t)n); memcpy(&j->unpx[2 * ob], &un[2 * o], 8 * (size_t)n); }
            for (int i = 0; i < n; i++) {
                j->hp[ob + i] = s.age[i] > 0;
                j->p3[2 * (ob + i)] = s.kps[2 * i] - (float)s.C.disparity + s.gauss(s.rng); j->p3[2 * (ob + i) + 1] = s.kps[2 * i + 1] + s.gauss(s.rng);
            }
            o += (size_t)n;
        }
        map_q.push(std::move(j));
    };

    // Pipeline (ov2_btracker_upload / _prepare / _track_frame): while frame f is tracked, frame f + 1 is pre-processed on the tracker's
    // prep stream, frame f + 2 travels on its copy stream and the loader threads fill the staging set of frame f + 3.
    // wait until the mappers have consumed keyframe g (its pyramid set is about to be overwritten)
    auto wait_mappers = [&](int g) {
        if (g < 0 || g % kf_every != 0) return;
        const double tw = now();
        { std::unique_lock<std::mutex> l(done_m); done_cv.wait(l, [&] { return mapper_done_kf >= g; }); }
        wait_mapper += now() - tw;
    };
    for (int f = 0; f < 3 && f < F; f++) { load_kick(f); load_wait(); }                 // the three staging sets start full
    const double t_begin = wall();
    const double t0 = now();
    {
        const int na = N;
        for (int b = 0; b < na; b++) imgs[(size_t)b] = ov2_btracker_image_buffer(trk, 0, b, nullptr);
        std::fill(nper.begin(), nper.end(), 0);
        double tl = now();
        CK(ov2_btracker_upload(trk, 0, na)); CK(ov2_btracker_prepare(trk, 0, na));        // (prologue of the pipeline)
        if (F > 1) CK(ov2_btracker_upload(trk, 1, n_active_at(1)));
        if (F > 2) CK(ov2_btracker_upload(trk, 2, n_active_at(2)));
        if (F > 1) CK(ov2_btracker_prepare(trk, 1, n_active_at(1)));
        CK(ov2_btracker_track_frame(trk, na, imgs.data(), pitch, kps.data(), pri.data(), hp.data(), nper.data(), 1, out.data(), st.data(), p3p.data()));
        lib_s += now() - tl;
        if (F > 3) load_kick(3);                                                        // staging set 0 is free again
        for (int b = 0; b < na; b++) S[(size_t)b]->frames = 1;
        keyframe(0, na);
    }
    long steps = 1;
    Pool pool; pool.start(n_work);
    // priors of step f for sequence b (the motion model's stand-in): true flow + noise for the keypoints that were tracked before
    auto make_priors = [&](int b, int f) {
        Seq &s = *S[(size_t)b];
        const int n = (int)s.age.size();
        const size_t o = (size_t)b * NM;
        nper[(size_t)b] = n;
        const Flow flow(s.C, f - 1, f);
        for (int i = 0; i < n; i++) {
            const float x = s.kps[2 * i], y = s.kps[2 * i + 1];
            double gx, gy; flow(x, y, gx, gy);
            gt[2 * (o + i)] = gx; gt[2 * (o + i) + 1] = gy;
            const uint8_t hpi = s.age[i] > 0;
            hp[o + i] = hpi;
            kps[2 * (o + i)] = x; kps[2 * (o + i) + 1] = y;
            pri[2 * (o + i)] = hpi ? (float)(gx + s.C.prior_sigma * s.gauss(s.rng)) : x;
            pri[2 * (o + i) + 1] = hpi ? (float)(gy + s.C.prior_sigma * s.gauss(s.rng)) : y;
        }
    };
    // results of step f for sequence b: digests, counters, the surviving keypoints
    auto take_results = [&](int b, int f) {
        Seq &s = *S[(size_t)b];
        const int n = nper[(size_t)b];
        const size_t o = (size_t)b * NM;
        std::vector<float> &unpx = s.tmp_unpx, &nk = s.tmp_kps; std::vector<double> &bv = s.tmp_bv; std::vector<int> &na_ = s.tmp_age;
        unpx.resize(2 * (size_t)n); bv.resize(3 * (size_t)n);
        if (n) CK(ov2_btracker_last_keypoints(trk, b, n, unpx.data(), bv.data()));
        s.tdig.val(f); s.tdig.val(n); s.tdig.val(p3p[(size_t)b]); s.tdig.add(&out[2 * o], 8 * (size_t)n); s.tdig.add(&st[o], (size_t)n);
        s.tdig.add(unpx.data(), 8 * (size_t)n); s.tdig.add(bv.data(), 24 * (size_t)n);
        s.frames++; s.attempted += n;
        nk.clear(); na_.clear();
        for (int i = 0; i < n; i++) {
            if (!(st[o + i] & 1)) continue;
            s.tracked++;
            const float x = out[2 * (o + i)], y = out[2 * (o + i) + 1];
            const double ex = x - gt[2 * (o + i)], ey = y - gt[2 * (o + i) + 1];
            s.err_sq += ex * ex + ey * ey; s.err_n++;
            if (x > 8 && x < w - 9 && y > 8 && y < h - 9) { nk.push_back(x); nk.push_back(y); na_.push_back(s.age[i] + 1); }
        }
        s.kps.swap(nk); s.age.swap(na_);
        if (f == s.C.n_frames - 1) s.t_last_frame = now();
    };
    if (F > 1) pool.run(n_active_at(1), [&](int b) { make_priors(b, 1); });
    for (int f = 1; f < F; f++) {
        const int na = n_active_at(f);
        // the step is enqueued first (frame f was prepared during step f - 1); the enqueue cost of the frames to come -- upload of f + 2,
        // pre-processing of f + 1 -- then runs beside its kernels instead of before them
        for (int b = 0; b < na; b++) imgs[(size_t)b] = ov2_btracker_image_buffer(trk, f % 3, b, nullptr);
        double tl = now();
        CK(ov2_btracker_track_frame_begin(trk, na, imgs.data(), pitch, kps.data(), pri.data(), hp.data(), nper.data(), 1));
        lib_s += now() - tl; t_begin_s += now() - tl;
        if (f + 2 < F) {                                                                // frame f + 2 has been loaded since step f - 1 returned
            const double tw = now(); load_wait(); wait_loader += now() - tw;
            tl = now();
            CK(ov2_btracker_upload(trk, (f + 2) % 3, n_active_at(f + 2)));
            lib_s += now() - tl; t_upload_s += now() - tl;
        }
        if (f + 1 < F) {
            wait_mappers(f + 1 - sets);                                                  // the pyramid set frame f + 1 goes into
            tl = now();
            CK(ov2_btracker_prepare(trk, (f + 1) % 3, n_active_at(f + 1)));
            lib_s += now() - tl; t_prepare_s += now() - tl;
        } else wait_mappers(f - sets);
        tl = now();
        CK(ov2_btracker_track_frame_end(trk, out.data(), st.data(), p3p.data()));
        lib_s += now() - tl; t_end_s += now() - tl;
        if (f + 3 < F) load_kick(f + 3);                                                // staging set f % 3 is free again
        steps++;
        // per-sequence host work on the pool: the results of this step and -- unless a keyframe's detection comes in between -- the priors of the next
        const int na1 = f + 1 < F ? n_active_at(f + 1) : 0;
        tl = now();
        if (f % kf_every == 0) {
            pool.run(na, [&](int b) { take_results(b, f); });
            const double lib0 = lib_s, tk = now();
            keyframe(f, na);
            t_detect_s += lib_s - lib0; tl += now() - tk;                                // (the keyframe's library calls are counted there)
            pool.run(na1, [&](int b) { make_priors(b, f + 1); });
        } else
            pool.run(na, [&](int b) { take_results(b, f); if (b < na1) make_priors(b, f + 1); });
        t_host_s += now() - tl;
    }
    pool.stop();
    CK(ov2_ctx_sync(ctxA));
    const double slam_s = now() - t0;
    map_q.close(); mapper.join();
    if (est_batch) for (auto &t : estimator_b) t.join();
    else for (auto &s : S) s->estimator.join();
    const double total_s = now() - t0;
    const double t_end = wall();
    { std::lock_guard<std::mutex> l(LD.m); LD.quit = true; }
    LD.cv_go.notify_all();
    for (auto &t : LD.th) t.join();

    // ---- results: one line per sequence in input order, then the rank's summary ---------------------------------------------
    std::vector<std::string> lines((size_t)N);
    long frames = 0;
    for (auto &sp : S) {
        const Seq &s = *sp;
        frames += s.frames;
        char line[2048];
        snprintf(line, sizeof(line), "{\"frames\": %ld, \"seconds\": %.6f, \"tracked\": %ld, \"attempted\": %ld, \"err_sq_sum\": %.6f, \"err_n\": %ld, "
                 "\"keyframes\": %ld, \"stereo_kfs\": %ld, \"stereo_ok\": %ld, \"stereo_kps\": %ld, \"mapper_busy_s\": %.6f, \"ba_solves\": %ld, "
                 "\"ba_skipped_kfs\": %ld, \"ba_iterations\": %ld, \"ba_busy_s\": %.6f, \"ba_device_ms\": %.4f, \"ba_policy\": \"%s\", \"device\": %d, "
                 "\"t_begin\": %.6f, \"t_end\": %.6f, \"mode\": \"lockstep\", \"batch_item\": %d, \"track_digest\": \"%016llx\", \"detect_digest\": \"%016llx\", "
                 "\"stereo_digest\": \"%016llx\"}",
                 s.frames, (s.t_drained > 0 ? s.t_drained : now()) - t0, s.tracked, s.attempted, s.err_sq, s.err_n, s.keyframes, s.stereo_kfs, s.stereo_ok,
                 s.stereo_kps, s.mapper_busy, s.ba_solves, s.ba_skipped, s.ba_iterations, s.ba_busy, s.ba_device_ms, ba_all ? "all" : "newest", device,
                 t_begin, t_end, (int)(&sp - &S[0]), (unsigned long long)s.tdig.h, (unsigned long long)s.ddig.h, (unsigned long long)s.sdig.h);
        lines[(size_t)s.id] = line;
    }
    for (auto &l : lines) printf("%s\n", l.c_str());
    printf("{\"lockstep_summary\": true, \"sequences\": %d, \"frames\": %ld, \"steps\": %ld, \"seconds\": %.6f, \"slam_thread_seconds\": %.6f, "
           "\"slam_library_s\": %.6f, \"slam_wait_for_loader_s\": %.6f, \"slam_wait_for_mapper_s\": %.6f, \"loader_threads\": %d, \"stream_priorities\": %d, \"device\": %d, "
           "\"batched_estimator\": %d, \"ba_batches\": %ld, \"ba_problems\": %ld, \"host_workers\": %d, "
           "\"slam_step_breakdown_s\": {\"track_frame_begin\": %.6f, \"upload\": %.6f, \"prepare\": %.6f, \"track_frame_end\": %.6f, \"keyframe_library_calls\": %.6f, "
           "\"per_sequence_host_work_on_the_pool\": %.6f}, \"t_begin\": %.6f, \"t_end\": %.6f}\n",
           N, frames, steps, total_s, slam_s, lib_s, wait_loader, wait_mapper, n_load, use_prio ? 1 : 0, device, est_groups, est_batches.load(), est_problems.load(), n_work,
           t_begin_s, t_upload_s, t_prepare_s, t_end_s, t_detect_s, t_host_s, t_begin, t_end);

    for (auto &s : S) if (s->ctxC) ov2_ctx_destroy(s->ctxC);
    for (ov2_ctx *c : ctxE) if (c) ov2_ctx_destroy(c);
    ov2_pyr_destroy(pyrR); ov2_ctx_destroy(ctxB);
    ov2_btracker_destroy(trk);
    ov2_ctx_destroy(ctxA);
    return 0;
}
