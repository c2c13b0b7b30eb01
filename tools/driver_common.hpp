// driver_common.hpp -- what the two native host programs share (tools/stream_driver.cpp: one camera stream per SLAM thread;
// tools/lockstep_driver.cpp: a rank's sequences in lock-step): the case file written by ov2slam_amd/stream.py:write_case, the
// synthetic ground-truth flow, a blocking queue, and the FNV-1a digest both programs print so that a test can compare what they
// computed bit for bit.  Nothing but the C ABI of include/ov2slam_hip.h.
#pragma once
#include "include/ov2slam_hip.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#ifndef OV2_DRIVER_NAME
#define OV2_DRIVER_NAME "driver"
#endif

static void die(const char *what, int rc) { fprintf(stderr, OV2_DRIVER_NAME ": %s failed (%d): %s\n", what, rc, ov2_last_error()); exit(3); }
#define CK(call) do { const int rc_ = (call); if (rc_ != OV2_OK) die(#call, rc_); } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double wall() { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); }   // epoch seconds: comparable across processes

struct BAProb {
    int n_kf, n_lm, n_res;
    std::vector<double> poses, invdepth, lm_auv, res_uv, res_sigma;
    std::vector<uint8_t> kf_const, res_type;
    std::vector<int> lm_anchor, res_kf, res_lm;
    double calib_l[4], calib_r[4], T_rl[7];
};
struct Case {
    int w, h, n_views, n_frames, kf_every, cell, nbmaxkps;
    double disparity, prior_sigma;
    std::vector<std::vector<uint8_t>> left, right;
    std::vector<double> offs;                      // n_views x (ox, oy, theta)
    std::vector<BAProb> ba;
};

template <class T> static void rd(FILE *f, T *p, size_t n) { if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, OV2_DRIVER_NAME ": short case file\n"); exit(2); } }
static Case read_case(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    Case c;
    int hdr[8]; rd(f, hdr, 8);
    c.w = hdr[0]; c.h = hdr[1]; c.n_views = hdr[2]; c.n_frames = hdr[3]; c.kf_every = hdr[4]; c.cell = hdr[5]; c.nbmaxkps = hdr[6];
    const int n_ba = hdr[7];
    double dd[2]; rd(f, dd, 2); c.disparity = dd[0]; c.prior_sigma = dd[1];
    c.offs.resize(3 * (size_t)c.n_views); rd(f, c.offs.data(), c.offs.size());
    for (int side = 0; side < 2; side++)
        for (int v = 0; v < c.n_views; v++) {
            std::vector<uint8_t> img((size_t)c.w * c.h); rd(f, img.data(), img.size());
            (side ? c.right : c.left).push_back(std::move(img));
        }
    for (int b = 0; b < n_ba; b++) {
        BAProb p; int s[3]; rd(f, s, 3); p.n_kf = s[0]; p.n_lm = s[1]; p.n_res = s[2];
        p.poses.resize(7 * (size_t)p.n_kf); rd(f, p.poses.data(), p.poses.size());
        p.kf_const.resize(p.n_kf); rd(f, p.kf_const.data(), p.kf_const.size());
        p.invdepth.resize(p.n_lm); rd(f, p.invdepth.data(), p.invdepth.size());
        p.lm_anchor.resize(p.n_lm); rd(f, p.lm_anchor.data(), p.lm_anchor.size());
        p.lm_auv.resize(2 * (size_t)p.n_lm); rd(f, p.lm_auv.data(), p.lm_auv.size());
        p.res_type.resize(p.n_res); rd(f, p.res_type.data(), p.res_type.size());
        p.res_kf.resize(p.n_res); rd(f, p.res_kf.data(), p.res_kf.size());
        p.res_lm.resize(p.n_res); rd(f, p.res_lm.data(), p.res_lm.size());
        p.res_uv.resize(2 * (size_t)p.n_res); rd(f, p.res_uv.data(), p.res_uv.size());
        p.res_sigma.resize(p.n_res); rd(f, p.res_sigma.data(), p.res_sigma.size());
        rd(f, p.calib_l, 4); rd(f, p.calib_r, 4); rd(f, p.T_rl, 7);
        c.ba.push_back(std::move(p));
    }
    fclose(f);
    return c;
}

static int view_index(const Case &c, int f) { const int n = c.n_views, k = f % (2 * n - 2); return k < n ? k : 2 * n - 2 - k; }
// ground-truth position in frame fb of pixel (x, y) of frame fa (batch.SyntheticSequence.flow); the trigonometry is per frame pair
struct Flow {
    double cx, cy, ca, sa, ax, ay, cb, sb, bx, by;
    Flow(const Case &c, int fa, int fb)
    {
        cx = (c.w - 1) / 2.0; cy = (c.h - 1) / 2.0;
        const double *a = &c.offs[3 * (size_t)view_index(c, fa)], *b = &c.offs[3 * (size_t)view_index(c, fb)];
        ca = cos(a[2]); sa = sin(a[2]); ax = a[0]; ay = a[1];
        cb = cos(-b[2]); sb = sin(-b[2]); bx = b[0]; by = b[1];
    }
    void operator()(float x, float y, double &ox, double &oy) const
    {
        double dx = x - cx, dy = y - cy;
        const double tx = ca * dx - sa * dy + cx + ax, ty = sa * dx + ca * dy + cy + ay;
        dx = tx - cx - bx; dy = ty - cy - by;
        ox = cb * dx - sb * dy + cx; oy = sb * dx + cb * dy + cy;
    }
};

struct KfJob { int f; const ov2_pyr *left; const uint8_t *right_img; std::vector<float> kps, unpx, p3; std::vector<uint8_t> hp; };
template <class T> struct Queue {
    std::mutex m; std::condition_variable cv; std::deque<T> q; bool closed = false;
    void push(T v) { { std::lock_guard<std::mutex> l(m); q.push_back(std::move(v)); } cv.notify_one(); }
    void close() { { std::lock_guard<std::mutex> l(m); closed = true; } cv.notify_all(); }
    bool pop(T &v) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty() || closed; }); if (q.empty()) return false; v = std::move(q.front()); q.pop_front(); return true; }
    bool try_pop(T &v) { std::lock_guard<std::mutex> l(m); if (q.empty()) return false; v = std::move(q.front()); q.pop_front(); return true; }
};


// FNV-1a style digest over the bytes the library returned, eight bytes per round (a byte-serial FNV cost 0.14 ms per lock-step frame
// step -- as much as the GPU work it was checking): equal digests <=> the two drivers saw bit-identical results
struct Fnv {
    uint64_t h = 1469598103934665603ull;
    void add(const void *p, size_t n)
    {
        const uint8_t *b = (const uint8_t *)p;
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, b + i, 8); h ^= w; h *= 1099511628211ull; }
        if (i < n) { uint64_t w = 0; memcpy(&w, b + i, n - i); h ^= w; h *= 1099511628211ull; }
        h ^= (uint64_t)n; h *= 1099511628211ull;
    }
    template <class T> void val(T v) { add(&v, sizeof(v)); }
};

static void fill_ba_problem(const BAProb &p, ov2_ba_problem &P)
{
    memset(&P, 0, sizeof(P));
    P.n_kf = p.n_kf; P.poses = p.poses.data(); P.kf_const = p.kf_const.data(); P.n_lm = p.n_lm; P.invdepth = p.invdepth.data();
    P.lm_anchor_kf = p.lm_anchor.data(); P.lm_anchor_uv = p.lm_auv.data(); P.n_res = p.n_res; P.res_type = p.res_type.data();
    P.res_kf = p.res_kf.data(); P.res_lm = p.res_lm.data(); P.res_uv = p.res_uv.data(); P.res_sigma = p.res_sigma.data();
    memcpy(P.calib_l, p.calib_l, 32); memcpy(P.calib_r, p.calib_r, 32); memcpy(P.T_rl, p.T_rl, 56);
}
