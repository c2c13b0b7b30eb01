// verbatim_run.cpp -- test driver for ov2slam_amd/host/verbatim.hpp (the reference's exact cv:: signatures, -DOV2_WITH_OPENCV against the
// stand-in <opencv2/core.hpp> of tests/fake_opencv: the image has no OpenCV).  Same case file as adapter_run.cpp.  Checks inside: the
// verbatim route returns the bits of the Context / Pyramid route; the per-thread pyramid cache hits on unchanged buffers and refreshes a
// buffer whose pixels changed in place (the reference re-uses its pyramid Mats).  Dumps the verbatim results for the Python side.
#include <cstdio>
#include <cstring>
#include "../../ov2slam_amd/host/verbatim.hpp"

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
static std::vector<cv::Point2f> pts(const std::vector<float> &v) { std::vector<cv::Point2f> p(v.size() / 2); for (size_t i = 0; i < p.size(); i++) p[i] = cv::Point2f(v[2 * i], v[2 * i + 1]); return p; }
static bool same(const std::vector<cv::Point2f> &a, const std::vector<cv::Point2f> &b) { return a.size() == b.size() && (a.empty() || !memcmp(&a[0], &b[0], a.size() * sizeof(cv::Point2f))); }
// what cv::buildOpticalFlowPyramid(img, pyr, Size(9, 9), 3) returns, as far as the adapter looks: 2 Mats per level, [0] = the level-0 image
static std::vector<cv::Mat> fake_pyramid(std::vector<uint8_t> &img, int w, int h)
{
    std::vector<cv::Mat> v(8);
    for (cv::Mat &m : v) { m.data = nullptr; m.cols = m.rows = 0; m.step.v = 0; }
    v[0].data = img.data(); v[0].cols = w; v[0].rows = h; v[0].step.v = (size_t)w;
    return v;
}
#define CHECK(c, msg) do { if (!(c)) { fprintf(stderr, "verbatim_run: %s\n", msg); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: verbatim_run <case> <result>\n"); return 2; }
    try {
        FILE *fi = fopen(argv[1], "rb"), *fo = fopen(argv[2], "wb");
        if (!fi || !fo) throw std::runtime_error("cannot open files");
        const std::vector<int> dims = rd<int>(fi);
        const int w = dims[0], h = dims[1], cell = dims[2];
        std::vector<uint8_t> img0 = rd<uint8_t>(fi), img1 = rd<uint8_t>(fi);
        const std::vector<cv::Point2f> kps = pts(rd<float>(fi)), pri = pts(rd<float>(fi));
        fclose(fi);
        std::vector<cv::Mat> vprev = fake_pyramid(img0, w, h), vcur = fake_pyramid(img1, w, h);

        // ---- fbKltTracking: the reference's signature against the Context / Pyramid route
        ov2::verbatim::FeatureTracker vt(30, 0.01f);
        std::vector<cv::Point2f> k1 = kps, p1 = pri; std::vector<bool> s1;
        vt.fbKltTracking(vprev, vcur, 9, 3, 30.f, 0.5f, k1, p1, s1);
        ov2::Context ctx(0);
        ov2::Pyramid P0, P1;
        CHECK(P0.build(ctx, ov2::Image8(img0.data(), w, h, w), 9, 3) == OV2_OK && P1.build(ctx, ov2::Image8(img1.data(), w, h, w), 9, 3) == OV2_OK, "pyramid build");
        ov2::FeatureTracker ft(30, 0.01f);
        std::vector<cv::Point2f> k2 = kps, p2 = pri; std::vector<bool> s2;
        ft.fbKltTracking(ctx, P0, P1, 9, 3, 30.f, 0.5f, k2, p2, s2);
        CHECK(same(p1, p2) && s1 == s2 && s1.size() == kps.size(), "verbatim fbKltTracking differs from the Context route");
        const ov2::verbatim::CacheStats a = ov2::verbatim::threadPyramids().stats();
        CHECK(a.misses == 2 && a.hits == 0, "first call: two uploads expected");
        // unchanged buffers: both pyramids come from the cache
        std::vector<cv::Point2f> k3 = kps, p3 = pri; std::vector<bool> s3;
        vt.fbKltTracking(vprev, vcur, 9, 3, 30.f, 0.5f, k3, p3, s3);
        const ov2::verbatim::CacheStats b = ov2::verbatim::threadPyramids().stats();
        CHECK(same(p3, p1) && s3 == s1 && b.hits == 2 && b.misses == 2, "second call: cache hits expected, same result");
        // the reference writes the next frame into the SAME Mat: new pixels behind an old pointer must be uploaded again
        std::vector<uint8_t> keep = img1;
        memcpy(img1.data(), img0.data(), img0.size());
        std::vector<cv::Point2f> k4 = kps, p4 = kps; std::vector<bool> s4;
        vt.fbKltTracking(vprev, vcur, 9, 3, 30.f, 0.5f, k4, p4, s4);
        const ov2::verbatim::CacheStats c = ov2::verbatim::threadPyramids().stats();
        CHECK(c.misses == 3 && c.hits == 3, "changed pixels behind the same pointer: one refresh, one hit expected");
        std::vector<cv::Point2f> k5 = kps, p5 = kps; std::vector<bool> s5;
        ft.fbKltTracking(ctx, P0, P0, 9, 3, 30.f, 0.5f, k5, p5, s5);
        CHECK(same(p4, p5) && s4 == s5, "refreshed pyramid: result differs from the Context route on the new pixels");
        memcpy(img1.data(), keep.data(), keep.size());
        // fewer levels than the pyramid holds, another window: entries of their own
        std::vector<cv::Point2f> k6 = kps, p6 = pri; std::vector<bool> s6;
        vt.fbKltTracking(vprev, vcur, 9, 1, 30.f, 0.5f, k6, p6, s6);
        std::vector<cv::Point2f> k7 = kps, p7 = pri; std::vector<bool> s7;
        ft.fbKltTracking(ctx, P0, P1, 9, 1, 30.f, 0.5f, k7, p7, s7);
        CHECK(same(p6, p7) && s6 == s7, "nbpyrlvl 1: verbatim differs from the Context route");

        // ---- detectSingleScale / detectGridFAST
        cv::Mat im; im.data = img1.data(); im.cols = w; im.rows = h; im.step.v = (size_t)w;
        const cv::Rect roi = {5, 5, w - 10, h - 10};
        std::vector<cv::Point2f> cur(kps.begin(), kps.begin() + (long)(kps.size() / 3));
        ov2::verbatim::FeatureExtractor vx(500, 35, 0.001, 10);
        ov2::FeatureExtractor fx(500, 35, 0.001, 10);
        const std::vector<cv::Point2f> d1 = vx.detectSingleScale(im, cell, cur, roi), d2 = fx.detectSingleScale(ctx, ov2::Image8(im), cell, cur, roi);
        CHECK(same(d1, d2) && d1.size() > 20 && vx.dmaxquality() == fx.dmaxquality_, "verbatim detectSingleScale differs");
        const std::vector<cv::Point2f> f1 = vx.detectGridFAST(im, cell, cur, roi), f2 = fx.detectGridFAST(ctx, ov2::Image8(im), cell, cur, roi);
        CHECK(same(f1, f2) && vx.nfast_th() == fx.nfast_th_, "verbatim detectGridFAST differs");

        wr(fo, p1.empty() ? nullptr : &p1[0].x, 2 * p1.size());
        std::vector<uint8_t> sb(s1.begin(), s1.end()); wr(fo, sb.data(), sb.size());
        wr(fo, d1.empty() ? nullptr : &d1[0].x, 2 * d1.size());
        fclose(fo);
        printf("verbatim ok: %zu keypoints, %zu detections, cache %ld hits / %ld misses\n", kps.size(), d1.size(), c.hits, c.misses);
    } catch (const std::exception &e) { fprintf(stderr, "verbatim_run: %s\n", e.what()); return 1; }
    return 0;
}
