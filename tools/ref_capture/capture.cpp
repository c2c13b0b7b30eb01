// capture.cpp -- runs the REAL reference front-end (the reference's own feature_tracker.cpp / feature_extractor.cpp
// linked against a system OpenCV) on fixed inputs and dumps inputs' outputs as .npy fixtures.  Not built in this repo's
// image (no OpenCV / Eigen there); see CMakeLists.txt.  What is captured, per input set <tag> (euroc, kitti):
//   <tag>_clahe.npy                       cv::createCLAHE(3.0, Size(w/50, h/50))->apply(cur)        (ov2slam.cpp:85-89)
//   <tag>_pyr{prev,cur}_L<l>_{img,der}    cv::buildOpticalFlowPyramid(img, pyr, Size(9,9), 3)        (visual_front_end.cpp:1172)
//   <tag>_fbklt_lvl<n>_{out,status}       FeatureTracker::fbKltTracking(prevpyr, curpyr, 9, n, 30, 0.5, kps, priors, status)
//   <tag>_lk_fwd_{out,status,err}         the forward cv::calcOpticalFlowPyrLK alone (feature_tracker.cpp:66-69)
//   <tag>_singlescale_{pts,quality}       FeatureExtractor::detectSingleScale(im, 35, curkps, roi)   (2 consecutive calls)
//   <tag>_gridfast_{pts,th}               FeatureExtractor::detectGridFAST(im, 50, curkps, roi)      (2 consecutive calls)
//   <tag>_linesad_{xprior,l1err}          FeatureTracker::getLineMinSAD on level 3 of (prev, cur)
// cv::setNumThreads(0): the grid detectors mutate a shared mask inside cv::parallel_for_; serial order is
// the canonical semantics the oracle restates.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include <opencv2/core.hpp>
#include <opencv2/imgproc.hpp>
#include <opencv2/video/tracking.hpp>

#include "feature_extractor.hpp"
#include "feature_tracker.hpp"

// ---- minimal .npy (version 1.0, C order, little endian) -----------------------------------------------------------
struct Npy { std::string descr; std::vector<size_t> shape; std::vector<uint8_t> data; };

static bool npy_load(const std::string &path, Npy &a)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { std::cerr << "cannot open " << path << "\n"; return false; }
    char magic[10];
    f.read(magic, 10);
    if (std::memcmp(magic, "\x93NUMPY", 6) != 0) return false;
    const size_t hlen = (uint8_t)magic[8] | ((size_t)(uint8_t)magic[9] << 8);
    std::string hdr(hlen, ' ');
    f.read(&hdr[0], (std::streamsize)hlen);
    const size_t d0 = hdr.find("'descr': '") + 10;
    a.descr = hdr.substr(d0, hdr.find("'", d0) - d0);
    const size_t s0 = hdr.find("(", hdr.find("'shape'")) + 1, s1 = hdr.find(")", s0);
    a.shape.clear();
    std::string sh = hdr.substr(s0, s1 - s0);
    size_t p = 0;
    while (p < sh.size()) {
        while (p < sh.size() && (sh[p] == ' ' || sh[p] == ',')) p++;
        if (p >= sh.size()) break;
        a.shape.push_back((size_t)std::stoul(sh.substr(p)));
        while (p < sh.size() && sh[p] != ',') p++;
    }
    size_t n = 1;
    for (size_t s : a.shape) n *= s;
    const size_t item = (size_t)std::stoul(a.descr.substr(2));
    a.data.resize(n * item);
    f.read((char *)a.data.data(), (std::streamsize)a.data.size());
    return (bool)f;
}

static void npy_save(const std::string &path, const std::string &descr, const std::vector<size_t> &shape, const void *data, size_t bytes)
{
    std::string sh = "(";
    for (size_t i = 0; i < shape.size(); i++) sh += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
    sh += ")";
    std::string hdr = "{'descr': '" + descr + "', 'fortran_order': False, 'shape': " + sh + ", }";
    while ((10 + hdr.size() + 1) % 64 != 0) hdr += ' ';
    hdr += '\n';
    std::ofstream f(path, std::ios::binary);
    const char magic[8] = {(char)0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    f.write(magic, 8);
    const uint16_t hl = (uint16_t)hdr.size();
    f.write((const char *)&hl, 2);
    f.write(hdr.data(), (std::streamsize)hdr.size());
    f.write((const char *)data, (std::streamsize)bytes);
}

static void save_mat_u8(const std::string &p, const cv::Mat &m)
{
    cv::Mat c = m.clone();                                           // contiguous copy of an ROI
    npy_save(p, "|u1", {(size_t)c.rows, (size_t)c.cols}, c.data, (size_t)c.rows * c.cols);
}
static void save_mat_s16c2(const std::string &p, const cv::Mat &m)
{
    cv::Mat c = m.clone();
    npy_save(p, "<i2", {(size_t)c.rows, (size_t)c.cols, 2}, c.data, (size_t)c.rows * c.cols * 4);
}
static void save_pts(const std::string &p, const std::vector<cv::Point2f> &v)
{
    npy_save(p, "<f4", {v.size(), 2}, v.empty() ? nullptr : (const void *)&v[0].x, v.size() * 8);
}
static std::vector<cv::Point2f> load_pts(const std::string &p)
{
    Npy a;
    std::vector<cv::Point2f> v;
    if (!npy_load(p, a)) return v;
    v.resize(a.shape[0]);
    std::memcpy(v.data(), a.data.data(), a.data.size());
    return v;
}
static cv::Mat load_img(const std::string &p)
{
    Npy a;
    if (!npy_load(p, a)) return cv::Mat();
    cv::Mat m((int)a.shape[0], (int)a.shape[1], CV_8UC1);
    std::memcpy(m.data, a.data.data(), a.data.size());
    return m;
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::cerr << "usage: ov2_ref_capture <inputs dir> <output dir>\n"; return 2; }
    const std::string in = std::string(argv[1]) + "/", out = std::string(argv[2]) + "/";
    cv::setNumThreads(0);
    std::cout << "OpenCV " << CV_VERSION << "\n";
    {
        const std::string v = CV_VERSION;
        npy_save(out + "opencv_version.npy", "|u1", {v.size()}, v.data(), v.size());
    }
    for (const std::string tag : {"euroc", "kitti"}) {
        cv::Mat prev = load_img(in + tag + "_prev.npy"), cur = load_img(in + tag + "_cur.npy");
        if (prev.empty() || cur.empty()) { std::cerr << "missing inputs for " << tag << "\n"; return 1; }
        std::vector<cv::Point2f> kps = load_pts(in + tag + "_kps.npy"), pri = load_pts(in + tag + "_pri.npy");
        std::vector<cv::Point2f> curkps = load_pts(in + tag + "_curkps.npy");
        const int w = prev.cols, h = prev.rows;

        // CLAHE exactly as SlamManager builds it (ov2slam.cpp:85-89), applied like visual_front_end.cpp:1159
        cv::Ptr<cv::CLAHE> clahe = cv::createCLAHE(3.0, cv::Size(w / 50, h / 50));
        cv::Mat eq;
        clahe->apply(cur, eq);
        save_mat_u8(out + tag + "_clahe.npy", eq);

        // pyramids (raw frames, so that LK fixtures do not depend on CLAHE parity)
        std::vector<cv::Mat> ppyr, cpyr;
        cv::buildOpticalFlowPyramid(prev, ppyr, cv::Size(9, 9), 3);
        cv::buildOpticalFlowPyramid(cur, cpyr, cv::Size(9, 9), 3);
        for (size_t l = 0; l * 2 + 1 < cpyr.size(); l++) {
            save_mat_u8(out + tag + "_pyrcur_L" + std::to_string(l) + "_img.npy", cpyr[2 * l]);
            save_mat_s16c2(out + tag + "_pyrcur_L" + std::to_string(l) + "_der.npy", cpyr[2 * l + 1]);
            save_mat_u8(out + tag + "_pyrprev_L" + std::to_string(l) + "_img.npy", ppyr[2 * l]);
        }

        FeatureTracker tracker(30, 0.01f, clahe);
        for (int lvl : {3, 1, 0}) {
            std::vector<cv::Point2f> k = kps, p = pri;
            std::vector<bool> st;
            tracker.fbKltTracking(ppyr, cpyr, 9, lvl, 30.f, 0.5f, k, p, st);
            std::vector<uint8_t> s8(st.begin(), st.end());
            save_pts(out + tag + "_fbklt_lvl" + std::to_string(lvl) + "_out.npy", p);
            npy_save(out + tag + "_fbklt_lvl" + std::to_string(lvl) + "_status.npy", "|u1", {s8.size()}, s8.data(), s8.size());
        }
        {   // the forward call alone, with err (OPTFLOW_LK_GET_MIN_EIGENVALS) -- feature_tracker.cpp:66-69
            std::vector<cv::Point2f> p = pri;
            std::vector<uchar> st; std::vector<float> err;
            cv::calcOpticalFlowPyrLK(ppyr, cpyr, kps, p, st, err, cv::Size(9, 9), 3, tracker.klt_convg_crit_,
                                     cv::OPTFLOW_USE_INITIAL_FLOW + cv::OPTFLOW_LK_GET_MIN_EIGENVALS);
            save_pts(out + tag + "_lk_fwd_out.npy", p);
            npy_save(out + tag + "_lk_fwd_status.npy", "|u1", {st.size()}, st.data(), st.size());
            npy_save(out + tag + "_lk_fwd_err.npy", "<f4", {err.size()}, err.data(), err.size() * 4);
        }

        // detectors: two consecutive calls each so that the threshold adaptation is captured too
        const cv::Rect roi(cv::Point2i(5, 5), cv::Point2i(w - 5, h - 5));            // camera_calibration.cpp:72-73
        {
            FeatureExtractor fx(1000, 35, 0.001, 10);
            std::vector<double> q;
            for (int call = 0; call < 2; call++) {
                std::vector<cv::Point2f> pts = fx.detectSingleScale(prev, 35, call == 0 ? std::vector<cv::Point2f>() : curkps, roi);
                save_pts(out + tag + "_singlescale_call" + std::to_string(call) + "_pts.npy", pts);
                q.push_back(fx.dmaxquality_);
            }
            npy_save(out + tag + "_singlescale_quality.npy", "<f8", {q.size()}, q.data(), q.size() * 8);
        }
        {
            FeatureExtractor fx(1000, 50, 0.001, 10);
            std::vector<int32_t> th;
            for (int call = 0; call < 2; call++) {
                std::vector<cv::Point2f> pts = fx.detectGridFAST(prev, 50, call == 0 ? std::vector<cv::Point2f>() : curkps, roi);
                save_pts(out + tag + "_gridfast_call" + std::to_string(call) + "_pts.npy", pts);
                th.push_back(fx.nfast_th_);
            }
            npy_save(out + tag + "_gridfast_th.npy", "<i4", {th.size()}, th.data(), th.size() * 4);
        }
        {   // getLineMinSAD on the coarsest level like map_manager.cpp:421-439 (points scaled by 1 / 2^3)
            const cv::Mat &l3 = ppyr[6], &r3 = cpyr[6];
            std::vector<float> xp(kps.size()), l1(kps.size());
            for (size_t i = 0; i < kps.size(); i++) {
                float x = -1.f, e = 255.f;
                tracker.getLineMinSAD(l3, r3, kps[i] * 0.125f, 7, x, e, true);
                xp[i] = x; l1[i] = e;
            }
            npy_save(out + tag + "_linesad_xprior.npy", "<f4", {xp.size()}, xp.data(), xp.size() * 4);
            npy_save(out + tag + "_linesad_l1err.npy", "<f4", {l1.size()}, l1.data(), l1.size() * 4);
        }
        std::cout << tag << ": captured\n";
    }
    return 0;
}
