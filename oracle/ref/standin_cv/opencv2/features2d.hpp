// Stand-in for <opencv2/features2d.hpp> (see core.hpp): FAST = the oracle's orc_fast9_16 + OpenCV's keypoint mask filter restated; the
// detectors the two grid functions never touch (GFTT, ORB / BRIEF) are declarations that abort when called
#pragma once
#include "core.hpp"
extern "C" int orc_fast9_16(const uint8_t *img, int w, int h, int stride, int threshold, int nonmax, int *xs, int *ys, int *scores, int cap);
namespace cv {
struct KeyPoint {
    Point2f pt; float size = 7.f, angle = -1.f, response = 0.f; int octave = 0, class_id = -1;
    static void convert(const std::vector<KeyPoint> &k, std::vector<Point2f> &p) { p.clear(); for (const auto &q : k) p.push_back(q.pt); }
    static void convert(const std::vector<Point2f> &p, std::vector<KeyPoint> &k) { k.clear(); for (const auto &q : p) { KeyPoint a; a.pt = q; k.push_back(a); } }
};
struct KeyPointsFilter {
    // features2d/src/keypoint.cpp MaskPredicate: mask.at<uchar>((int)(pt.y + 0.5f), (int)(pt.x + 0.5f)) == 0 -- a BYTE read whatever the mask's type
    static void runByPixelsMask(std::vector<KeyPoint> &k, const Mat &mask)
    {
        if (mask.empty()) return;
        k.erase(std::remove_if(k.begin(), k.end(), [&](const KeyPoint &q) { return mask.data[(size_t)(int)(q.pt.y + 0.5f) * mask.step + (size_t)(int)(q.pt.x + 0.5f)] == 0; }), k.end());
    }
};
struct Feature2D {
    virtual ~Feature2D() {}
    virtual void detect(const Mat &, std::vector<KeyPoint> &, const Mat & = Mat()) { abort(); }
    virtual void compute(const Mat &, std::vector<KeyPoint> &, Mat &) { abort(); }
};
typedef Feature2D DescriptorExtractor;
struct FastFeatureDetector : Feature2D {
    int th;
    explicit FastFeatureDetector(int t) : th(t) {}
    static Ptr<FastFeatureDetector> create(int threshold = 10) { return std::make_shared<FastFeatureDetector>(threshold); }
    void setThreshold(int t) { th = t; }
    int getThreshold() const { return th; }
    // FAST 9/16 with non-maximum suppression on the (ROI) image, keypoints in scan order, response = score; then the mask filter
    void detect(const Mat &im, std::vector<KeyPoint> &kps, const Mat &mask = Mat()) override
    {
        const int cap = im.rows * im.cols;
        std::vector<int> xs((size_t)cap), ys((size_t)cap), sc((size_t)cap);
        const int n = orc_fast9_16(im.data, im.cols, im.rows, (int)im.step, th, 1, xs.data(), ys.data(), sc.data(), cap);
        kps.clear();
        for (int i = 0; i < n; i++) { KeyPoint k; k.pt = Point2f((float)xs[(size_t)i], (float)ys[(size_t)i]); k.response = (float)sc[(size_t)i]; kps.push_back(k); }
        KeyPointsFilter::runByPixelsMask(kps, mask);
    }
};
struct GFTTDetector : Feature2D {
    static Ptr<GFTTDetector> create(int = 1000, double = 0.01, double = 1) { return std::make_shared<GFTTDetector>(); }
    void setQualityLevel(double) {} void setMinDistance(double) {} void setMaxFeatures(int) {}
};
struct ORB : Feature2D { static Ptr<ORB> create(int = 500, float = 1.2f, int = 8) { return std::make_shared<ORB>(); } };
}   // namespace cv
