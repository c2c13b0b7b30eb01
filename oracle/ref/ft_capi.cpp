// ft_capi.cpp -- C entry points over the REFERENCE'S OWN FeatureTracker (/root/reference/src/feature_tracker.cpp, compiled from where it lies
// against the stand-in OpenCV of oracle/ref/standin_cv, whose three algorithms are the oracle's restatements): fbKltTracking, getLineMinSAD
// and inBorder as the reference wrote them.  tests/test_reference_factors.py compares them with oracle/frontend.c: orc_fb_klt and
// oracle/stereo.c: orc_line_min_sad -- status bytes and float bits.  TEST INFRASTRUCTURE ONLY.
#include "feature_tracker.hpp"

#include <opencv2/video/tracking.hpp>

extern "C" {
int orc_pyr_level_size(const void *p, int level, int *w, int *h);

// the std::vector<cv::Mat> cv::buildOpticalFlowPyramid(withDerivatives = true) returns: two entries per level
static std::vector<cv::Mat> pyr_vector(const void *orc_pyr, int n_levels)
{
    std::vector<cv::Mat> v;
    for (int l = 0; l < n_levels; l++) {
        int w = 0, h = 0;
        orc_pyr_level_size(orc_pyr, l, &w, &h);
        cv::Mat m; m.rows = h; m.cols = w; m.orc_pyr_handle = orc_pyr;
        v.push_back(m); v.push_back(m);                         // image, derivatives
    }
    return v;
}

// FeatureTracker::fbKltTracking on two oracle pyramids of n_levels levels.  prior_xy: in = priors, out = tracked positions.
int ref_fb_klt(const void *prev, const void *cur, int n_levels, int win, int nbpyrlvl, int max_iter, float eps_px, float ferr, float fmax_fbklt_dist,
               const float *kps_xy, float *prior_xy, int n, uint8_t *status_out)
{
    FeatureTracker trk(max_iter, eps_px, nullptr);
    std::vector<cv::Point2f> vkps((size_t)n), vpri((size_t)n);
    for (int i = 0; i < n; i++) { vkps[(size_t)i] = cv::Point2f(kps_xy[2 * i], kps_xy[2 * i + 1]); vpri[(size_t)i] = cv::Point2f(prior_xy[2 * i], prior_xy[2 * i + 1]); }
    std::vector<bool> st;
    trk.fbKltTracking(pyr_vector(prev, n_levels), pyr_vector(cur, n_levels), win, nbpyrlvl, ferr, fmax_fbklt_dist, vkps, vpri, st);
    if ((int)st.size() != n) return n == 0 ? 0 : -1;
    for (int i = 0; i < n; i++) { status_out[i] = st[(size_t)i] ? 1 : 0; prior_xy[2 * i] = vpri[(size_t)i].x; prior_xy[2 * i + 1] = vpri[(size_t)i].y; }
    return 0;
}

void ref_line_min_sad(const uint8_t *iml, int lstride, const uint8_t *imr, int rstride, int w, int h, float x, float y, int nwinsize, int go_left,
                      float *xprior, float *l1err)
{
    FeatureTracker trk(30, 0.01f, nullptr);
    const cv::Mat L = cv::Mat::wrap(iml, h, w, (size_t)lstride), R = cv::Mat::wrap(imr, h, w, (size_t)rstride);
    *l1err = 255.f;                                             // (the reference leaves it unset on its early returns; the oracle writes 255)
    trk.getLineMinSAD(L, R, cv::Point2f(x, y), nwinsize, *xprior, *l1err, go_left != 0);
}

int ref_in_border(float x, float y, int w, int h)
{
    FeatureTracker trk(30, 0.01f, nullptr);
    cv::Mat m; m.rows = h; m.cols = w;
    return trk.inBorder(cv::Point2f(x, y), m) ? 1 : 0;
}
}
