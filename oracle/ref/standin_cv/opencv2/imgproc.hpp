// Stand-in for <opencv2/imgproc.hpp> (see core.hpp): the algorithms are the oracle's restatements
#pragma once
#include "core.hpp"
extern "C" {
void orc_get_rect_subpix_8u(const uint8_t *src, int src_step, int sw, int sh, uint8_t *dst, int pw, int ph, float cx_f, float cy_f);
void orc_circle_fill0(uint8_t *mask, int w, int h, int cx, int cy, int radius);
void orc_cell_mineig(const uint8_t *img, int w, int h, int stride, int x0, int y0, int cell, float *hmap);
void orc_corner_subpix(const uint8_t *img, int w, int h, int stride, float *xy, int n, int half_win, int max_iters, double eps);
}
namespace cv {
inline void getRectSubPix(const Mat &src, Size sz, Point2f c, Mat &dst)
{
    dst.create(sz.height, sz.width, CV_8U);
    orc_get_rect_subpix_8u(src.data, (int)src.step, src.cols, src.rows, dst.data, sz.width, sz.height, c.x, c.y);
}
// cv::circle(img, centre, radius, 0, FILLED) on a whole u8 or float image: the pixel set of the oracle's midpoint circle
inline void circle(Mat &img, Point c, int radius, const Scalar &color, int thickness)
{
    assert(thickness == -1 && img.data == img.whole);
    std::vector<uint8_t> m((size_t)img.rows * img.cols, 1);
    orc_circle_fill0(m.data(), img.cols, img.rows, c.x, c.y, radius);
    for (int i = 0; i < img.rows; i++) for (int j = 0; j < img.cols; j++)
        if (!m[(size_t)i * img.cols + j]) { if (img.type_ == CV_32F) img.f(i, j) = (float)color.v; else img.data[i * img.step + j] = (uint8_t)color.v; }
}
// GaussianBlur(im(roi), dst, 3x3) then cornerMinEigenVal(dst, hmap, 3, 3): the oracle restates the PAIR (the blur of a ROI view reads its
// neighbours in the parent image, the eigenvalue map treats the blurred cell as a standalone image): the blur records where the cell lies
struct BlurredCell { const uint8_t *img; int w, h, stride, x0, y0, cell; };
inline void GaussianBlur(const Mat &src, Mat &dst, Size k, double)
{
    assert(k.width == 3 && k.height == 3 && src.type_ == CV_8U && src.rows == src.cols);
    int x0, y0; src.roi_origin(x0, y0);
    dst.create(1, (int)sizeof(BlurredCell), CV_8U);
    BlurredCell b{src.whole, src.whole_cols, src.whole_rows, (int)src.step, x0, y0, src.rows};
    memcpy(dst.data, &b, sizeof(b));
}
inline void cornerMinEigenVal(const Mat &blurred, Mat &hmap, int block, int ksize)
{
    assert(block == 3 && ksize == 3 && blurred.cols == (int)sizeof(BlurredCell));
    BlurredCell b; memcpy(&b, blurred.data, sizeof(b));
    hmap.create(b.cell, b.cell, CV_32F);
    orc_cell_mineig(b.img, b.w, b.h, b.stride, b.x0, b.y0, b.cell, (float *)hmap.data);
}
inline void cornerSubPix(const Mat &im, std::vector<Point2f> &pts, Size win, Size, TermCriteria crit)
{
    static_assert(sizeof(Point2f) == 8, "points are float pairs");
    if (!pts.empty()) orc_corner_subpix(im.data, im.cols, im.rows, (int)im.step, &pts[0].x, (int)pts.size(), win.width, crit.maxCount, crit.epsilon);
}
}   // namespace cv
#include "features2d.hpp"     // (the real imgproc.hpp does not pull it in; feature_extractor.cpp gets it through frame.hpp -> ... in the reference tree)
