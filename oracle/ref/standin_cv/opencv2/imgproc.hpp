// Stand-in for <opencv2/imgproc.hpp> (see core.hpp): cv::getRectSubPix through the oracle's restatement
#pragma once
#include "core.hpp"
extern "C" void orc_get_rect_subpix_8u(const uint8_t *src, int src_step, int sw, int sh, uint8_t *dst, int pw, int ph, float cx_f, float cy_f);
namespace cv {
inline void getRectSubPix(const Mat &src, Size sz, Point2f c, Mat &dst)
{
    dst.own = std::make_shared<std::vector<uint8_t>>((size_t)sz.width * sz.height);
    dst.rows = sz.height; dst.cols = sz.width; dst.step = (size_t)sz.width; dst.data = dst.own->data();
    orc_get_rect_subpix_8u(src.data, (int)src.step, src.cols, src.rows, dst.own->data(), sz.width, sz.height, c.x, c.y);
}
}   // namespace cv
