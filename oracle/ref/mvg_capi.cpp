// mvg_capi.cpp -- C entry point over the REFERENCE'S OWN MultiViewGeometry::computeSampsonDistance (/root/reference/src/multi_view_geometry.cpp:
// 798-814).  That translation unit also holds the RANSAC / OpenGV code, which cannot be compiled here, so the Makefile extracts the text of
// this one function AT BUILD TIME into oracle/_ref/gen_mvg_sampson.inc (git-ignored build output: nothing of the reference is committed)
// and this file compiles it against the stand-in Eigen of oracle/ref/standin.  tests/test_reference_factors.py compares it bit for bit
// with oracle/stereo.c: orc_sampson_distance -- the epipolar gate of MapManager::stereoMatching (src/map_manager.cpp:595), whose float /
// double narrowing points are easy to restate wrongly.  TEST INFRASTRUCTURE ONLY.
#include <Eigen/Core>

#include <cmath>

struct MultiViewGeometry {          // include/multi_view_geometry.hpp:121
    static float computeSampsonDistance(const Eigen::Matrix3d &Frl, const Eigen::Vector3d &leftpt, const Eigen::Vector3d &rightpt);
};

#include "gen_mvg_sampson.inc"

extern "C" float ref_sampson_distance(const double F[9], float lx, float ly, float rx, float ry)
{
    Eigen::Matrix3d Frl;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Frl(i, j) = F[3 * i + j];
    // the cv::Point2f overload (src/multi_view_geometry.cpp:816-821): float coordinates widened into homogeneous doubles
    Eigen::Vector3d lpt(lx, ly, 1.), rpt(rx, ry, 1.);
    return MultiViewGeometry::computeSampsonDistance(Frl, lpt, rpt);
}
