// compile_check.cpp -- keeps the C++ adapters compiling (g++ -fsyntax-only, see tests/test_abi.py)
#include "feature_tracker.hpp"
#include "feature_extractor.hpp"
#include "optimizer.hpp"
#include "camera_calibration.hpp"
#include "visual_front_end.hpp"
#include "slam_gpu.hpp"
int main() { return 0; }
