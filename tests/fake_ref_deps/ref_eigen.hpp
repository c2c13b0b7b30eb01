// Stand-in Eigen for the SYNTAX CHECK of the patched reference sources (tests/test_integration_patch.py): oracle/ref/standin's
// mini Eigen plus the few names the reference's headers mention.  Not a substitute for a real build.
#pragma once
#include <memory>
#include "../../oracle/ref/standin/mini_eigen.hpp"
namespace Eigen {
template <class T> using aligned_allocator = std::allocator<T>;
typedef Matrix<double, 3, 4> Matrix34d;
typedef Matrix<float, 3, 1> Vector3f;
}
#include <ostream>
namespace Eigen {
template <class D, int R, int C> std::ostream &operator<<(std::ostream &os, const Dense<D, R, C> &m) { for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) os << m(i, j) << ' '; return os; }
}
