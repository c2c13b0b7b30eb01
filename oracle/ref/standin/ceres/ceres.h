// ceres/ceres.h stand-in: the two interfaces the reference's factor file derives from (include/ceres/sized_cost_function.h,
// include/ceres/local_parameterization.h of the vendored Ceres 2.0.0).  TEST INFRASTRUCTURE ONLY -- see mini_eigen.hpp.
#pragma once
#include <cstddef>

namespace ceres {

class CostFunction {
public:
    virtual ~CostFunction() {}
    virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
};

template <int kNumResiduals, int... Ns>
class SizedCostFunction : public CostFunction {
public:
    static const int num_residuals = kNumResiduals;
};

class LocalParameterization {
public:
    virtual ~LocalParameterization() {}
    virtual bool Plus(const double *x, const double *delta, double *x_plus_delta) const = 0;
    virtual bool ComputeJacobian(const double *x, double *jacobian) const = 0;
    virtual int GlobalSize() const = 0;
    virtual int LocalSize() const = 0;
};

}  // namespace ceres
