// Stand-in <ceres/ceres.h> for the SYNTAX CHECK of the patched reference sources: the names src/optimizer.cpp and
// src/multi_view_geometry.cpp use, with the signatures of the vendored Ceres 2.0.0 (include/ceres/*.h).  Nothing is implemented.
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "../../../oracle/ref/standin/ceres/ceres.h"
namespace ceres {
enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };
enum DoglegType { TRADITIONAL_DOGLEG, SUBSPACE_DOGLEG };
class LossFunction { public: virtual ~LossFunction() {} };
class HuberLoss : public LossFunction { public: explicit HuberLoss(double a) : a_(a) {} double a_; };
class LossFunctionWrapper : public LossFunction { public: LossFunctionWrapper(LossFunction *, Ownership) {} void Reset(LossFunction *, Ownership) {} };
struct ResidualBlock;
typedef ResidualBlock *ResidualBlockId;
class ParameterBlockOrdering { public: bool AddElementToGroup(const double *, int) { return true; } };
class Problem {
public:
    void AddParameterBlock(double *, int) {}
    void AddParameterBlock(double *, int, LocalParameterization *) {}
    void SetParameterBlockConstant(const double *) {}
    void RemoveResidualBlock(ResidualBlockId) {}
    template <class... Ts> ResidualBlockId AddResidualBlock(CostFunction *, LossFunction *, Ts *...) { return nullptr; }
};
class Solver {
public:
    struct Options {
        std::shared_ptr<ParameterBlockOrdering> linear_solver_ordering;
        LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
        TrustRegionStrategyType trust_region_strategy_type = LEVENBERG_MARQUARDT;
        DoglegType dogleg_type = TRADITIONAL_DOGLEG;
        bool use_nonmonotonic_steps = false, minimizer_progress_to_stdout = false;
        int num_threads = 1, max_num_iterations = 50;
        double function_tolerance = 1e-6, max_solver_time_in_seconds = 1e9;
    };
    struct Summary { std::string FullReport() const { return ""; } std::string BriefReport() const { return ""; } bool IsSolutionUsable() const { return true; } };
};
inline void Solve(const Solver::Options &, Problem *, Solver::Summary *) {}
}  // namespace ceres
