// trlm_capi.cpp -- Ceres' OWN trust-region loop, executed (TEST INFRASTRUCTURE ONLY: nothing under ov2slam_amd/ loads this library).
//
// What runs from the reference tree, unchanged and from where it lies (/root/reference/Thirdparty/ceres-solver/internal/ceres):
//     trust_region_minimizer.cc      TrustRegionMinimizer::Minimize and every step of it (:67-829)
//     trust_region_step_evaluator.cc the step-quality bookkeeping
//     levenberg_marquardt_strategy.cc the LM diagonal, radius schedule
//     corrector.cc, loss_function.cc  HuberLoss, LossFunctionWrapper, the robustifier's corrector
//     minimizer.cc, array_utils.cc, types.cc, sparse_matrix.cc, linear_operator.cc, function_sample.cc, stringprintf.cc, wall_time.cc,
//     file.cc, miniglog/glog/logging.cc
// against Ceres' own headers and the stand-in Eigen of standin_dyn/ (Eigen is absent from this image), plus -- through the C entry
// points of libref_factors.so -- the reference's own factors and SE(3) parameterisation (src/ceres_parametrization.cpp).
//
// What this file supplies, because the Ceres classes that do it need all of Eigen (small_blas, LLT) and most of Ceres' program
// machinery (none of it part of the loop under test):
//   * FlatEvaluator : ceres::internal::Evaluator -- what ProgramEvaluator::Evaluate (program_evaluator.h:104-310) + ResidualBlock::Evaluate
//     (residual_block.cc:68-205) do for the flat problem of orc_ba_problem: reference factor -> local Jacobians (global Jacobian x
//     SE3LeftParameterization::ComputeJacobian) -> Ceres' Corrector with the reference's LossFunctionWrapper(HuberLoss) -> cost,
//     residuals, gradient J^T r, Jacobian.  Program layout as Ceres builds it for the reference (src/optimizer.cpp:95-407): constant
//     blocks removed, landmarks (elimination group 0) in front of the poses.
//   * RowJacobian : SparseMatrix -- a row-compressed Jacobian with the six operations the loop calls.
//   * ScalarSchurSolver : LinearSolver -- min |Ax - b|^2 + |Dx|^2 by eliminating the scalar landmark columns and a dense Cholesky of the
//     reduced system (what DENSE_SCHUR computes, schur_complement_solver.cc:130-180 / schur_eliminator_impl.h); written here
//     independently of oracle/ba.c, so the two agree to rounding only.
//   * abort()-ing definitions of the symbols trust_region_minimizer.cc / minimizer.cc link against but never reach with the reference's
//     options (inner iterations, bounds line search, LINE_SEARCH minimizer, problem dumps), and three empty virtual destructors whose
//     own translation units pull in every solver of the library.
// Neither a reference build nor a baseline: never timed, never shipped.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "ceres/coordinate_descent_minimizer.h"
#include "ceres/corrector.h"
#include "ceres/evaluator.h"
#include "ceres/levenberg_marquardt_strategy.h"
#include "ceres/line_search.h"
#include "ceres/line_search_minimizer.h"
#include "ceres/linear_least_squares_problems.h"
#include "ceres/linear_solver.h"
#include "ceres/loss_function.h"
#include "ceres/minimizer.h"
#include "ceres/sparse_matrix.h"
#include "ceres/trust_region_minimizer.h"
#include "ceres/trust_region_strategy.h"

#include "../ov2_oracle.h"          // orc_ba_problem / orc_ba_options / orc_ba_result: the flat layout the oracle and the tests use

extern "C" {
// libref_factors.so (factors_capi.cpp): the reference's factors and parameterisation
int ref_factor_eval(int type, const double *const *params, const double uv[2], const double anch_uv[2], double sigma, const double pnp_K[4],
                    const double pnp_xyz[3], double *residuals, double **jacobians, double *chi2, int *depthpos);
int ref_se3_plus(const double x[7], const double delta[6], double out[7]);
int ref_se3_plus_jacobian(const double x[7], double J[42]);
}

namespace ceres {
namespace internal {

// ---- never reached with the reference's options (src/optimizer.cpp:436-467): defined so that the library loads ----------------------
#define TRLM_UNREACHED(what) do { fprintf(stderr, "libref_trlm: %s is outside the loop under test\n", what); abort(); } while (0)
CoordinateDescentMinimizer::~CoordinateDescentMinimizer() {}
void CoordinateDescentMinimizer::Minimize(const Minimizer::Options &, double *, Solver::Summary *) { TRLM_UNREACHED("CoordinateDescentMinimizer (inner iterations)"); }
void LineSearchMinimizer::Minimize(const Minimizer::Options &, double *, Solver::Summary *) { TRLM_UNREACHED("LineSearchMinimizer"); }
LineSearch *LineSearch::Create(const LineSearchType, const LineSearch::Options &, std::string *) { TRLM_UNREACHED("LineSearch::Create (bounds constraints)"); }
void LineSearch::Search(double, double, double, Summary *) const { TRLM_UNREACHED("LineSearch::Search"); }
LineSearchFunction::LineSearchFunction(Evaluator *evaluator) : evaluator_(evaluator), position_(0), direction_(0), scaled_direction_(0), initial_evaluator_residual_time_in_seconds(0), initial_evaluator_jacobian_time_in_seconds(0) { TRLM_UNREACHED("LineSearchFunction"); }
void LineSearchFunction::Init(const Vector &, const Vector &) { TRLM_UNREACHED("LineSearchFunction::Init"); }
bool DumpLinearLeastSquaresProblem(const std::string &, DumpFormatType, const SparseMatrix *, const double *, const double *, const double *, int) { TRLM_UNREACHED("DumpLinearLeastSquaresProblem"); }
// (evaluator.cc / linear_solver.cc / trust_region_strategy.cc define these next to factories that reference every evaluator, solver and
// strategy of the library)
Evaluator::~Evaluator() {}
LinearSolver::~LinearSolver() {}
TrustRegionStrategy::~TrustRegionStrategy() {}

// ---- Jacobian ---------------------------------------------------------------------------------------------------------------------------
class RowJacobian : public SparseMatrix {
public:
    int nrows, ncols;
    std::vector<int> ptr, col;        // row r: entries ptr[r] .. ptr[r + 1]
    std::vector<double> val;
    RowJacobian(int r, int c) : nrows(r), ncols(c), ptr((size_t)r + 1, 0) {}
    void RightMultiply(const double *x, double *y) const override { for (int r = 0; r < nrows; r++) { double s = 0; for (int k = ptr[r]; k < ptr[r + 1]; k++) s += val[k] * x[col[k]]; y[r] += s; } }
    void LeftMultiply(const double *x, double *y) const override { for (int r = 0; r < nrows; r++) for (int k = ptr[r]; k < ptr[r + 1]; k++) y[col[k]] += val[k] * x[r]; }
    void SquaredColumnNorm(double *x) const override { for (int c = 0; c < ncols; c++) x[c] = 0; for (size_t k = 0; k < val.size(); k++) x[col[k]] += val[k] * val[k]; }
    void ScaleColumns(const double *scale) override { for (size_t k = 0; k < val.size(); k++) val[k] *= scale[col[k]]; }
    void SetZero() override { for (double &v : val) v = 0; }
    void ToDenseMatrix(Matrix *dense) const override { dense->resize(nrows, ncols); dense->setZero(); for (int r = 0; r < nrows; r++) for (int k = ptr[r]; k < ptr[r + 1]; k++) (*dense)(r, col[k]) = val[k]; }
    void ToTextFile(FILE *f) const override { for (int r = 0; r < nrows; r++) for (int k = ptr[r]; k < ptr[r + 1]; k++) fprintf(f, "%d %d %.17g\n", r, col[k], val[k]); }
    double *mutable_values() override { return val.data(); }
    const double *values() const override { return val.data(); }
    int num_rows() const override { return nrows; }
    int num_cols() const override { return ncols; }
    int num_nonzeros() const override { return (int)val.size(); }
};

// ---- evaluator over the flat problem ----------------------------------------------------------------------------------------------------
struct FlatProgram {
    const orc_ba_problem *p;
    std::vector<int> act;             // active residual blocks, in the order of the flat arrays (= AddResidualBlock order)
    std::vector<int> lm_state, lm_col;       // per landmark: offset in the state vector / Jacobian column, -1 = not in the program
    std::vector<int> kf_state, kf_col;       // per keyframe: -1 = constant
    int n_state = 0, n_cols = 0, n_e = 0;    // n_e: landmark columns (all in front)
    std::vector<double> chi2; std::vector<unsigned char> depthpos;      // the factors' mutable members after their LAST Evaluate
};

class FlatEvaluator : public Evaluator {
public:
    FlatProgram &P;
    std::unique_ptr<LossFunction> loss;       // LossFunctionWrapper(HuberLoss(delta), TAKE_OWNERSHIP) as src/optimizer.cpp:49; NULL: trivial loss (:52)
    int n_evals = 0, n_jac_evals = 0;
    FlatEvaluator(FlatProgram &prog, double huber_delta) : P(prog)
    {
        if (huber_delta > 0) loss.reset(new LossFunctionWrapper(new HuberLoss(huber_delta), TAKE_OWNERSHIP));
    }
    SparseMatrix *CreateJacobian() const override
    {
        RowJacobian *J = new RowJacobian(2 * (int)P.act.size(), P.n_cols);
        const orc_ba_problem *p = P.p;
        for (size_t k = 0; k < P.act.size(); k++) {
            const int i = P.act[k];
            int cols[13], nc = 0;
            block_columns(i, cols, &nc);
            for (int row = 0; row < 2; row++) {
                for (int c = 0; c < nc; c++) { J->col.push_back(cols[c]); J->val.push_back(0.0); }
                J->ptr[2 * k + row + 1] = (int)J->col.size();
            }
            (void)p;
        }
        return J;
    }
    // Jacobian columns of residual block i in the order Evaluate writes them: [lambda][anchor pose 6][observing pose 6]
    void block_columns(int i, int *cols, int *nc) const
    {
        const orc_ba_problem *p = P.p;
        int n = 0;
        if (p->res_type[i] == ORC_RES_PNP) { const int co = P.kf_col[p->res_kf[i]]; if (co >= 0) for (int c = 0; c < 6; c++) cols[n++] = co + c; *nc = n; return; }
        const int lm = p->res_lm[i];
        cols[n++] = P.lm_col[lm];
        if (p->res_type[i] != ORC_RES_RIGHT_ANCH) {
            const int ca = P.kf_col[p->lm_anchor_kf[lm]], co = P.kf_col[p->res_kf[i]];
            if (ca >= 0) for (int c = 0; c < 6; c++) cols[n++] = ca + c;
            if (co >= 0) for (int c = 0; c < 6; c++) cols[n++] = co + c;
        }
        *nc = n;
    }
    const double *pose_of(int kf, const double *state) const { return P.kf_state[kf] >= 0 ? state + P.kf_state[kf] : P.p->poses + 7 * kf; }

    bool Evaluate(const EvaluateOptions &eo, const double *state, double *cost, double *residuals, double *gradient, SparseMatrix *jacobian) override
    {
        const orc_ba_problem *p = P.p;
        RowJacobian *J = static_cast<RowJacobian *>(jacobian);
        const bool want_jac = J != nullptr || gradient != nullptr;                  // program_evaluator.h:148-160
        n_evals++; n_jac_evals += want_jac;
        *cost = 0.0;
        if (gradient) for (int c = 0; c < P.n_cols; c++) gradient[c] = 0.0;
        for (size_t k = 0; k < P.act.size(); k++) {
            const int i = P.act[k], type = p->res_type[i];
            const int lm = type == ORC_RES_PNP ? -1 : p->res_lm[i];
            const int a = lm >= 0 ? p->lm_anchor_kf[lm] : -1, o = p->res_kf[i];
            double lam = lm >= 0 ? state[P.lm_state[lm]] : 0.0;
            // parameter blocks in the factor's order (factors_capi.cpp) and which of them are variable
            const double *par[6]; int size[6], var_col[6], np = 0;
            auto add = [&](const double *q, int sz, int col) { par[np] = q; size[np] = sz; var_col[np] = col; np++; };
            if (type == ORC_RES_LEFT) { add(p->calib_l, 4, -1); add(pose_of(a, state), 7, P.kf_col[a]); add(pose_of(o, state), 7, P.kf_col[o]); add(&lam, 1, P.lm_col[lm]); }
            else if (type == ORC_RES_RIGHT) { add(p->calib_l, 4, -1); add(p->calib_r, 4, -1); add(pose_of(a, state), 7, P.kf_col[a]); add(pose_of(o, state), 7, P.kf_col[o]); add(p->T_rl, 7, -1); add(&lam, 1, P.lm_col[lm]); }
            else if (type == ORC_RES_RIGHT_ANCH) { add(p->calib_l, 4, -1); add(p->calib_r, 4, -1); add(p->T_rl, 7, -1); add(&lam, 1, P.lm_col[lm]); }
            else if (type == ORC_RES_PNP) { add(pose_of(o, state), 7, P.kf_col[o]); }
            else return false;
            // global Jacobians of the variable blocks only (a NULL entry = "not wanted", residual_block.cc:84-96; constant blocks are not in
            // the reduced program)
            double Jg[6][14]; double *jg[6];
            for (int b = 0; b < np; b++) jg[b] = (want_jac && var_col[b] >= 0) ? Jg[b] : nullptr;
            double r[2], chi2 = 0; int dpos = 0;
            const double zero3[3] = {0, 0, 0};
            if (ref_factor_eval(type, par, p->res_uv + 2 * i, lm >= 0 ? p->lm_anchor_uv + 2 * lm : zero3, p->res_sigma[i], p->calib_l,
                                type == ORC_RES_PNP ? p->res_xyz + 3 * i : zero3, r, want_jac ? jg : nullptr, &chi2, &dpos) != 0) return false;
            P.chi2[i] = chi2; P.depthpos[i] = (unsigned char)dpos;
            const double sq = r[0] * r[0] + r[1] * r[1];                              // residual_block.cc:131
            // local Jacobians (residual_block.cc:134-156), laid out [lambda][anchor][obs] = the block's columns
            double Jl[2 * 13]; int cols[13], nc = 0;
            if (want_jac) {
                block_columns(i, cols, &nc);
                int at = 0;
                auto put_pose = [&](int b) {
                    double Jp[42];
                    ref_se3_plus_jacobian(par[b], Jp);                              // 7 x 6 row-major (se3left_parametrization.hpp:59-69)
                    for (int row = 0; row < 2; row++) for (int c = 0; c < 6; c++) { double s = 0; for (int q = 0; q < 7; q++) s += Jg[b][row * 7 + q] * Jp[q * 6 + c]; Jl[row * nc + at + c] = s; }
                    at += 6;
                };
                // the lambda block first (it is the LAST block of every factor), then the poses in factor order
                if (type != ORC_RES_PNP) { Jl[0 * nc + 0] = Jg[np - 1][0]; Jl[1 * nc + 0] = Jg[np - 1][1]; at = 1; }
                for (int b = 0; b < np; b++) if (size[b] == 7 && var_col[b] >= 0) put_pose(b);
            }
            if (!loss || !eo.apply_loss_function) *cost += 0.5 * sq;                 // residual_block.cc:158-161
            else {
                double rho[3];
                loss->Evaluate(sq, rho);
                *cost += 0.5 * rho[0];
                if (want_jac || residuals) {                                         // :167-171
                    Corrector correct(sq, rho);
                    // one call per parameter block in Ceres (:175-188); the correction is column-wise, so one call over the block row is the same
                    if (want_jac) correct.CorrectJacobian(2, nc, r, Jl);
                    correct.CorrectResiduals(2, r);                                  // :191
                }
            }
            if (residuals) { residuals[2 * k] = r[0]; residuals[2 * k + 1] = r[1]; }
            if (J) for (int row = 0; row < 2; row++) for (int c = 0; c < nc; c++) J->val[(size_t)J->ptr[2 * k + row] + c] = Jl[row * nc + c];
            if (gradient) for (int c = 0; c < nc; c++) gradient[cols[c]] += Jl[c] * r[0] + Jl[nc + c] * r[1];      // program_evaluator.h:258-276
        }
        return true;
    }
    bool Plus(const double *state, const double *delta, double *out) const override
    {
        const orc_ba_problem *p = P.p;
        for (int l = 0; l < p->n_lm; l++) if (P.lm_state[l] >= 0) out[P.lm_state[l]] = state[P.lm_state[l]] + delta[P.lm_col[l]];
        for (int k = 0; k < p->n_kf; k++) if (P.kf_state[k] >= 0 && ref_se3_plus(state + P.kf_state[k], delta + P.kf_col[k], out + P.kf_state[k]) != 0) return false;
        return true;
    }
    int NumParameters() const override { return P.n_state; }
    int NumEffectiveParameters() const override { return P.n_cols; }
    int NumResiduals() const override { return 2 * (int)P.act.size(); }
};

// ---- linear solver: Schur complement over the scalar landmark columns, dense Cholesky of the pose system ------------------------------
class ScalarSchurSolver : public LinearSolver {
public:
    int n_e;
    int n_solves = 0;
    explicit ScalarSchurSolver(int ne) : n_e(ne) {}
    Summary Solve(LinearOperator *A_, const double *b, const PerSolveOptions &ps, double *x) override
    {
        n_solves++;
        const RowJacobian *A = static_cast<const RowJacobian *>(A_);
        const int n = A->ncols, nf = n - n_e;
        const double *D = ps.D;
        std::vector<double> ee((size_t)n_e, 0.0), eb((size_t)n_e, 0.0), S((size_t)nf * nf, 0.0), g((size_t)nf, 0.0);
        std::vector<std::vector<std::pair<int, double> > > W((size_t)n_e);              // E^T F, per landmark: (pose column, value), accumulated
        for (int e = 0; e < n_e; e++) ee[e] = D ? D[e] * D[e] : 0.0;
        for (int f = 0; f < nf; f++) S[(size_t)f * nf + f] = D ? D[n_e + f] * D[n_e + f] : 0.0;
        std::vector<double> wrow((size_t)nf, 0.0);
        for (int r = 0; r < A->nrows; r++) {
            int e = -1; double ev = 0;
            for (int k = A->ptr[r]; k < A->ptr[r + 1]; k++) if (A->col[k] < n_e) { e = A->col[k]; ev = A->val[k]; }
            for (int k = A->ptr[r]; k < A->ptr[r + 1]; k++) {
                const int c = A->col[k]; const double v = A->val[k];
                if (c < n_e) { ee[c] += v * v; eb[c] += v * b[r]; continue; }
                g[c - n_e] += v * b[r];
                for (int k2 = A->ptr[r]; k2 < A->ptr[r + 1]; k2++) if (A->col[k2] >= n_e) S[(size_t)(c - n_e) * nf + A->col[k2] - n_e] += v * A->val[k2];
                if (e >= 0) {
                    std::vector<std::pair<int, double> > &w = W[(size_t)e];
                    size_t q = 0; for (; q < w.size(); q++) if (w[q].first == c - n_e) break;
                    if (q == w.size()) w.push_back(std::make_pair(c - n_e, 0.0));
                    w[q].second += ev * v;
                }
            }
        }
        for (int e = 0; e < n_e; e++) {
            const std::vector<std::pair<int, double> > &w = W[(size_t)e];
            const double inv = 1.0 / ee[e];
            for (size_t a = 0; a < w.size(); a++) {
                g[w[a].first] -= w[a].second * inv * eb[e];
                for (size_t c = 0; c < w.size(); c++) S[(size_t)w[a].first * nf + w[c].first] -= w[a].second * inv * w[c].second;
            }
        }
        Summary s;
        s.num_iterations = 1;
        // dense Cholesky (lower), in place
        for (int j = 0; j < nf; j++) {
            double d = S[(size_t)j * nf + j];
            for (int k = 0; k < j; k++) d -= S[(size_t)j * nf + k] * S[(size_t)j * nf + k];
            if (!(d > 0.0)) { s.termination_type = LINEAR_SOLVER_FAILURE; s.message = "reduced system not positive definite"; return s; }
            d = std::sqrt(d);
            S[(size_t)j * nf + j] = d;
            for (int i = j + 1; i < nf; i++) {
                double v = S[(size_t)i * nf + j];
                for (int k = 0; k < j; k++) v -= S[(size_t)i * nf + k] * S[(size_t)j * nf + k];
                S[(size_t)i * nf + j] = v / d;
            }
        }
        std::vector<double> y(g);
        for (int i = 0; i < nf; i++) { double v = y[i]; for (int k = 0; k < i; k++) v -= S[(size_t)i * nf + k] * y[k]; y[i] = v / S[(size_t)i * nf + i]; }
        for (int i = nf - 1; i >= 0; i--) { double v = y[i]; for (int k = i + 1; k < nf; k++) v -= S[(size_t)k * nf + i] * y[k]; y[i] = v / S[(size_t)i * nf + i]; }
        for (int f = 0; f < nf; f++) x[n_e + f] = y[f];
        for (int e = 0; e < n_e; e++) {
            double v = eb[e];
            for (const std::pair<int, double> &w : W[(size_t)e]) v -= w.second * y[w.first];
            x[e] = v / ee[e];
        }
        s.termination_type = LINEAR_SOLVER_SUCCESS;
        s.message = "Success.";
        return s;
    }
};

}  // namespace internal
}  // namespace ceres

// ---- C entry point ----------------------------------------------------------------------------------------------------------------------------
extern "C" {

typedef struct {
    int iteration, step_is_valid, step_is_successful, linear_solver_iterations;
    double cost, cost_change, gradient_max_norm, gradient_norm, step_norm, relative_decrease, trust_region_radius, eta;
} ref_trlm_iter;

// One ceres::Solve of the reference's local BA on the flat problem, by Ceres' own TrustRegionMinimizer.  res: as orc_ba_solve fills it
// (iterations = linear solves, i.e. trust-region steps computed; chi2 / depthpos = the factors' members after their last Evaluate);
// trace[0 .. *n_trace): the IterationSummary entries Ceres recorded (entry 0 = the starting point); counts[0..3] = num_successful_steps,
// num_unsuccessful_steps, cost evaluations, Jacobian evaluations; msg: Solver::Summary::message.  Returns 0, -1 on bad input.
int ref_trlm_solve(const orc_ba_problem *p, const orc_ba_options *o, orc_ba_result *res, ref_trlm_iter *trace, int trace_cap, int *n_trace,
                   int counts[4], char *msg, int msg_cap)
{
    using namespace ceres;
    using namespace ceres::internal;
    if (!p || !o || !res || p->n_kf <= 0) return -1;
    FlatProgram P;
    P.p = p;
    P.lm_state.assign((size_t)p->n_lm, -1); P.lm_col.assign((size_t)p->n_lm, -1);
    P.kf_state.assign((size_t)p->n_kf, -1); P.kf_col.assign((size_t)p->n_kf, -1);
    P.chi2.assign((size_t)p->n_res + 1, 0.0); P.depthpos.assign((size_t)p->n_res + 1, 0);
    for (int i = 0; i < p->n_res; i++) {
        if (res->chi2_last_eval) P.chi2[i] = res->chi2_last_eval[i];
        if (res->depthpos_last_eval) P.depthpos[i] = res->depthpos_last_eval[i];
        if (p->res_active && !p->res_active[i]) continue;
        if (p->res_type[i] != ORC_RES_PNP) { const int lm = p->res_lm[i]; if (lm < 0 || lm >= p->n_lm) return -1; P.lm_state[lm] = 0; }
        P.act.push_back(i);
    }
    // the reduced program: landmarks with a residual block (elimination group 0) in front, then the non-constant keyframes (group 1)
    for (int l = 0; l < p->n_lm; l++) if (P.lm_state[l] == 0) { P.lm_state[l] = P.n_state++; P.lm_col[l] = P.n_cols++; }
    P.n_e = P.n_cols;
    for (int k = 0; k < p->n_kf; k++) if (!p->kf_const[k]) { P.kf_state[k] = P.n_state; P.n_state += 7; P.kf_col[k] = P.n_cols; P.n_cols += 6; }
    std::vector<double> x((size_t)P.n_state + 1, 0.0);
    for (int l = 0; l < p->n_lm; l++) if (P.lm_state[l] >= 0) x[(size_t)P.lm_state[l]] = p->invdepth[l];
    for (int k = 0; k < p->n_kf; k++) if (P.kf_state[k] >= 0) memcpy(&x[(size_t)P.kf_state[k]], p->poses + 7 * k, 7 * sizeof(double));

    // Solver::Options as src/optimizer.cpp:436-467 sets them (everything else: Ceres' defaults), then what TrustRegionPreprocessor does
    // with them (trust_region_preprocessor.cc:330-372)
    Solver::Options so;
    so.linear_solver_type = DENSE_SCHUR;
    so.trust_region_strategy_type = LEVENBERG_MARQUARDT;
    so.num_threads = 1;
    so.max_num_iterations = o->max_iter;
    so.function_tolerance = o->function_tolerance;
    so.gradient_tolerance = o->gradient_tolerance;
    so.parameter_tolerance = o->parameter_tolerance;
    so.initial_trust_region_radius = o->initial_radius;
    so.max_trust_region_radius = o->max_radius;
    so.min_trust_region_radius = o->min_radius;
    so.min_lm_diagonal = o->min_lm_diagonal;
    so.max_lm_diagonal = o->max_lm_diagonal;
    so.min_relative_decrease = o->min_relative_decrease;
    so.jacobi_scaling = o->jacobi_scaling != 0;
    so.max_num_consecutive_invalid_steps = o->max_consecutive_invalid_steps;
    so.max_solver_time_in_seconds = 1e9;               // results independent of machine load (the library's default too)
    so.minimizer_progress_to_stdout = false;
    so.logging_type = SILENT;
    Minimizer::Options mo(so);
    mo.is_silent = true;
    std::shared_ptr<FlatEvaluator> ev(new FlatEvaluator(P, o->huber_delta));
    mo.evaluator = ev;
    mo.jacobian.reset(ev->CreateJacobian());
    ScalarSchurSolver linear_solver(P.n_e);
    TrustRegionStrategy::Options ts;
    ts.linear_solver = &linear_solver;
    ts.initial_radius = so.initial_trust_region_radius;
    ts.max_radius = so.max_trust_region_radius;
    ts.min_lm_diagonal = so.min_lm_diagonal;
    ts.max_lm_diagonal = so.max_lm_diagonal;
    ts.trust_region_strategy_type = so.trust_region_strategy_type;
    ts.dogleg_type = so.dogleg_type;
    mo.trust_region_strategy.reset(new LevenbergMarquardtStrategy(ts));
    Solver::Summary summary;
    summary.fixed_cost = 0.0;                          // solver.cc: pp.fixed_cost, the cost of residual blocks with constant blocks only (none here)
    TrustRegionMinimizer minimizer;
    minimizer.Minimize(mo, x.data(), &summary);

    for (int l = 0; l < p->n_lm; l++) if (res->invdepth_out) res->invdepth_out[l] = P.lm_state[l] >= 0 ? x[(size_t)P.lm_state[l]] : p->invdepth[l];
    for (int k = 0; k < p->n_kf; k++) if (res->poses_out) memcpy(res->poses_out + 7 * k, P.kf_state[k] >= 0 ? &x[(size_t)P.kf_state[k]] : p->poses + 7 * k, 7 * sizeof(double));
    for (int i = 0; i < p->n_res; i++) {
        if (res->chi2_last_eval) res->chi2_last_eval[i] = P.chi2[i];
        if (res->depthpos_last_eval) res->depthpos_last_eval[i] = P.depthpos[i];
    }
    res->iterations = linear_solver.n_solves;
    res->num_successful_steps = summary.num_successful_steps;
    res->initial_cost = summary.iterations.empty() ? 0.0 : summary.iterations.front().cost;
    double fc = res->initial_cost;                     // the minimum over the accepted points (trust_region_minimizer.cc:609-617)
    for (const IterationSummary &it : summary.iterations) if (it.step_is_successful && it.cost < fc) fc = it.cost;
    res->final_cost = fc;
    const std::string &m = summary.message;
    int term = ORC_TERM_FAILURE;
    if (m.find("Function tolerance reached") != std::string::npos) term = ORC_TERM_FUNCTION_TOL;
    else if (m.find("Parameter tolerance reached") != std::string::npos) term = ORC_TERM_PARAMETER_TOL;
    else if (m.find("Gradient tolerance reached") != std::string::npos) term = ORC_TERM_GRADIENT_TOL;
    else if (m.find("Maximum number of iterations reached") != std::string::npos) term = ORC_TERM_NO_CONVERGENCE;
    else if (m.find("Minimum trust region radius reached") != std::string::npos) term = ORC_TERM_MIN_RADIUS;
    else if (m.find("Number of consecutive invalid steps") != std::string::npos) term = ORC_TERM_INVALID_STEPS;
    res->termination = term;
    if (n_trace) {
        int n = 0;
        for (const IterationSummary &it : summary.iterations) {
            if (trace && n < trace_cap) {
                ref_trlm_iter &t = trace[n];
                t.iteration = it.iteration; t.step_is_valid = it.step_is_valid; t.step_is_successful = it.step_is_successful;
                t.linear_solver_iterations = it.linear_solver_iterations;
                t.cost = it.cost; t.cost_change = it.cost_change; t.gradient_max_norm = it.gradient_max_norm; t.gradient_norm = it.gradient_norm;
                t.step_norm = it.step_norm; t.relative_decrease = it.relative_decrease; t.trust_region_radius = it.trust_region_radius; t.eta = it.eta;
            }
            n++;
        }
        *n_trace = n;
    }
    if (counts) { counts[0] = summary.num_successful_steps; counts[1] = summary.num_unsuccessful_steps; counts[2] = ev->n_evals; counts[3] = ev->n_jac_evals; }
    if (msg && msg_cap > 0) { strncpy(msg, m.c_str(), (size_t)msg_cap - 1); msg[msg_cap - 1] = 0; }
    return 0;
}

}  // extern "C"
