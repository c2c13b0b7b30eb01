/*
 * ba.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Anchored-inverse-depth local bundle adjustment: the two ceres::Solve calls of
 * Optimizer::localBA (src/optimizer.cpp:479, :618) restated without Ceres/Eigen/Sophus.
 * Files followed (all under /root/reference):
 *   residuals / Jacobians : src/ceres_parametrization.cpp:361-473 (left), :476-577 (right at
 *                           anchor), :579-712 (right); local parameterisation
 *                           include/ceres_parametrization/ceres_parametrization/se3left_parametrization.hpp:39-73
 *   pose retraction       : Thirdparty/Sophus/sophus/se3.hpp:763-784, so3.hpp:585-621
 *   Ceres 2.0.0 (Thirdparty/ceres-solver/internal/ceres/):
 *     loss_function.cc:48-62 (Huber), corrector.cc:42-156, residual_block.cc:69-200,
 *     trust_region_minimizer.cc (whole loop), trust_region_step_evaluator.cc:52-100,
 *     levenberg_marquardt_strategy.cc:66-160, schur_eliminator_impl.h:179-377,
 *     schur_complement_solver.cc:118-176 (+ dense Cholesky of the reduced system)
 * Pinned by the known-answer values of loss_function_test.cc:92-104,
 * corrector_test.cc:58-140, levenberg_marquardt_strategy_test.cc:81-111 (tests/test_oracle_ba.py)
 * and by a dense numpy LM cross-check; localBA outputs themselves have no golden vectors.
 * Wall-clock limits (max_solver_time_in_seconds) are deliberately not restated.
 */
#include "ov2_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* ---------------- small algebra ---------------- */
static void quat_normalize(double q[4])
{
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* q = [x y z w] -> row-major R (Eigen::Quaternion::toRotationMatrix) */
static void quat_to_R(const double q[4], double R[9])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

static void mat3_vec(const double R[9], const double v[3], double o[3])
{
    o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}

static void mat3T_vec(const double R[9], const double v[3], double o[3])
{
    o[0] = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
    o[1] = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
    o[2] = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
}

void orc_se3_exp(const double a[6], double t_out[3], double q_out[4])
{
    const double *omega = a + 3;
    const double theta_sq = omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2];
    double theta, imag, real;
    if (theta_sq < 1e-10 * 1e-10) {
        theta = 0;
        const double t4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * t4;
    } else {
        theta = sqrt(theta_sq);
        const double half = 0.5 * theta;
        imag = sin(half) / theta;
        real = cos(half);
    }
    q_out[0] = imag * omega[0]; q_out[1] = imag * omega[1]; q_out[2] = imag * omega[2]; q_out[3] = real;
    /* V = I + (1-cos)/theta^2 * Omega + (theta - sin)/theta^3 * Omega^2, or R if theta tiny */
    double V[9];
    const double O[9] = {0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0};
    if (theta < 1e-10) {
        quat_to_R(q_out, V);
    } else {
        double O2[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
        const double c1 = (1.0 - cos(theta)) / theta_sq, c2 = (theta - sin(theta)) / (theta_sq * theta);
        for (int k = 0; k < 9; k++) V[k] = c1 * O[k] + c2 * O2[k];
        V[0] += 1; V[4] += 1; V[8] += 1;
    }
    mat3_vec(V, a, t_out);
}

void orc_se3_left_plus(const double pose[7], const double delta[6], double out[7])
{
    double te[3], qe[4];
    orc_se3_exp(delta, te, qe);
    double q[4] = {pose[3], pose[4], pose[5], pose[6]};
    quat_normalize(q);                       /* Sophus::SE3d(q, t) normalises */
    /* quaternion product a*b with a = qe, b = q (Sophus so3.hpp operator*) */
    const double ax = qe[0], ay = qe[1], az = qe[2], aw = qe[3];
    const double bx = q[0], by = q[1], bz = q[2], bw = q[3];
    double r[4];
    r[3] = aw * bw - ax * bx - ay * by - az * bz;
    r[0] = aw * bx + ax * bw + ay * bz - az * by;
    r[1] = aw * by + ay * bw + az * bx - ax * bz;
    r[2] = aw * bz + az * bw + ax * by - ay * bx;
    quat_normalize(r);
    double Re[9], rt[3];
    quat_to_R(qe, Re);
    mat3_vec(Re, pose, rt);
    out[0] = te[0] + rt[0]; out[1] = te[1] + rt[1]; out[2] = te[2] + rt[2];
    out[3] = r[0]; out[4] = r[1]; out[5] = r[2]; out[6] = r[3];
}

/* ---------------- loss / corrector / LM radius ---------------- */
void orc_huber(double a, double s, double rho[3])
{
    const double b = a * a;
    if (s > b) {
        const double r = sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = a / r > DBL_MIN ? a / r : DBL_MIN;   /* std::max(numeric_limits<double>::min(), a_/r) */
        rho[2] = -rho[1] / (2.0 * s);
    } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
}

void orc_corrector(double sq_norm, const double rho[3], int n_rows, int n_cols,
                   double *residuals, double *jacobian)
{
    const double sqrt_rho1 = sqrt(rho[1]);
    double residual_scaling, alpha_sq_norm;
    if (sq_norm == 0.0 || rho[2] <= 0.0) {
        residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0;
    } else {
        const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
        const double alpha = 1.0 - sqrt(D);
        residual_scaling = sqrt_rho1 / (1 - alpha);
        alpha_sq_norm = alpha / sq_norm;
    }
    if (jacobian) {
        if (alpha_sq_norm == 0.0) {
            for (int i = 0; i < n_rows * n_cols; i++) jacobian[i] *= sqrt_rho1;
        } else {
            for (int c = 0; c < n_cols; c++) {
                double r_transpose_j = 0.0;
                for (int r = 0; r < n_rows; r++) r_transpose_j += jacobian[r * n_cols + c] * residuals[r];
                for (int r = 0; r < n_rows; r++)
                    jacobian[r * n_cols + c] = sqrt_rho1 * (jacobian[r * n_cols + c] - alpha_sq_norm * residuals[r] * r_transpose_j);
            }
        }
    }
    for (int r = 0; r < n_rows; r++) residuals[r] *= residual_scaling;
}

void orc_lm_step_accepted(double step_quality, double *radius, double *decrease_factor, double max_radius)
{
    double d = 1.0 - pow(2.0 * step_quality - 1.0, 3);
    if (d < 1.0 / 3.0) d = 1.0 / 3.0;
    *radius = *radius / d;
    if (*radius > max_radius) *radius = max_radius;
    *decrease_factor = 2.0;
}

void orc_lm_step_rejected(double *radius, double *decrease_factor)
{
    *radius = *radius / *decrease_factor;
    *decrease_factor *= 2.0;
}

/* LevenbergMarquardtStrategy::ComputeStep (levenberg_marquardt_strategy.cc:76-88): the regulariser handed to the linear
 * solver is  D_i = sqrt( clamp(diag(J^T J)_i, min_diagonal, max_diagonal) / radius ).  jtj_diag is clamped IN PLACE (Ceres
 * keeps the clamped diagonal_ and re-uses it while reuse_diagonal is set).  Pinned by the vendored unit test
 * LevenbergMarquardtStrategy.CorrectDiagonalToLinearSolver (levenberg_marquardt_strategy_test.cc:112-135).          */
void orc_lm_diagonal(double *jtj_diag, int n, double radius, double min_diagonal, double max_diagonal, int clamp, double *D_out)
{
    for (int i = 0; i < n; i++) {
        if (clamp) jtj_diag[i] = fmin(fmax(jtj_diag[i], min_diagonal), max_diagonal);
        D_out[i] = sqrt(jtj_diag[i] / radius);
    }
}

/* Generic dense restatement of SchurEliminator::Eliminate + BackSubstitute (schur_eliminator_impl.h:179-377) for scalar
 * e-blocks, used to pin the elimination algebra against Ceres' own fixed problems (schur_eliminator_test.cc:82-225 on
 * linear_least_squares_problems.cc problem 2) and to cross-check ba_schur_solve on small BA problems.
 *   J: m x n row-major, columns [0, n_e) are the eliminated (e) columns, every row touches at most ONE e column
 *   (row_e[r] = that column or -1: a row without e-block, NoEBlockRowsUpdate), D: n regulariser entries (or NULL = 0).
 *   lhs (s x s, s = n - n_e, full symmetric), rhs (s):  S = F^T F + D_f^2 - sum_e (E^T F)^T (E^T E + D_e^2)^-1 (E^T F)
 *   sol (n): solution of (J^T J + D^2) x = J^T b through the reduced system + back substitution.  Returns 0 / -1.    */
static int chol_lower(double *A, int n);
static void chol_solve(const double *L, int n, double *b);
int orc_schur_eliminate_dense(const double *J, const double *b, const double *D, int m, int n, int n_e, const int *row_e,
                              double *lhs, double *rhs, double *sol)
{
    const int s = n - n_e;
    if (m <= 0 || n <= 0 || n_e < 0 || s < 0) return -1;
    memset(lhs, 0, sizeof(double) * (size_t)s * s);
    memset(rhs, 0, sizeof(double) * (size_t)s);
    for (int c = 0; c < s; c++) lhs[(size_t)c * s + c] = D ? D[n_e + c] * D[n_e + c] : 0.0;
    double *buf = (double *)malloc(sizeof(double) * (size_t)(s + 1));
    double *ete_inv = (double *)calloc((size_t)n_e + 1, sizeof(double)), *g_e = (double *)calloc((size_t)n_e + 1, sizeof(double));
    for (int e = 0; e < n_e; e++) {                              /* one chunk per e-block */
        double ete = D ? D[e] * D[e] : 0.0, g = 0;
        memset(buf, 0, sizeof(double) * (size_t)s);
        for (int r = 0; r < m; r++) {
            if (row_e[r] != e) continue;
            const double *row = J + (size_t)r * n;
            ete += row[e] * row[e]; g += row[e] * b[r];
            for (int c = 0; c < s; c++) {
                buf[c] += row[e] * row[n_e + c];                 /* E^T F */
                rhs[c] += row[n_e + c] * b[r];                   /* F^T b */
                for (int d = 0; d < s; d++) lhs[(size_t)c * s + d] += row[n_e + c] * row[n_e + d];
            }
        }
        if (!(ete > 0)) { free(buf); free(ete_inv); free(g_e); return -1; }
        ete_inv[e] = 1.0 / ete; g_e[e] = g;
        for (int c = 0; c < s; c++) {
            rhs[c] -= buf[c] * ete_inv[e] * g;
            for (int d = 0; d < s; d++) lhs[(size_t)c * s + d] -= buf[c] * ete_inv[e] * buf[d];
        }
    }
    for (int r = 0; r < m; r++) {                                /* rows without an e-block */
        if (row_e[r] >= 0) continue;
        const double *row = J + (size_t)r * n;
        for (int c = 0; c < s; c++) {
            rhs[c] += row[n_e + c] * b[r];
            for (int d = 0; d < s; d++) lhs[(size_t)c * s + d] += row[n_e + c] * row[n_e + d];
        }
    }
    int rc = 0;
    if (sol) {
        double *L = (double *)malloc(sizeof(double) * (size_t)(s * s + 1));
        memcpy(L, lhs, sizeof(double) * (size_t)s * s);
        rc = s > 0 ? chol_lower(L, s) : 0;
        if (rc == 0) {
            memcpy(sol + n_e, rhs, sizeof(double) * (size_t)s);
            if (s > 0) chol_solve(L, s, sol + n_e);
            for (int e = 0; e < n_e; e++) {                      /* y_e = (E^T b - E^T F y_f) / (E^T E + D_e^2) */
                double acc = g_e[e];
                for (int r = 0; r < m; r++) {
                    if (row_e[r] != e) continue;
                    const double *row = J + (size_t)r * n;
                    for (int c = 0; c < s; c++) acc -= row[e] * row[n_e + c] * sol[n_e + c];
                }
                sol[e] = acc * ete_inv[e];
            }
        }
        free(L);
    }
    free(buf); free(ete_inv); free(g_e);
    return rc;
}

void orc_ba_default_options(orc_ba_options *o)
{
    o->max_iter = 5; o->function_tolerance = 1e-3; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->huber_delta = sqrt(5.9915); o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32;
    o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32; o->min_relative_decrease = 1e-3; o->jacobi_scaling = 1;
    o->max_consecutive_invalid_steps = 5;
}

/* ---------------- one residual block ---------------- */
int orc_ba_residual(int type, const double calib_l[4], const double calib_r[4], const double T_rl[7],
                    const double anchor_pose[7], const double obs_pose[7], double invdepth,
                    const double anchor_uv[2], const double uv[2], double sigma,
                    double r[2], double Ja[12], double Jo[12], double Jl[2], double *chi2)
{
    const double sqrt_info = 1.0 / sigma;
    const double zanch = 1.0 / invdepth;
    /* anchpt = zanch * K^-1 * (u, v, 1) */
    const double fx = calib_l[0], fy = calib_l[1], cx = calib_l[2], cy = calib_l[3];
    double anchpt[3] = {zanch * ((anchor_uv[0] - cx) / fx), zanch * ((anchor_uv[1] - cy) / fy), zanch};
    double qa[4] = {anchor_pose[3], anchor_pose[4], anchor_pose[5], anchor_pose[6]};
    quat_normalize(qa);
    double Rwa[9];
    quat_to_R(qa, Rwa);
    double Rrl[9], qrl[4] = {T_rl[3], T_rl[4], T_rl[5], T_rl[6]};
    quat_normalize(qrl);
    quat_to_R(qrl, Rrl);

    double campt[3];      /* point in the projecting camera */
    double JR[6];         /* J_cam * (rotation chain), 2x3 row-major */
    double wpt[3] = {0, 0, 0}, Rcw[9];
    const double *K = (type == ORC_RES_LEFT) ? calib_l : calib_r;
    if (type == ORC_RES_RIGHT_ANCH) {
        mat3_vec(Rrl, anchpt, campt);
        campt[0] += T_rl[0]; campt[1] += T_rl[1]; campt[2] += T_rl[2];
    } else {
        double tmp[3];
        mat3_vec(Rwa, anchpt, tmp);
        wpt[0] = tmp[0] + anchor_pose[0]; wpt[1] = tmp[1] + anchor_pose[1]; wpt[2] = tmp[2] + anchor_pose[2];
        double qo[4] = {obs_pose[3], obs_pose[4], obs_pose[5], obs_pose[6]};
        quat_normalize(qo);
        double Rwc[9];
        quat_to_R(qo, Rwc);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rcw[3 * i + j] = Rwc[3 * j + i];
        double d[3] = {wpt[0] - obs_pose[0], wpt[1] - obs_pose[1], wpt[2] - obs_pose[2]};
        double lcam[3];
        mat3_vec(Rcw, d, lcam);
        if (type == ORC_RES_LEFT) { campt[0] = lcam[0]; campt[1] = lcam[1]; campt[2] = lcam[2]; }
        else {
            mat3_vec(Rrl, lcam, campt);
            campt[0] += T_rl[0]; campt[1] += T_rl[1]; campt[2] += T_rl[2];
        }
    }
    const double invz = 1.0 / campt[2];
    const double pu = K[0] * campt[0] * invz + K[2], pv = K[1] * campt[1] * invz + K[3];
    r[0] = sqrt_info * (pu - uv[0]);
    r[1] = sqrt_info * (pv - uv[1]);
    *chi2 = r[0] * r[0] + r[1] * r[1];
    const int depthpos = campt[2] > 0;
    if (!Ja && !Jo && !Jl) return depthpos;

    const double invz2 = invz * invz;
    const double Jc[6] = {invz * K[0], 0, -campt[0] * invz2 * K[0], 0, invz * K[1], -campt[1] * invz2 * K[1]};
    /* rotation chain */
    double M[9];
    if (type == ORC_RES_LEFT) memcpy(M, Rcw, sizeof(M));
    else if (type == ORC_RES_RIGHT_ANCH) memcpy(M, Rrl, sizeof(M));
    else
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
            M[3 * i + j] = Rrl[3 * i] * Rcw[j] + Rrl[3 * i + 1] * Rcw[3 + j] + Rrl[3 * i + 2] * Rcw[6 + j];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++)
        JR[3 * i + j] = Jc[3 * i] * M[j] + Jc[3 * i + 1] * M[3 + j] + Jc[3 * i + 2] * M[6 + j];

    if (type == ORC_RES_RIGHT_ANCH) {
        if (Ja) memset(Ja, 0, sizeof(double) * 12);
        if (Jo) memset(Jo, 0, sizeof(double) * 12);
        if (Jl) {
            const double jl[3] = {-zanch * anchpt[0], -zanch * anchpt[1], -zanch * anchpt[2]};
            Jl[0] = sqrt_info * (JR[0] * jl[0] + JR[1] * jl[1] + JR[2] * jl[2]);
            Jl[1] = sqrt_info * (JR[3] * jl[0] + JR[4] * jl[1] + JR[5] * jl[2]);
        }
        return depthpos;
    }
    /* skew(wpt) */
    const double S[9] = {0, -wpt[2], wpt[1], wpt[2], 0, -wpt[0], -wpt[1], wpt[0], 0};
    double JRS[6];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++)
        JRS[3 * i + j] = JR[3 * i] * S[j] + JR[3 * i + 1] * S[3 + j] + JR[3 * i + 2] * S[6 + j];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) {
            if (Ja) { Ja[6 * i + j] = sqrt_info * JR[3 * i + j]; Ja[6 * i + 3 + j] = sqrt_info * (-JRS[3 * i + j]); }
            if (Jo) { Jo[6 * i + j] = sqrt_info * (-JR[3 * i + j]); Jo[6 * i + 3 + j] = sqrt_info * JRS[3 * i + j]; }
        }
    if (Jl) {
        double Ra[3];
        mat3_vec(Rwa, anchpt, Ra);
        const double jl[3] = {-zanch * Ra[0], -zanch * Ra[1], -zanch * Ra[2]};
        Jl[0] = sqrt_info * (JR[0] * jl[0] + JR[1] * jl[1] + JR[2] * jl[2]);
        Jl[1] = sqrt_info * (JR[3] * jl[0] + JR[4] * jl[1] + JR[5] * jl[2]);
    }
    (void)mat3T_vec;
    return depthpos;
}

/* DirectLeftSE3::ReprojectionErrorSE3::Evaluate (src/ceres_parametrization.cpp:301-358): fixed world point */
static int pnp_residual(const double calib[4], const double pose[7], const double xyz[3], const double uv[2], double sigma,
                        double r[2], double Jo[12], double *chi2)
{
    const double sqrt_info = 1.0 / sigma;
    double q[4] = {pose[3], pose[4], pose[5], pose[6]};
    quat_normalize(q);
    double Rwc[9], Rcw[9];
    quat_to_R(q, Rwc);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rcw[3 * i + j] = Rwc[3 * j + i];
    const double d[3] = {xyz[0] - pose[0], xyz[1] - pose[1], xyz[2] - pose[2]};
    double c[3];
    mat3_vec(Rcw, d, c);
    const double invz = 1.0 / c[2];
    r[0] = sqrt_info * (calib[0] * c[0] * invz + calib[2] - uv[0]);
    r[1] = sqrt_info * (calib[1] * c[1] * invz + calib[3] - uv[1]);
    *chi2 = r[0] * r[0] + r[1] * r[1];
    if (Jo) {
        const double invz2 = invz * invz;
        const double Jc[6] = {invz * calib[0], 0, -c[0] * invz2 * calib[0], 0, invz * calib[1], -c[1] * invz2 * calib[1]};
        double JR[6];
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) JR[3 * i + j] = Jc[3 * i] * Rcw[j] + Jc[3 * i + 1] * Rcw[3 + j] + Jc[3 * i + 2] * Rcw[6 + j];
        const double S[9] = {0, -xyz[2], xyz[1], xyz[2], 0, -xyz[0], -xyz[1], xyz[0], 0};
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++) {
                const double jrs = JR[3 * i] * S[j] + JR[3 * i + 1] * S[3 + j] + JR[3 * i + 2] * S[6 + j];
                Jo[6 * i + j] = -sqrt_info * JR[3 * i + j];
                Jo[6 * i + 3 + j] = sqrt_info * jrs;
            }
    }
    return c[2] > 0;
}

/* ---------------- solver ---------------- */
typedef struct {
    const orc_ba_problem *p;
    const orc_ba_options *o;
    int n_act;            /* active residuals */
    int *act;             /* indices of active residuals */
    int *pose_col;        /* n_kf: first column (6 per pose) in the F part, -1 if constant */
    int n_opt;            /* number of variable poses */
    int nf;               /* 6 * n_opt */
    int *lm_ptr, *lm_idx; /* CSR: landmark -> positions in act[] */
    /* linearisation at x (scaled, corrected) */
    double *r, *Ja, *Jo, *Jl;   /* per active residual: 2, 12, 12, 2 */
    double *scale_f, *scale_l;  /* jacobi scaling */
} ba_ws;

static double ba_evaluate(ba_ws *w, const double *poses, const double *lam, int want_jac,
                          double *chi2_out, uint8_t *dpos_out, double *grad_f, double *grad_l)
{
    const orc_ba_problem *p = w->p;
    double cost = 0;
    if (grad_f) memset(grad_f, 0, sizeof(double) * (size_t)w->nf);
    if (grad_l) memset(grad_l, 0, sizeof(double) * (size_t)p->n_lm);
    for (int k = 0; k < w->n_act; k++) {
        const int i = w->act[k];
        const int type = p->res_type[i];
        const int lm = type == ORC_RES_PNP ? -1 : p->res_lm[i];
        const int a = lm >= 0 ? p->lm_anchor_kf[lm] : p->res_kf[i];
        const int o = type == ORC_RES_RIGHT_ANCH ? a : p->res_kf[i];
        double r[2], Ja[12], Jo[12], Jl[2], chi2;
        int dp;
        if (type == ORC_RES_PNP) {
            dp = pnp_residual(p->calib_l, poses + 7 * o, p->res_xyz + 3 * i, p->res_uv + 2 * i, p->res_sigma[i], r, want_jac ? Jo : NULL, &chi2);
            memset(Ja, 0, sizeof(Ja)); Jl[0] = Jl[1] = 0;
        } else
            dp = orc_ba_residual(type, p->calib_l, p->calib_r, p->T_rl, poses + 7 * a, poses + 7 * o, lam[lm],
                                 p->lm_anchor_uv + 2 * lm, p->res_uv + 2 * i, p->res_sigma[i],
                                 r, want_jac ? Ja : NULL, want_jac ? Jo : NULL, want_jac ? Jl : NULL, &chi2);
        if (chi2_out) chi2_out[i] = chi2;
        if (dpos_out) dpos_out[i] = (uint8_t)dp;
        const double s = r[0] * r[0] + r[1] * r[1];
        double rho[3];
        if (w->o->huber_delta > 0) orc_huber(w->o->huber_delta, s, rho);
        else { rho[0] = s; rho[1] = 1; rho[2] = 0; }
        cost += 0.5 * rho[0];
        if (!want_jac) continue;
        /* corrector: jacobians first (with the un-corrected residual), then the residual */
        double rr[2] = {r[0], r[1]};
        orc_corrector(s, rho, 2, 6, rr, Ja); rr[0] = r[0]; rr[1] = r[1];
        orc_corrector(s, rho, 2, 6, rr, Jo); rr[0] = r[0]; rr[1] = r[1];
        orc_corrector(s, rho, 2, 1, rr, Jl);
        /* an anchor==obs residual (possible only through bad input) would alias; the reference never builds one */
        const int ca = w->pose_col[a], co = w->pose_col[o];
        if (type == ORC_RES_RIGHT_ANCH || type == ORC_RES_PNP || ca < 0) memset(Ja, 0, sizeof(Ja));
        if (type == ORC_RES_RIGHT_ANCH || co < 0) memset(Jo, 0, sizeof(Jo));
        /* gradient with the un-scaled jacobian (evaluator), then jacobi scaling */
        if (grad_f) {
            for (int c = 0; c < 6; c++) {
                if (ca >= 0) grad_f[ca + c] += Ja[c] * rr[0] + Ja[6 + c] * rr[1];
                if (co >= 0) grad_f[co + c] += Jo[c] * rr[0] + Jo[6 + c] * rr[1];
            }
        }
        if (grad_l && lm >= 0) grad_l[lm] += Jl[0] * rr[0] + Jl[1] * rr[1];
        memcpy(w->r + 2 * k, rr, sizeof(rr));
        memcpy(w->Ja + 12 * k, Ja, sizeof(Ja));
        memcpy(w->Jo + 12 * k, Jo, sizeof(Jo));
        memcpy(w->Jl + 2 * k, Jl, sizeof(Jl));
    }
    return cost;
}

/* landmark (-1 for pose-only blocks), anchor / observer keyframes and their first columns (-1 = constant / absent) */
static void ba_res_blocks(const ba_ws *w, int i, int *lm, int *ca, int *co)
{
    const orc_ba_problem *p = w->p;
    const int type = p->res_type[i];
    if (type == ORC_RES_PNP) { *lm = -1; *ca = -1; *co = w->pose_col[p->res_kf[i]]; return; }
    *lm = p->res_lm[i];
    const int a = p->lm_anchor_kf[*lm];
    if (type == ORC_RES_RIGHT_ANCH) { *ca = -1; *co = -1; return; }
    *ca = w->pose_col[a]; *co = w->pose_col[p->res_kf[i]];
}

static void ba_col_sqnorm(const ba_ws *w, double *nf, double *nl)
{
    const orc_ba_problem *p = w->p;
    memset(nf, 0, sizeof(double) * (size_t)w->nf);
    memset(nl, 0, sizeof(double) * (size_t)p->n_lm);
    for (int k = 0; k < w->n_act; k++) {
        int lm, ca, co;
        ba_res_blocks(w, w->act[k], &lm, &ca, &co);
        const double *Ja = w->Ja + 12 * k, *Jo = w->Jo + 12 * k, *Jl = w->Jl + 2 * k;
        for (int c = 0; c < 6; c++) {
            if (ca >= 0) nf[ca + c] += Ja[c] * Ja[c] + Ja[6 + c] * Ja[6 + c];
            if (co >= 0) nf[co + c] += Jo[c] * Jo[c] + Jo[6 + c] * Jo[6 + c];
        }
        if (lm >= 0) nl[lm] += Jl[0] * Jl[0] + Jl[1] * Jl[1];
    }
}

static void ba_scale_columns(ba_ws *w)
{
    const orc_ba_problem *p = w->p;
    for (int k = 0; k < w->n_act; k++) {
        int lm, ca, co;
        ba_res_blocks(w, w->act[k], &lm, &ca, &co);
        double *Ja = w->Ja + 12 * k, *Jo = w->Jo + 12 * k, *Jl = w->Jl + 2 * k;
        for (int c = 0; c < 6; c++) {
            if (ca >= 0) { Ja[c] *= w->scale_f[ca + c]; Ja[6 + c] *= w->scale_f[ca + c]; }
            if (co >= 0) { Jo[c] *= w->scale_f[co + c]; Jo[6 + c] *= w->scale_f[co + c]; }
        }
        if (lm >= 0) { Jl[0] *= w->scale_l[lm]; Jl[1] *= w->scale_l[lm]; }
    }
    (void)p;
}

/* dense Cholesky (lower) in place; returns 0 on success */
static int chol_lower(double *A, int n)
{
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0) || !isfinite(d)) return -1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return 0;
}

static void chol_solve(const double *L, int n, double *b)
{
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i * n + k] * b[k];
        b[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= L[k * n + i] * b[k];
        b[i] = s / L[i * n + i];
    }
}

/* Schur-complement solve of  min |J y - r|^2 + |D y|^2 ; y_f, y_l out.  returns 0 / -1 (Cholesky failed) */
static int ba_schur_solve(const ba_ws *w, const double *Df, const double *Dl, double *yf, double *yl)
{
    const orc_ba_problem *p = w->p;
    const int nf = w->nf;
    double *S = (double *)calloc((size_t)nf * nf + 1, sizeof(double));
    double *rhs = (double *)calloc((size_t)nf + 1, sizeof(double));
    double *wrow = (double *)malloc(sizeof(double) * (size_t)(nf + 1));
    int *touched = (int *)malloc(sizeof(int) * (size_t)(w->n_opt + 1));
    uint8_t *flag = (uint8_t *)calloc((size_t)w->n_opt + 1, 1);
    double *ete_inv = (double *)malloc(sizeof(double) * (size_t)p->n_lm);
    double *etb = (double *)malloc(sizeof(double) * (size_t)p->n_lm);
    for (int c = 0; c < nf; c++) S[(size_t)c * nf + c] = Df[c] * Df[c];
    for (int lm = 0; lm < p->n_lm; lm++) {
        ete_inv[lm] = 0; etb[lm] = 0;
        if (w->lm_ptr[lm] == w->lm_ptr[lm + 1]) continue;
        double ete = Dl[lm] * Dl[lm], g = 0;
        int nt = 0;
        for (int q = w->lm_ptr[lm]; q < w->lm_ptr[lm + 1]; q++) {
            const int k = w->lm_idx[q], i = w->act[k];
            const int a = p->lm_anchor_kf[lm];
            const int o = p->res_type[i] == ORC_RES_RIGHT_ANCH ? a : p->res_kf[i];
            const int ca = w->pose_col[a], co = w->pose_col[o];
            const double *Ja = w->Ja + 12 * k, *Jo = w->Jo + 12 * k, *Jl = w->Jl + 2 * k, *r = w->r + 2 * k;
            ete += Jl[0] * Jl[0] + Jl[1] * Jl[1];
            g += Jl[0] * r[0] + Jl[1] * r[1];
            const int is_pose_res = p->res_type[i] != ORC_RES_RIGHT_ANCH;
            if (is_pose_res && ca >= 0 && !flag[ca / 6]) { flag[ca / 6] = 1; touched[nt++] = ca; for (int c = 0; c < 6; c++) wrow[ca + c] = 0; }
            if (is_pose_res && co >= 0 && !flag[co / 6]) { flag[co / 6] = 1; touched[nt++] = co; for (int c = 0; c < 6; c++) wrow[co + c] = 0; }
            if (!is_pose_res) continue;
            /* F^T F and F^T b */
            for (int c = 0; c < 6; c++) {
                if (ca >= 0) {
                    wrow[ca + c] += Jl[0] * Ja[c] + Jl[1] * Ja[6 + c];
                    rhs[ca + c] += Ja[c] * r[0] + Ja[6 + c] * r[1];
                    for (int d = 0; d < 6; d++) {
                        S[(size_t)(ca + c) * nf + ca + d] += Ja[c] * Ja[d] + Ja[6 + c] * Ja[6 + d];
                        if (co >= 0) {
                            const double v = Ja[c] * Jo[d] + Ja[6 + c] * Jo[6 + d];
                            S[(size_t)(ca + c) * nf + co + d] += v;
                            S[(size_t)(co + d) * nf + ca + c] += v;
                        }
                    }
                }
                if (co >= 0) {
                    wrow[co + c] += Jl[0] * Jo[c] + Jl[1] * Jo[6 + c];
                    rhs[co + c] += Jo[c] * r[0] + Jo[6 + c] * r[1];
                    for (int d = 0; d < 6; d++) S[(size_t)(co + c) * nf + co + d] += Jo[c] * Jo[d] + Jo[6 + c] * Jo[6 + d];
                }
            }
        }
        const double inv = 1.0 / ete;
        ete_inv[lm] = inv; etb[lm] = g;
        /* S -= w^T inv w ; rhs -= w^T inv g */
#ifdef ORC_FAST
        /* speed baseline only (Makefile `fast`): the block rows are gathered contiguously and only the blocks on or below the
         * diagonal -- all chol_lower reads -- are updated: half the flops, unit-stride inner loops the compiler vectorises */
        {
            double wc6[6 * 64], wv[6 * 64];
            int ord[64];
            const int ntc = nt > 64 ? 64 : nt;
            for (int x = 0; x < ntc; x++) ord[x] = touched[x];
            for (int x = 1; x < ntc; x++) { int v = ord[x], y = x; while (y > 0 && ord[y - 1] > v) { ord[y] = ord[y - 1]; y--; } ord[y] = v; }
            for (int x = 0; x < ntc; x++)
                for (int c = 0; c < 6; c++) { wv[6 * x + c] = wrow[ord[x] + c]; wc6[6 * x + c] = wv[6 * x + c] * inv; rhs[ord[x] + c] -= wc6[6 * x + c] * g; }
            for (int x = 0; x < ntc; x++)
                for (int c = 0; c < 6; c++) {
                    const double wc = wc6[6 * x + c];
                    double *Srow = S + (size_t)(ord[x] + c) * nf;
                    for (int y = 0; y <= x; y++) {
                        double *dst = Srow + ord[y];
                        const double *wy = wv + 6 * y;
                        for (int d = 0; d < 6; d++) dst[d] -= wc * wy[d];
                    }
                }
            if (nt > 64) return -1;
        }
#else
        for (int x = 0; x < nt; x++)
            for (int c = 0; c < 6; c++) {
                const double wc = wrow[touched[x] + c] * inv;
                rhs[touched[x] + c] -= wc * g;
                for (int y = 0; y < nt; y++)
                    for (int d = 0; d < 6; d++)
                        S[(size_t)(touched[x] + c) * nf + touched[y] + d] -= wc * wrow[touched[y] + d];
            }
#endif
        for (int x = 0; x < nt; x++) flag[touched[x] / 6] = 0;
    }
    /* rows without an e-block (SchurEliminator::NoEBlockRowsUpdate): pose-only residual blocks */
    for (int k = 0; k < w->n_act; k++) {
        const int i = w->act[k];
        if (p->res_type[i] != ORC_RES_PNP) continue;
        const int co = w->pose_col[p->res_kf[i]];
        if (co < 0) continue;
        const double *Jo = w->Jo + 12 * k, *r = w->r + 2 * k;
        for (int c = 0; c < 6; c++) {
            rhs[co + c] += Jo[c] * r[0] + Jo[6 + c] * r[1];
            for (int d = 0; d < 6; d++) S[(size_t)(co + c) * nf + co + d] += Jo[c] * Jo[d] + Jo[6 + c] * Jo[6 + d];
        }
    }
    int rc = 0;
    if (nf > 0) {
        rc = chol_lower(S, nf);
        if (rc == 0) { memcpy(yf, rhs, sizeof(double) * (size_t)nf); chol_solve(S, nf, yf); }
    }
    if (rc == 0) {
        /* back substitution: y_l = (E^T b - E^T F y_f) / (E^T E + D^2) */
        for (int lm = 0; lm < p->n_lm; lm++) {
            yl[lm] = 0;
            if (w->lm_ptr[lm] == w->lm_ptr[lm + 1]) continue;
            double acc = etb[lm];
            for (int q = w->lm_ptr[lm]; q < w->lm_ptr[lm + 1]; q++) {
                const int k = w->lm_idx[q], i = w->act[k];
                if (p->res_type[i] == ORC_RES_RIGHT_ANCH) continue;
                const int a = p->lm_anchor_kf[lm], o = p->res_kf[i];
                const int ca = w->pose_col[a], co = w->pose_col[o];
                const double *Ja = w->Ja + 12 * k, *Jo = w->Jo + 12 * k, *Jl = w->Jl + 2 * k;
                for (int c = 0; c < 6; c++) {
                    if (ca >= 0) acc -= (Jl[0] * Ja[c] + Jl[1] * Ja[6 + c]) * yf[ca + c];
                    if (co >= 0) acc -= (Jl[0] * Jo[c] + Jl[1] * Jo[6 + c]) * yf[co + c];
                }
            }
            yl[lm] = acc * ete_inv[lm];
        }
    }
    free(S); free(rhs); free(wrow); free(touched); free(flag); free(ete_inv); free(etb);
    return rc;
}

static double vec_norm_diff(const double *a, const double *b, int n)
{
    double s = 0;
    for (int i = 0; i < n; i++) s += (a[i] - b[i]) * (a[i] - b[i]);
    return s;
}

/* per-iteration trace (tests only): what Ceres records in Solver::Summary::iterations (trust_region_minimizer.cc:313-337) */
static __thread orc_ba_iter *g_trace = NULL;
static __thread int g_trace_cap = 0;
static __thread int *g_trace_n = NULL;
void orc_ba_set_trace(orc_ba_iter *buf, int cap, int *n) { g_trace = buf; g_trace_cap = cap; g_trace_n = n; if (n) *n = 0; }

int orc_ba_solve(const orc_ba_problem *p, const orc_ba_options *o, orc_ba_result *res)
{
    if (!p || !o || !res || p->n_kf <= 0 || p->n_lm < 0 || p->n_res < 0) return -1;
    ba_ws w; memset(&w, 0, sizeof(w));
    w.p = p; w.o = o;
    w.pose_col = (int *)malloc(sizeof(int) * (size_t)p->n_kf);
    for (int k = 0; k < p->n_kf; k++) {
        if (p->kf_const[k]) w.pose_col[k] = -1;
        else { w.pose_col[k] = 6 * w.n_opt; w.n_opt++; }
    }
    w.nf = 6 * w.n_opt;
    w.act = (int *)malloc(sizeof(int) * (size_t)(p->n_res + 1));
    for (int i = 0; i < p->n_res; i++) {
        if (p->res_active && !p->res_active[i]) continue;
        if (p->res_type[i] == ORC_RES_PNP) {
            if (!p->res_xyz || p->res_kf[i] < 0 || p->res_kf[i] >= p->n_kf) { free(w.pose_col); free(w.act); return -1; }
        } else {
            const int lm = p->res_lm[i];
            if (lm < 0 || lm >= p->n_lm || p->lm_anchor_kf[lm] < 0 || p->lm_anchor_kf[lm] >= p->n_kf) { free(w.pose_col); free(w.act); return -1; }
            if (p->res_type[i] != ORC_RES_RIGHT_ANCH && (p->res_kf[i] < 0 || p->res_kf[i] >= p->n_kf)) { free(w.pose_col); free(w.act); return -1; }
        }
        w.act[w.n_act++] = i;
    }
    /* landmark -> residual CSR */
    w.lm_ptr = (int *)calloc((size_t)p->n_lm + 2, sizeof(int));
    w.lm_idx = (int *)malloc(sizeof(int) * (size_t)(w.n_act + 1));
    for (int k = 0; k < w.n_act; k++) if (p->res_type[w.act[k]] != ORC_RES_PNP) w.lm_ptr[p->res_lm[w.act[k]] + 1]++;
    for (int l = 0; l < p->n_lm; l++) w.lm_ptr[l + 1] += w.lm_ptr[l];
    {
        int *fill = (int *)malloc(sizeof(int) * (size_t)(p->n_lm + 1));
        memcpy(fill, w.lm_ptr, sizeof(int) * (size_t)(p->n_lm + 1));
        for (int k = 0; k < w.n_act; k++) if (p->res_type[w.act[k]] != ORC_RES_PNP) w.lm_idx[fill[p->res_lm[w.act[k]]]++] = k;
        free(fill);
    }
    const size_t na = (size_t)w.n_act + 1;
    w.r = (double *)malloc(sizeof(double) * 2 * na);
    w.Ja = (double *)malloc(sizeof(double) * 12 * na);
    w.Jo = (double *)malloc(sizeof(double) * 12 * na);
    w.Jl = (double *)malloc(sizeof(double) * 2 * na);
    w.scale_f = (double *)malloc(sizeof(double) * (size_t)(w.nf + 1));
    w.scale_l = (double *)malloc(sizeof(double) * (size_t)(p->n_lm + 1));
    for (int c = 0; c < w.nf; c++) w.scale_f[c] = 1.0;
    for (int l = 0; l < p->n_lm; l++) w.scale_l[l] = 1.0;

    const int NP = 7 * p->n_kf;
    double *x_pose = (double *)malloc(sizeof(double) * (size_t)NP), *x_lam = (double *)malloc(sizeof(double) * (size_t)(p->n_lm + 1));
    double *c_pose = (double *)malloc(sizeof(double) * (size_t)NP), *c_lam = (double *)malloc(sizeof(double) * (size_t)(p->n_lm + 1));
    double *gf = (double *)malloc(sizeof(double) * (size_t)(w.nf + 1)), *gl = (double *)malloc(sizeof(double) * (size_t)(p->n_lm + 1));
    double *diag_f = (double *)malloc(sizeof(double) * (size_t)(w.nf + 1)), *diag_l = (double *)malloc(sizeof(double) * (size_t)(p->n_lm + 1));
    double *Df = (double *)malloc(sizeof(double) * (size_t)(w.nf + 1)), *Dl = (double *)malloc(sizeof(double) * (size_t)(p->n_lm + 1));
    double *yf = (double *)malloc(sizeof(double) * (size_t)(w.nf + 1)), *yl = (double *)malloc(sizeof(double) * (size_t)(p->n_lm + 1));
    memcpy(x_pose, p->poses, sizeof(double) * (size_t)NP);
    memcpy(x_lam, p->invdepth, sizeof(double) * (size_t)p->n_lm);

    /* which landmarks are part of the (reduced) program */
    /* iteration 0 */
    double x_cost = ba_evaluate(&w, x_pose, x_lam, 1, res->chi2_last_eval, res->depthpos_last_eval, gf, gl);
    if (o->jacobi_scaling) {
        ba_col_sqnorm(&w, diag_f, diag_l);
        for (int c = 0; c < w.nf; c++) w.scale_f[c] = 1.0 / (1.0 + sqrt(diag_f[c]));
        for (int l = 0; l < p->n_lm; l++) w.scale_l[l] = 1.0 / (1.0 + sqrt(diag_l[l]));
        ba_scale_columns(&w);
    }
    res->initial_cost = x_cost;
    double minimum_cost = x_cost;
    double x_norm = -1.0;                 /* "Invalid value" until the first successful step */
    double radius = o->initial_radius, decrease_factor = 2.0;
    int reuse_diagonal = 0;
    int num_invalid = 0;
    /* TrustRegionStepEvaluator with max_consecutive_nonmonotonic_steps = 0 */
    double ev_min = x_cost, ev_cur = x_cost, ev_ref = x_cost, ev_cand = x_cost, ev_acc_ref = 0, ev_acc_cand = 0;
    int ev_nonmono = 0;

    int iteration = 0;                   /* index of the last finalized iteration */
    int step_successful = 1;             /* iteration 0 counts as successful */
    int term = ORC_TERM_NO_CONVERGENCE;
    int n_success = 0, n_steps = 0;
    double gmax = 0, gnorm = 0;
    /* the iteration summary being filled (IterationZero, trust_region_minimizer.cc:195-231) */
    orc_ba_iter cur; memset(&cur, 0, sizeof(cur));
    /* gradient max norm at x: |x - Plus(x, -g)|_inf */
    #define GRAD_MAX_NORM()                                                                         \
        do {                                                                                       \
            gmax = 0; gnorm = 0;                                                                   \
            for (int k_ = 0; k_ < p->n_kf; k_++) {                                                \
                if (w.pose_col[k_] < 0) continue;                                                  \
                double d_[6], out_[7];                                                             \
                for (int c_ = 0; c_ < 6; c_++) d_[c_] = -gf[w.pose_col[k_] + c_];                  \
                orc_se3_left_plus(x_pose + 7 * k_, d_, out_);                                      \
                for (int c_ = 0; c_ < 7; c_++) { double v_ = fabs(x_pose[7 * k_ + c_] - out_[c_]); if (v_ > gmax) gmax = v_; gnorm += v_ * v_; } \
            }                                                                                      \
            for (int l_ = 0; l_ < p->n_lm; l_++) { if (w.lm_ptr[l_] == w.lm_ptr[l_ + 1]) continue; double v_ = fabs(gl[l_]); if (v_ > gmax) gmax = v_; gnorm += v_ * v_; } \
            gnorm = sqrt(gnorm);                                                                   \
        } while (0)
    GRAD_MAX_NORM();
    cur.iteration = 0; cur.step_is_valid = 1; cur.step_is_successful = 1; cur.cost = x_cost; cur.gradient_max_norm = gmax; cur.gradient_norm = gnorm;

    for (;;) {
        /* FinalizeIterationAndCheckIfMinimizerCanContinue */
        if (step_successful) {
            n_success++;
            if (x_cost < minimum_cost) minimum_cost = x_cost;
        }
        cur.trust_region_radius = radius;
        if (g_trace_n) { if (g_trace && *g_trace_n < g_trace_cap) g_trace[*g_trace_n] = cur; (*g_trace_n)++; }
        if (iteration >= o->max_iter) { term = ORC_TERM_NO_CONVERGENCE; break; }
        if (step_successful && gmax <= o->gradient_tolerance) { term = ORC_TERM_GRADIENT_TOL; break; }
        if (radius <= o->min_radius) { term = ORC_TERM_MIN_RADIUS; break; }
        iteration++;
        step_successful = 0;
        { const double pg = cur.gradient_norm, pgm = cur.gradient_max_norm; memset(&cur, 0, sizeof(cur)); cur.iteration = iteration; cur.gradient_norm = pg; cur.gradient_max_norm = pgm; }

        /* ComputeTrustRegionStep (LevenbergMarquardtStrategy::ComputeStep) */
        if (!reuse_diagonal) ba_col_sqnorm(&w, diag_f, diag_l);
        orc_lm_diagonal(diag_f, w.nf, radius, o->min_lm_diagonal, o->max_lm_diagonal, !reuse_diagonal, Df);
        orc_lm_diagonal(diag_l, p->n_lm, radius, o->min_lm_diagonal, o->max_lm_diagonal, !reuse_diagonal, Dl);
        n_steps++;
        int lin_ok = ba_schur_solve(&w, Df, Dl, yf, yl) == 0;
        reuse_diagonal = 1;
        int step_valid = 0;
        double model_cost_change = 0;
        if (lin_ok) {
            for (int c = 0; c < w.nf; c++) { if (!isfinite(yf[c])) lin_ok = 0; yf[c] = -yf[c]; }
            for (int l = 0; l < p->n_lm; l++) { if (!isfinite(yl[l])) lin_ok = 0; yl[l] = -yl[l]; }
        }
        if (lin_ok) {
            /* model_cost_change = -(J step) . (r + J step / 2) */
            for (int k = 0; k < w.n_act; k++) {
                int lm, ca, co;
                ba_res_blocks(&w, w.act[k], &lm, &ca, &co);
                const double *Ja = w.Ja + 12 * k, *Jo = w.Jo + 12 * k, *Jl = w.Jl + 2 * k, *r = w.r + 2 * k;
                double m0 = 0, m1 = 0;
                if (lm >= 0) { m0 = Jl[0] * yl[lm]; m1 = Jl[1] * yl[lm]; }
                for (int c = 0; c < 6; c++) {
                    if (ca >= 0) { m0 += Ja[c] * yf[ca + c]; m1 += Ja[6 + c] * yf[ca + c]; }
                    if (co >= 0) { m0 += Jo[c] * yf[co + c]; m1 += Jo[6 + c] * yf[co + c]; }
                }
                model_cost_change -= m0 * (r[0] + m0 / 2.0) + m1 * (r[1] + m1 / 2.0);
            }
            step_valid = model_cost_change > 0.0;
        }
        if (!step_valid) {
            /* HandleInvalidStep */
            if (++num_invalid >= o->max_consecutive_invalid_steps) { term = ORC_TERM_INVALID_STEPS; break; }
            orc_lm_step_rejected(&radius, &decrease_factor);
            reuse_diagonal = 1;
            cur.cost = x_cost;                         /* a step of length zero and no progress (:476-484) */
            continue;
        }
        cur.step_is_valid = 1;
        num_invalid = 0;
        /* delta = step .* scale ; candidate = Plus(x, delta) */
        memcpy(c_pose, x_pose, sizeof(double) * (size_t)NP);
        for (int k = 0; k < p->n_kf; k++) {
            if (w.pose_col[k] < 0) continue;
            double d[6];
            for (int c = 0; c < 6; c++) d[c] = yf[w.pose_col[k] + c] * w.scale_f[w.pose_col[k] + c];
            orc_se3_left_plus(x_pose + 7 * k, d, c_pose + 7 * k);
        }
        for (int l = 0; l < p->n_lm; l++) c_lam[l] = x_lam[l] + yl[l] * w.scale_l[l];
        const double cand_cost = ba_evaluate(&w, c_pose, c_lam, 0, res->chi2_last_eval, res->depthpos_last_eval, NULL, NULL);
        /* ParameterToleranceReached */
        double step_sq = 0;
        for (int k = 0; k < p->n_kf; k++) if (w.pose_col[k] >= 0) step_sq += vec_norm_diff(x_pose + 7 * k, c_pose + 7 * k, 7);
        for (int l = 0; l < p->n_lm; l++) if (w.lm_ptr[l] != w.lm_ptr[l + 1]) step_sq += (x_lam[l] - c_lam[l]) * (x_lam[l] - c_lam[l]);
        cur.step_norm = sqrt(step_sq);
        if (sqrt(step_sq) <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) { term = ORC_TERM_PARAMETER_TOL; break; }
        /* FunctionToleranceReached */
        cur.cost_change = x_cost - cand_cost;
        if (fabs(x_cost - cand_cost) <= o->function_tolerance * x_cost) { term = ORC_TERM_FUNCTION_TOL; break; }
        /* IsStepSuccessful */
        double rel;
        {
            const double r1 = (ev_cur - cand_cost) / model_cost_change;
            const double r2 = (ev_ref - cand_cost) / (ev_acc_ref + model_cost_change);
            rel = r1 > r2 ? r1 : r2;
        }
        cur.relative_decrease = rel;
        if (rel > o->min_relative_decrease) {
            /* HandleSuccessfulStep */
            memcpy(x_pose, c_pose, sizeof(double) * (size_t)NP);
            memcpy(x_lam, c_lam, sizeof(double) * (size_t)p->n_lm);
            double xn = 0;
            for (int k = 0; k < p->n_kf; k++) if (w.pose_col[k] >= 0) for (int c = 0; c < 7; c++) xn += x_pose[7 * k + c] * x_pose[7 * k + c];
            for (int l = 0; l < p->n_lm; l++) if (w.lm_ptr[l] != w.lm_ptr[l + 1]) xn += x_lam[l] * x_lam[l];
            x_norm = sqrt(xn);
            x_cost = ba_evaluate(&w, x_pose, x_lam, 1, res->chi2_last_eval, res->depthpos_last_eval, gf, gl);
            if (o->jacobi_scaling) ba_scale_columns(&w);
            GRAD_MAX_NORM();
            step_successful = 1;
            cur.step_is_successful = 1; cur.cost = x_cost; cur.gradient_max_norm = gmax; cur.gradient_norm = gnorm;
            orc_lm_step_accepted(rel, &radius, &decrease_factor, o->max_radius);
            reuse_diagonal = 0;
            /* step_evaluator_->StepAccepted(candidate_cost, model_cost_change) */
            ev_cur = cand_cost; ev_acc_cand += model_cost_change; ev_acc_ref += model_cost_change;
            if (ev_cur < ev_min) { ev_min = ev_cur; ev_nonmono = 0; ev_cand = ev_cur; ev_acc_cand = 0; }
            else { ev_nonmono++; if (ev_cur > ev_cand) { ev_cand = ev_cur; ev_acc_cand = 0; } }
            if (ev_nonmono == 0) { ev_ref = ev_cand; ev_acc_ref = ev_acc_cand; }
        } else {
            cur.cost = cand_cost;                      /* :119-127: the rejected candidate's cost, the gradient norms of the last accepted point */
            orc_lm_step_rejected(&radius, &decrease_factor);
            reuse_diagonal = 1;
        }
    }
    #undef GRAD_MAX_NORM
    if (res->poses_out) memcpy(res->poses_out, x_pose, sizeof(double) * (size_t)NP);
    if (res->invdepth_out) memcpy(res->invdepth_out, x_lam, sizeof(double) * (size_t)p->n_lm);
    res->iterations = n_steps;
    res->num_successful_steps = n_success;
    res->final_cost = minimum_cost;
    res->termination = term;
    free(w.pose_col); free(w.act); free(w.lm_ptr); free(w.lm_idx); free(w.r); free(w.Ja); free(w.Jo); free(w.Jl);
    free(w.scale_f); free(w.scale_l); free(x_pose); free(x_lam); free(c_pose); free(c_lam); free(gf); free(gl);
    free(diag_f); free(diag_l); free(Df); free(Dl); free(yf); free(yl);
    return 0;
}
