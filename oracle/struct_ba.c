/*
 * struct_ba.c -- CPU ORACLE (test infrastructure) for Optimizer::structureOnlyBA
 * (/root/reference/src/optimizer.cpp:2594-2781): every keyframe pose, both calibrations and the stereo
 * extrinsic are constant parameter blocks; the variables are 3-D world points (PointXYZParametersBlock),
 * observed through
 *     DirectLeftSE3::ReprojectionErrorKSE3XYZ          (src/ceres_parametrization.cpp, "KSE3XYZ::Evaluate")
 *     DirectLeftSE3::ReprojectionErrorRightCamKSE3XYZ  (same file, right camera through Trl)
 * with Huber(sqrt(robust_mono_th)) and Ceres options DENSE_SCHUR / LEVENBERG_MARQUARDT, 10 iterations,
 * function_tolerance 1e-3 (:2742-2758; the 10-20 ms wall-clock cap has no counterpart here).
 * With every pose constant Ceres removes them from the program; all remaining blocks are e-blocks, the
 * reduced system is empty and the LM step is the block-diagonal solve (J_p^T J_p + D^2) y_p = J_p^T r per
 * point -- inside the SAME trust-region loop as orc_ba_solve (ba.c), restated here for 3-parameter blocks.
 */
#include "ov2_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void q_to_R(const double q[4], double R[9])       /* q = (x, y, z, w), Eigen coefficient order */
{
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

/* residual of one block; J = d r / d X (2x3, row-major) if non-NULL.  returns depth > 0 */
int orc_xyz_residual(int type, const double calib_l[4], const double calib_r[4], const double T_rl[7], const double pose[7],
                     const double X[3], const double uv[2], double sigma, double r[2], double *J, double *chi2)
{
    double Rwc[9], Rcw[9], tcw[3], c[3];
    q_to_R(pose + 3, Rwc);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rcw[3 * i + j] = Rwc[3 * j + i];
    for (int i = 0; i < 3; i++) tcw[i] = -(Rcw[3 * i] * pose[0] + Rcw[3 * i + 1] * pose[1] + Rcw[3 * i + 2] * pose[2]);
    for (int i = 0; i < 3; i++) c[i] = Rcw[3 * i] * X[0] + Rcw[3 * i + 1] * X[1] + Rcw[3 * i + 2] * X[2] + tcw[i];
    double M[9];                                    /* d(cam point) / dX */
    const double *K = calib_l;
    memcpy(M, Rcw, sizeof(M));
    if (type == ORC_XYZ_RIGHT) {
        double Rrl[9], rc[3];
        q_to_R(T_rl + 3, Rrl);
        for (int i = 0; i < 3; i++) rc[i] = Rrl[3 * i] * c[0] + Rrl[3 * i + 1] * c[1] + Rrl[3 * i + 2] * c[2] + T_rl[i];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
            M[3 * i + j] = Rrl[3 * i] * Rcw[j] + Rrl[3 * i + 1] * Rcw[3 + j] + Rrl[3 * i + 2] * Rcw[6 + j];
        memcpy(c, rc, sizeof(rc));
        K = calib_r;
    }
    const double invz = 1. / c[2], si = 1. / sigma;
    r[0] = si * (K[0] * c[0] * invz + K[2] - uv[0]);
    r[1] = si * (K[1] * c[1] * invz + K[3] - uv[1]);
    *chi2 = r[0] * r[0] + r[1] * r[1];
    if (J) {
        const double invz2 = invz * invz;
        const double Jc[6] = {invz * K[0], 0., -c[0] * invz2 * K[0], 0., invz * K[1], -c[1] * invz2 * K[1]};
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++)
            J[3 * i + j] = si * (Jc[3 * i] * M[j] + Jc[3 * i + 1] * M[3 + j] + Jc[3 * i + 2] * M[6 + j]);
    }
    return c[2] > 0;
}

typedef struct {
    const orc_sba_problem *p;
    const orc_ba_options *o;
    int n_act, *act, *pt_ptr;
    double *r, *J;              /* per active residual: corrected residual (2) and scaled jacobian (6) */
    double *scale;              /* 3 per point */
} sba_ws;

static double sba_evaluate(sba_ws *w, const double *x, int want_jac, double *chi2_out, uint8_t *dpos_out, double *grad)
{
    const orc_sba_problem *p = w->p;
    double cost = 0;
    if (grad) memset(grad, 0, sizeof(double) * 3 * (size_t)p->n_pts);
    for (int k = 0; k < w->n_act; k++) {
        const int i = w->act[k], pt = p->res_pt[i];
        double r[2], J[6], chi2;
        const int dp = orc_xyz_residual(p->res_type[i], p->calib_l, p->calib_r, p->T_rl, p->poses + 7 * p->res_kf[i], x + 3 * pt,
                                        p->res_uv + 2 * i, p->res_sigma[i], r, want_jac ? J : NULL, &chi2);
        if (chi2_out) chi2_out[i] = chi2;
        if (dpos_out) dpos_out[i] = (uint8_t)dp;
        const double s = r[0] * r[0] + r[1] * r[1];
        double rho[3];
        if (w->o->huber_delta > 0) orc_huber(w->o->huber_delta, s, rho);
        else { rho[0] = s; rho[1] = 1; rho[2] = 0; }
        cost += 0.5 * rho[0];
        if (!want_jac) continue;
        double rr[2] = {r[0], r[1]};
        orc_corrector(s, rho, 2, 3, rr, J);
        if (grad) for (int c = 0; c < 3; c++) grad[3 * pt + c] += J[c] * rr[0] + J[3 + c] * rr[1];
        memcpy(w->r + 2 * k, rr, sizeof(rr));
        memcpy(w->J + 6 * k, J, sizeof(J));
    }
    return cost;
}

static void sba_col_sqnorm(const sba_ws *w, double *n)
{
    memset(n, 0, sizeof(double) * 3 * (size_t)w->p->n_pts);
    for (int k = 0; k < w->n_act; k++) {
        const int pt = w->p->res_pt[w->act[k]];
        const double *J = w->J + 6 * k;
        for (int c = 0; c < 3; c++) n[3 * pt + c] += J[c] * J[c] + J[3 + c] * J[3 + c];
    }
}

static void sba_scale_columns(sba_ws *w)
{
    for (int k = 0; k < w->n_act; k++) {
        const int pt = w->p->res_pt[w->act[k]];
        double *J = w->J + 6 * k;
        for (int c = 0; c < 3; c++) { J[c] *= w->scale[3 * pt + c]; J[3 + c] *= w->scale[3 * pt + c]; }
    }
}

/* block-diagonal solve (J^T J + D^2) y = J^T r per point; returns 0 / -1 (some block not positive definite) */
static int sba_solve_blocks(const sba_ws *w, const double *D, double *y)
{
    const int n = w->p->n_pts;
    double *H = (double *)calloc((size_t)n * 9 + 1, sizeof(double)), *g = (double *)calloc((size_t)n * 3 + 1, sizeof(double));
    for (int k = 0; k < w->n_act; k++) {
        const int pt = w->p->res_pt[w->act[k]];
        const double *J = w->J + 6 * k, *r = w->r + 2 * k;
        for (int a = 0; a < 3; a++) {
            g[3 * pt + a] += J[a] * r[0] + J[3 + a] * r[1];
            for (int b = 0; b < 3; b++) H[9 * pt + 3 * a + b] += J[a] * J[b] + J[3 + a] * J[3 + b];
        }
    }
    int rc = 0;
    for (int pt = 0; pt < n; pt++) {
        if (w->pt_ptr[pt] == w->pt_ptr[pt + 1]) { y[3 * pt] = y[3 * pt + 1] = y[3 * pt + 2] = 0; continue; }
        double *A = H + 9 * pt, L[9] = {0};
        for (int a = 0; a < 3; a++) A[4 * a] += D[3 * pt + a] * D[3 * pt + a];
        /* 3x3 Cholesky */
        int ok = 1;
        for (int j = 0; j < 3 && ok; j++) {
            double d = A[4 * j];
            for (int k = 0; k < j; k++) d -= L[3 * j + k] * L[3 * j + k];
            if (!(d > 0.0) || !isfinite(d)) { ok = 0; break; }
            L[4 * j] = sqrt(d);
            for (int i = j + 1; i < 3; i++) {
                double s = A[3 * i + j];
                for (int k = 0; k < j; k++) s -= L[3 * i + k] * L[3 * j + k];
                L[3 * i + j] = s / L[4 * j];
            }
        }
        if (!ok) { rc = -1; break; }
        double b[3] = {g[3 * pt], g[3 * pt + 1], g[3 * pt + 2]};
        for (int i = 0; i < 3; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[3 * i + k] * b[k]; b[i] = s / L[4 * i]; }
        for (int i = 2; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < 3; k++) s -= L[3 * k + i] * b[k]; b[i] = s / L[4 * i]; }
        y[3 * pt] = b[0]; y[3 * pt + 1] = b[1]; y[3 * pt + 2] = b[2];
    }
    free(H); free(g);
    return rc;
}

int orc_structure_ba(const orc_sba_problem *p, const orc_ba_options *o, orc_sba_result *res)
{
    if (!p || !o || !res || p->n_kf <= 0 || p->n_pts < 0 || p->n_res < 0) return -1;
    sba_ws w; memset(&w, 0, sizeof(w));
    w.p = p; w.o = o;
    w.act = (int *)malloc(sizeof(int) * (size_t)(p->n_res + 1));
    w.pt_ptr = (int *)calloc((size_t)p->n_pts + 2, sizeof(int));
    for (int i = 0; i < p->n_res; i++) {
        if (p->res_active && !p->res_active[i]) continue;
        if (p->res_pt[i] < 0 || p->res_pt[i] >= p->n_pts || p->res_kf[i] < 0 || p->res_kf[i] >= p->n_kf ||
            (p->res_type[i] != ORC_XYZ_LEFT && p->res_type[i] != ORC_XYZ_RIGHT)) { free(w.act); free(w.pt_ptr); return -1; }
        w.act[w.n_act++] = i;
        w.pt_ptr[p->res_pt[i] + 1]++;
    }
    for (int l = 0; l < p->n_pts; l++) w.pt_ptr[l + 1] += w.pt_ptr[l];
    const size_t na = (size_t)w.n_act + 1, N = 3 * (size_t)p->n_pts + 1;
    w.r = (double *)malloc(sizeof(double) * 2 * na);
    w.J = (double *)malloc(sizeof(double) * 6 * na);
    w.scale = (double *)malloc(sizeof(double) * N);
    for (size_t c = 0; c + 1 < N; c++) w.scale[c] = 1.0;
    double *x = (double *)malloc(sizeof(double) * N), *cand = (double *)malloc(sizeof(double) * N), *g = (double *)malloc(sizeof(double) * N);
    double *diag = (double *)malloc(sizeof(double) * N), *D = (double *)malloc(sizeof(double) * N), *y = (double *)malloc(sizeof(double) * N);
    memcpy(x, p->xyz, sizeof(double) * 3 * (size_t)p->n_pts);
    #define IN_PROGRAM(l) (w.pt_ptr[l] != w.pt_ptr[(l) + 1])

    double x_cost = sba_evaluate(&w, x, 1, res->chi2_last_eval, res->depthpos_last_eval, g);
    if (o->jacobi_scaling) {
        sba_col_sqnorm(&w, diag);
        for (size_t c = 0; c + 1 < N; c++) w.scale[c] = 1.0 / (1.0 + sqrt(diag[c]));
        sba_scale_columns(&w);
    }
    res->initial_cost = x_cost;
    double minimum_cost = x_cost, x_norm = -1.0, radius = o->initial_radius, decrease_factor = 2.0;
    int reuse_diagonal = 0, num_invalid = 0;
    double ev_min = x_cost, ev_cur = x_cost, ev_ref = x_cost, ev_cand = x_cost, ev_acc_ref = 0, ev_acc_cand = 0;
    int ev_nonmono = 0, iteration = 0, step_successful = 1, term = ORC_TERM_NO_CONVERGENCE, n_success = 0, n_steps = 0;
    double gmax = 0;
    #define GRAD_MAX_NORM() do { gmax = 0; for (int l_ = 0; l_ < p->n_pts; l_++) if (IN_PROGRAM(l_)) \
        for (int c_ = 0; c_ < 3; c_++) { const double v_ = fabs(g[3 * l_ + c_]); if (v_ > gmax) gmax = v_; } } while (0)
    GRAD_MAX_NORM();
    for (;;) {
        if (step_successful) { n_success++; if (x_cost < minimum_cost) minimum_cost = x_cost; }
        if (iteration >= o->max_iter) { term = ORC_TERM_NO_CONVERGENCE; break; }
        if (step_successful && gmax <= o->gradient_tolerance) { term = ORC_TERM_GRADIENT_TOL; break; }
        if (radius <= o->min_radius) { term = ORC_TERM_MIN_RADIUS; break; }
        iteration++;
        step_successful = 0;
        if (!reuse_diagonal) {
            sba_col_sqnorm(&w, diag);
            for (size_t c = 0; c + 1 < N; c++) diag[c] = fmin(fmax(diag[c], o->min_lm_diagonal), o->max_lm_diagonal);
        }
        for (size_t c = 0; c + 1 < N; c++) D[c] = sqrt(diag[c] / radius);
        n_steps++;
        int lin_ok = sba_solve_blocks(&w, D, y) == 0;
        reuse_diagonal = 1;
        int step_valid = 0;
        double model_cost_change = 0;
        if (lin_ok) for (size_t c = 0; c + 1 < N; c++) { if (!isfinite(y[c])) lin_ok = 0; y[c] = -y[c]; }
        if (lin_ok) {
            for (int k = 0; k < w.n_act; k++) {
                const int pt = p->res_pt[w.act[k]];
                const double *J = w.J + 6 * k, *r = w.r + 2 * k;
                const double m0 = J[0] * y[3 * pt] + J[1] * y[3 * pt + 1] + J[2] * y[3 * pt + 2];
                const double m1 = J[3] * y[3 * pt] + J[4] * y[3 * pt + 1] + J[5] * y[3 * pt + 2];
                model_cost_change -= m0 * (r[0] + m0 / 2.0) + m1 * (r[1] + m1 / 2.0);
            }
            step_valid = model_cost_change > 0.0;
        }
        if (!step_valid) {
            if (++num_invalid >= o->max_consecutive_invalid_steps) { term = ORC_TERM_INVALID_STEPS; break; }
            orc_lm_step_rejected(&radius, &decrease_factor);
            reuse_diagonal = 1;
            continue;
        }
        num_invalid = 0;
        for (size_t c = 0; c + 1 < N; c++) cand[c] = x[c] + y[c] * w.scale[c];
        const double cand_cost = sba_evaluate(&w, cand, 0, res->chi2_last_eval, res->depthpos_last_eval, NULL);
        double step_sq = 0;
        for (int l = 0; l < p->n_pts; l++) if (IN_PROGRAM(l)) for (int c = 0; c < 3; c++) step_sq += (x[3 * l + c] - cand[3 * l + c]) * (x[3 * l + c] - cand[3 * l + c]);
        if (sqrt(step_sq) <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) { term = ORC_TERM_PARAMETER_TOL; break; }
        if (fabs(x_cost - cand_cost) <= o->function_tolerance * x_cost) { term = ORC_TERM_FUNCTION_TOL; break; }
        double rel;
        {
            const double r1 = (ev_cur - cand_cost) / model_cost_change, r2 = (ev_ref - cand_cost) / (ev_acc_ref + model_cost_change);
            rel = r1 > r2 ? r1 : r2;
        }
        if (rel > o->min_relative_decrease) {
            memcpy(x, cand, sizeof(double) * 3 * (size_t)p->n_pts);
            double xn = 0;
            for (int l = 0; l < p->n_pts; l++) if (IN_PROGRAM(l)) for (int c = 0; c < 3; c++) xn += x[3 * l + c] * x[3 * l + c];
            x_norm = sqrt(xn);
            x_cost = sba_evaluate(&w, x, 1, res->chi2_last_eval, res->depthpos_last_eval, g);
            if (o->jacobi_scaling) sba_scale_columns(&w);
            GRAD_MAX_NORM();
            step_successful = 1;
            orc_lm_step_accepted(rel, &radius, &decrease_factor, o->max_radius);
            reuse_diagonal = 0;
            ev_cur = cand_cost; ev_acc_cand += model_cost_change; ev_acc_ref += model_cost_change;
            if (ev_cur < ev_min) { ev_min = ev_cur; ev_nonmono = 0; ev_cand = ev_cur; ev_acc_cand = 0; }
            else { ev_nonmono++; if (ev_cur > ev_cand) { ev_cand = ev_cur; ev_acc_cand = 0; } }
            if (ev_nonmono == 0) { ev_ref = ev_cand; ev_acc_ref = ev_acc_cand; }
        } else {
            orc_lm_step_rejected(&radius, &decrease_factor);
            reuse_diagonal = 1;
        }
    }
    #undef GRAD_MAX_NORM
    #undef IN_PROGRAM
    if (res->xyz_out) memcpy(res->xyz_out, x, sizeof(double) * 3 * (size_t)p->n_pts);
    res->iterations = n_steps; res->num_successful_steps = n_success; res->final_cost = minimum_cost; res->termination = term;
    free(w.act); free(w.pt_ptr); free(w.r); free(w.J); free(w.scale); free(x); free(cand); free(g); free(diag); free(D); free(y);
    return 0;
}
