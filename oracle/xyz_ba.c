/*
 * xyz_ba.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Bundle adjustment with 3-D point landmarks and VARIABLE poses: the `buse_inv_depth: 0` branch of Optimizer::localBA /
 * looseBA / fullBA (src/optimizer.cpp:207-209 PointXYZParametersBlock in group 0, :333-384 residual blocks; no shipped
 * parameter file selects it, the code path exists).  Factors (src/ceres_parametrization.cpp):
 *   ORC_XYZ_LEFT   DirectLeftSE3::ReprojectionErrorKSE3XYZ          :107-195   parameters {calib, pose, X}
 *   ORC_XYZ_RIGHT  DirectLeftSE3::ReprojectionErrorRightCamKSE3XYZ  :198-298   parameters {calib_r, pose, T_rl, X}
 * d r / d pose = [-J_R, J_R hat(X)] (left-multiplicative SE3 perturbation, tangent [upsilon, omega], :162-170), d r / d X
 * = J_R (:171-178); calibration and extrinsic blocks are constant.  Solver = the same restatement of Ceres 2.0.0's
 * trust-region Levenberg-Marquardt as oracle/ba.c (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc) with the
 * Schur complement over 3 x 3 e-blocks (schur_eliminator_impl.h:179-377: the e-block inverse is InvertPSDMatrix<3>, a
 * Cholesky solve of the identity).  Pinned like oracle/ba.c: finite-difference Jacobians, equality with orc_structure_ba
 * when every pose is constant, and an independent dense numpy LM (tests/test_oracle_xyz_ba.py).
 */
#include "ov2_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void hat_mul(const double JR[6], const double X[3], double out[6])
{
    /* J_R (2x3) * hat(X) (3x3),  hat(X) = [0 -z y; z 0 -x; -y x 0] */
    for (int i = 0; i < 2; i++) {
        const double a = JR[3 * i], b = JR[3 * i + 1], c = JR[3 * i + 2];
        out[3 * i]     = b * X[2] - c * X[1];
        out[3 * i + 1] = c * X[0] - a * X[2];
        out[3 * i + 2] = a * X[1] - b * X[0];
    }
}

/* residual of one block; Jp (2x6 row-major, local pose parameterisation) and Jx (2x3) when non-NULL */
int orc_xyzba_residual(int type, const double calib_l[4], const double calib_r[4], const double T_rl[7], const double pose[7],
                       const double X[3], const double uv[2], double sigma, double r[2], double *Jp, double *Jx, double *chi2)
{
    double JR[6];
    const int dp = orc_xyz_residual(type, calib_l, calib_r, T_rl, pose, X, uv, sigma, r, (Jp || Jx) ? JR : NULL, chi2);
    if (Jx) memcpy(Jx, JR, sizeof(JR));
    if (Jp) {
        double JH[6];
        hat_mul(JR, X, JH);
        for (int i = 0; i < 2; i++)
            for (int c = 0; c < 3; c++) { Jp[6 * i + c] = -JR[3 * i + c]; Jp[6 * i + 3 + c] = JH[3 * i + c]; }
    }
    return dp;
}

typedef struct {
    const orc_xyzba_problem *p;
    const orc_ba_options *o;
    int n_act, *act;
    int *pose_col, n_opt, nf;
    int *pt_ptr, *pt_idx;       /* CSR: point -> positions in act[] */
    double *r, *Jp, *Jx;        /* per active residual: corrected residual 2, scaled Jacobians 12 and 6 */
    double *scale_f, *scale_x;  /* jacobi scaling: nf and 3 * n_pts */
} xb_ws;

static double xb_evaluate(xb_ws *w, const double *poses, const double *xyz, int want_jac, double *chi2_out, uint8_t *dpos_out,
                          double *grad_f, double *grad_x)
{
    const orc_xyzba_problem *p = w->p;
    double cost = 0;
    if (grad_f) memset(grad_f, 0, sizeof(double) * (size_t)w->nf);
    if (grad_x) memset(grad_x, 0, sizeof(double) * 3 * (size_t)p->n_pts);
    for (int k = 0; k < w->n_act; k++) {
        const int i = w->act[k], pt = p->res_pt[i], kf = p->res_kf[i];
        double r[2], Jp[12], Jx[6], chi2;
        const int dp = orc_xyzba_residual(p->res_type[i], p->calib_l, p->calib_r, p->T_rl, poses + 7 * kf, xyz + 3 * pt,
                                          p->res_uv + 2 * i, p->res_sigma[i], r, want_jac ? Jp : NULL, want_jac ? Jx : NULL, &chi2);
        if (chi2_out) chi2_out[i] = chi2;
        if (dpos_out) dpos_out[i] = (uint8_t)dp;
        const double s = r[0] * r[0] + r[1] * r[1];
        double rho[3];
        if (w->o->huber_delta > 0) orc_huber(w->o->huber_delta, s, rho);
        else { rho[0] = s; rho[1] = 1; rho[2] = 0; }
        cost += 0.5 * rho[0];
        if (!want_jac) continue;
        double rr[2] = {r[0], r[1]};
        orc_corrector(s, rho, 2, 6, rr, Jp); rr[0] = r[0]; rr[1] = r[1];
        orc_corrector(s, rho, 2, 3, rr, Jx);
        const int co = w->pose_col[kf];
        if (co < 0) memset(Jp, 0, sizeof(Jp));
        if (grad_f && co >= 0) for (int c = 0; c < 6; c++) grad_f[co + c] += Jp[c] * rr[0] + Jp[6 + c] * rr[1];
        if (grad_x) for (int c = 0; c < 3; c++) grad_x[3 * pt + c] += Jx[c] * rr[0] + Jx[3 + c] * rr[1];
        memcpy(w->r + 2 * k, rr, sizeof(rr));
        memcpy(w->Jp + 12 * k, Jp, sizeof(Jp));
        memcpy(w->Jx + 6 * k, Jx, sizeof(Jx));
    }
    return cost;
}

static void xb_col_sqnorm(const xb_ws *w, double *nf, double *nx)
{
    const orc_xyzba_problem *p = w->p;
    memset(nf, 0, sizeof(double) * (size_t)w->nf);
    memset(nx, 0, sizeof(double) * 3 * (size_t)p->n_pts);
    for (int k = 0; k < w->n_act; k++) {
        const int i = w->act[k], co = w->pose_col[p->res_kf[i]], pt = p->res_pt[i];
        const double *Jp = w->Jp + 12 * k, *Jx = w->Jx + 6 * k;
        if (co >= 0) for (int c = 0; c < 6; c++) nf[co + c] += Jp[c] * Jp[c] + Jp[6 + c] * Jp[6 + c];
        for (int c = 0; c < 3; c++) nx[3 * pt + c] += Jx[c] * Jx[c] + Jx[3 + c] * Jx[3 + c];
    }
}

static void xb_scale_columns(xb_ws *w)
{
    const orc_xyzba_problem *p = w->p;
    for (int k = 0; k < w->n_act; k++) {
        const int i = w->act[k], co = w->pose_col[p->res_kf[i]], pt = p->res_pt[i];
        double *Jp = w->Jp + 12 * k, *Jx = w->Jx + 6 * k;
        if (co >= 0) for (int c = 0; c < 6; c++) { Jp[c] *= w->scale_f[co + c]; Jp[6 + c] *= w->scale_f[co + c]; }
        for (int c = 0; c < 3; c++) { Jx[c] *= w->scale_x[3 * pt + c]; Jx[3 + c] *= w->scale_x[3 * pt + c]; }
    }
}

static int chol_lower_n(double *A, int n)
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

static void chol_solve_n(const double *L, int n, double *b)
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

/* inverse of a 3x3 SPD matrix through its Cholesky factor (InvertPSDMatrix<3>); returns 0 / -1 */
static int inv3_spd(const double M[9], double inv[9])
{
    double L[9];
    memcpy(L, M, sizeof(L));
    if (chol_lower_n(L, 3) != 0) return -1;
    for (int c = 0; c < 3; c++) {
        double e[3] = {0, 0, 0};
        e[c] = 1.0;
        chol_solve_n(L, 3, e);
        for (int r = 0; r < 3; r++) inv[3 * r + c] = e[r];
    }
    return 0;
}

/* Schur-complement solve of  min |J y - r|^2 + |D y|^2  with 3x3 e-blocks */
static int xb_schur_solve(const xb_ws *w, const double *Df, const double *Dx, double *yf, double *yx)
{
    const orc_xyzba_problem *p = w->p;
    const int nf = w->nf;
    double *S = (double *)calloc((size_t)nf * nf + 1, sizeof(double));
    double *rhs = (double *)calloc((size_t)nf + 1, sizeof(double));
    double *wrow = (double *)malloc(sizeof(double) * 3 * (size_t)(nf + 1));      /* E^T F: 3 rows */
    int *touched = (int *)malloc(sizeof(int) * (size_t)(w->n_opt + 1));
    uint8_t *flag = (uint8_t *)calloc((size_t)w->n_opt + 1, 1);
    double *einv = (double *)calloc(9 * (size_t)p->n_pts + 1, sizeof(double));
    double *etb = (double *)calloc(3 * (size_t)p->n_pts + 1, sizeof(double));
    int rc = 0;
    for (int c = 0; c < nf; c++) S[(size_t)c * nf + c] = Df[c] * Df[c];
    for (int pt = 0; pt < p->n_pts && rc == 0; pt++) {
        if (w->pt_ptr[pt] == w->pt_ptr[pt + 1]) continue;
        double ete[9] = {Dx[3 * pt] * Dx[3 * pt], 0, 0, 0, Dx[3 * pt + 1] * Dx[3 * pt + 1], 0, 0, 0, Dx[3 * pt + 2] * Dx[3 * pt + 2]};
        double g[3] = {0, 0, 0};
        int nt = 0;
        for (int q = w->pt_ptr[pt]; q < w->pt_ptr[pt + 1]; q++) {
            const int k = w->pt_idx[q], i = w->act[k], co = w->pose_col[p->res_kf[i]];
            const double *Jp = w->Jp + 12 * k, *Jx = w->Jx + 6 * k, *r = w->r + 2 * k;
            for (int a = 0; a < 3; a++) {
                g[a] += Jx[a] * r[0] + Jx[3 + a] * r[1];
                for (int b = 0; b < 3; b++) ete[3 * a + b] += Jx[a] * Jx[b] + Jx[3 + a] * Jx[3 + b];
            }
            if (co < 0) continue;
            if (!flag[co / 6]) { flag[co / 6] = 1; touched[nt++] = co; for (int a = 0; a < 3; a++) for (int c = 0; c < 6; c++) wrow[(size_t)a * nf + co + c] = 0; }
            for (int c = 0; c < 6; c++) {
                rhs[co + c] += Jp[c] * r[0] + Jp[6 + c] * r[1];
                for (int d = 0; d < 6; d++) S[(size_t)(co + c) * nf + co + d] += Jp[c] * Jp[d] + Jp[6 + c] * Jp[6 + d];
                for (int a = 0; a < 3; a++) wrow[(size_t)a * nf + co + c] += Jx[a] * Jp[c] + Jx[3 + a] * Jp[6 + c];
            }
        }
        double inv[9];
        if (inv3_spd(ete, inv) != 0) { rc = -1; break; }
        memcpy(einv + 9 * (size_t)pt, inv, sizeof(inv));
        memcpy(etb + 3 * (size_t)pt, g, sizeof(g));
        /* S -= W^T inv W ; rhs -= W^T inv g */
        double ig[3];
        for (int a = 0; a < 3; a++) ig[a] = inv[3 * a] * g[0] + inv[3 * a + 1] * g[1] + inv[3 * a + 2] * g[2];
        for (int x = 0; x < nt; x++)
            for (int c = 0; c < 6; c++) {
                const int cc = touched[x] + c;
                double wi[3];                                   /* (W^T inv) row cc */
                for (int b = 0; b < 3; b++) wi[b] = wrow[cc] * inv[b] + wrow[(size_t)nf + cc] * inv[3 + b] + wrow[2 * (size_t)nf + cc] * inv[6 + b];
                rhs[cc] -= wrow[cc] * ig[0] + wrow[(size_t)nf + cc] * ig[1] + wrow[2 * (size_t)nf + cc] * ig[2];
                for (int y = 0; y < nt; y++)
                    for (int d = 0; d < 6; d++) {
                        const int dd = touched[y] + d;
                        S[(size_t)cc * nf + dd] -= wi[0] * wrow[dd] + wi[1] * wrow[(size_t)nf + dd] + wi[2] * wrow[2 * (size_t)nf + dd];
                    }
            }
        for (int x = 0; x < nt; x++) flag[touched[x] / 6] = 0;
    }
    if (rc == 0 && nf > 0) {
        rc = chol_lower_n(S, nf);
        if (rc == 0) { memcpy(yf, rhs, sizeof(double) * (size_t)nf); chol_solve_n(S, nf, yf); }
    }
    if (rc == 0) {
        for (int pt = 0; pt < p->n_pts; pt++) {
            yx[3 * pt] = yx[3 * pt + 1] = yx[3 * pt + 2] = 0;
            if (w->pt_ptr[pt] == w->pt_ptr[pt + 1]) continue;
            double acc[3] = {etb[3 * pt], etb[3 * pt + 1], etb[3 * pt + 2]};
            for (int q = w->pt_ptr[pt]; q < w->pt_ptr[pt + 1]; q++) {
                const int k = w->pt_idx[q], i = w->act[k], co = w->pose_col[p->res_kf[i]];
                if (co < 0) continue;
                const double *Jp = w->Jp + 12 * k, *Jx = w->Jx + 6 * k;
                double m0 = 0, m1 = 0;
                for (int c = 0; c < 6; c++) { m0 += Jp[c] * yf[co + c]; m1 += Jp[6 + c] * yf[co + c]; }
                for (int a = 0; a < 3; a++) acc[a] -= Jx[a] * m0 + Jx[3 + a] * m1;
            }
            const double *inv = einv + 9 * (size_t)pt;
            for (int a = 0; a < 3; a++) yx[3 * pt + a] = inv[3 * a] * acc[0] + inv[3 * a + 1] * acc[1] + inv[3 * a + 2] * acc[2];
        }
    }
    free(S); free(rhs); free(wrow); free(touched); free(flag); free(einv); free(etb);
    return rc;
}

int orc_xyzba_solve(const orc_xyzba_problem *p, const orc_ba_options *o, orc_xyzba_result *res)
{
    if (!p || !o || !res || p->n_kf <= 0 || p->n_pts < 0 || p->n_res < 0) return -1;
    xb_ws w; memset(&w, 0, sizeof(w));
    w.p = p; w.o = o;
    w.pose_col = (int *)malloc(sizeof(int) * (size_t)p->n_kf);
    for (int k = 0; k < p->n_kf; k++) {
        if (p->kf_const && p->kf_const[k]) w.pose_col[k] = -1;
        else { w.pose_col[k] = 6 * w.n_opt; w.n_opt++; }
    }
    w.nf = 6 * w.n_opt;
    w.act = (int *)malloc(sizeof(int) * (size_t)(p->n_res + 1));
    for (int i = 0; i < p->n_res; i++) {
        if (p->res_active && !p->res_active[i]) continue;
        if (p->res_pt[i] < 0 || p->res_pt[i] >= p->n_pts || p->res_kf[i] < 0 || p->res_kf[i] >= p->n_kf || p->res_type[i] > ORC_XYZ_RIGHT) {
            free(w.pose_col); free(w.act); return -1;
        }
        w.act[w.n_act++] = i;
    }
    w.pt_ptr = (int *)calloc((size_t)p->n_pts + 2, sizeof(int));
    w.pt_idx = (int *)malloc(sizeof(int) * (size_t)(w.n_act + 1));
    for (int k = 0; k < w.n_act; k++) w.pt_ptr[p->res_pt[w.act[k]] + 1]++;
    for (int l = 0; l < p->n_pts; l++) w.pt_ptr[l + 1] += w.pt_ptr[l];
    {
        int *fill = (int *)malloc(sizeof(int) * (size_t)(p->n_pts + 1));
        memcpy(fill, w.pt_ptr, sizeof(int) * (size_t)(p->n_pts + 1));
        for (int k = 0; k < w.n_act; k++) w.pt_idx[fill[p->res_pt[w.act[k]]]++] = k;
        free(fill);
    }
    const size_t na = (size_t)w.n_act + 1;
    const int NX = 3 * p->n_pts, NP = 7 * p->n_kf;
    w.r = (double *)malloc(sizeof(double) * 2 * na);
    w.Jp = (double *)malloc(sizeof(double) * 12 * na);
    w.Jx = (double *)malloc(sizeof(double) * 6 * na);
    w.scale_f = (double *)malloc(sizeof(double) * (size_t)(w.nf + 1));
    w.scale_x = (double *)malloc(sizeof(double) * (size_t)(NX + 1));
    for (int c = 0; c < w.nf; c++) w.scale_f[c] = 1.0;
    for (int l = 0; l < NX; l++) w.scale_x[l] = 1.0;
#define XB_ALLOC(n) ((double *)malloc(sizeof(double) * (size_t)((n) + 1)))
    double *x_pose = XB_ALLOC(NP), *c_pose = XB_ALLOC(NP), *x_pt = XB_ALLOC(NX), *c_pt = XB_ALLOC(NX);
    double *gf = XB_ALLOC(w.nf), *gx = XB_ALLOC(NX), *diag_f = XB_ALLOC(w.nf), *diag_x = XB_ALLOC(NX);
    double *Df = XB_ALLOC(w.nf), *Dx = XB_ALLOC(NX), *yf = XB_ALLOC(w.nf), *yx = XB_ALLOC(NX);
#undef XB_ALLOC
    memcpy(x_pose, p->poses, sizeof(double) * (size_t)NP);
    memcpy(x_pt, p->xyz, sizeof(double) * (size_t)NX);

    double x_cost = xb_evaluate(&w, x_pose, x_pt, 1, res->chi2_last_eval, res->depthpos_last_eval, gf, gx);
    if (o->jacobi_scaling) {
        xb_col_sqnorm(&w, diag_f, diag_x);
        for (int c = 0; c < w.nf; c++) w.scale_f[c] = 1.0 / (1.0 + sqrt(diag_f[c]));
        for (int l = 0; l < NX; l++) w.scale_x[l] = 1.0 / (1.0 + sqrt(diag_x[l]));
        xb_scale_columns(&w);
    }
    res->initial_cost = x_cost;
    double minimum_cost = x_cost, x_norm = -1.0, radius = o->initial_radius, decrease_factor = 2.0;
    int reuse_diagonal = 0, num_invalid = 0;
    double ev_min = x_cost, ev_cur = x_cost, ev_ref = x_cost, ev_cand = x_cost, ev_acc_ref = 0, ev_acc_cand = 0;
    int ev_nonmono = 0;
    int iteration = 0, step_successful = 1, term = ORC_TERM_NO_CONVERGENCE, n_success = 0, n_steps = 0;
    double gmax = 0;
#define XB_HAS(l) (w.pt_ptr[(l)] != w.pt_ptr[(l) + 1])
#define GRAD_MAX_NORM()                                                                                     \
    do {                                                                                                    \
        gmax = 0;                                                                                           \
        for (int k_ = 0; k_ < p->n_kf; k_++) {                                                              \
            if (w.pose_col[k_] < 0) continue;                                                               \
            double d_[6], out_[7];                                                                          \
            for (int c_ = 0; c_ < 6; c_++) d_[c_] = -gf[w.pose_col[k_] + c_];                               \
            orc_se3_left_plus(x_pose + 7 * k_, d_, out_);                                                   \
            for (int c_ = 0; c_ < 7; c_++) { double v_ = fabs(x_pose[7 * k_ + c_] - out_[c_]); if (v_ > gmax) gmax = v_; } \
        }                                                                                                   \
        for (int l_ = 0; l_ < p->n_pts; l_++) {                                                             \
            if (!XB_HAS(l_)) continue;                                                                      \
            for (int c_ = 0; c_ < 3; c_++) { double v_ = fabs(gx[3 * l_ + c_]); if (v_ > gmax) gmax = v_; } \
        }                                                                                                   \
    } while (0)
    GRAD_MAX_NORM();

    for (;;) {
        if (step_successful) { n_success++; if (x_cost < minimum_cost) minimum_cost = x_cost; }
        if (iteration >= o->max_iter) { term = ORC_TERM_NO_CONVERGENCE; break; }
        if (step_successful && gmax <= o->gradient_tolerance) { term = ORC_TERM_GRADIENT_TOL; break; }
        if (radius <= o->min_radius) { term = ORC_TERM_MIN_RADIUS; break; }
        iteration++;
        step_successful = 0;
        if (!reuse_diagonal) xb_col_sqnorm(&w, diag_f, diag_x);
        orc_lm_diagonal(diag_f, w.nf, radius, o->min_lm_diagonal, o->max_lm_diagonal, !reuse_diagonal, Df);
        orc_lm_diagonal(diag_x, NX, radius, o->min_lm_diagonal, o->max_lm_diagonal, !reuse_diagonal, Dx);
        n_steps++;
        int lin_ok = xb_schur_solve(&w, Df, Dx, yf, yx) == 0;
        reuse_diagonal = 1;
        int step_valid = 0;
        double model_cost_change = 0;
        if (lin_ok) {
            for (int c = 0; c < w.nf; c++) { if (!isfinite(yf[c])) lin_ok = 0; yf[c] = -yf[c]; }
            for (int l = 0; l < NX; l++) { if (!isfinite(yx[l])) lin_ok = 0; yx[l] = -yx[l]; }
        }
        if (lin_ok) {
            for (int k = 0; k < w.n_act; k++) {
                const int i = w.act[k], co = w.pose_col[p->res_kf[i]], pt = p->res_pt[i];
                const double *Jp = w.Jp + 12 * k, *Jx = w.Jx + 6 * k, *r = w.r + 2 * k;
                double m0 = 0, m1 = 0;
                for (int c = 0; c < 3; c++) { m0 += Jx[c] * yx[3 * pt + c]; m1 += Jx[3 + c] * yx[3 * pt + c]; }
                if (co >= 0) for (int c = 0; c < 6; c++) { m0 += Jp[c] * yf[co + c]; m1 += Jp[6 + c] * yf[co + c]; }
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
        memcpy(c_pose, x_pose, sizeof(double) * (size_t)NP);
        for (int k = 0; k < p->n_kf; k++) {
            if (w.pose_col[k] < 0) continue;
            double d[6];
            for (int c = 0; c < 6; c++) d[c] = yf[w.pose_col[k] + c] * w.scale_f[w.pose_col[k] + c];
            orc_se3_left_plus(x_pose + 7 * k, d, c_pose + 7 * k);
        }
        for (int l = 0; l < NX; l++) c_pt[l] = x_pt[l] + yx[l] * w.scale_x[l];
        const double cand_cost = xb_evaluate(&w, c_pose, c_pt, 0, res->chi2_last_eval, res->depthpos_last_eval, NULL, NULL);
        double step_sq = 0;
        for (int k = 0; k < p->n_kf; k++) if (w.pose_col[k] >= 0) for (int c = 0; c < 7; c++) { const double d = x_pose[7 * k + c] - c_pose[7 * k + c]; step_sq += d * d; }
        for (int l = 0; l < p->n_pts; l++) if (XB_HAS(l)) for (int c = 0; c < 3; c++) { const double d = x_pt[3 * l + c] - c_pt[3 * l + c]; step_sq += d * d; }
        if (sqrt(step_sq) <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) { term = ORC_TERM_PARAMETER_TOL; break; }
        if (fabs(x_cost - cand_cost) <= o->function_tolerance * x_cost) { term = ORC_TERM_FUNCTION_TOL; break; }
        double rel;
        {
            const double r1 = (ev_cur - cand_cost) / model_cost_change;
            const double r2 = (ev_ref - cand_cost) / (ev_acc_ref + model_cost_change);
            rel = r1 > r2 ? r1 : r2;
        }
        if (rel > o->min_relative_decrease) {
            memcpy(x_pose, c_pose, sizeof(double) * (size_t)NP);
            memcpy(x_pt, c_pt, sizeof(double) * (size_t)NX);
            double xn = 0;
            for (int k = 0; k < p->n_kf; k++) if (w.pose_col[k] >= 0) for (int c = 0; c < 7; c++) xn += x_pose[7 * k + c] * x_pose[7 * k + c];
            for (int l = 0; l < p->n_pts; l++) if (XB_HAS(l)) for (int c = 0; c < 3; c++) xn += x_pt[3 * l + c] * x_pt[3 * l + c];
            x_norm = sqrt(xn);
            x_cost = xb_evaluate(&w, x_pose, x_pt, 1, res->chi2_last_eval, res->depthpos_last_eval, gf, gx);
            if (o->jacobi_scaling) xb_scale_columns(&w);
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
#undef XB_HAS
    if (res->poses_out) memcpy(res->poses_out, x_pose, sizeof(double) * (size_t)NP);
    if (res->xyz_out) memcpy(res->xyz_out, x_pt, sizeof(double) * (size_t)NX);
    res->iterations = n_steps;
    res->num_successful_steps = n_success;
    res->final_cost = minimum_cost;
    res->termination = term;
    free(w.pose_col); free(w.act); free(w.pt_ptr); free(w.pt_idx); free(w.r); free(w.Jp); free(w.Jx); free(w.scale_f); free(w.scale_x);
    free(x_pose); free(c_pose); free(x_pt); free(c_pt); free(gf); free(gx); free(diag_f); free(diag_x); free(Df); free(Dx); free(yf); free(yx);
    return 0;
}
