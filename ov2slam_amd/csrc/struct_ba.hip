// struct_ba.hip -- Optimizer::structureOnlyBA for gfx950 (/root/reference/src/optimizer.cpp:2594-2781).
//
// Every keyframe pose, both calibrations and the stereo extrinsic are constant; the variables are 3-D world
// points observed through DirectLeftSE3::ReprojectionErrorKSE3XYZ / ReprojectionErrorRightCamKSE3XYZ
// (src/ceres_parametrization.cpp) with a Huber loss.  Ceres removes the constant blocks, so every remaining
// block is an e-block: the reduced camera system is empty and one LM step is the block-diagonal solve
// (J_p^T J_p + D_p^2) y_p = J_p^T r_p per point -- but all points share ONE trust region (radius, accept /
// reject, tolerances), so this is a single coupled Ceres TrustRegionMinimizer loop, not n independent solves.
//
// Design: ONE launch, one 1024-thread workgroup.  A thread owns points t, t+1024, ... and everything that is
// per point (residuals, corrected Jacobians, Jacobi scaling, LM diagonal, 3x3 Cholesky, candidate) -- no
// atomics; the loop-level scalars (costs, model cost change, norms, gradient max-norm) are workgroup
// reductions in a fixed order.  The problem this is called on (the map points merged at a loop closure,
// src/loop_closer.cpp:353) has 10^2..10^4 points: latency, not throughput, is what matters, and a single
// launch with no host round trips is the shortest path.
#include "common.hpp"
#include <float.h>
#include <math.h>

#pragma clang fp contract(off)

struct SbaOut { int iterations, num_successful_steps, termination, pad; double initial_cost, final_cost; };

struct SbaDev {
    int n_kf, n_pts, n_res;
    const double *poses;          // 7 * n_kf
    double *kf_rt;                // 12 * n_kf: Rcw (row-major), tcw
    const int *pt_ptr, *pt_res;   // CSR: point -> active residual indices (ascending)
    const uint8_t *res_type;
    const int *res_kf;
    const double *res_uv, *res_sigma;
    double *x, *cand, *g, *scale, *diag, *y;   // 3 * n_pts each
    double *jr;                   // 8 per CSR slot: corrected (scaled) Jacobian 2x3 + corrected residual
    double *chi2; uint8_t *dpos;  // n_res
    double calib_l[4], calib_r[4], Rrl[9], trl[3];
    ov2_ba_options o;
    SbaOut *out;
};

__device__ __forceinline__ void sba_quat_to_R(const double *q, double *R)
{
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

// one residual block; J = d r / d X (2x3) when JAC
template <bool JAC>
__device__ __forceinline__ int sba_residual(const SbaDev &D, int type, const double *rt, const double *X, const double *uv, double sigma,
                                            double *r, double *J)
{
    double c[3];
    for (int i = 0; i < 3; i++) c[i] = rt[3 * i] * X[0] + rt[3 * i + 1] * X[1] + rt[3 * i + 2] * X[2] + rt[9 + i];
    double M[9];
    const double *K = D.calib_l;
    for (int k = 0; k < 9; k++) M[k] = rt[k];
    if (type == OV2_XYZ_RIGHT) {
        double rc[3];
        for (int i = 0; i < 3; i++) rc[i] = D.Rrl[3 * i] * c[0] + D.Rrl[3 * i + 1] * c[1] + D.Rrl[3 * i + 2] * c[2] + D.trl[i];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
            M[3 * i + j] = D.Rrl[3 * i] * rt[j] + D.Rrl[3 * i + 1] * rt[3 + j] + D.Rrl[3 * i + 2] * rt[6 + j];
        c[0] = rc[0]; c[1] = rc[1]; c[2] = rc[2];
        K = D.calib_r;
    }
    const double invz = 1. / c[2], si = 1. / sigma;
    r[0] = si * (K[0] * c[0] * invz + K[2] - uv[0]);
    r[1] = si * (K[1] * c[1] * invz + K[3] - uv[1]);
    if (JAC) {
        const double invz2 = invz * invz;
        const double Jc[6] = {invz * K[0], 0., -c[0] * invz2 * K[0], 0., invz * K[1], -c[1] * invz2 * K[1]};
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++)
            J[3 * i + j] = si * (Jc[3 * i] * M[j] + Jc[3 * i + 1] * M[3 + j] + Jc[3 * i + 2] * M[6 + j]);
    }
    return c[2] > 0;
}

__device__ __forceinline__ void sba_huber(double a, double s, double &rho0, double &rho1)
{
    if (a > 0 && s > a * a) {
        const double r = sqrt(s);
        rho0 = 2.0 * a * r - a * a;
        rho1 = fmax(DBL_MIN, a / r);
    } else { rho0 = s; rho1 = 1.0; }
}

// deterministic workgroup reductions (result in every thread)
__device__ __forceinline__ double sba_block_sum(double v, double *sh)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0;
    for (int w = 0; w < nw; w++) t += sh[w];
    return t;
}
__device__ __forceinline__ double sba_block_max(double v, double *sh)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0;
    for (int w = 0; w < nw; w++) t = fmax(t, sh[w]);
    return t;
}

// cost of the active residuals at `x` (thread-local partial); JAC: also store the corrected Jacobians / residuals
// (columns scaled by D.scale) and the gradient J^T r of the un-scaled Jacobian.
template <bool JAC>
__device__ __forceinline__ double sba_evaluate(const SbaDev &D, const double *x)
{
    double cost = 0;
    for (int pt = threadIdx.x; pt < D.n_pts; pt += blockDim.x) {
        const double X[3] = {x[3 * pt], x[3 * pt + 1], x[3 * pt + 2]};
        double g[3] = {0, 0, 0};
        const double sc0 = D.scale[3 * pt], sc1 = D.scale[3 * pt + 1], sc2 = D.scale[3 * pt + 2];
        for (int s = D.pt_ptr[pt]; s < D.pt_ptr[pt + 1]; s++) {
            const int i = D.pt_res[s];
            double r[2], J[6];
            const int dp = sba_residual<JAC>(D, D.res_type[i], D.kf_rt + 12 * D.res_kf[i], X, D.res_uv + 2 * i, D.res_sigma[i], r, J);
            const double sq = r[0] * r[0] + r[1] * r[1];
            D.chi2[i] = sq; D.dpos[i] = (uint8_t)dp;
            double rho0, rho1;
            sba_huber(D.o.huber_delta, sq, rho0, rho1);
            cost += 0.5 * rho0;
            if (JAC) {
                const double k = sqrt(rho1);            // corrector.cc: rho'' <= 0 for Huber / trivial loss
                for (int q = 0; q < 6; q++) J[q] *= k;
                r[0] *= k; r[1] *= k;
                for (int c = 0; c < 3; c++) g[c] += J[c] * r[0] + J[3 + c] * r[1];
                double *o = D.jr + 8 * (size_t)s;
                o[0] = J[0] * sc0; o[1] = J[1] * sc1; o[2] = J[2] * sc2; o[3] = J[3] * sc0; o[4] = J[4] * sc1; o[5] = J[5] * sc2;
                o[6] = r[0]; o[7] = r[1];
            }
        }
        if (JAC) { D.g[3 * pt] = g[0]; D.g[3 * pt + 1] = g[1]; D.g[3 * pt + 2] = g[2]; }
    }
    return cost;
}

// squared column norms of the stored (scaled) Jacobian of a point
__device__ __forceinline__ void sba_colnorm(const SbaDev &D, int pt, double n[3])
{
    n[0] = n[1] = n[2] = 0;
    for (int s = D.pt_ptr[pt]; s < D.pt_ptr[pt + 1]; s++) {
        const double *J = D.jr + 8 * (size_t)s;
        for (int c = 0; c < 3; c++) n[c] += J[c] * J[c] + J[3 + c] * J[3 + c];
    }
}

__global__ __launch_bounds__(1024) void k_structure_ba(SbaDev D)
{
    __shared__ double sh[16];
    const ov2_ba_options &o = D.o;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int k = tid; k < D.n_kf; k += nt) {
        double Rwc[9];
        const double *p = D.poses + 7 * k;
        sba_quat_to_R(p + 3, Rwc);
        double *rt = D.kf_rt + 12 * k;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rt[3 * i + j] = Rwc[3 * j + i];
        for (int i = 0; i < 3; i++) rt[9 + i] = -(rt[3 * i] * p[0] + rt[3 * i + 1] * p[1] + rt[3 * i + 2] * p[2]);
    }
    for (int c = tid; c < 3 * D.n_pts; c += nt) D.scale[c] = 1.0;
    __syncthreads();
    auto in_program = [&](int pt) { return D.pt_ptr[pt] != D.pt_ptr[pt + 1]; };
    auto grad_max = [&]() {
        double m = 0;
        for (int pt = tid; pt < D.n_pts; pt += nt) if (in_program(pt)) for (int c = 0; c < 3; c++) m = fmax(m, fabs(D.g[3 * pt + c]));
        return sba_block_max(m, sh);
    };

    // iteration 0
    double x_cost = sba_block_sum(sba_evaluate<true>(D, D.x), sh);
    if (o.jacobi_scaling) {
        for (int pt = tid; pt < D.n_pts; pt += nt) {
            double n[3];
            sba_colnorm(D, pt, n);
            for (int c = 0; c < 3; c++) D.scale[3 * pt + c] = 1.0 / (1.0 + sqrt(n[c]));
            for (int s = D.pt_ptr[pt]; s < D.pt_ptr[pt + 1]; s++) {
                double *J = D.jr + 8 * (size_t)s;
                for (int c = 0; c < 3; c++) { J[c] *= D.scale[3 * pt + c]; J[3 + c] *= D.scale[3 * pt + c]; }
            }
        }
    }
    const double initial_cost = x_cost;
    double minimum_cost = x_cost, x_norm = -1.0, radius = o.initial_radius, decrease_factor = 2.0;
    int reuse_diagonal = 0, num_invalid = 0;
    double ev_min = x_cost, ev_cur = x_cost, ev_ref = x_cost, ev_cand = x_cost, ev_acc_ref = 0, ev_acc_cand = 0;
    int ev_nonmono = 0, iteration = 0, step_successful = 1, term = OV2_TERM_NO_CONVERGENCE, n_success = 0, n_steps = 0;
    double gmax = grad_max();

    for (;;) {                                         // every scalar below is identical in all threads
        if (step_successful) { n_success++; if (x_cost < minimum_cost) minimum_cost = x_cost; }
        if (iteration >= o.max_iter) { term = OV2_TERM_NO_CONVERGENCE; break; }
        if (step_successful && gmax <= o.gradient_tolerance) { term = OV2_TERM_GRADIENT_TOL; break; }
        if (radius <= o.min_radius) { term = OV2_TERM_MIN_RADIUS; break; }
        iteration++;
        step_successful = 0;
        n_steps++;
        // LM diagonal, block-diagonal solve and the model cost change, all per point
        double mcc = 0, bad = 0;
        for (int pt = tid; pt < D.n_pts; pt += nt) {
            if (!reuse_diagonal) {
                double n[3];
                sba_colnorm(D, pt, n);
                for (int c = 0; c < 3; c++) D.diag[3 * pt + c] = fmin(fmax(n[c], o.min_lm_diagonal), o.max_lm_diagonal);
            }
            if (!in_program(pt)) { D.y[3 * pt] = D.y[3 * pt + 1] = D.y[3 * pt + 2] = 0; continue; }
            double A[9] = {0}, b[3] = {0, 0, 0};
            for (int s = D.pt_ptr[pt]; s < D.pt_ptr[pt + 1]; s++) {
                const double *J = D.jr + 8 * (size_t)s, *r = J + 6;
                for (int a = 0; a < 3; a++) {
                    b[a] += J[a] * r[0] + J[3 + a] * r[1];
                    for (int c = 0; c < 3; c++) A[3 * a + c] += J[a] * J[c] + J[3 + a] * J[3 + c];
                }
            }
            for (int a = 0; a < 3; a++) { const double d = sqrt(D.diag[3 * pt + a] / radius); A[4 * a] += d * d; }
            double L[9] = {0};
            bool ok = true;
            for (int j = 0; j < 3 && ok; j++) {
                double d = A[4 * j];
                for (int k = 0; k < j; k++) d -= L[3 * j + k] * L[3 * j + k];
                if (!(d > 0.0) || !isfinite(d)) { ok = false; break; }
                L[4 * j] = sqrt(d);
                for (int i = j + 1; i < 3; i++) {
                    double s = A[3 * i + j];
                    for (int k = 0; k < j; k++) s -= L[3 * i + k] * L[3 * j + k];
                    L[3 * i + j] = s / L[4 * j];
                }
            }
            if (ok) {
                for (int i = 0; i < 3; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[3 * i + k] * b[k]; b[i] = s / L[4 * i]; }
                for (int i = 2; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < 3; k++) s -= L[3 * k + i] * b[k]; b[i] = s / L[4 * i]; }
                for (int c = 0; c < 3; c++) { if (!isfinite(b[c])) ok = false; b[c] = -b[c]; }
            }
            if (!ok) { bad = 1; continue; }
            D.y[3 * pt] = b[0]; D.y[3 * pt + 1] = b[1]; D.y[3 * pt + 2] = b[2];
            for (int s = D.pt_ptr[pt]; s < D.pt_ptr[pt + 1]; s++) {
                const double *J = D.jr + 8 * (size_t)s, *r = J + 6;
                const double m0 = J[0] * b[0] + J[1] * b[1] + J[2] * b[2], m1 = J[3] * b[0] + J[4] * b[1] + J[5] * b[2];
                mcc -= m0 * (r[0] + m0 / 2.0) + m1 * (r[1] + m1 / 2.0);
            }
        }
        reuse_diagonal = 1;
        const bool lin_ok = sba_block_max(bad, sh) == 0;
        const double model_cost_change = sba_block_sum(mcc, sh);
        if (!(lin_ok && model_cost_change > 0.0)) {
            if (++num_invalid >= o.max_consecutive_invalid_steps) { term = OV2_TERM_INVALID_STEPS; break; }
            radius = radius / decrease_factor; decrease_factor *= 2.0;
            continue;
        }
        num_invalid = 0;
        double step_sq = 0;
        for (int pt = tid; pt < D.n_pts; pt += nt)
            for (int c = 0; c < 3; c++) {
                const double xv = D.x[3 * pt + c], cv = xv + D.y[3 * pt + c] * D.scale[3 * pt + c];
                D.cand[3 * pt + c] = cv;
                if (in_program(pt)) step_sq += (xv - cv) * (xv - cv);
            }
        const double cand_cost = sba_block_sum(sba_evaluate<false>(D, D.cand), sh);
        step_sq = sba_block_sum(step_sq, sh);
        if (sqrt(step_sq) <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) { term = OV2_TERM_PARAMETER_TOL; break; }
        if (fabs(x_cost - cand_cost) <= o.function_tolerance * x_cost) { term = OV2_TERM_FUNCTION_TOL; break; }
        const double r1 = (ev_cur - cand_cost) / model_cost_change, r2 = (ev_ref - cand_cost) / (ev_acc_ref + model_cost_change);
        const double rel = r1 > r2 ? r1 : r2;
        if (rel > o.min_relative_decrease) {
            double xn = 0;
            for (int pt = tid; pt < D.n_pts; pt += nt)
                for (int c = 0; c < 3; c++) { const double v = D.cand[3 * pt + c]; D.x[3 * pt + c] = v; if (in_program(pt)) xn += v * v; }
            x_norm = sqrt(sba_block_sum(xn, sh));
            x_cost = sba_block_sum(sba_evaluate<true>(D, D.x), sh);          // stores the Jacobians already column-scaled
            gmax = grad_max();
            step_successful = 1;
            const double t3 = 2.0 * rel - 1.0;
            double d = 1.0 - t3 * t3 * t3;
            if (d < 1.0 / 3.0) d = 1.0 / 3.0;
            radius = radius / d;
            if (radius > o.max_radius) radius = o.max_radius;
            decrease_factor = 2.0;
            reuse_diagonal = 0;
            ev_cur = cand_cost; ev_acc_cand += model_cost_change; ev_acc_ref += model_cost_change;
            if (ev_cur < ev_min) { ev_min = ev_cur; ev_nonmono = 0; ev_cand = ev_cur; ev_acc_cand = 0; }
            else { ev_nonmono++; if (ev_cur > ev_cand) { ev_cand = ev_cur; ev_acc_cand = 0; } }
            if (ev_nonmono == 0) { ev_ref = ev_cand; ev_acc_ref = ev_acc_cand; }
        } else {
            radius = radius / decrease_factor; decrease_factor *= 2.0;
        }
    }
    if (tid == 0) {
        D.out->iterations = n_steps; D.out->num_successful_steps = n_success; D.out->termination = term;
        D.out->initial_cost = initial_cost; D.out->final_cost = minimum_cost;
    }
}

extern "C" {

int ov2_structure_ba(ov2_ctx *ctx, const ov2_sba_problem *p, const ov2_ba_options *opt, ov2_sba_result *res)
{
    OV2_REQUIRE(ctx && p && opt && res, OV2_EINVAL, "NULL argument");
    OV2_REQUIRE(p->n_kf > 0 && p->n_pts >= 0 && p->n_res >= 0, OV2_EINVAL, "bad problem size");
    OV2_REQUIRE(p->poses && (p->n_pts == 0 || p->xyz), OV2_EINVAL, "NULL parameter array");
    OV2_REQUIRE(p->n_res == 0 || (p->res_type && p->res_kf && p->res_pt && p->res_uv && p->res_sigma), OV2_EINVAL, "NULL residual array");
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    const int n_kf = p->n_kf, n_pts = p->n_pts, n_res = p->n_res;
    // host: validate, build the point -> active residual CSR (ascending residual index inside a point)
    std::vector<int> pt_ptr((size_t)n_pts + 2, 0), pt_res;
    int n_act = 0;
    for (int i = 0; i < n_res; i++) {
        if (p->res_active && !p->res_active[i]) continue;
        OV2_REQUIRE(p->res_pt[i] >= 0 && p->res_pt[i] < n_pts && p->res_kf[i] >= 0 && p->res_kf[i] < n_kf &&
                    (p->res_type[i] == OV2_XYZ_LEFT || p->res_type[i] == OV2_XYZ_RIGHT), OV2_EINVAL, "residual block out of range");
        pt_ptr[p->res_pt[i] + 1]++; n_act++;
    }
    for (int l = 0; l < n_pts; l++) pt_ptr[l + 1] += pt_ptr[l];
    pt_res.resize((size_t)n_act + 1);
    {
        std::vector<int> fill(pt_ptr.begin(), pt_ptr.end());
        for (int i = 0; i < n_res; i++) if (!p->res_active || p->res_active[i]) pt_res[fill[p->res_pt[i]]++] = i;
    }
    // one device arena, one H2D
    auto up8 = [](size_t v) { return (v + 7) & ~(size_t)7; };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += up8(bytes); return o; };
    const size_t N3 = 3 * (size_t)n_pts + 1, NR = (size_t)n_res + 1;
    const size_t o_poses = take(56 * (size_t)n_kf), o_x = take(8 * N3), o_uv = take(16 * NR), o_sig = take(8 * NR), o_kf = take(4 * NR);
    const size_t o_type = take(NR), o_ptr = take(4 * ((size_t)n_pts + 2)), o_pres = take(4 * ((size_t)n_act + 1));
    const size_t h2d = off;
    const size_t o_rt = take(96 * (size_t)n_kf), o_cand = take(8 * N3), o_g = take(8 * N3), o_scale = take(8 * N3), o_diag = take(8 * N3), o_y = take(8 * N3);
    const size_t o_jr = take(64 * ((size_t)n_act + 1));
    const size_t o_chi2 = take(8 * NR), o_dpos = take(NR), o_out = take(sizeof(SbaOut)), total = off;
    int rc = ctx->reserve_device(total);  if (rc) return rc;
    rc = ctx->reserve_host(total);        if (rc) return rc;
    uint8_t *hs = (uint8_t *)ctx->h_scratch, *ds = (uint8_t *)ctx->d_scratch;
    memcpy(hs + o_poses, p->poses, 56 * (size_t)n_kf);
    if (n_pts) memcpy(hs + o_x, p->xyz, 24 * (size_t)n_pts);
    if (n_res) {
        memcpy(hs + o_uv, p->res_uv, 16 * (size_t)n_res); memcpy(hs + o_sig, p->res_sigma, 8 * (size_t)n_res);
        memcpy(hs + o_kf, p->res_kf, 4 * (size_t)n_res); memcpy(hs + o_type, p->res_type, (size_t)n_res);
    }
    memcpy(hs + o_ptr, pt_ptr.data(), 4 * ((size_t)n_pts + 1));
    if (n_act) memcpy(hs + o_pres, pt_res.data(), 4 * (size_t)n_act);
    // chi2 / depthpos are in/out: residual blocks that are never evaluated keep the caller's values
    for (int i = 0; i < n_res; i++) {
        ((double *)(hs + o_chi2))[i] = res->chi2_last_eval ? res->chi2_last_eval[i] : 0.0;
        (hs + o_dpos)[i] = res->depthpos_last_eval ? res->depthpos_last_eval[i] : 0;
    }
    OV2_HIP_CHECK(hipMemcpyAsync(ds, hs, h2d, hipMemcpyHostToDevice, ctx->stream));
    OV2_HIP_CHECK(hipMemcpyAsync(ds + o_chi2, hs + o_chi2, o_out - o_chi2, hipMemcpyHostToDevice, ctx->stream));
    SbaDev D;
    D.n_kf = n_kf; D.n_pts = n_pts; D.n_res = n_res;
    D.poses = (const double *)(ds + o_poses); D.kf_rt = (double *)(ds + o_rt);
    D.pt_ptr = (const int *)(ds + o_ptr); D.pt_res = (const int *)(ds + o_pres);
    D.res_type = ds + o_type; D.res_kf = (const int *)(ds + o_kf); D.res_uv = (const double *)(ds + o_uv); D.res_sigma = (const double *)(ds + o_sig);
    D.x = (double *)(ds + o_x); D.cand = (double *)(ds + o_cand); D.g = (double *)(ds + o_g); D.scale = (double *)(ds + o_scale);
    D.diag = (double *)(ds + o_diag); D.y = (double *)(ds + o_y); D.jr = (double *)(ds + o_jr);
    D.chi2 = (double *)(ds + o_chi2); D.dpos = ds + o_dpos; D.out = (SbaOut *)(ds + o_out);
    for (int i = 0; i < 4; i++) { D.calib_l[i] = p->calib_l[i]; D.calib_r[i] = p->calib_r[i]; }
    {
        const double *q = p->T_rl + 3;
        const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        const double x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
        double *R = D.Rrl;
        R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
        R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
        R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
        D.trl[0] = p->T_rl[0]; D.trl[1] = p->T_rl[1]; D.trl[2] = p->T_rl[2];
    }
    D.o = *opt;
    hipEvent_t e0, e1;
    OV2_HIP_CHECK(hipEventCreate(&e0)); OV2_HIP_CHECK(hipEventCreate(&e1));
    OV2_HIP_CHECK(hipEventRecord(e0, ctx->stream));
    hipLaunchKernelGGL(k_structure_ba, dim3(1), dim3(1024), 0, ctx->stream, D);
    OV2_HIP_CHECK(hipGetLastError());
    OV2_HIP_CHECK(hipEventRecord(e1, ctx->stream));
    OV2_HIP_CHECK(hipMemcpyAsync(hs + o_x, ds + o_x, 8 * N3, hipMemcpyDeviceToHost, ctx->stream));
    OV2_HIP_CHECK(hipMemcpyAsync(hs + o_chi2, ds + o_chi2, total - o_chi2, hipMemcpyDeviceToHost, ctx->stream));
    OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    OV2_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (res->xyz_out && n_pts) memcpy(res->xyz_out, hs + o_x, 24 * (size_t)n_pts);
    if (res->chi2_last_eval && n_res) memcpy(res->chi2_last_eval, hs + o_chi2, 8 * (size_t)n_res);
    if (res->depthpos_last_eval && n_res) memcpy(res->depthpos_last_eval, hs + o_dpos, (size_t)n_res);
    const SbaOut *O = (const SbaOut *)(hs + o_out);
    res->iterations = O->iterations; res->num_successful_steps = O->num_successful_steps; res->termination = O->termination;
    res->initial_cost = O->initial_cost; res->final_cost = O->final_cost; res->solve_ms = ms;
    return OV2_OK;
}

} // extern "C"
