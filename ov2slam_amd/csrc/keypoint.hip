// keypoint.hip -- per-keypoint undistortion + bearing vector for gfx950.
//
// Replaces the per-keypoint scalar loop of Frame::computeKeypoint (/root/reference/src/frame.cpp:246-254):
//     kp.unpx_ = pcalib_leftcam_->undistortImagePoint(pt)      (camera_calibration.cpp:313-333:
//                cv::undistortPoints / cv::fisheye::undistortPoints with P = K, one point per call)
//     kp.bv_   = (iK_ * (unpx.x, unpx.y, 1)).normalized()
// which the reference runs ~300 times per frame (addKeypoint / updateKeypoint after detection and
// tracking).  One thread per keypoint, fp64 like OpenCV / Eigen, no FMA contraction: the pinhole model
// (rational + tangential + thin-prism, 5 fixed-point iterations) and the bearing are +,-,*,/ and sqrt
// only and match the oracle bit for bit; the fisheye model calls tan(), whose last bit may differ
// between libm implementations (the float output hides it except at rounding boundaries).
#include "keypoint_dev.hpp"

#pragma clang fp contract(off)

__global__ __launch_bounds__(256) void k_compute_keypoints(KpCalib c, const float2 *__restrict__ px, int n,
                                                           float2 *__restrict__ unpx, double *__restrict__ bv, const int *__restrict__ n_dev)
{
    const int il = blockIdx.x * blockDim.x + threadIdx.x;
    const int item = blockIdx.y;                          // lock-step tracker (trackb.hip): n point slots and one count per batch item
    const int slots = n;                                  // point slots per item
    if (n_dev) n = min(n, n_dev[item]);                   // the tracker's graph is launched over its capacity: the count lives on the device
    if (il >= n) return;
    const int i = item * slots + il;
    const float2 p = px[i];
    const float2 u = kp_undistort_image_point(c, p);
    unpx[i] = u;
    if (bv) {
        const double x = (double)u.x, y = (double)u.y;
        double b[3];
#pragma unroll
        for (int r = 0; r < 3; r++) b[r] = (c.iK[3 * r] * x + c.iK[3 * r + 1] * y) + c.iK[3 * r + 2] * 1.;
        const double nrm = sqrt((b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]);
        bv[3 * (long long)i] = b[0] / nrm; bv[3 * (long long)i + 1] = b[1] / nrm; bv[3 * (long long)i + 2] = b[2] / nrm;
    }
}

int ov2_kp_calib(int model, const double K[4], const double *D, int nD, const double iK[9], KpCalib &c)
{
    OV2_REQUIRE(K && iK, OV2_EINVAL, "K / iK == NULL");
    OV2_REQUIRE(model == OV2_CAM_PINHOLE || model == OV2_CAM_FISHEYE, OV2_EINVAL, "unknown camera model");
    OV2_REQUIRE(nD >= 0 && nD <= 14 && (nD == 0 || D), OV2_EINVAL, "bad distortion vector");
    OV2_REQUIRE(model != OV2_CAM_FISHEYE || nD == 0 || nD == 4, OV2_EINVAL, "the fisheye model takes 4 coefficients");
    OV2_REQUIRE(nD == 0 || model == OV2_CAM_FISHEYE || (nD == 4 || nD == 5 || nD == 8 || nD == 12), OV2_EUNSUPPORTED,
                "pinhole distortion vectors of 4, 5, 8 or 12 coefficients (the tilted-sensor terms of the 14-vector are not implemented)");
    memset(&c, 0, sizeof(c));
    c.fx = K[0]; c.fy = K[1]; c.cx = K[2]; c.cy = K[3];
    for (int i = 0; i < nD; i++) c.k[i] = D[i];
    for (int i = 0; i < 9; i++) c.iK[i] = iK[i];
    c.nD = nD; c.model = model;
    return OV2_OK;
}

// launcher for track.hip: Frame::computeKeypoint of the tracker's output positions inside its per-frame enqueue
int ov2_launch_compute_keypoints(hipStream_t s, const KpCalib &c, const float *px_d, int n_max, const int *n_dev, float *unpx_d, double *bv_d, int items)
{
    hipLaunchKernelGGL(k_compute_keypoints, dim3((n_max + 255) / 256, items), dim3(256), 0, s, c, (const float2 *)px_d, n_max, (float2 *)unpx_d, bv_d, n_dev);
    OV2_HIP_CHECK(hipGetLastError());
    return OV2_OK;
}

extern "C" {

int ov2_compute_keypoints_d(ov2_ctx *ctx, int model, const double K[4], const double *D, int nD, const double iK[9],
                            const float *px_xy_d, int n, float *unpx_xy_d, double *bv_xyz_d)
{
    OV2_REQUIRE(ctx, OV2_EINVAL, "ctx == NULL");
    if (n <= 0) return OV2_OK;
    OV2_REQUIRE(px_xy_d && unpx_xy_d, OV2_EINVAL, "NULL point buffer");
    KpCalib c;
    const int rc = ov2_kp_calib(model, K, D, nD, iK, c);
    if (rc != OV2_OK) return rc;
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_compute_keypoints, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, c, (const float2 *)px_xy_d, n,
                       (float2 *)unpx_xy_d, bv_xyz_d, (const int *)nullptr);
    OV2_HIP_CHECK(hipGetLastError());
    return OV2_OK;
}

int ov2_compute_keypoints(ov2_ctx *ctx, int model, const double K[4], const double *D, int nD, const double iK[9],
                          const float *px_xy_h, int n, float *unpx_xy_h, double *bv_xyz_h)
{
    OV2_REQUIRE(ctx, OV2_EINVAL, "ctx == NULL");
    if (n <= 0) return OV2_OK;
    OV2_REQUIRE(px_xy_h && unpx_xy_h, OV2_EINVAL, "NULL point buffer");
    KpCalib c;
    int rc = ov2_kp_calib(model, K, D, nD, iK, c);
    if (rc != OV2_OK) return rc;
    OV2_HIP_CHECK(hipSetDevice(ctx->device));
    // layout: [bv 24n][px 8n][unpx 8n]
    const size_t o_bv = 0, o_px = 24 * (size_t)n, o_un = 32 * (size_t)n, total = 40 * (size_t)n;
    rc = ctx->reserve_device(total);  if (rc) return rc;
    rc = ctx->reserve_host(total);    if (rc) return rc;
    uint8_t *hs = (uint8_t *)ctx->h_scratch, *ds = (uint8_t *)ctx->d_scratch;
    memcpy(hs + o_px, px_xy_h, 8 * (size_t)n);
    OV2_HIP_CHECK(hipMemcpyAsync(ds + o_px, hs + o_px, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_compute_keypoints, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, c, (const float2 *)(ds + o_px), n,
                       (float2 *)(ds + o_un), bv_xyz_h ? (double *)(ds + o_bv) : nullptr, (const int *)nullptr);
    OV2_HIP_CHECK(hipGetLastError());
    OV2_HIP_CHECK(hipMemcpyAsync(hs + o_un, ds + o_un, 8 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (bv_xyz_h) OV2_HIP_CHECK(hipMemcpyAsync(hs + o_bv, ds + o_bv, 24 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    OV2_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    memcpy(unpx_xy_h, hs + o_un, 8 * (size_t)n);
    if (bv_xyz_h) memcpy(bv_xyz_h, hs + o_bv, 24 * (size_t)n);
    return OV2_OK;
}

} // extern "C"
