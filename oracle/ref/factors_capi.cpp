// factors_capi.cpp -- C entry points over the REFERENCE'S OWN factor classes (/root/reference/src/ceres_parametrization.cpp compiled from
// where it lies, against the stand-in headers of oracle/ref/standin): tests/test_reference_factors.py compares their residuals and
// Jacobians with oracle/ba.c, oracle/xyz_ba.c and (on the GPU box, where the prebuilt library travels) the device lineariser.
// TEST INFRASTRUCTURE ONLY: nothing under ov2slam_amd/ loads this library.
#include "ceres_parametrization.hpp"

#include <memory>

extern "C" {

// type: 0 ReprojectionErrorKSE3AnchInvDepth {4,7,7,1}        1 ReprojectionErrorRightCamKSE3AnchInvDepth {4,4,7,7,7,1}
//       2 ReprojectionErrorRightAnchCamKSE3AnchInvDepth {4,4,7,1}   3 ReprojectionErrorSE3 {7} (K, world point in the constructor)
//       4 ReprojectionErrorKSE3XYZ {4,7,3}                   5 ReprojectionErrorRightCamKSE3XYZ {4,7,7,3}
// params: the parameter blocks in the factor's order.  jacobians: NULL, or one pointer per block (each NULL or 2 x size doubles,
// row-major, global size like Ceres hands them to CostFunction::Evaluate).  chi2 / depthpos: the mutable members the reference
// reads back after the solve (chi2err_, isdepthpositive_).  Returns 0, or -1 for an unknown type.
int ref_factor_eval(int type, const double *const *params, const double uv[2], const double anch_uv[2], double sigma, const double pnp_K[4],
                    const double pnp_xyz[3], double *residuals, double **jacobians, double *chi2, int *depthpos)
{
    using namespace DirectLeftSE3;
    bool ok = false;
    switch (type) {
    case 0: { ReprojectionErrorKSE3AnchInvDepth f(uv[0], uv[1], anch_uv[0], anch_uv[1], sigma); ok = f.Evaluate(params, residuals, jacobians); *chi2 = f.chi2err_; *depthpos = f.isdepthpositive_; break; }
    case 1: { ReprojectionErrorRightCamKSE3AnchInvDepth f(uv[0], uv[1], anch_uv[0], anch_uv[1], sigma); ok = f.Evaluate(params, residuals, jacobians); *chi2 = f.chi2err_; *depthpos = f.isdepthpositive_; break; }
    case 2: { ReprojectionErrorRightAnchCamKSE3AnchInvDepth f(uv[0], uv[1], anch_uv[0], anch_uv[1], sigma); ok = f.Evaluate(params, residuals, jacobians); *chi2 = f.chi2err_; *depthpos = f.isdepthpositive_; break; }
    case 3: { ReprojectionErrorSE3 f(uv[0], uv[1], pnp_K[0], pnp_K[1], pnp_K[2], pnp_K[3], Eigen::Vector3d(pnp_xyz[0], pnp_xyz[1], pnp_xyz[2]), sigma);
              ok = f.Evaluate(params, residuals, jacobians); *chi2 = f.chi2err_; *depthpos = f.isdepthpositive_; break; }
    case 4: { ReprojectionErrorKSE3XYZ f(uv[0], uv[1], sigma); ok = f.Evaluate(params, residuals, jacobians); *chi2 = f.chi2err_; *depthpos = f.isdepthpositive_; break; }
    case 5: { ReprojectionErrorRightCamKSE3XYZ f(uv[0], uv[1], sigma); ok = f.Evaluate(params, residuals, jacobians); *chi2 = f.chi2err_; *depthpos = f.isdepthpositive_; break; }
    default: return -1;
    }
    return ok ? 0 : -2;
}

// SE3LeftParameterization (include/ceres_parametrization/ceres_parametrization/se3left_parametrization.hpp:39-73)
int ref_se3_plus(const double x[7], const double delta[6], double out[7]) { SE3LeftParameterization p; return p.Plus(x, delta, out) ? 0 : -2; }
int ref_se3_plus_jacobian(const double x[7], double J[42]) { SE3LeftParameterization p; return p.ComputeJacobian(x, J) ? 0 : -2; }
int ref_se3_sizes(int *global_size, int *local_size) { SE3LeftParameterization p; *global_size = p.GlobalSize(); *local_size = p.LocalSize(); return 0; }

// the parameter-block holders (se3_param_block.hpp:33-77, inverse_depth_param_block.hpp:34-76, pointxyz_param_block.hpp): round trips
int ref_pose_block_roundtrip(const double pose_in[7], double values_out[7], double pose_out[7])
{
    Eigen::Map<const Eigen::Vector3d> t(pose_in);
    Eigen::Map<const Eigen::Quaterniond> q(pose_in + 3);
    PoseParametersBlock blk(7, Sophus::SE3d(q, t));
    for (int i = 0; i < 7; i++) values_out[i] = blk.values()[i];
    const Sophus::SE3d T = blk.getPose();
    pose_out[0] = T.translation().x(); pose_out[1] = T.translation().y(); pose_out[2] = T.translation().z();
    pose_out[3] = T.unit_quaternion().x(); pose_out[4] = T.unit_quaternion().y(); pose_out[5] = T.unit_quaternion().z(); pose_out[6] = T.unit_quaternion().w();
    return 0;
}
double ref_invdepth_block(double anch_depth) { InvDepthParametersBlock b(1, 2, anch_depth); return b.getInvDepth(); }

// LeftSE3RelativePoseError (pose-graph factor of the loop closer: outside SURVEY section 8, exported because it compiles with the file)
int ref_relpose_eval(const double Tc0c1[7], double sigma, const double *const *params, double residuals[6], double **jacobians, double *chi2)
{
    Eigen::Map<const Eigen::Vector3d> t(Tc0c1);
    Eigen::Map<const Eigen::Quaterniond> q(Tc0c1 + 3);
    LeftSE3RelativePoseError f(Sophus::SE3d(q, t), sigma);
    const bool ok = f.Evaluate(params, residuals, jacobians);
    *chi2 = f.chi2err_;
    return ok ? 0 : -2;
}

}  // extern "C"
