// sophus/so3.hpp + sophus/se3.hpp stand-in: the members of Sophus::SO3d / SE3d that the reference's factor code calls, restated
// from the vendored Sophus (/root/reference/Thirdparty/Sophus/sophus/so3.hpp, se3.hpp; cited per function) on top of mini_eigen.hpp.
// TEST INFRASTRUCTURE ONLY -- see mini_eigen.hpp.
#pragma once
#include "../mini_eigen.hpp"

namespace Sophus {

template <class S> struct Constants { static S epsilon() { return S(1e-10); } static S pi() { return S(3.141592653589793238462643383279502884); } };   // common.hpp:109-121

class SO3d {
public:
    typedef Eigen::Vector3d Tangent;
    typedef Eigen::Matrix3d Transformation;
    Eigen::Quaterniond q_;
    SO3d() {}
    // so3.hpp:480-489: takes the quaternion and normalises it
    explicit SO3d(const Eigen::Quaterniond &q) : q_(q) { normalize(); }
    void normalize() { const double len = q_.norm(); for (int i = 0; i < 4; i++) q_.c[i] /= len; }                 // so3.hpp:297-303
    const Eigen::Quaterniond &unit_quaternion() const { return q_; }
    SO3d inverse() const { return SO3d(q_.conjugate()); }                                                             // so3.hpp:229-231
    Transformation matrix() const { return q_.toRotationMatrix(); }                                                   // so3.hpp:310-312
    SO3d operator*(const SO3d &o) const                                                                               // so3.hpp:328-343
    {
        const Eigen::Quaterniond &a = q_, &b = o.q_;
        return SO3d(Eigen::Quaterniond(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                                       a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                                       a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                                       a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x()));
    }
    template <class P> Eigen::Vector3d operator*(const Eigen::Dense<P, 3, 1> &p) const                               // so3.hpp:358-369
    {
        const Eigen::Vector3d pv(p), qv = q_.vec();
        Eigen::Vector3d uv = qv.cross(pv);
        uv += uv;
        return pv + q_.w() * uv + qv.cross(uv);
    }
    static Transformation hat(const Tangent &omega)                                                                   // so3.hpp:673-682
    {
        Transformation Omega;
        Omega << 0., -omega(2), omega(1), omega(2), 0., -omega(0), -omega(1), omega(0), 0.;
        return Omega;
    }
    struct TangentAndTheta { Tangent tangent; double theta; };
    TangentAndTheta logAndTheta() const                                                                               // so3.hpp:247-290
    {
        TangentAndTheta J;
        const double squared_n = q_.vec().squaredNorm(), w = q_.w();
        double two_atan_nbyw_by_n;
        if (squared_n < Constants<double>::epsilon() * Constants<double>::epsilon()) {
            const double squared_w = w * w;
            two_atan_nbyw_by_n = 2. / w - (2.0 / 3.0) * (squared_n) / (w * squared_w);
            J.theta = 2. * squared_n / w;
        } else {
            const double n = std::sqrt(squared_n);
            if (std::fabs(w) < Constants<double>::epsilon()) two_atan_nbyw_by_n = (w > 0. ? 1. : -1.) * Constants<double>::pi() / n;
            else two_atan_nbyw_by_n = 2. * std::atan(n / w) / n;
            J.theta = two_atan_nbyw_by_n * n;
        }
        J.tangent = two_atan_nbyw_by_n * q_.vec();
        return J;
    }
    Tangent log() const { return logAndTheta().tangent; }
    static SO3d expAndTheta(const Tangent &omega, double *theta)                                                      // so3.hpp:585-617
    {
        const double theta_sq = omega.squaredNorm();
        double imag_factor, real_factor;
        if (theta_sq < Constants<double>::epsilon() * Constants<double>::epsilon()) {
            *theta = 0.;
            const double theta_po4 = theta_sq * theta_sq;
            imag_factor = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
            real_factor = 1. - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
        } else {
            *theta = std::sqrt(theta_sq);
            const double half_theta = 0.5 * (*theta);
            imag_factor = std::sin(half_theta) / (*theta);
            real_factor = std::cos(half_theta);
        }
        SO3d q;
        q.q_ = Eigen::Quaterniond(real_factor, imag_factor * omega.x(), imag_factor * omega.y(), imag_factor * omega.z());   // (not re-normalised)
        return q;
    }
    static SO3d exp(const Tangent &omega) { double th; return expAndTheta(omega, &th); }
};

class SE3d {
public:
    typedef Eigen::Matrix<double, 6, 1> Tangent;
    typedef Eigen::Matrix<double, 6, 6> Adjoint;
    SO3d so3_;
    Eigen::Vector3d t_;
    SE3d() {}
    SE3d(const SO3d &so3, const Eigen::Vector3d &t) : so3_(so3), t_(t) {}                                              // se3.hpp:468-471
    template <class Q, class T> SE3d(const Q &q, const Eigen::Dense<T, 3, 1> &t) : so3_(Eigen::Quaterniond(q)), t_(t) {}   // se3.hpp:490-492
    const SO3d &so3() const { return so3_; }
    const Eigen::Vector3d &translation() const { return t_; }
    const Eigen::Quaterniond &unit_quaternion() const { return so3_.unit_quaternion(); }
    Eigen::Matrix3d rotationMatrix() const { return so3_.matrix(); }
    SE3d inverse() const { const SO3d invR = so3_.inverse(); return SE3d(invR, invR * (t_ * -1.)); }                     // se3.hpp:208-211
    SE3d operator*(const SE3d &o) const { return SE3d(so3_ * o.so3_, t_ + so3_ * o.t_); }                                // se3.hpp:308-312
    template <class P> Eigen::Vector3d operator*(const Eigen::Dense<P, 3, 1> &p) const { return so3_ * p + t_; }        // se3.hpp:326-329
    Adjoint Adj() const                                                                                               // se3.hpp:103-111
    {
        const Eigen::Matrix3d R = so3_.matrix();
        Adjoint res;
        res.block<3, 3>(0, 0) = R;
        res.block<3, 3>(3, 3) = R;
        res.block<3, 3>(0, 3) = SO3d::hat(t_) * R;
        res.block<3, 3>(3, 0) = Eigen::Matrix3d::Zero();
        return res;
    }
    Tangent log() const                                                                                               // se3.hpp:223-255
    {
        Tangent upsilon_omega;
        const SO3d::TangentAndTheta ot = so3_.logAndTheta();
        const double theta = ot.theta;
        upsilon_omega.tail<3>() = ot.tangent;
        const Eigen::Matrix3d Omega = SO3d::hat(ot.tangent);
        if (std::fabs(theta) < Constants<double>::epsilon()) {
            const Eigen::Matrix3d V_inv = Eigen::Matrix3d::Identity() - 0.5 * Omega + (1. / 12.) * (Omega * Omega);
            upsilon_omega.head<3>() = V_inv * t_;
        } else {
            const double half_theta = 0.5 * theta;
            const Eigen::Matrix3d V_inv = (Eigen::Matrix3d::Identity() - 0.5 * Omega +
                                           (1. - theta * std::cos(half_theta) / (2. * std::sin(half_theta))) / (theta * theta) * (Omega * Omega));
            upsilon_omega.head<3>() = V_inv * t_;
        }
        return upsilon_omega;
    }
    template <class A> static SE3d exp(const Eigen::Dense<A, 6, 1> &a)                                               // se3.hpp:763-783
    {
        const Eigen::Vector3d omega(a(3), a(4), a(5)), ups(a(0), a(1), a(2));
        double theta;
        const SO3d so3 = SO3d::expAndTheta(omega, &theta);
        const Eigen::Matrix3d Omega = SO3d::hat(omega), Omega_sq = Omega * Omega;
        Eigen::Matrix3d V;
        if (theta < Constants<double>::epsilon()) V = so3.matrix();
        else {
            const double theta_sq = theta * theta;
            V = (Eigen::Matrix3d::Identity() + (1. - std::cos(theta)) / (theta_sq) * Omega + (theta - std::sin(theta)) / (theta_sq * theta) * Omega_sq);
        }
        return SE3d(so3, V * ups);
    }
};

}  // namespace Sophus
