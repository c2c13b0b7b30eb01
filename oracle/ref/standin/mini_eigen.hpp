// mini_eigen.hpp -- a stand-in for the small part of Eigen 3 that /root/reference/src/ceres_parametrization.cpp and the headers
// under /root/reference/include/ceres_parametrization use (fixed-size dense matrices of doubles, Map, Block, Quaternion).
//
// TEST INFRASTRUCTURE ONLY (oracle/): neither Eigen nor Sophus nor Ceres exist in this image, so the reference's own factor file
// cannot be compiled against them; with this header (and sophus/se3.hpp, ceres/ceres.h beside it) the reference's SOURCE FILE
// compiles unchanged, from where it lies, into oracle/_ref/libref_factors.so -- the checker of the oracle's and the device's
// residuals and Jacobians (tests/test_reference_factors.py).  What this file defines is the meaning of the expressions, not Eigen's
// evaluation order: products are plain triple loops, so agreement is to rounding (1e-12), not bit for bit.
// Everything evaluates eagerly into a Matrix; aliasing is handled by value semantics of the temporaries.
#pragma once
#include <cmath>
#include <type_traits>
#include <cstddef>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace Eigen {

enum { ColMajor = 0, RowMajor = 1 };

template <class S, int R, int C, int Opt = ColMajor> class Matrix;
template <class D, int BR, int BC> class Block;

// ---- CRTP base: anything with rows x cols doubles reachable through at(i, j) ---------------------------------------------------
template <class Derived, int R, int C>
struct Dense {
    enum { Rows = R, Cols = C };
    const Derived &d() const { return *static_cast<const Derived *>(this); }
    Derived &d() { return *static_cast<Derived *>(this); }
    double operator()(int i, int j) const { return d().at(i, j); }
    double &operator()(int i, int j) { return d().at(i, j); }
    double operator()(int i) const { return C == 1 ? d().at(i, 0) : d().at(0, i); }
    double &operator()(int i) { return C == 1 ? d().at(i, 0) : d().at(0, i); }
    double operator[](int i) const { return (*this)(i); }
    double &operator[](int i) { return (*this)(i); }
    double x() const { return (*this)(0); }
    double y() const { return (*this)(1); }
    double z() const { return (*this)(2); }
    double w() const { return (*this)(3); }
    double &x() { return (*this)(0); }
    double &y() { return (*this)(1); }
    double &z() { return (*this)(2); }
    double &w() { return (*this)(3); }
    // a 1 x 1 result is a scalar (float num = r.transpose() * F * l;)
    operator double() const { static_assert(R == 1 && C == 1, "only a 1 x 1 expression converts to a scalar"); return (*this)(0, 0); }
    int rows() const { return R; }
    int cols() const { return C; }
    Matrix<double, R, C> eval() const;
    Derived &noalias() { return d(); }
    Derived &setZero() { for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) d().at(i, j) = 0.; return d(); }
    Derived &setIdentity() { for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) d().at(i, j) = i == j ? 1. : 0.; return d(); }
    template <class O> Derived &assign(const Dense<O, R, C> &o)
    {
        double t[R * C];                                             // (the source may alias the destination)
        for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) t[i * C + j] = o(i, j);
        for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) d().at(i, j) = t[i * C + j];
        return d();
    }
    template <int BR, int BC> Block<Derived, BR, BC> block(int i, int j) { return Block<Derived, BR, BC>(d(), i, j); }
    template <int BR, int BC> Matrix<double, BR, BC> block(int i, int j) const;
    Block<Derived, 3, 3> block(int i, int j, int, int) { return Block<Derived, 3, 3>(d(), i, j); }      // (Sophus' Adj(): 3 x 3 blocks only)
    template <int N> Block<Derived, N, C> topRows() { return Block<Derived, N, C>(d(), 0, 0); }
    template <int N> Block<Derived, N, C> bottomRows() { return Block<Derived, N, C>(d(), R - N, 0); }
    template <int N> Block<Derived, N, 1> head() { return Block<Derived, N, 1>(d(), 0, 0); }
    template <int N> Block<Derived, N, 1> tail() { return Block<Derived, N, 1>(d(), R - N, 0); }
    template <int N> Matrix<double, N, 1> head() const;
    template <int N> Matrix<double, N, 1> tail() const;
    Matrix<double, C, R> transpose() const;
    Matrix<double, R, C> inverse() const;
    double squaredNorm() const { double s = 0; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) s += (*this)(i, j) * (*this)(i, j); return s; }
    double norm() const { return std::sqrt(squaredNorm()); }
    template <class O> double dot(const Dense<O, R, C> &o) const { double s = 0; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) s += (*this)(i, j) * o(i, j); return s; }
    template <class O> Matrix<double, 3, 1> cross(const Dense<O, 3, 1> &o) const;
    template <class O> Derived &operator+=(const Dense<O, R, C> &o) { for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) d().at(i, j) += o(i, j); return d(); }
    template <class O> Derived &operator-=(const Dense<O, R, C> &o) { for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) d().at(i, j) -= o(i, j); return d(); }
    Derived &operator/=(double s) { for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) d().at(i, j) /= s; return d(); }
    Derived &operator*=(double s) { for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) d().at(i, j) *= s; return d(); }
    // comma initialiser: m << a, b, c, ...  fills in row-major order whatever the storage order
    struct Comma {
        Derived &m; int k;
        Comma &operator,(double v) { m.at(k / C, k % C) = v; k++; return *this; }
    };
    Comma operator<<(double v) { d().at(0, 0) = v; return Comma{d(), 1}; }
};

// ---- owning matrix -----------------------------------------------------------------------------------------------------------------
template <int R, int C, int Opt>
class Matrix<double, R, C, Opt> : public Dense<Matrix<double, R, C, Opt>, R, C> {
public:
    typedef Dense<Matrix<double, R, C, Opt>, R, C> Base;
    using Base::operator();
    using Base::operator<<;
    double a[R * C];
    Matrix() { for (int i = 0; i < R * C; i++) a[i] = 0.; }
    Matrix(double x, double y) { static_assert(R * C == 2, "2-vector"); a[0] = x; a[1] = y; }
    Matrix(double x, double y, double z) { static_assert(R * C == 3, "3-vector"); a[0] = x; a[1] = y; a[2] = z; }
    Matrix(double x, double y, double z, double w) { static_assert(R * C == 4, "4-vector"); a[0] = x; a[1] = y; a[2] = z; a[3] = w; }
    template <class O> Matrix(const Dense<O, R, C> &o) { for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) at(i, j) = o(i, j); }
    template <class O> Matrix &operator=(const Dense<O, R, C> &o) { return this->assign(o); }
    double at(int i, int j) const { return (int)Opt == (int)RowMajor ? a[i * C + j] : a[j * R + i]; }
    double &at(int i, int j) { return (int)Opt == (int)RowMajor ? a[i * C + j] : a[j * R + i]; }
    double *data() { return a; }
    const double *data() const { return a; }
    static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
    static Matrix Zero() { return Matrix(); }
    static Matrix Zero(int, int) { return Matrix(); }
};

typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;

// ---- block of another dense object (a reference) ----------------------------------------------------------------------------------
template <class P, int BR, int BC>
class Block : public Dense<Block<P, BR, BC>, BR, BC> {
public:
    typedef Dense<Block<P, BR, BC>, BR, BC> Base;
    using Base::operator();
    P &p; int i0, j0;
    Block(P &p_, int i, int j) : p(p_), i0(i), j0(j) {}
    double at(int i, int j) const { return static_cast<const P &>(p).at(i0 + i, j0 + j); }
    double &at(int i, int j) { return p.at(i0 + i, j0 + j); }
    template <class O> Block &operator=(const Dense<O, BR, BC> &o) { return this->assign(o); }
    Block &operator=(const Block &o) { return this->assign(o); }
};

// ---- Map: a matrix (or quaternion) over caller memory ---------------------------------------------------------------------------
template <class M> struct map_traits;
template <int R, int C, int Opt> struct map_traits<Matrix<double, R, C, Opt>> { enum { Rows = R, Cols = C, Order = Opt }; typedef double *Ptr; };
template <int R, int C, int Opt> struct map_traits<const Matrix<double, R, C, Opt>> { enum { Rows = R, Cols = C, Order = Opt }; typedef const double *Ptr; };

template <class M>
class Map : public Dense<Map<M>, map_traits<M>::Rows, map_traits<M>::Cols> {
public:
    enum { R = map_traits<M>::Rows, C = map_traits<M>::Cols, Opt = map_traits<M>::Order };
    typedef Dense<Map<M>, R, C> Base;
    using Base::operator();
    typename map_traits<M>::Ptr p;
    explicit Map(typename map_traits<M>::Ptr q) : p(q) {}
    Map(typename map_traits<M>::Ptr q, int, int) : p(q) {}
    double at(int i, int j) const { return (int)Opt == (int)RowMajor ? p[i * C + j] : p[j * R + i]; }
    double &at(int i, int j) { return const_cast<double &>((int)Opt == (int)RowMajor ? p[i * C + j] : p[j * R + i]); }
    template <class O> Map &operator=(const Dense<O, R, C> &o) { return this->assign(o); }
    Map &operator=(const Map &o) { return this->assign(o); }
};

// ---- definitions that need Matrix -------------------------------------------------------------------------------------------------
template <class D, int R, int C> Matrix<double, R, C> Dense<D, R, C>::eval() const { return Matrix<double, R, C>(*this); }
template <class D, int R, int C> template <int BR, int BC> Matrix<double, BR, BC> Dense<D, R, C>::block(int i, int j) const
{
    Matrix<double, BR, BC> m;
    for (int a = 0; a < BR; a++) for (int b = 0; b < BC; b++) m(a, b) = (*this)(i + a, j + b);
    return m;
}
template <class D, int R, int C> template <int N> Matrix<double, N, 1> Dense<D, R, C>::head() const { Matrix<double, N, 1> m; for (int i = 0; i < N; i++) m(i) = (*this)(i); return m; }
template <class D, int R, int C> template <int N> Matrix<double, N, 1> Dense<D, R, C>::tail() const { Matrix<double, N, 1> m; for (int i = 0; i < N; i++) m(i) = (*this)(R * C - N + i); return m; }
template <class D, int R, int C> Matrix<double, C, R> Dense<D, R, C>::transpose() const
{
    Matrix<double, C, R> m;
    for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) m(j, i) = (*this)(i, j);
    return m;
}
template <class D, int R, int C> template <class O> Matrix<double, 3, 1> Dense<D, R, C>::cross(const Dense<O, 3, 1> &o) const
{
    return Matrix<double, 3, 1>((*this)(1) * o(2) - (*this)(2) * o(1), (*this)(2) * o(0) - (*this)(0) * o(2), (*this)(0) * o(1) - (*this)(1) * o(0));
}
// inverse: cofactors for 2 x 2 and 3 x 3 (what Eigen does for these sizes), Gauss-Jordan with partial pivoting otherwise
template <class D, int R, int C> Matrix<double, R, C> Dense<D, R, C>::inverse() const
{
    static_assert(R == C, "square");
    Matrix<double, R, C> m;
    if (R == 2) {
        const double det = (*this)(0, 0) * (*this)(1, 1) - (*this)(1, 0) * (*this)(0, 1), id = 1. / det;
        m(0, 0) = (*this)(1, 1) * id; m(0, 1) = -(*this)(0, 1) * id; m(1, 0) = -(*this)(1, 0) * id; m(1, 1) = (*this)(0, 0) * id;
        return m;
    }
    if (R == 3) {
        const Dense &A = *this;
        auto cof = [&](int i, int j) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            return A(i1, j1) * A(i2, j2) - A(i1, j2) * A(i2, j1);
        };
        const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
        const double det = A(0, 0) * c00 + A(1, 0) * c10 + A(2, 0) * c20, id = 1. / det;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m(i, j) = cof(j, i) * id;
        return m;
    }
    double w[R][2 * C];
    for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) { w[i][j] = (*this)(i, j); w[i][C + j] = i == j ? 1. : 0.; }
    for (int k = 0; k < R; k++) {
        int piv = k;
        for (int i = k + 1; i < R; i++) if (std::fabs(w[i][k]) > std::fabs(w[piv][k])) piv = i;
        if (piv != k) for (int j = 0; j < 2 * C; j++) { const double t = w[k][j]; w[k][j] = w[piv][j]; w[piv][j] = t; }
        const double ip = 1. / w[k][k];
        for (int j = 0; j < 2 * C; j++) w[k][j] *= ip;
        for (int i = 0; i < R; i++) if (i != k) { const double f = w[i][k]; if (f != 0.) for (int j = 0; j < 2 * C; j++) w[i][j] -= f * w[k][j]; }
    }
    for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) m(i, j) = w[i][C + j];
    return m;
}

// ---- arithmetic -------------------------------------------------------------------------------------------------------------------
template <class A, class B, int R, int K, int C>
Matrix<double, R, C> operator*(const Dense<A, R, K> &a, const Dense<B, K, C> &b)
{
    Matrix<double, R, C> m;
    for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) { double s = 0; for (int k = 0; k < K; k++) s += a(i, k) * b(k, j); m(i, j) = s; }
    return m;
}
template <class A, int R, int C> Matrix<double, R, C> operator*(double s, const Dense<A, R, C> &a) { Matrix<double, R, C> m; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) m(i, j) = s * a(i, j); return m; }
template <class A, int R, int C> Matrix<double, R, C> operator*(const Dense<A, R, C> &a, double s) { Matrix<double, R, C> m; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) m(i, j) = a(i, j) * s; return m; }
template <class A, int R, int C> Matrix<double, R, C> operator/(const Dense<A, R, C> &a, double s) { Matrix<double, R, C> m; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) m(i, j) = a(i, j) / s; return m; }
template <class A, class B, int R, int C> Matrix<double, R, C> operator+(const Dense<A, R, C> &a, const Dense<B, R, C> &b) { Matrix<double, R, C> m; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) m(i, j) = a(i, j) + b(i, j); return m; }
template <class A, class B, int R, int C> Matrix<double, R, C> operator-(const Dense<A, R, C> &a, const Dense<B, R, C> &b) { Matrix<double, R, C> m; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) m(i, j) = a(i, j) - b(i, j); return m; }
template <class A, int R, int C> Matrix<double, R, C> operator-(const Dense<A, R, C> &a) { Matrix<double, R, C> m; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) m(i, j) = -a(i, j); return m; }

// ---- quaternion, coefficients stored (x, y, z, w) like Eigen ------------------------------------------------------------------
template <class S> class Quaternion;
template <> class Quaternion<double> {
public:
    double c[4];                                                        // x y z w
    Quaternion() { c[0] = c[1] = c[2] = 0.; c[3] = 1.; }
    Quaternion(double w, double x, double y, double z) { c[0] = x; c[1] = y; c[2] = z; c[3] = w; }
    template <class QM> Quaternion(const QM &m, decltype(&QM::is_quaternion_map) = nullptr) { for (int i = 0; i < 4; i++) c[i] = m.p[i]; }
    double x() const { return c[0]; } double y() const { return c[1]; } double z() const { return c[2]; } double w() const { return c[3]; }
    Vector3d vec() const { return Vector3d(c[0], c[1], c[2]); }
    Map<Vector4d> coeffs() { return Map<Vector4d>(c); }
    Vector4d coeffs() const { return Vector4d(c[0], c[1], c[2], c[3]); }
    double squaredNorm() const { return c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]; }
    double norm() const { return std::sqrt(squaredNorm()); }
    Quaternion conjugate() const { return Quaternion(c[3], -c[0], -c[1], -c[2]); }
    // Eigen's QuaternionBase::toRotationMatrix
    Matrix3d toRotationMatrix() const
    {
        Matrix3d res;
        const double tx = 2. * x(), ty = 2. * y(), tz = 2. * z();
        const double twx = tx * w(), twy = ty * w(), twz = tz * w();
        const double txx = tx * x(), txy = ty * x(), txz = tz * x();
        const double tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
        res(0, 0) = 1. - (tyy + tzz); res(0, 1) = txy - twz; res(0, 2) = txz + twy;
        res(1, 0) = txy + twz; res(1, 1) = 1. - (txx + tzz); res(1, 2) = tyz - twx;
        res(2, 0) = txz - twy; res(2, 1) = tyz + twx; res(2, 2) = 1. - (txx + tyy);
        return res;
    }
};
typedef Quaternion<double> Quaterniond;

template <> class Map<const Quaterniond> {
public:
    static const int is_quaternion_map = 1;
    const double *p;
    explicit Map(const double *q) : p(q) {}
    double x() const { return p[0]; } double y() const { return p[1]; } double z() const { return p[2]; } double w() const { return p[3]; }
};
template <> class Map<Quaterniond> {
public:
    static const int is_quaternion_map = 1;
    double *p;
    explicit Map(double *q) : p(q) {}
    double x() const { return p[0]; } double y() const { return p[1]; } double z() const { return p[2]; } double w() const { return p[3]; }
    Map &operator=(const Quaterniond &q) { for (int i = 0; i < 4; i++) p[i] = q.c[i]; return *this; }
};

}  // namespace Eigen
