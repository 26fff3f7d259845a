// eigen_shim.hpp — TEST INFRASTRUCTURE ONLY (oracle/ref_frontend): the subset of the Eigen 3 interface that the reference's
// DecompUtil headers and jps3d sources use, so that those sources compile UNTOUCHED where they lie under /root/reference although
// Eigen itself is absent from this image.  Nothing under faster_amd/ includes this file.
//
// What it is: small dense matrices (fixed sizes up to 6, dynamic rows) with the elementary operations the reference calls —
// norms, dot/cross products, 2x2/3x3 inverses, products, comma initialisers, row views, quaternion -> rotation.  What it is not:
// an algorithm of the reference.  Which points are kept, which point is closest, where a plane goes, which node is expanded next
// is all decided by the reference's own code; this file only supplies the arithmetic underneath (formulas as Eigen documents
// them: cofactor inverse, |v| = sqrt(sum of squares), Hamilton product, the 2x/2y/2z rotation-matrix expansion), so results
// agree with a real Eigen build to rounding, not bit for bit.
#pragma once
#include <algorithm>  // (Eigen brings <algorithm> in transitively; geometric_utils.h relies on it for std::sort)
#include <array>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <iostream>
#include <memory>
#include <type_traits>
#include <vector>

namespace Eigen {

constexpr int Dynamic = -1;
constexpr int Infinity = -1;
enum TransformTraits { Isometry = 1, Affine = 2 };
template <class T>
using aligned_allocator = std::allocator<T>;

template <typename T>
class Quaternion;

namespace shim_detail {
template <typename T, int R, int C, bool Dyn = (R == Dynamic || C == Dynamic)>
struct Storage;
template <typename T, int R, int C>
struct Storage<T, R, C, false> {
  std::array<T, (size_t)(R * C)> v{};
  int rows() const { return R; }
  int cols() const { return C; }
  void resize(int r, int c) { assert(r == R && c == C); (void)r; (void)c; }
  T* data() { return v.data(); }
  const T* data() const { return v.data(); }
};
template <typename T, int R, int C>
struct Storage<T, R, C, true> {
  std::vector<T> v;
  int r_ = (R == Dynamic ? 0 : R), c_ = (C == Dynamic ? 0 : C);
  int rows() const { return r_; }
  int cols() const { return c_; }
  void resize(int r, int c) { r_ = r; c_ = c; v.assign((size_t)r * (size_t)c, T()); }
  T* data() { return v.data(); }
  const T* data() const { return v.data(); }
};
}  // namespace shim_detail

// Row-major storage internally (the interface used by the reference never exposes the storage order).
template <typename T, int R, int C>
class Matrix {
public:
  typedef T Scalar;
  static constexpr bool IsVector = (C == 1 || R == 1);
  Matrix() {}
  // (rows, cols) for matrices with a dynamic dimension; two coefficients for fixed 2-vectors
  template <typename A, typename B, typename = typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>::type>
  Matrix(A a, B b) {
    if (R == Dynamic || C == Dynamic) s_.resize(R == Dynamic ? (int)a : R, C == Dynamic ? (int)b : C);
    else { (*this)(0) = (T)a; if (size() > 1) (*this)(1) = (T)b; }
  }
  // (size) for dynamic vectors
  template <typename A, typename = typename std::enable_if<std::is_integral<A>::value>::type>
  explicit Matrix(A n) {
    if (R == Dynamic && C != Dynamic) s_.resize((int)n, C);
    else if (C == Dynamic && R != Dynamic) s_.resize(R, (int)n);
    else if (size() > 0) (*this)(0) = (T)n;
  }
  Matrix(T a, T b, T c) { set3(a, b, c); }
  Matrix(T a, T b, T c, T d) { (*this)(0) = a; (*this)(1) = b; (*this)(2) = c; (*this)(3) = d; }
  explicit Matrix(const Quaternion<T>& q);
  template <int R2, int C2>
  Matrix(const Matrix<T, R2, C2>& o) { assign(o); }
  template <int R2, int C2>
  Matrix& operator=(const Matrix<T, R2, C2>& o) { assign(o); return *this; }

  int rows() const { return s_.rows(); }
  int cols() const { return s_.cols(); }
  int size() const { return rows() * cols(); }
  T* data() { return s_.data(); }
  const T* data() const { return s_.data(); }
  T& operator()(int i, int j) { return s_.data()[i * cols() + j]; }
  const T& operator()(int i, int j) const { return s_.data()[i * cols() + j]; }
  T& operator()(int i) { return s_.data()[i]; }
  const T& operator()(int i) const { return s_.data()[i]; }
  T& operator[](int i) { return s_.data()[i]; }
  const T& operator[](int i) const { return s_.data()[i]; }
  T& x() { return (*this)(0); }
  T& y() { return (*this)(1); }
  T& z() { return (*this)(2); }
  const T& x() const { return (*this)(0); }
  const T& y() const { return (*this)(1); }
  const T& z() const { return (*this)(2); }
  void resize(int r, int c) { s_.resize(r, c); }

  static Matrix Zero() { Matrix m; m.fill(T(0)); return m; }
  static Matrix Zero(int n) { Matrix m(n); m.fill(T(0)); return m; }
  static Matrix Constant(T v) { Matrix m; m.fill(v); return m; }
  static Matrix Identity() {
    Matrix m;
    m.fill(T(0));
    for (int i = 0; i < m.rows() && i < m.cols(); i++) m(i, i) = T(1);
    return m;
  }
  static Matrix UnitX() { Matrix m = Zero(); m(0) = T(1); return m; }
  static Matrix UnitY() { Matrix m = Zero(); m(1) = T(1); return m; }
  static Matrix UnitZ() { Matrix m = Zero(); m(2) = T(1); return m; }
  void fill(T v) { for (int i = 0; i < size(); i++) (*this)(i) = v; }

  // ---- comma initialiser: m << a, b, c; ----
  struct Comma {
    Matrix& m;
    int k;
    Comma& operator,(T v) { m(k++) = v; return *this; }
  };
  Comma operator<<(T v) { (*this)(0) = v; return Comma{*this, 1}; }

  // ---- row view: A.row(i) = v; ----
  struct RowRef {
    Matrix& m;
    int i;
    template <int R2, int C2>
    RowRef& operator=(const Matrix<T, R2, C2>& v) {
      assert(v.size() == m.cols());
      for (int j = 0; j < m.cols(); j++) m(i, j) = v(j);
      return *this;
    }
    operator Matrix<T, 1, C>() const {
      Matrix<T, 1, C> r;
      if (C == Dynamic) r.resize(1, m.cols());
      for (int j = 0; j < m.cols(); j++) r(j) = m(i, j);
      return r;
    }
  };
  RowRef row(int i) { return RowRef{*this, i}; }
  template <int N>
  Matrix<T, N, C> topRows() const {
    Matrix<T, N, C> r;
    if (C == Dynamic) r.resize(N, cols());
    for (int i = 0; i < N; i++)
      for (int j = 0; j < cols(); j++) r(i, j) = (*this)(i, j);
    return r;
  }
  void conservativeResize(int r, int c) {
    Matrix old = *this;
    s_.resize(r, c);
    for (int i = 0; i < r && i < old.rows(); i++)
      for (int j = 0; j < c && j < old.cols(); j++) (*this)(i, j) = old(i, j);
  }

  // ---- element-wise arithmetic ----
  Matrix operator-() const { Matrix r = *this; for (int i = 0; i < size(); i++) r(i) = -r(i); return r; }
  Matrix& operator+=(const Matrix& o) { for (int i = 0; i < size(); i++) (*this)(i) += o(i); return *this; }
  Matrix& operator-=(const Matrix& o) { for (int i = 0; i < size(); i++) (*this)(i) -= o(i); return *this; }
  Matrix& operator*=(T s) { for (int i = 0; i < size(); i++) (*this)(i) *= s; return *this; }
  Matrix& operator/=(T s) { for (int i = 0; i < size(); i++) (*this)(i) /= s; return *this; }
  friend Matrix operator+(Matrix a, const Matrix& b) { a += b; return a; }
  friend Matrix operator-(Matrix a, const Matrix& b) { a -= b; return a; }
  template <typename S, typename = typename std::enable_if<std::is_arithmetic<S>::value>::type>
  friend Matrix operator*(Matrix a, S s) { a *= (T)s; return a; }
  template <typename S, typename = typename std::enable_if<std::is_arithmetic<S>::value>::type>
  friend Matrix operator*(S s, Matrix a) { a *= (T)s; return a; }
  template <typename S, typename = typename std::enable_if<std::is_arithmetic<S>::value>::type>
  friend Matrix operator/(Matrix a, S s) { a /= (T)s; return a; }
  bool operator==(const Matrix& o) const {
    if (rows() != o.rows() || cols() != o.cols()) return false;
    for (int i = 0; i < size(); i++) if (!((*this)(i) == o(i))) return false;
    return true;
  }
  bool operator!=(const Matrix& o) const { return !(*this == o); }

  // ---- reductions ----
  T dot(const Matrix& o) const { T s = T(0); for (int i = 0; i < size(); i++) s += (*this)(i) * o(i); return s; }
  T squaredNorm() const { return dot(*this); }
  T norm() const { return std::sqrt(squaredNorm()); }
  Matrix normalized() const {
    const T z = squaredNorm();
    return z > T(0) ? Matrix(*this / std::sqrt(z)) : *this;
  }
  void normalize() { *this = normalized(); }
  template <int P>
  T lpNorm() const {
    static_assert(P == Infinity, "only the infinity norm is provided");
    T m = T(0);
    for (int i = 0; i < size(); i++) m = std::max(m, (T)std::abs((*this)(i)));
    return m;
  }
  bool isApprox(const Matrix& o, T prec = T(1e-12)) const {
    T d = T(0);
    for (int i = 0; i < size(); i++) d += ((*this)(i) - o(i)) * ((*this)(i) - o(i));
    return d <= prec * prec * std::min(squaredNorm(), o.squaredNorm());
  }
  Matrix cross(const Matrix& o) const {
    Matrix r;
    r(0) = (*this)(1) * o(2) - (*this)(2) * o(1);
    r(1) = (*this)(2) * o(0) - (*this)(0) * o(2);
    r(2) = (*this)(0) * o(1) - (*this)(1) * o(0);
    return r;
  }
  template <typename U>
  Matrix<U, R, C> cast() const {
    Matrix<U, R, C> r;
    if (R == Dynamic || C == Dynamic) r.resize(rows(), cols());
    for (int i = 0; i < size(); i++) r(i) = (U)(*this)(i);
    return r;
  }

  // ---- matrix algebra ----
  Matrix<T, C, R> transpose() const {
    Matrix<T, C, R> r;
    if (R == Dynamic || C == Dynamic) r.resize(cols(), rows());
    for (int i = 0; i < rows(); i++)
      for (int j = 0; j < cols(); j++) r(j, i) = (*this)(i, j);
    return r;
  }
  T determinant() const {
    const Matrix& m = *this;
    if (rows() == 2) return m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0);
    assert(rows() == 3 && cols() == 3);
    return m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
           m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
  }
  Matrix inverse() const {  // cofactors times 1 / det
    const Matrix& m = *this;
    Matrix r;
    if (rows() == 2) {
      const T id = T(1) / determinant();
      r(0, 0) = m(1, 1) * id; r(0, 1) = -m(0, 1) * id; r(1, 0) = -m(1, 0) * id; r(1, 1) = m(0, 0) * id;
      return r;
    }
    assert(rows() == 3 && cols() == 3);
    const T c00 = m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1), c01 = m(1, 2) * m(2, 0) - m(1, 0) * m(2, 2), c02 = m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0);
    const T id = T(1) / (m(0, 0) * c00 + m(0, 1) * c01 + m(0, 2) * c02);
    r(0, 0) = c00 * id; r(1, 0) = c01 * id; r(2, 0) = c02 * id;
    r(0, 1) = (m(0, 2) * m(2, 1) - m(0, 1) * m(2, 2)) * id;
    r(1, 1) = (m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0)) * id;
    r(2, 1) = (m(0, 1) * m(2, 0) - m(0, 0) * m(2, 1)) * id;
    r(0, 2) = (m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1)) * id;
    r(1, 2) = (m(0, 2) * m(1, 0) - m(0, 0) * m(1, 2)) * id;
    r(2, 2) = (m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0)) * id;
    return r;
  }

private:
  void set3(T a, T b, T c) { (*this)(0) = a; (*this)(1) = b; (*this)(2) = c; }
  template <int R2, int C2>
  void assign(const Matrix<T, R2, C2>& o) {
    if (R == Dynamic || C == Dynamic) {
      // vector <- vector of the other orientation is allowed (Eigen transposes implicitly)
      if (IsVector && o.cols() != 1 && C == 1) s_.resize(o.size(), 1);
      else s_.resize(o.rows(), o.cols());
    }
    assert(size() == o.size());
    if (rows() == o.rows())
      for (int i = 0; i < size(); i++) (*this)(i) = o(i);
    else
      for (int i = 0; i < size(); i++) (*this)(i) = o(i);  // (vector orientation change: same linear order)
  }
  shim_detail::Storage<T, R, C> s_;
};

template <typename T, int R, int K, int C>
Matrix<T, R, C> operator*(const Matrix<T, R, K>& a, const Matrix<T, K, C>& b) {
  Matrix<T, R, C> r;
  if (R == Dynamic || C == Dynamic) r.resize(a.rows(), b.cols());
  assert(a.cols() == b.rows());
  for (int i = 0; i < a.rows(); i++)
    for (int j = 0; j < b.cols(); j++) {
      T s = T(0);
      for (int k = 0; k < a.cols(); k++) s += a(i, k) * b(k, j);
      r(i, j) = s;
    }
  return r;
}

template <typename T, int R, int C>
std::ostream& operator<<(std::ostream& os, const Matrix<T, R, C>& m) {
  for (int i = 0; i < m.rows(); i++) {
    for (int j = 0; j < m.cols(); j++) os << (j ? " " : "") << m(i, j);
    if (i + 1 < m.rows()) os << "\n";
  }
  return os;
}

template <typename T>
class Quaternion {
public:
  Quaternion() : w_(1), x_(0), y_(0), z_(0) {}
  Quaternion(T w, T x, T y, T z) : w_(w), x_(x), y_(y), z_(z) {}
  T w() const { return w_; }
  T x() const { return x_; }
  T y() const { return y_; }
  T z() const { return z_; }
  Quaternion operator*(const Quaternion& b) const {  // Hamilton product
    const Quaternion& a = *this;
    return Quaternion(a.w_ * b.w_ - a.x_ * b.x_ - a.y_ * b.y_ - a.z_ * b.z_, a.w_ * b.x_ + a.x_ * b.w_ + a.y_ * b.z_ - a.z_ * b.y_,
                      a.w_ * b.y_ + a.y_ * b.w_ + a.z_ * b.x_ - a.x_ * b.z_, a.w_ * b.z_ + a.z_ * b.w_ + a.x_ * b.y_ - a.y_ * b.x_);
  }
  Matrix<T, 3, 3> toRotationMatrix() const {
    Matrix<T, 3, 3> res;
    const T tx = T(2) * x_, ty = T(2) * y_, tz = T(2) * z_;
    const T twx = tx * w_, twy = ty * w_, twz = tz * w_;
    const T txx = tx * x_, txy = ty * x_, txz = tz * x_;
    const T tyy = ty * y_, tyz = tz * y_, tzz = tz * z_;
    res(0, 0) = T(1) - (tyy + tzz); res(0, 1) = txy - twz; res(0, 2) = txz + twy;
    res(1, 0) = txy + twz; res(1, 1) = T(1) - (txx + tzz); res(1, 2) = tyz - twx;
    res(2, 0) = txz - twy; res(2, 1) = tyz + twx; res(2, 2) = T(1) - (txx + tyy);
    return res;
  }
  static Quaternion FromTwoVectors(const Matrix<T, 3, 1>& a, const Matrix<T, 3, 1>& b) {  // (visualisation helpers only)
    const Matrix<T, 3, 1> v0 = a.normalized(), v1 = b.normalized();
    const T c = v1.dot(v0);
    if (c < T(-1) + T(1e-12)) return Quaternion(0, 1, 0, 0);
    const Matrix<T, 3, 1> axis = v0.cross(v1);
    const T s = std::sqrt((T(1) + c) * T(2)), invs = T(1) / s;
    return Quaternion(s * T(0.5), axis(0) * invs, axis(1) * invs, axis(2) * invs);
  }

private:
  T w_, x_, y_, z_;
};
template <typename T, int R, int C>
Matrix<T, R, C>::Matrix(const Quaternion<T>& q) { *this = q.toRotationMatrix(); }
// matrix * rotation: the rotation acts as its matrix (Eigen's RotationBase)
template <typename T>
Matrix<T, 3, 3> operator*(const Matrix<T, 3, 3>& m, const Quaternion<T>& q) { return m * q.toRotationMatrix(); }

template <typename T, int Dim, int Mode>
class Transform {};  // (typedefs in the reference's data_type.h only)
template <typename M>
class SelfAdjointEigenSolver {  // (named by a function template that is never instantiated on the path)
public:
  explicit SelfAdjointEigenSolver(const M&) {}
  M eigenvalues() const { return M(); }
};

typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<int, 3, 1> Vector3i;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, Dynamic, 1> VectorXd;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;
typedef Quaternion<double> Quaterniond;

}  // namespace Eigen
