// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Minimal FP64 dense-matrix helper standing in for the Eigen3 types the
// reference uses (Eigen::MatrixXd / VectorXd / Quaterniond).  Eigen is a
// third-party dependency of the reference that is absent from /root/reference
// and from this image; version unpinned by the reference
// (CMakeModules/FindEigen3.cmake:19-27, minimum 2.91.0).  Only the documented
// semantics of the call sites are restated here (column-major storage, plain
// triple-loop products, Eigen's quaternion product / inverse() /
// toRotationMatrix() formulas, lower Cholesky LLT).  Eigen's internal
// summation order is not reproducible, hence FP64 tolerances on EKF outputs.
#pragma once
#include <cassert>
#include <cmath>
#include <cstring>
#include <vector>

namespace oracle {

struct Mat {
  int r = 0, c = 0;
  std::vector<double> a;  // column-major, like Eigen's default
  Mat() {}
  Mat(int rows, int cols) : r(rows), c(cols), a((size_t)rows * cols, 0.0) {}
  inline double& operator()(int i, int j) { return a[(size_t)i + (size_t)j * r]; }
  inline double operator()(int i, int j) const { return a[(size_t)i + (size_t)j * r]; }
  inline double& operator()(int i) { return a[i]; }
  inline double operator()(int i) const { return a[i]; }
  void resize(int rows, int cols) { r = rows; c = cols; a.assign((size_t)rows * cols, 0.0); }
  void setZero() { std::fill(a.begin(), a.end(), 0.0); }
  void setIdentity() {
    setZero();
    for (int i = 0; i < (r < c ? r : c); ++i) (*this)(i, i) = 1.0;
  }
  int size() const { return r * c; }
};

typedef Mat Vec;  // column vector: c == 1

inline Vec make_vec(int n) { return Mat(n, 1); }

// C = A * B.  j-k-i loop order: the inner loop runs down a column of A and C
// (contiguous), which g++ -O3 vectorises without reassociating any sum: every
// C(i,j) is accumulated in increasing k, one rounding per multiply and per add.
inline Mat mul(const Mat& A, const Mat& B) {
  assert(A.c == B.r);
  Mat C(A.r, B.c);
  const int M = A.r, K = A.c, N = B.c;
  for (int j = 0; j < N; ++j) {
    double* cj = &C.a[(size_t)j * M];
    for (int k = 0; k < K; ++k) {
      const double b = B.a[(size_t)k + (size_t)j * K];
      const double* ak = &A.a[(size_t)k * M];
      for (int i = 0; i < M; ++i) cj[i] += ak[i] * b;
    }
  }
  return C;
}

inline Mat transpose(const Mat& A) {
  Mat T(A.c, A.r);
  for (int j = 0; j < A.c; ++j)
    for (int i = 0; i < A.r; ++i) T(j, i) = A(i, j);
  return T;
}

inline Mat add(const Mat& A, const Mat& B) {
  assert(A.r == B.r && A.c == B.c);
  Mat C(A.r, A.c);
  for (size_t i = 0; i < A.a.size(); ++i) C.a[i] = A.a[i] + B.a[i];
  return C;
}

inline Mat sub(const Mat& A, const Mat& B) {
  assert(A.r == B.r && A.c == B.c);
  Mat C(A.r, A.c);
  for (size_t i = 0; i < A.a.size(); ++i) C.a[i] = A.a[i] - B.a[i];
  return C;
}

inline Mat scaled(const Mat& A, double s) {
  Mat C(A.r, A.c);
  for (size_t i = 0; i < A.a.size(); ++i) C.a[i] = A.a[i] * s;
  return C;
}

inline Mat get_block(const Mat& A, int i0, int j0, int rows, int cols) {
  Mat B(rows, cols);
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < rows; ++i) B(i, j) = A(i0 + i, j0 + j);
  return B;
}

inline void set_block(Mat& A, int i0, int j0, const Mat& B) {
  for (int j = 0; j < B.c; ++j)
    for (int i = 0; i < B.r; ++i) A(i0 + i, j0 + j) = B(i, j);
}

inline double trace(const Mat& A) {
  double t = 0.0;
  for (int i = 0; i < A.r; ++i) t += A(i, i);
  return t;
}

// Lower Cholesky factor (Eigen::LLT semantics: only the lower triangle of A is
// read; unblocked left-looking column sweep).  Returns false if a pivot is
// not positive (Eigen would set info()==NumericalIssue and carry on; the
// reference never checks).
inline bool llt_lower(const Mat& A, Mat& L) {
  const int n = A.r;
  L = Mat(n, n);
  bool ok = true;
  for (int k = 0; k < n; ++k) {
    double x = A(k, k);
    for (int p = 0; p < k; ++p) x -= L(k, p) * L(k, p);
    if (!(x > 0.0)) ok = false;
    const double d = std::sqrt(x);
    L(k, k) = d;
    for (int i = k + 1; i < n; ++i) {
      double s = A(i, k);
      for (int p = 0; p < k; ++p) s -= L(i, p) * L(k, p);
      L(i, k) = s / d;
    }
  }
  return ok;
}

// `S_L.inverse()` at kalman.cpp:106, monoslam.cpp:373 and feature_init_info.cpp:61 is called on a plain
// Eigen::MatrixXd (the triangular factor has been copied into a dense matrix first), so Eigen runs its general
// dynamic-size inverse: LU with partial (row) pivoting, then P, L and U solves against the identity.  Restated here
// with the same pivot rule (first row of largest magnitude in the column); a triangular-aware inverse would differ
// from it in the last bits and, on badly conditioned S, in the 11th digit of the update.
inline Mat general_inverse(const Mat& A) {
  const int n = A.r;
  Mat lu = A;
  std::vector<int> row_of(n);
  for (int i = 0; i < n; ++i) row_of[i] = i;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double big = std::fabs(lu(k, k));
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(lu(i, k)) > big) { big = std::fabs(lu(i, k)); piv = i; }
    if (piv != k) {
      for (int j = 0; j < n; ++j) { const double t = lu(k, j); lu(k, j) = lu(piv, j); lu(piv, j) = t; }
      const int t = row_of[k]; row_of[k] = row_of[piv]; row_of[piv] = t;
    }
    if (lu(k, k) != 0.0)
      for (int i = k + 1; i < n; ++i) lu(i, k) /= lu(k, k);
    for (int j = k + 1; j < n; ++j) {
      const double ukj = lu(k, j);
      for (int i = k + 1; i < n; ++i) lu(i, j) -= lu(i, k) * ukj;
    }
  }
  Mat X(n, n);
  for (int j = 0; j < n; ++j) {
    for (int i = 0; i < n; ++i) X(i, j) = (row_of[i] == j) ? 1.0 : 0.0;
    for (int k = 0; k < n; ++k) {           // unit-lower forward sweep
      const double xk = X(k, j);
      if (xk != 0.0)
        for (int i = k + 1; i < n; ++i) X(i, j) -= lu(i, k) * xk;
    }
    for (int k = n - 1; k >= 0; --k) {      // upper backward sweep (Eigen's matrix solver scales by 1 / diagonal)
      X(k, j) *= 1.0 / lu(k, k);
      const double xk = X(k, j);
      if (xk != 0.0)
        for (int i = 0; i < k; ++i) X(i, j) -= lu(i, k) * xk;
    }
  }
  return X;
}

// ---- quaternion (Eigen::Quaterniond semantics) ----
struct Quat {
  double w = 1, x = 0, y = 0, z = 0;
  Quat() {}
  Quat(double w_, double x_, double y_, double z_) : w(w_), x(x_), y(y_), z(z_) {}
};

inline Quat qmul(const Quat& a, const Quat& b) {
  return Quat(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
              a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x);
}

// Eigen: conjugate / squaredNorm (no normalisation assumption).
inline Quat qinverse(const Quat& q) {
  const double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  if (n2 > 0.0) return Quat(q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2);
  return Quat(0, 0, 0, 0);
}

// Eigen::QuaternionBase::toRotationMatrix — does NOT normalise (Q11).
inline Mat qrot(const Quat& q) {
  Mat R(3, 3);
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R(0, 0) = 1.0 - (tyy + tzz);
  R(0, 1) = txy - twz;
  R(0, 2) = txz + twy;
  R(1, 0) = txy + twz;
  R(1, 1) = 1.0 - (txx + tzz);
  R(1, 2) = tyz - twx;
  R(2, 0) = txz - twy;
  R(2, 1) = tyz + twx;
  R(2, 2) = 1.0 - (txx + tyy);
  return R;
}

}  // namespace oracle
