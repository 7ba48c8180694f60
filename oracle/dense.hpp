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

// C = A * B.  Every C(i,j) is accumulated in increasing k starting from zero, one rounding per multiply and one per add
// (built with -ffp-contract=off: no FMA) - the order of a plain triple loop.  The loop nest is register- and cache-blocked
// (8 rows x 4 columns of C live in registers over the whole k loop, the 8 x K panel of A stays in L1): blocking over i
// and j does not change the order in which any single C(i,j) is summed, so the result is bit-identical to the plain
// loop; it only stops the CPU baseline from being a memory-bandwidth benchmark (2.5 x faster at n = 313).
typedef double dense_v4d __attribute__((vector_size(32), aligned(8)));
inline Mat mul(const Mat& A, const Mat& B) {
  assert(A.c == B.r);
  Mat C(A.r, B.c);
  const int M = A.r, K = A.c, N = B.c;
  const double* a = A.a.data();
  const double* b = B.a.data();
  double* c = C.a.data();
  const int M8 = M & ~7, N4 = N & ~3;
  for (int i = 0; i < M8; i += 8) {
    for (int j = 0; j < N4; j += 4) {
      dense_v4d c00 = {0, 0, 0, 0}, c01 = c00, c10 = c00, c11 = c00, c20 = c00, c21 = c00, c30 = c00, c31 = c00;
      const double* b0 = b + (size_t)j * K;
      const double* b1 = b0 + K;
      const double* b2 = b1 + K;
      const double* b3 = b2 + K;
      const double* ap = a + i;
      for (int k = 0; k < K; ++k, ap += M) {
        const dense_v4d a0 = *(const dense_v4d*)ap, a1 = *(const dense_v4d*)(ap + 4);
        const double s0 = b0[k], s1 = b1[k], s2 = b2[k], s3 = b3[k];
        c00 += a0 * s0; c01 += a1 * s0;
        c10 += a0 * s1; c11 += a1 * s1;
        c20 += a0 * s2; c21 += a1 * s2;
        c30 += a0 * s3; c31 += a1 * s3;
      }
      double* cp = c + i + (size_t)j * M;
      *(dense_v4d*)cp = c00; *(dense_v4d*)(cp + 4) = c01; cp += M;
      *(dense_v4d*)cp = c10; *(dense_v4d*)(cp + 4) = c11; cp += M;
      *(dense_v4d*)cp = c20; *(dense_v4d*)(cp + 4) = c21; cp += M;
      *(dense_v4d*)cp = c30; *(dense_v4d*)(cp + 4) = c31;
    }
    for (int j = N4; j < N; ++j) {
      dense_v4d c0 = {0, 0, 0, 0}, c1 = c0;
      const double* bj = b + (size_t)j * K;
      const double* ap = a + i;
      for (int k = 0; k < K; ++k, ap += M) {
        c0 += *(const dense_v4d*)ap * bj[k];
        c1 += *(const dense_v4d*)(ap + 4) * bj[k];
      }
      *(dense_v4d*)(c + i + (size_t)j * M) = c0;
      *(dense_v4d*)(c + i + 4 + (size_t)j * M) = c1;
    }
  }
  for (int j = 0; j < N; ++j) {            // remaining rows
    const double* bj = b + (size_t)j * K;
    for (int i = M8; i < M; ++i) {
      double acc = 0.0;
      for (int k = 0; k < K; ++k) acc += a[(size_t)i + (size_t)k * M] * bj[k];
      c[(size_t)i + (size_t)j * M] = acc;
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
