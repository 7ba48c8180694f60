// Scalar FP64 model math of the MonoSLAM per-frame path, written for the device.
// Every function is SL2_HD so that the same source is (a) inlined into the HIP
// kernels and (b) compilable by plain g++ for formula-level unit tests on a
// machine without a GPU (tests/test_device_math_host.py).  The host build is a
// TEST vehicle only: the engine never runs these on the CPU.
//
// Operation ORDER follows the reference expression by expression (file:line
// cited per function; paths relative to /root/reference/scenelib2/) so that,
// with FP contraction off, results equal the reference's scalar arithmetic up
// to libm differences in sin/cos/acos.  Build with -ffp-contract=off.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SL2_HD __host__ __device__ __forceinline__
#else
#define SL2_HD inline
#endif

namespace sl2 {

struct CameraParams {  // camera.cpp:49-62 (SetCameraParameters)
  int width, height;
  double fku, fkv, u0, v0, kd1;
  int sd;
};

constexpr double kSdAComponentFilter = 4.0;      // motion_model.cpp:45
constexpr double kSdAlphaComponentFilter = 6.0;  // motion_model.cpp:45
constexpr double kNoSigma = 3.0;                 // monoslam.cpp:48
constexpr double kCorrThresh2 = 0.40;            // monoslam.cpp:48
constexpr double kCorrelationSigmaThreshold = 10.0;  // monoslam.cpp:49
constexpr int kBoxSize = 11;                         // monoslam.cpp:48
constexpr double kMaximumLengthRatio = 2.0;          // full_feature_model.cpp:49
constexpr double kImageSearchBoundary = 20.0;        // full_feature_model.cpp:51
constexpr double kPi = 3.14159265358979323846;

// Eigen::Quaterniond::toRotationMatrix (no normalisation, Q11). q = (w,x,y,z), R row-major.
SL2_HD void quat_to_rot(const double q[4], double R[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0 - (txx + tyy);
}

// Eigen::Quaterniond::inverse(): conjugate / squared norm.
SL2_HD void quat_inverse(const double q[4], double qi[4]) {
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (n2 > 0.0) { qi[0] = q[0] / n2; qi[1] = -q[1] / n2; qi[2] = -q[2] / n2; qi[3] = -q[3] / n2; }
  else { qi[0] = qi[1] = qi[2] = qi[3] = 0.0; }
}

// ---------------------------------------------------------------------------
// Motion model: f(xv), and the two non-trivial blocks of F = dfv/dxv:
//   A44 = F[3:7,3:7] = dq3_by_dq2(qwt)          (math_util.cpp:99-114)
//   B43 = F[3:7,10:13] = dq3_by_dq1(qold) * D   (math_util.cpp:82-97,
//                                                motion_model.cpp:290-349)
// motion_model.cpp:84-146.  u (acceleration control) is zero (monoslam.cpp:114).
// ---------------------------------------------------------------------------
SL2_HD void motion_f_and_blocks(const double xv[13], double dt, double f[13], double A44[16], double B43[12]) {
  const double qold[4] = {xv[3], xv[4], xv[5], xv[6]};
  const double om[3] = {xv[10], xv[11], xv[12]};
  for (int i = 0; i < 3; ++i) f[i] = xv[i] + xv[7 + i] * dt;
  // qwt = q(omega*dt)  (math_util.cpp:61-80)
  const double av[3] = {om[0] * dt, om[1] * dt, om[2] * dt};
  const double angle = sqrt(av[0] * av[0] + av[1] * av[1] + av[2] * av[2]);
  double qw, qx, qy, qz;
  if (angle > 0.0) {
    const double s = sin(angle / 2.0) / angle;
    const double c = cos(angle / 2.0);
    qx = s * av[0]; qy = s * av[1]; qz = s * av[2]; qw = c;
  } else { qx = qy = qz = 0.0; qw = 1.0; }
  // qnew = qold * qwt (Eigen product)
  const double aw = qold[0], ax = qold[1], ay = qold[2], az = qold[3];
  f[3] = aw * qw - ax * qx - ay * qy - az * qz;
  f[4] = aw * qx + ax * qw + ay * qz - az * qy;
  f[5] = aw * qy + ay * qw + az * qx - ax * qz;
  f[6] = aw * qz + az * qw + ax * qy - ay * qx;
  for (int i = 0; i < 3; ++i) f[7 + i] = xv[7 + i] + 0.0 * dt;  // vnew = vold + u*dt, u = 0
  for (int i = 0; i < 3; ++i) f[10 + i] = om[i];
  // dq3_by_dq2(qwt)
  {
    const double w = qw, x = qx, y = qy, z = qz;
    const double v[16] = {w, -x, -y, -z, x, w, z, -y, y, -z, w, x, z, y, -x, w};
    for (int i = 0; i < 16; ++i) A44[i] = v[i];
  }
  // D = dqomegadt_by_domega(omega, dt)  (4x3), no |omega|==0 guard (Q10)
  double D[12];
  {
    const double omega = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double sn = sin(omega * dt / 2.0), cs = cos(omega * dt / 2.0);
    for (int j = 0; j < 3; ++j) D[0 * 3 + j] = (-dt / 2.0) * (om[j] / omega) * sn;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        if (i == j)
          D[(1 + i) * 3 + j] = (dt / 2.0) * om[i] * om[i] / (omega * omega) * cs +
                               (1.0 / omega) * (1.0 - om[i] * om[i] / (omega * omega)) * sn;
        else
          D[(1 + i) * 3 + j] = (om[i] * om[j] / (omega * omega)) * ((dt / 2.0) * cs - (1.0 / omega) * sn);
      }
  }
  // dq3_by_dq1(qold) * D
  {
    const double w = aw, x = ax, y = ay, z = az;
    const double M[16] = {w, -x, -y, -z, x, w, -z, y, y, z, w, -x, z, -y, x, w};
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 3; ++j) {
        double acc = 0.0;
        for (int k = 0; k < 4; ++k) acc += M[i * 4 + k] * D[k * 3 + j];
        B43[i * 3 + j] = acc;
      }
  }
}

// f applied `steps` times (FindNonOverlappingRegion predicts ten steps ahead, monoslam.cpp:888-893): the velocities do not
// change under f (u = 0), so q(omega dt) - a square root, a sine and a cosine - is the SAME quaternion in every step and is
// formed once; every step then is the position update and the quaternion product, in the expressions of
// motion_f_and_blocks.  Bit-identical to calling that `steps` times (k_map_find's region stage did: ten serial sin / cos on one lane).
SL2_HD void motion_f_repeated(const double xv[13], double dt, int steps, double out[13]) {
  for (int i = 0; i < 13; ++i) out[i] = xv[i];
  const double om[3] = {xv[10], xv[11], xv[12]};
  const double av[3] = {om[0] * dt, om[1] * dt, om[2] * dt};
  const double angle = sqrt(av[0] * av[0] + av[1] * av[1] + av[2] * av[2]);
  double qw, qx, qy, qz;
  if (angle > 0.0) {
    const double s = sin(angle / 2.0) / angle;
    const double c = cos(angle / 2.0);
    qx = s * av[0]; qy = s * av[1]; qz = s * av[2]; qw = c;
  } else { qx = qy = qz = 0.0; qw = 1.0; }
  for (int it = 0; it < steps; ++it) {
    for (int i = 0; i < 3; ++i) out[i] = out[i] + out[7 + i] * dt;
    const double aw = out[3], ax = out[4], ay = out[5], az = out[6];
    out[3] = aw * qw - ax * qx - ay * qy - az * qz;
    out[4] = aw * qx + ax * qw + ay * qz - az * qy;
    out[5] = aw * qy + ay * qw + az * qx - ax * qz;
    out[6] = aw * qz + az * qw + ax * qy - ay * qx;
    for (int i = 0; i < 3; ++i) out[7 + i] = out[7 + i] + 0.0 * dt;
  }
}

// One row of F applied to a 13-vector: sum_k F[i][k] v[k], nonzero terms only,
// in increasing k (== the dense product's rounding, zeros add exactly).
SL2_HD double frow_dot(int i, double dt, const double A44[16], const double B43[12], const double v[13]) {
  if (i < 3) return v[i] + dt * v[7 + i];
  if (i < 7) {
    const int a = i - 3;
    double acc = A44[a * 4 + 0] * v[3];
    acc += A44[a * 4 + 1] * v[4];
    acc += A44[a * 4 + 2] * v[5];
    acc += A44[a * 4 + 3] * v[6];
    acc += B43[a * 3 + 0] * v[10];
    acc += B43[a * 3 + 1] * v[11];
    acc += B43[a * 3 + 2] * v[12];
    return acc;
  }
  return v[i];
}

// Q[i][j] of Q = (G Pnn) G^T (motion_model.cpp:148-217) from the same B43 block.
SL2_HD double process_noise_entry(int i, int j, double dt, const double B43[12]) {
  const double lin = kSdAComponentFilter * kSdAComponentFilter * dt * dt;
  const double ang = kSdAlphaComponentFilter * kSdAlphaComponentFilter * dt * dt;
  // G rows: r (0..2): dt*e_i on noise cols 0..2 ; q (3..6): B43 on noise cols 3..5 ;
  //         v (7..9): e_i on cols 0..2 ; w (10..12): e_i on cols 3..5
  double acc = 0.0;
  for (int k = 0; k < 6; ++k) {
    double gi, gj;
    if (i < 3) gi = (k == i) ? dt : 0.0;
    else if (i < 7) gi = (k >= 3) ? B43[(i - 3) * 3 + (k - 3)] : 0.0;
    else if (i < 10) gi = (k == i - 7) ? 1.0 : 0.0;
    else gi = (k == i - 10 + 3) ? 1.0 : 0.0;
    if (j < 3) gj = (k == j) ? dt : 0.0;
    else if (j < 7) gj = (k >= 3) ? B43[(j - 3) * 3 + (k - 3)] : 0.0;
    else if (j < 10) gj = (k == j - 7) ? 1.0 : 0.0;
    else gj = (k == j - 10 + 3) ? 1.0 : 0.0;
    const double pnn = (k < 3) ? lin : ang;
    acc += (gi * pnn) * gj;
  }
  return acc;
}

// dqnorm_by_dq (motion_model.cpp:351-380): qq is the SQUARED norm (Q9). N row-major 4x4.
SL2_HD void dqnorm_by_dq(const double q[4], double N[16]) {
  const double qq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      N[i * 4 + j] = (i == j) ? (1 - q[i] * q[i] / (qq * qq)) / qq : -q[i] * q[j] / (qq * qq * qq);
}

// ---------------------------------------------------------------------------
// Measurement model of one fully-initialised feature.
//   full_feature_model.cpp:67-101 (zeroedyi + Jacobians), :178-195 (h, dh/dxp, dh/dy)
//   feature_model.cpp:152-238 (dqbar_by_dq, dRq_times_a_by_dq, dR_by_dq*)
//   camera.cpp:90-114 (Project), :183-215 (ProjectionJacobian), :282-300 (noise)
// Outputs: zeroed (3), RRW (9), h (2), Hx (2x7 row-major), Hy (2x3 row-major), R.
// ---------------------------------------------------------------------------
SL2_HD void zeroedyi_only(const double xp[7], const double y[3], double zeroed[3]) {
  const double d[3] = {y[0] - xp[0], y[1] - xp[1], y[2] - xp[2]};
  double qi[4], R[9];
  quat_inverse(&xp[3], qi);
  quat_to_rot(qi, R);
  for (int i = 0; i < 3; ++i) {
    double acc = 0.0;
    for (int k = 0; k < 3; ++k) acc += R[i * 3 + k] * d[k];
    zeroed[i] = acc;
  }
}

SL2_HD void measurement_model(const CameraParams& cam, const double xp[7], const double y[3], double zeroed[3],
                              double h[2], double Hx[14], double Hy[6], double* Rnoise) {
  const double d[3] = {y[0] - xp[0], y[1] - xp[1], y[2] - xp[2]};
  double qRW[4], RRW[9];
  quat_inverse(&xp[3], qRW);
  quat_to_rot(qRW, RRW);
  for (int i = 0; i < 3; ++i) {
    double acc = 0.0;
    for (int k = 0; k < 3; ++k) acc += RRW[i * 3 + k] * d[k];
    zeroed[i] = acc;
  }
  // dzeroedyi_by_dxp (3x7) = [ -RRW | (dR_by_dq{0,x,y,z}(qRW) d) * diag(1,-1,-1,-1) ]
  double dz[21];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) dz[i * 7 + j] = RRW[i * 3 + j] * -1.0;
  {
    const double w = qRW[0], x = qRW[1], yy = qRW[2], z = qRW[3];
    const double t[4][9] = {{2 * w, -2 * z, 2 * yy, 2 * z, 2 * w, -2 * x, -2 * yy, 2 * x, 2 * w},
                            {2 * x, 2 * yy, 2 * z, 2 * yy, -2 * x, -2 * w, 2 * z, 2 * w, -2 * x},
                            {-2 * yy, 2 * x, 2 * w, 2 * x, 2 * yy, 2 * z, -2 * w, 2 * z, -2 * yy},
                            {-2 * z, -2 * w, 2 * x, 2 * w, -2 * z, 2 * yy, 2 * x, 2 * yy, 2 * z}};
    for (int k = 0; k < 4; ++k) {
      const double sgn = (k == 0) ? 1.0 : -1.0;
      for (int i = 0; i < 3; ++i) {
        double acc = 0.0;
        for (int c = 0; c < 3; ++c) acc += t[k][i * 3 + c] * d[c];
        dz[i * 7 + 3 + k] = acc * sgn;
      }
    }
  }
  // Project
  const double ic0 = -cam.fku * zeroed[0] / zeroed[2];
  const double ic1 = -cam.fkv * zeroed[1] / zeroed[2];
  {
    const double radius2 = (ic0 * ic0 + ic1 * ic1);
    const double factor = sqrt(1 + 2 * cam.kd1 * radius2);
    h[0] = ic0 / factor + cam.u0;
    h[1] = ic1 / factor + cam.v0;
  }
  // ProjectionJacobian (uses the point just projected)
  double J[6];
  {
    const double fku_yz = cam.fku / zeroed[2];
    const double fkv_yz = cam.fkv / zeroed[2];
    const double du[6] = {-fku_yz, 0.0, fku_yz * zeroed[0] / zeroed[2], 0.0, -fkv_yz, fkv_yz * zeroed[1] / zeroed[2]};
    double d00 = ic0 * ic0, d01 = ic0 * ic1, d10 = ic1 * ic0, d11 = ic1 * ic1;
    const double radius2 = d00 + d11;
    const double distor = 1 + 2 * cam.kd1 * radius2;
    const double distor1_2 = sqrt(distor);
    const double distor3_2 = distor1_2 * distor;
    const double s = -2 * cam.kd1 / distor3_2;
    d00 *= s; d01 *= s; d10 *= s; d11 *= s;
    d00 += (1 / distor1_2);
    d11 += (1 / distor1_2);
    for (int c = 0; c < 3; ++c) {
      J[0 * 3 + c] = d00 * du[0 * 3 + c] + d01 * du[1 * 3 + c];
      J[1 * 3 + c] = d10 * du[0 * 3 + c] + d11 * du[1 * 3 + c];
    }
  }
  for (int r = 0; r < 2; ++r) {
    for (int c = 0; c < 7; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += J[r * 3 + k] * dz[k * 7 + c];
      Hx[r * 7 + c] = acc;
    }
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += J[r * 3 + k] * RRW[k * 3 + c];
      Hy[r * 3 + c] = acc;
    }
  }
  // MeasurementNoise
  {
    const double dx = h[0] - cam.u0, dy = h[1] - cam.v0;
    const double distance = sqrt(dx * dx + dy * dy);
    const double max_distance = sqrt(cam.u0 * cam.u0 + cam.v0 * cam.v0);
    const double ratio = distance / max_distance;
    const double sd_use = cam.sd * (1.0 + ratio);
    *Rnoise = sd_use * sd_use;
  }
}

// Innovation covariance S_i (feature_model.cpp:99-116).  Pxx7: rows/cols 0..6 of
// Pxx (row-major 7x7, general); Pxy7: rows 0..6 of Pxy (7x3); Pyy 3x3.
// Columns 7..12 of dh_by_dxv are zero (monoslam.cpp:298-300), so the 13-wide
// sums of the reference reduce to these 7-wide sums exactly.
SL2_HD void innovation_cov(const double Hx[14], const double Hy[6], double Rn, const double Pxx7[49],
                           const double Pxy7[21], const double Pyy[9], double S[4]) {
  double M1[4], T[4], M4[4];
  for (int r = 0; r < 2; ++r) {
    double t1[7];
    for (int c = 0; c < 7; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 7; ++k) acc += Hx[r * 7 + k] * Pxx7[k * 7 + c];
      t1[c] = acc;
    }
    for (int s = 0; s < 2; ++s) {
      double acc = 0.0;
      for (int c = 0; c < 7; ++c) acc += t1[c] * Hx[s * 7 + c];
      M1[r * 2 + s] = acc;
    }
    double t2[3];
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 7; ++k) acc += Hx[r * 7 + k] * Pxy7[k * 3 + c];
      t2[c] = acc;
    }
    for (int s = 0; s < 2; ++s) {
      double acc = 0.0;
      for (int c = 0; c < 3; ++c) acc += t2[c] * Hy[s * 3 + c];
      T[r * 2 + s] = acc;
    }
    double t3[3];
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += Hy[r * 3 + k] * Pyy[k * 3 + c];
      t3[c] = acc;
    }
    for (int s = 0; s < 2; ++s) {
      double acc = 0.0;
      for (int c = 0; c < 3; ++c) acc += t3[c] * Hy[s * 3 + c];
      M4[r * 2 + s] = acc;
    }
  }
  for (int r = 0; r < 2; ++r)
    for (int s = 0; s < 2; ++s) {
      double v = 0.0;
      v += M1[r * 2 + s];
      v += T[r * 2 + s];
      v += T[s * 2 + r];
      v += M4[r * 2 + s];
      S[r * 2 + s] = v;
    }
  S[0] += Rn;
  S[3] += Rn;
}

// visibility_test (full_feature_model.cpp:103-170).  0 = visible.
SL2_HD int visibility_test(const CameraParams& cam, const double xp[7], const double y[3], const double xp_orig[7],
                           const double h[2]) {
  int cant_see = 0;
  if (h[0] < 0.0 + kImageSearchBoundary || h[0] > (double)(cam.width - 1 - kImageSearchBoundary)) cant_see |= 1;
  if (h[1] < 0.0 + kImageSearchBoundary || h[1] > (double)(cam.height - 1 - kImageSearchBoundary)) cant_see |= 2;
  double z[3], R[9], a[3], b[3];
  zeroedyi_only(xp, y, z);
  if (z[2] <= 0) cant_see |= 16;  // kBehindCameraFail_ (full_feature_model.h:74-78)
  quat_to_rot(&xp[3], R);
  for (int i = 0; i < 3; ++i) { double acc = 0.0; for (int k = 0; k < 3; ++k) acc += R[i * 3 + k] * z[k]; a[i] = acc; }
  zeroedyi_only(xp_orig, y, z);
  quat_to_rot(&xp_orig[3], R);
  for (int i = 0; i < 3; ++i) { double acc = 0.0; for (int k = 0; k < 3; ++k) acc += R[i * 3 + k] * z[k]; b[i] = acc; }
  const double mod = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  const double mod_o = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
  const double length_ratio = mod / mod_o;
  if (length_ratio > kMaximumLengthRatio || length_ratio < (1.0 / kMaximumLengthRatio)) cant_see |= 4;   // kDistanceFail_
  const double dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  double angle = acos(dot / (mod * mod_o));
  angle = (angle >= 0.0 ? angle : -angle);
  if (angle > kPi * 45.0 / 180.0) cant_see |= 8;  // kAngleFail_
  return cant_see;
}

// S^-1 of the 2x2 innovation covariance as the reference forms it (monoslam.cpp:371-374): LLT of S (reads S00, S10, S11:
// Eigen::LLT reads the lower triangle), the factor copied into a dense MatrixXd, `.inverse()` of THAT - which for a
// dynamic-size matrix is Eigen's general inverse: LU with partial (row) pivoting, a unit-lower forward solve and an upper
// backward solve (scaled by the reciprocal of the diagonal) against the permuted identity - and S^-1 = X^T X.  Written out
// for the 2x2 factor [[p, 0], [q, r]]; oracle/dense.hpp (general_inverse) is the n x n statement of the same steps and
// tests/test_device_math_host.py holds the two equal bit for bit.  Returns (a, b, c) = (Sinv00, Sinv01, Sinv11).
SL2_HD void sinv_from_S(const double S[4], double* a, double* b, double* c) {
  const double p = sqrt(S[0]);
  const double q = S[2] / p;
  const double r = sqrt(S[3] - q * q);
  double X00, X01, X10, X11;
  if (fabs(q) > fabs(p)) {  // pivot row 1: LU of [[q, r], [p, 0]]
    const double l = p / q;
    const double u11 = 0.0 - l * r;
    const double iu = 1.0 / u11, iq = 1.0 / q;
    X10 = 1.0 * iu;
    X00 = (0.0 - r * X10) * iq;
    X11 = (0.0 - l * 1.0) * iu;
    X01 = (1.0 - r * X11) * iq;
  } else {
    const double l = q / p;
    const double ir = 1.0 / r, ip = 1.0 / p;   // u11 = r - l * 0
    X10 = (0.0 - l * 1.0) * ir;
    X00 = 1.0 * ip;
    X11 = 1.0 * ir;
    X01 = 0.0;
  }
  *a = X00 * X00 + X10 * X10;
  *b = X00 * X01 + X10 * X11;
  *c = X01 * X01 + X11 * X11;
}

// Search window of elliptical_search (monoslam.cpp:416-439).
struct SearchBounds {
  int ucentre, vcentre, urelstart, urelfinish, vrelstart, vrelfinish, halfwidth, halfheight;
};
SL2_HD SearchBounds search_bounds(const double centre[2], double a, double b, double c, int width, int height) {
  SearchBounds sb;
  const int BOXSIZE = kBoxSize;
  sb.halfwidth = (int)(kNoSigma / sqrt(a - b * b / c));
  sb.halfheight = (int)(kNoSigma / sqrt(c - b * b / a));
  sb.ucentre = int(centre[0] + 0.5);
  sb.vcentre = int(centre[1] + 0.5);
  sb.urelstart = -sb.halfwidth; sb.urelfinish = sb.halfwidth;
  sb.vrelstart = -sb.halfheight; sb.vrelfinish = sb.halfheight;
  if (sb.ucentre + sb.urelstart - (BOXSIZE - 1) / 2 < 0) sb.urelstart = (BOXSIZE - 1) / 2 - sb.ucentre;
  if (sb.ucentre + sb.urelfinish - (BOXSIZE - 1) / 2 > width - BOXSIZE)
    sb.urelfinish = width - BOXSIZE - sb.ucentre + (BOXSIZE - 1) / 2;
  if (sb.vcentre + sb.vrelstart - (BOXSIZE - 1) / 2 < 0) sb.vrelstart = (BOXSIZE - 1) / 2 - sb.vcentre;
  if (sb.vcentre + sb.vrelfinish - (BOXSIZE - 1) / 2 > height - BOXSIZE)
    sb.vrelfinish = height - BOXSIZE - sb.vcentre + (BOXSIZE - 1) / 2;
  return sb;
}

// Ellipse membership (monoslam.cpp:453-454): strict '<', C++ left-to-right order.
SL2_HD bool in_ellipse(double a, double b, double c, int urel, int vrel) {
  return a * urel * urel + 2 * b * urel * vrel + c * vrel * vrel < kNoSigma * kNoSigma;
}

// FP64 epilogue of correlate2_warning (improc.cpp:99-133) on the five exact
// integer sums over the 11x11 window (n = 121).
SL2_HD double ncc_score(int Sg0, int Sg1, int Sg0g1, int Sg0sq, int Sg1sq, double* sd0, double* sd1) {
  const double n = 121.0;
  const double Sg0d = Sg0, Sg1d = Sg1, Sg0g1d = Sg0g1, Sg0sqd = Sg0sq, Sg1sqd = Sg1sq;
  const double g0bar = Sg0d / n, g1bar = Sg1d / n;
  const double varg0 = Sg0sqd / n - (g0bar * g0bar);
  const double varg1 = Sg1sqd / n - (g1bar * g1bar);
  const double sigmag0 = sqrt(varg0), sigmag1 = sqrt(varg1);
  *sd0 = sigmag0; *sd1 = sigmag1;
  if (sigmag0 == 0.0) { if (sigmag1 == 0.0) return 0.0; else return 1.0; }
  if (sigmag1 == 0.0) return 1.0;
  const double k = g0bar / sigmag0 - g1bar / sigmag1;
  const double C = Sg0sqd / varg0 + Sg1sqd / varg1 + n * (k * k) - Sg0g1d * 2.0 / (sigmag0 * sigmag1) -
                   Sg0d * 2.0 * k / sigmag0 + Sg1d * 2.0 * k / sigmag1;
  return C / n;
}

}  // namespace sl2
