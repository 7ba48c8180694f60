// Scalar FP64 model math of the feature-initialisation path, inlined by
// the mapping kernels (sl2_mapping.hip).  Same discipline as sl2_math.hpp: every expression is
// written in the operation order of the reference so that, without FP contraction, results equal
// the oracle's bit for bit wherever no libm function is involved.
//   part_feature_model.cpp:80-333   PartFeatureModel (ray feature ypi = (r_W, hhat_W), depth lambda)
//   camera.cpp:133-154, 247-275     Unproject, UnprojectionJacobian
//   feature_model.cpp:99-116        func_Si (shared base class)
//   feature_init_info.cpp:57-65     Particle::set_S
//   monoslam.cpp:1466-1481          particle likelihood
#pragma once
#include "sl2_math.hpp"

namespace sl2 {

// POSIX drand48 (srand48(0) at monoslam.cpp:1968; drand48() at :989-990): X' = (a X + c) mod 2^48
constexpr unsigned long long kRand48Seed0 = 0x330EULL;   // srand48(0)
SL2_HD double rand48_next(unsigned long long* state) {
  *state = (0x5DEECE66DULL * *state + 0xBULL) & 0xFFFFFFFFFFFFULL;
  return (double)*state / 281474976710656.0;
}

// dRq_times_a_by_dq (feature_model.cpp:164-194): out (3x4 row-major), column k = dR/dq_k(q) a
SL2_HD void dRq_times_a_by_dq(const double q[4], const double a[3], double out[12]) {
  const double w = q[0], x = q[1], yy = q[2], z = q[3];
  const double t[4][9] = {{2 * w, -2 * z, 2 * yy, 2 * z, 2 * w, -2 * x, -2 * yy, 2 * x, 2 * w},
                          {2 * x, 2 * yy, 2 * z, 2 * yy, -2 * x, -2 * w, 2 * z, 2 * w, -2 * x},
                          {-2 * yy, 2 * x, 2 * w, 2 * x, 2 * yy, 2 * z, -2 * w, 2 * z, -2 * yy},
                          {-2 * z, -2 * w, 2 * x, 2 * w, -2 * z, 2 * yy, 2 * x, 2 * yy, 2 * z}};
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 3; ++i) {
      double acc = 0.0;
      for (int c = 0; c < 3; ++c) acc += t[k][i * 3 + c] * a[c];
      out[i * 4 + k] = acc;
    }
}

// func_ypi_and_dypi_by_dxp_and_dypi_by_dhi_and_Ri (part_feature_model.cpp:162-229).
// Outputs: ypi (6); Tq = dhLhatWi_by_dq (3x4 row-major; dypi_by_dxp = [I3 0; 0 Tq]); Dh = dhLhatWi_by_dhi (3x2); Ri.
SL2_HD void part_create_model(const CameraParams& cam, const double xp[7], const double hi[2], double ypi[6], double Tq[12],
                              double Dh[6], double* Ri) {
  // Unproject (camera.cpp:133-154)
  const double c0 = hi[0] - cam.u0, c1 = hi[1] - cam.v0;
  const double radius2 = (c0 * c0 + c1 * c1);
  const double factor = sqrt(1 - 2 * cam.kd1 * radius2);
  const double und0 = c0 / factor, und1 = c1 / factor;
  const double hLRi[3] = {und0 / -cam.fku, und1 / -cam.fkv, 1.0};
  const double nrm = sqrt(hLRi[0] * hLRi[0] + hLRi[1] * hLRi[1] + hLRi[2] * hLRi[2]);
  const double hhat[3] = {hLRi[0] / nrm, hLRi[1] / nrm, hLRi[2] / nrm};
  // dvnorm_by_dv (part_feature_model.cpp:300-333): vv is the squared norm
  double dn[9];
  {
    const double vv = hLRi[0] * hLRi[0] + hLRi[1] * hLRi[1] + hLRi[2] * hLRi[2];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        dn[i * 3 + j] = (i == j) ? (1 - hLRi[i] * hLRi[i] / (vv * vv)) / vv : -hLRi[i] * hLRi[j] / (vv * vv * vv);
  }
  double RWR[9];
  quat_to_rot(&xp[3], RWR);
  for (int i = 0; i < 3; ++i) {
    double acc = 0.0;
    for (int k = 0; k < 3; ++k) acc += RWR[i * 3 + k] * hhat[k];
    ypi[3 + i] = acc;
    ypi[i] = xp[i];
  }
  dRq_times_a_by_dq(&xp[3], hhat, Tq);
  // UnprojectionJacobian (camera.cpp:247-275), 3x2
  double UJ[6];
  {
    const double dy_by_du[6] = {-1 / cam.fku, 0.0, 0.0, -1 / cam.fkv, 0.0, 0.0};
    double d00 = c0 * c0, d01 = c0 * c1, d10 = c1 * c0, d11 = c1 * c1;
    const double r2 = d00 + d11;
    const double distor = 1 - 2 * cam.kd1 * r2;
    const double distor1_2 = sqrt(distor);
    const double distor3_2 = distor1_2 * distor;
    const double s = 2 * cam.kd1 / distor3_2;
    d00 *= s; d01 *= s; d10 *= s; d11 *= s;
    d00 += (1 / distor1_2);
    d11 += (1 / distor1_2);
    for (int r = 0; r < 3; ++r) {
      UJ[r * 2 + 0] = dy_by_du[r * 2 + 0] * d00 + dy_by_du[r * 2 + 1] * d10;
      UJ[r * 2 + 1] = dy_by_du[r * 2 + 0] * d01 + dy_by_du[r * 2 + 1] * d11;
    }
  }
  // (RWR * dn) * UJ
  double M[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += RWR[i * 3 + k] * dn[k * 3 + j];
      M[i * 3 + j] = acc;
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 2; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += M[i * 3 + k] * UJ[k * 2 + j];
      Dh[i * 2 + j] = acc;
    }
  // MeasurementNoise (camera.cpp:282-300)
  {
    const double dx = hi[0] - cam.u0, dy = hi[1] - cam.v0;
    const double distance = sqrt(dx * dx + dy * dy);
    const double max_distance = sqrt(cam.u0 * cam.u0 + cam.v0 * cam.v0);
    const double ratio = distance / max_distance;
    const double sd_use = cam.sd * (1.0 + ratio);
    *Ri = sd_use * sd_use;
  }
}

// func_hpi_and_dhpi_by_dxp_and_dhpi_by_dyi (part_feature_model.cpp:231-265) on top of
// func_zeroedyi_and_dzeroedyi_by_dxp_and_dzeroedyi_by_dyi (:80-146).
// Outputs: hpi (2), Hx = dhpi_by_dxp (2x7), Hy = dhpi_by_dyi (2x6), R = measurement noise at hpi.
SL2_HD void part_measurement_model(const CameraParams& cam, const double xp[7], const double ypi[6], double lambda, double hpi[2],
                                   double Hx[14], double Hy[12], double* Rnoise) {
  const double d[3] = {ypi[0] - xp[0], ypi[1] - xp[1], ypi[2] - xp[2]};
  const double hh[3] = {ypi[3], ypi[4], ypi[5]};
  double qRW[4], RRW[9];
  quat_inverse(&xp[3], qRW);
  quat_to_rot(qRW, RRW);
  double zr[3], zh[3];
  for (int i = 0; i < 3; ++i) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < 3; ++k) { a += RRW[i * 3 + k] * d[k]; b += RRW[i * 3 + k] * hh[k]; }
    zr[i] = a; zh[i] = b;
  }
  // dzeroedyi_by_dxp (6x7): rows 0-2 = [-RRW | dRq(qRW, d) dqbar], rows 3-5 = [0 | dRq(qRW, hh) dqbar]
  double dzx[42], A[12], Bm[12];
  dRq_times_a_by_dq(qRW, d, A);
  dRq_times_a_by_dq(qRW, hh, Bm);
  for (int i = 0; i < 42; ++i) dzx[i] = 0.0;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) dzx[i * 7 + j] = RRW[i * 3 + j] * -1.0;
    for (int k = 0; k < 4; ++k) {
      const double sgn = (k == 0) ? 1.0 : -1.0;      // dqbar_by_dq = diag(1, -1, -1, -1): one non-zero per column
      dzx[i * 7 + 3 + k] = A[i * 4 + k] * sgn;
      dzx[(3 + i) * 7 + 3 + k] = Bm[i * 4 + k] * sgn;
    }
  }
  // hLR = zeroedri + lambda * zeroedhhati ; Project (camera.cpp:90-114)
  const double hLR[3] = {zr[0] + lambda * zh[0], zr[1] + lambda * zh[1], zr[2] + lambda * zh[2]};
  const double ic0 = -cam.fku * hLR[0] / hLR[2];
  const double ic1 = -cam.fkv * hLR[1] / hLR[2];
  {
    const double radius2 = (ic0 * ic0 + ic1 * ic1);
    const double factor = sqrt(1 + 2 * cam.kd1 * radius2);
    hpi[0] = ic0 / factor + cam.u0;
    hpi[1] = ic1 / factor + cam.v0;
  }
  double J[6];
  {
    const double fku_yz = cam.fku / hLR[2];
    const double fkv_yz = cam.fkv / hLR[2];
    const double du[6] = {-fku_yz, 0.0, fku_yz * hLR[0] / hLR[2], 0.0, -fkv_yz, fkv_yz * hLR[1] / hLR[2]};
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
  // JM = J * [I3 | lambda I3]  (2x6): the dense product adds exact zeros around the one non-zero term of each column
  double JM[12];
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 3; ++c) {
      JM[r * 6 + c] = J[r * 3 + c] * 1.0;
      JM[r * 6 + 3 + c] = J[r * 3 + c] * lambda;
    }
  for (int r = 0; r < 2; ++r) {
    for (int c = 0; c < 7; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 6; ++k) acc += JM[r * 6 + k] * dzx[k * 7 + c];
      Hx[r * 7 + c] = acc;
    }
    // dzeroedyi_by_dyi = blockdiag(RRW, RRW)
    for (int c = 0; c < 6; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 6; ++k) {
        const double e = ((k < 3) == (c < 3)) ? RRW[(k % 3) * 3 + (c % 3)] : 0.0;
        acc += JM[r * 6 + k] * e;
      }
      Hy[r * 6 + c] = acc;
    }
  }
  {
    const double dx = hpi[0] - cam.u0, dy = hpi[1] - cam.v0;
    const double distance = sqrt(dx * dx + dy * dy);
    const double max_distance = sqrt(cam.u0 * cam.u0 + cam.v0 * cam.v0);
    const double ratio = distance / max_distance;
    const double sd_use = cam.sd * (1.0 + ratio);
    *Rnoise = sd_use * sd_use;
  }
}

// func_Si for a 6-state feature (feature_model.cpp:99-116): Pxx7 7x7, Pxy7 7x6, Pyy 6x6 (row-major).
SL2_HD void innovation_cov6(const double Hx[14], const double Hy[12], double Rn, const double Pxx7[49], const double Pxy7[42],
                            const double Pyy[36], double S[4]) {
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
    double t2[6];
    for (int c = 0; c < 6; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 7; ++k) acc += Hx[r * 7 + k] * Pxy7[k * 6 + c];
      t2[c] = acc;
    }
    for (int s = 0; s < 2; ++s) {
      double acc = 0.0;
      for (int c = 0; c < 6; ++c) acc += t2[c] * Hy[s * 6 + c];
      T[r * 2 + s] = acc;
    }
    double t3[6];
    for (int c = 0; c < 6; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 6; ++k) acc += Hy[r * 6 + k] * Pyy[k * 6 + c];
      t3[c] = acc;
    }
    for (int s = 0; s < 2; ++s) {
      double acc = 0.0;
      for (int c = 0; c < 6; ++c) acc += t3[c] * Hy[s * 6 + c];
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

// Particle::set_S (feature_init_info.cpp:57-65): determinant of a dynamic-size Eigen matrix = partial-pivot LU
SL2_HD double det2_partial_pivot_lu(const double S[4]) {
  const double a = S[0], b = S[1], c = S[2], d = S[3];
  if (fabs(c) > fabs(a)) {
    const double l = a / c;
    return -(c * (b - l * d));
  }
  const double l = c / a;
  return a * (d - l * b);
}

// monoslam.cpp:1466-1481
SL2_HD double particle_likelihood(const double z[2], const double h[2], const double sinv[3], double detS) {
  const double nu0 = z[0] - h[0], nu1 = z[1] - h[1];
  const double t0 = sinv[0] * nu0 + sinv[1] * nu1;
  const double t1 = sinv[1] * nu0 + sinv[2] * nu1;
  const double nuT_Sinv_nu = nu0 * t0 + nu1 * t1;
  return (1.0 / (sqrt(2.0 * kPi * detS))) * exp(-0.5 * nuT_Sinv_nu);
}

}  // namespace sl2
