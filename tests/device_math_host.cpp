// TEST VEHICLE: compiles the device-side scalar model math (scenelib2_amd/csrc/
// sl2_math.hpp, the functions the HIP kernels inline) with plain g++ so the
// formulas can be checked against the oracle on a machine without a GPU.
// The engine never runs this code on the CPU.
#include "../scenelib2_amd/csrc/sl2_math.hpp"

using namespace sl2;

static CameraParams cam_from(const double* c8) {
  CameraParams cam;
  cam.width = (int)c8[0]; cam.height = (int)c8[1]; cam.fku = c8[2]; cam.fkv = c8[3];
  cam.u0 = c8[4]; cam.v0 = c8[5]; cam.kd1 = c8[6]; cam.sd = (int)c8[7];
  return cam;
}

extern "C" {

// f (13), dense F (169 row-major), Q (169 row-major), built only through the device helpers
void dm_motion(const double* xv, double dt, double* f, double* F, double* Q) {
  double A44[16], B43[12];
  motion_f_and_blocks(xv, dt, f, A44, B43);
  for (int j = 0; j < 13; ++j) {
    double e[13] = {0};
    e[j] = 1.0;
    for (int i = 0; i < 13; ++i) F[i * 13 + j] = frow_dot(i, dt, A44, B43, e);
  }
  for (int i = 0; i < 13; ++i) for (int j = 0; j < 13; ++j) Q[i * 13 + j] = process_noise_entry(i, j, dt, B43);
}

// Pxx' = (F Pxx) F^T + Q exactly as k_predict evaluates it; strip' = F strip (13 x ncol, row-major)
void dm_predict_cov(const double* xv, double dt, const double* Pxx, const double* strip, int ncol, double* Pxx_out,
                    double* strip_out) {
  double f[13], A44[16], B43[12], T[169];
  motion_f_and_blocks(xv, dt, f, A44, B43);
  for (int i = 0; i < 13; ++i)
    for (int j = 0; j < 13; ++j) {
      double v[13];
      for (int k = 0; k < 13; ++k) v[k] = Pxx[k * 13 + j];
      T[i * 13 + j] = frow_dot(i, dt, A44, B43, v);
    }
  for (int i = 0; i < 13; ++i)
    for (int j = 0; j < 13; ++j) {
      double v[13];
      for (int k = 0; k < 13; ++k) v[k] = T[i * 13 + k];
      Pxx_out[i * 13 + j] = frow_dot(j, dt, A44, B43, v) + process_noise_entry(i, j, dt, B43);
    }
  for (int j = 0; j < ncol; ++j) {
    double v[13];
    for (int k = 0; k < 13; ++k) v[k] = strip[k * ncol + j];
    for (int i = 0; i < 13; ++i) strip_out[i * ncol + j] = frow_dot(i, dt, A44, B43, v);
  }
}

// f applied `steps` times: by repeated calls of motion_f_and_blocks (out_a) and by motion_f_repeated (out_b)
void dm_motion_repeated(const double* xv, double dt, int steps, double* out_a, double* out_b) {
  double cur[13], f[13], A44[16], B43[12];
  for (int i = 0; i < 13; ++i) cur[i] = xv[i];
  for (int it = 0; it < steps; ++it) {
    motion_f_and_blocks(cur, dt, f, A44, B43);
    for (int i = 0; i < 13; ++i) cur[i] = f[i];
  }
  for (int i = 0; i < 13; ++i) out_a[i] = cur[i];
  motion_f_repeated(xv, dt, steps, out_b);
}

void dm_dqnorm(const double* q, double* N16) { dqnorm_by_dq(q, N16); }

// out: h[2], Hx[14], Hy[6], R, vis  (24 doubles)
void dm_measurement(const double* cam8, const double* xp, const double* y, const double* xp_org, double* out) {
  const CameraParams cam = cam_from(cam8);
  double zeroed[3], h[2], Hx[14], Hy[6], Rn;
  measurement_model(cam, xp, y, zeroed, h, Hx, Hy, &Rn);
  int k = 0;
  out[k++] = h[0]; out[k++] = h[1];
  for (int i = 0; i < 14; ++i) out[k++] = Hx[i];
  for (int i = 0; i < 6; ++i) out[k++] = Hy[i];
  out[k++] = Rn;
  out[k++] = (double)visibility_test(cam, xp, y, xp_org, h);
}

void dm_innovation_cov(const double* Hx, const double* Hy, double Rn, const double* Pxx7, const double* Pxy7,
                       const double* Pyy, double* S) {
  innovation_cov(Hx, Hy, Rn, Pxx7, Pxy7, Pyy, S);
}

void dm_sinv(const double* S4, double* abc) { sinv_from_S(S4, &abc[0], &abc[1], &abc[2]); }

// out: ucentre, vcentre, urelstart, urelfinish, vrelstart, vrelfinish, halfwidth, halfheight
void dm_search_bounds(const double* centre, double a, double b, double c, int width, int height, int* out) {
  const SearchBounds sb = search_bounds(centre, a, b, c, width, height);
  out[0] = sb.ucentre; out[1] = sb.vcentre; out[2] = sb.urelstart; out[3] = sb.urelfinish;
  out[4] = sb.vrelstart; out[5] = sb.vrelfinish; out[6] = sb.halfwidth; out[7] = sb.halfheight;
}

int dm_in_ellipse(double a, double b, double c, int urel, int vrel) { return in_ellipse(a, b, c, urel, vrel) ? 1 : 0; }

double dm_ncc_score(int Sg0, int Sg1, int Sg0g1, int Sg0sq, int Sg1sq, double* sd0, double* sd1) {
  return ncc_score(Sg0, Sg1, Sg0g1, Sg0sq, Sg1sq, sd0, sd1);
}

// A scalar re-enactment of search_core_v0's candidate scan (same helpers, same
// accept rule) — checks the enumeration/tie logic the kernel uses.
int dm_search_scan(const uint8_t* image, int width, int height, const uint8_t* patch, const double* centre, double a,
                   double b, double c, int* uv, double* score, int* ncand_out) {
  const SearchBounds sb = search_bounds(centre, a, b, c, width, height);
  const int nu = sb.urelfinish - sb.urelstart + 1, nv = sb.vrelfinish - sb.vrelstart + 1;
  int Sg0 = 0, Sg0sq = 0;
  for (int p = 0; p < 121; ++p) { Sg0 += patch[p]; Sg0sq += patch[p] * patch[p]; }
  double best = 1000000.0;
  int best_order = -1, ncand = 0;
  if (nu > 0 && nv > 0)
    for (int idx = 0; idx < nu * nv; ++idx) {
      const int urel = sb.urelstart + idx / nv, vrel = sb.vrelstart + idx % nv;
      if (!in_ellipse(a, b, c, urel, vrel)) continue;
      ++ncand;
      const uint8_t* p1 = image + (size_t)(sb.vcentre + vrel - 5) * width + (sb.ucentre + urel - 5);
      int Sg1 = 0, Sg0g1 = 0, Sg1sq = 0;
      for (int r = 0; r < 11; ++r)
        for (int cc = 0; cc < 11; ++cc) {
          const int g0 = patch[r * 11 + cc], g1 = p1[r * width + cc];
          Sg1 += g1; Sg0g1 += g0 * g1; Sg1sq += g1 * g1;
        }
      double sd0, sd1;
      const double corr = ncc_score(Sg0, Sg1, Sg0g1, Sg0sq, Sg1sq, &sd0, &sd1);
      if (corr <= best && !(sd0 < kCorrelationSigmaThreshold) && !(sd1 < kCorrelationSigmaThreshold)) { best = corr; best_order = idx; }
    }
  *score = best; *ncand_out = ncand;
  if (best_order >= 0) { uv[0] = sb.ucentre + sb.urelstart + best_order / nv; uv[1] = sb.vcentre + sb.vrelstart + best_order % nv; }
  return (best_order >= 0 && !(best > kCorrThresh2)) ? 1 : 0;
}

}  // extern "C"
