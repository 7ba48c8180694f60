// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
// GoOneStep restatement + a flat C interface for ctypes (tests, smoke(), and the
// cpu_baseline leg of bench.py).  See slam_oracle.hpp for the parity status.
#include <cstdlib>
#include <malloc.h>
#include "slam_oracle.hpp"
#include "feature_init_oracle.hpp"

#include <atomic>
#include <chrono>
#include <thread>

namespace oracle {

double now_seconds() {
  using namespace std::chrono;
  return duration_cast<duration<double>>(steady_clock::now().time_since_epoch()).count();
}

// monoslam.cpp:108-180
bool MonoSLAM::GoOneStep(const uint8_t* frame, bool save_trajectory, bool enable_mapping) {
  location_selected_flag = false;
  init_feature_search_region_defined_flag = false;
  const double u[3] = {0, 0, 0};
  const double prev_xp_pos[3] = {xv(0), xv(1), xv(2)};

  double t0 = now_seconds();
  KalmanFilterPredict(u);
  double t1 = now_seconds();
  times.predict += t1 - t0;

  number_of_visible_features = auto_select_n_features(kNumberOfFeaturesToSelect);
  double t2 = now_seconds();
  times.select += t2 - t1;

  if (selected_feature_list.size() != 0) {
    make_measurements(frame);
    double t3 = now_seconds();
    times.search += t3 - t2;
    if (successful_measurement_vector_size != 0) {
      KalmanFilterUpdate();
      normalise_state();
    }
    times.update += now_seconds() - t3;
  }
  double t4 = now_seconds();

  delete_bad_features();

  // monoslam.cpp:143-150: enforce symmetry of the total covariance
  Mat P(total_state_size, total_state_size);
  construct_total_covariance(P);
  const Mat PT = transpose(P);
  P = add(scaled(P, 0.5), scaled(PT, 0.5));
  fill_covariances(P);

  // monoslam.cpp:152-170: speed gate -> AutoInitialiseFeature, then MatchPartiallyInitialisedFeatures (always)
  // (func_xp only fills xpRES_: the scratch rRES_ pushed into the trajectory below stays stale, Q12)
  const double vel[3] = {(xv(0) - prev_xp_pos[0]) / kDeltaT, (xv(1) - prev_xp_pos[1]) / kDeltaT, (xv(2) - prev_xp_pos[2]) / kDeltaT};
  const double speed = std::sqrt(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]);
  if (speed > 0.2 && enable_mapping) {
    if (number_of_visible_features < kNumberOfFeaturesToKeepVisible &&
        feature_init_info_vector.size() < (unsigned int)kMaxFeaturesToInitAtOnce)
      AutoInitialiseFeature(frame);
  }
  MatchPartiallyInitialisedFeatures(frame);

  if (save_trajectory) {  // monoslam.cpp:172-177 (stale scratch rRES_, Q12)
    for (int i = 0; i < 3; ++i) trajectory_store.push_back(motion_model.rRES[i]);
    if (trajectory_store.size() > 3000) trajectory_store.erase(trajectory_store.begin(), trajectory_store.begin() + 3);
  }
  times.rest += now_seconds() - t4;
  return true;
}

}  // namespace oracle

using oracle::Feature;
using oracle::Mat;
using oracle::MonoSLAM;
using oracle::Vec;

// Every dense temporary of the filter (n x n doubles, several per frame) is above glibc's mmap threshold: left alone each is
// an mmap / munmap pair plus a page fault per 4 KB, and with one MonoSLAM object per hardware thread those take the process's
// address-space lock in turn.  The timing harness keeps such blocks on the heap instead (set at load time, before any object
// is built).  Timing infrastructure only: the arithmetic is untouched.  SL2_HARNESS_NO_MALLOPT=1 leaves glibc's defaults.
__attribute__((constructor)) static void sl2_harness_malloc_setup() {
  if (getenv("SL2_HARNESS_NO_MALLOPT")) return;
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
}

extern "C" {

void* orc_create(int width, int height, double fku, double fkv, double u0, double v0, double kd1, int sd,
                 double delta_t, int n_select) {
  MonoSLAM* m = new MonoSLAM();
  m->camera.width = width; m->camera.height = height;
  m->camera.fku = fku; m->camera.fkv = fkv; m->camera.u0 = u0; m->camera.v0 = v0;
  m->camera.kd1 = kd1; m->camera.sd = sd;
  m->kDeltaT = delta_t;
  m->kNumberOfFeaturesToSelect = n_select;
  return m;
}
void orc_destroy(void* h) { delete (MonoSLAM*)h; }

// Pxx row-major [13][13]
void orc_set_state(void* h, const double* xv, const double* Pxx) {
  MonoSLAM* m = (MonoSLAM*)h;
  for (int i = 0; i < 13; ++i) m->xv(i) = xv[i];
  for (int i = 0; i < 13; ++i) for (int j = 0; j < 13; ++j) m->Pxx(i, j) = Pxx[i * 13 + j];
}
void orc_get_state(void* h, double* xv, double* Pxx) {
  MonoSLAM* m = (MonoSLAM*)h;
  for (int i = 0; i < 13; ++i) xv[i] = m->xv(i);
  for (int i = 0; i < 13; ++i) for (int j = 0; j < 13; ++j) Pxx[i * 13 + j] = m->Pxx(i, j);
}
void orc_add_known_feature(void* h, const double* y, const double* xp, const uint8_t* patch) {
  ((MonoSLAM*)h)->AddNewKnownFeature(y, xp, patch);
}
int orc_go_one_step(void* h, const uint8_t* frame, int save_trajectory, int enable_mapping) {
  return ((MonoSLAM*)h)->GoOneStep(frame, save_trajectory != 0, enable_mapping != 0) ? 1 : 0;
}
int orc_num_features(void* h) { return (int)((MonoSLAM*)h)->feature_list.size(); }
int orc_num_selected(void* h) { return (int)((MonoSLAM*)h)->selected_feature_list.size(); }
int orc_total_state_size(void* h) { return ((MonoSLAM*)h)->total_state_size; }
int orc_num_visible(void* h) { return ((MonoSLAM*)h)->number_of_visible_features; }
int orc_measurement_size(void* h) { return ((MonoSLAM*)h)->successful_measurement_vector_size; }
void orc_get_total_state(void* h, double* x) {
  MonoSLAM* m = (MonoSLAM*)h;
  Vec v(m->total_state_size, 1);
  m->construct_total_state(v);
  for (int i = 0; i < m->total_state_size; ++i) x[i] = v(i);
}
// row-major [n][n]
void orc_get_total_covariance(void* h, double* P) {
  MonoSLAM* m = (MonoSLAM*)h;
  const int n = m->total_state_size;
  Mat M(n, n);
  m->construct_total_covariance(M);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) P[(size_t)i * n + j] = M(i, j);
}
// ints: label, selected, successful_flag, attempted, successful, position_in_total_state_vector
// doubles: y[3], h[2], z[2], nu[2], R, S[4] row-major, dh_by_dxv[26] row-major 2x13, dh_by_dy[6] row-major
void orc_get_feature(void* h, int idx, int* ints, double* dbl) {
  MonoSLAM* m = (MonoSLAM*)h;
  const Feature* f = m->feature_list[idx];
  ints[0] = f->label; ints[1] = f->selected_flag; ints[2] = f->successful_measurement_flag;
  ints[3] = f->attempted; ints[4] = f->successful; ints[5] = f->position_in_total_state_vector;
  int k = 0;
  for (int i = 0; i < 3; ++i) dbl[k++] = f->y[i];
  for (int i = 0; i < 2; ++i) dbl[k++] = f->h[i];
  for (int i = 0; i < 2; ++i) dbl[k++] = f->z[i];
  for (int i = 0; i < 2; ++i) dbl[k++] = f->nu[i];
  dbl[k++] = f->R;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) dbl[k++] = f->S(i, j);
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 13; ++j) dbl[k++] = f->dh_by_dxv(i, j);
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) dbl[k++] = f->dh_by_dy(i, j);
}
// ---- feature initialisation state ----
// params: keep_visible, max_init_at_once, n_particles, min_particles, erase_after ; min_lambda, max_lambda, sd_ratio, prune_threshold
void orc_set_mapping_params(void* h, const int* ip, const double* dp) {
  MonoSLAM* m = (MonoSLAM*)h;
  m->kNumberOfFeaturesToKeepVisible = ip[0]; m->kMaxFeaturesToInitAtOnce = ip[1]; m->kNumberOfParticles = ip[2];
  m->kMinNumberOfParticles = ip[3]; m->kErasePartiallyInitFeatureAfterThisManyAttempts = ip[4];
  m->kMinLambda = dp[0]; m->kMaxLambda = dp[1]; m->kStandardDeviationDepthRatio = dp[2]; m->kPruneProbabilityThreshold = dp[3];
}
// ints: n_partial, features_initialised, features_converted, partial_features_deleted, uu, vv, location_selected,
//       region_defined, ustart, vstart, ufinish, vfinish
void orc_get_mapping_info(void* h, int* ints) {
  MonoSLAM* m = (MonoSLAM*)h;
  ints[0] = (int)m->feature_init_info_vector.size(); ints[1] = m->features_initialised; ints[2] = m->features_converted;
  ints[3] = m->partial_features_deleted; ints[4] = m->uu; ints[5] = m->vv; ints[6] = m->location_selected_flag;
  ints[7] = m->init_feature_search_region_defined_flag; ints[8] = m->init_feature_search_ustart;
  ints[9] = m->init_feature_search_vstart; ints[10] = m->init_feature_search_ufinish; ints[11] = m->init_feature_search_vfinish;
}
// partial feature k: ints = label, n_particles, number_of_match_attempts, making_measurement ; dbl = mean, covariance, y[6]
// particles [n][10] = lambda, probability, cumulative, h0, h1, z0, z1, SInv00, SInv01, SInv11 (+ detS, success in [10], [11])
int orc_get_partial_feature(void* h, int k, int* ints, double* dbl, double* particles, int max_particles) {
  MonoSLAM* m = (MonoSLAM*)h;
  if (k < 0 || k >= (int)m->feature_init_info_vector.size()) return 0;
  const oracle::FeatureInitInfo& f = m->feature_init_info_vector[k];
  ints[0] = f.fp->label; ints[1] = (int)f.particle_vector.size(); ints[2] = f.number_of_match_attempts;
  ints[3] = f.making_measurement_on_this_step_flag;
  dbl[0] = f.mean; dbl[1] = f.covariance;
  for (int i = 0; i < 6; ++i) dbl[2 + i] = f.fp->y[i];
  for (int i = 0; i < (int)f.particle_vector.size() && i < max_particles; ++i) {
    const oracle::Particle& p = f.particle_vector[i];
    double* o = particles + 12 * (size_t)i;
    o[0] = p.lambda; o[1] = p.probability; o[2] = p.cumulative_probability; o[3] = p.m_h[0]; o[4] = p.m_h[1];
    o[5] = p.m_z[0]; o[6] = p.m_z[1]; o[7] = p.SInv[0]; o[8] = p.SInv[1]; o[9] = p.SInv[2]; o[10] = p.detS;
    o[11] = p.m_successful_measurement_flag;
  }
  return 1;
}
// per feature of feature_list_: state_size, fully_initialised, label
void orc_get_feature_kinds(void* h, int* out3) {
  MonoSLAM* m = (MonoSLAM*)h;
  for (size_t i = 0; i < m->feature_list.size(); ++i) {
    out3[3 * i] = m->feature_list[i]->state_size; out3[3 * i + 1] = m->feature_list[i]->fully_initialised_flag;
    out3[3 * i + 2] = m->feature_list[i]->label;
  }
}
void orc_get_feature_patch(void* h, int idx, uint8_t* patch121) {
  std::memcpy(patch121, ((MonoSLAM*)h)->feature_list[idx]->patch, 121);
}
// labels of the selected features in selected_feature_list_ order
void orc_get_selected_labels(void* h, int* labels) {
  MonoSLAM* m = (MonoSLAM*)h;
  for (size_t i = 0; i < m->selected_feature_list.size(); ++i) labels[i] = m->selected_feature_list[i]->label;
}
int orc_trajectory(void* h, double* out, int max_entries) {
  MonoSLAM* m = (MonoSLAM*)h;
  int n = (int)m->trajectory_store.size() / 3;
  if (n > max_entries) n = max_entries;
  for (int i = 0; i < 3 * n; ++i) out[i] = m->trajectory_store[i];
  return n;
}
void orc_get_diag(void* h, long long* cand, long long* window_bytes, double* times5) {
  MonoSLAM* m = (MonoSLAM*)h;
  *cand = m->total_candidates; *window_bytes = m->total_window_bytes;
  times5[0] = m->times.predict; times5[1] = m->times.select; times5[2] = m->times.search;
  times5[3] = m->times.update; times5[4] = m->times.rest;
}

// ---- seams, for stage-level parity tests ------------------------------------
void orc_kalman_filter_predict(void* h) { const double u[3] = {0, 0, 0}; ((MonoSLAM*)h)->KalmanFilterPredict(u); }
int orc_auto_select_n_features(void* h, int n) {
  MonoSLAM* m = (MonoSLAM*)h;
  m->number_of_visible_features = m->auto_select_n_features(n);
  return m->number_of_visible_features;
}
int orc_make_measurements(void* h, const uint8_t* frame) { return ((MonoSLAM*)h)->make_measurements(frame); }
void orc_kalman_filter_update(void* h) { ((MonoSLAM*)h)->KalmanFilterUpdate(); }
void orc_normalise_state(void* h) { ((MonoSLAM*)h)->normalise_state(); }
void orc_delete_bad_features(void* h) { ((MonoSLAM*)h)->delete_bad_features(); }
// mark_feature_by_lab(label) + delete_feature() (monoslam.cpp:743-812)
int orc_delete_feature(void* h, int label) {
  MonoSLAM* m = (MonoSLAM*)h;
  m->mark_feature_by_lab(label);
  return m->delete_feature() ? 1 : 0;
}
// Feature::Pyy_ (feature.h:84), row-major 3x3: a known feature with a prior uncertainty
void orc_set_feature_Pyy(void* h, int idx, const double* Pyy9) {
  Feature* f = ((MonoSLAM*)h)->feature_list[idx];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) f->Pyy(i, j) = Pyy9[i * 3 + j];
}
void orc_set_feature_counters(void* h, int idx, int attempted, int successful) {
  Feature* f = ((MonoSLAM*)h)->feature_list[idx];
  f->attempted = attempted; f->successful = successful;
}

// test hook: Feature::position_in_total_state_vector_ written directly - what feature.cpp:254 (Q28) does to it over several
// conversions, without running them (tests/test_gpu_slam.py: the dh_by_dy block landing inside the vehicle state)
void orc_set_feature_position(void* h, int idx, int pos) {
  ((MonoSLAM*)h)->feature_list[idx]->position_in_total_state_vector = pos;
}

// MonoSLAM::InitialiseFeature at (uu_, vv_) = (u, v) (monoslam.cpp:1211-1235: what the "initialise manual feature" button of
// examples/MonoSlamSceneLib1.cpp:191-192 calls after a mouse click set uu_ / vv_) and InitialiseAutoFeature (:1535-1541)
void orc_initialise_feature(void* h, const uint8_t* frame, int u, int v) {
  MonoSLAM* m = (MonoSLAM*)h;
  m->uu = u; m->vv = v;
  m->location_selected_flag = true;
  m->InitialiseFeature(frame);
}
void orc_initialise_auto_feature(void* h, const uint8_t* frame) { ((MonoSLAM*)h)->AutoInitialiseFeature(frame); }

// ---- stateless functions ------------------------------------------------------
double orc_correlate2_warning(int x0, int y0, int x0lim, int y0lim, int x1, int y1, const uint8_t* p0, int w0,
                              const uint8_t* p1, int w1, double* sd0, double* sd1) {
  return oracle::correlate2_warning(x0, y0, x0lim, y0lim, x1, y1, p0, w0, p1, w1, sd0, sd1);
}
// out_i: u, v, n_candidates, halfwidth, halfheight ; returns ok
int orc_elliptical_search(const uint8_t* image, int width, int height, const uint8_t* patch, const double* centre,
                          double a, double b, double c, int* out_i, double* best_corr) {
  int u = -1, v = -1, nc = 0, hw = 0, hh = 0;
  const bool ok = oracle::elliptical_search(image, width, height, patch, centre, a, b, c, &u, &v, 11, &nc, best_corr, &hw, &hh);
  out_i[0] = u; out_i[1] = v; out_i[2] = nc; out_i[3] = hw; out_i[4] = hh;
  return ok ? 1 : 0;
}
// monoslam.cpp:1070-1192.  io_uv holds (ubest, vbest) on entry and exit (left untouched when nothing scores).
void orc_find_best_patch(const uint8_t* image, int width, int height, int ustart, int vstart, int ufinish, int vfinish,
                         int* io_uv, double* evbest) {
  oracle::find_best_patch_inside_region(image, width, height, &io_uv[0], &io_uv[1], evbest, 11, ustart, vstart, ufinish, vfinish);
}
// SearchMultipleOverlappingEllipses over n ellipses: puinv [n][3] = (PuInv(0,0), PuInv(0,1), PuInv(1,1)), centre [n][2].
// out_i [n][3] = (result_flag, result_u, result_v); out_corr [n] = corrmax (diagnostic).  Returns the number of
// positions actually correlated (cache misses).
long long orc_search_multiple_ellipses(const uint8_t* image, int width, int height, const uint8_t* patch, int n,
                                       const double* puinv, const double* centre, int* out_i, double* out_corr) {
  oracle::MultiEllipseSearch s(image, width, height, patch, 11);
  for (int i = 0; i < n; ++i) s.add_ellipse(puinv[3 * i], puinv[3 * i + 1], puinv[3 * i + 2], centre[2 * i], centre[2 * i + 1]);
  s.search();
  for (int i = 0; i < n; ++i) {
    out_i[3 * i] = s.data[i].result_flag ? 1 : 0;
    out_i[3 * i + 1] = s.data[i].result_u;
    out_i[3 * i + 2] = s.data[i].result_v;
    if (out_corr) out_corr[i] = s.data[i].corrmax;
  }
  return s.correlations;
}
void orc_drand48_sequence(long seed, int n, double* out) {
  oracle::Rand48 r;
  r.seed(seed);
  for (int i = 0; i < n; ++i) out[i] = r.next();
}
void orc_sinv_from_S(const double* S4, double* abc) {
  Mat S(2, 2);
  S(0, 0) = S4[0]; S(0, 1) = S4[1]; S(1, 0) = S4[2]; S(1, 1) = S4[3];
  MonoSLAM::sinv_from_S(S, abc[0], abc[1], abc[2]);
}
// motion model: f (13), F row-major (169), Q row-major (169)
void orc_motion_model(const double* xv, double dt, double* f, double* F, double* Q) {
  oracle::MotionModel mm;
  Vec x(13, 1);
  for (int i = 0; i < 13; ++i) x(i) = xv[i];
  const double u[3] = {0, 0, 0};
  mm.func_fv_and_dfv_by_dxv(x, u, dt);
  mm.func_Q(x, dt);
  for (int i = 0; i < 13; ++i) f[i] = mm.fvRES(i);
  for (int i = 0; i < 13; ++i) for (int j = 0; j < 13; ++j) { F[i * 13 + j] = mm.dfv_by_dxv(i, j); Q[i * 13 + j] = mm.Qx(i, j); }
}
void orc_dqnorm_by_dq(const double* q, double* J16) {
  const Mat M = oracle::MotionModel::dqnorm_by_dq(oracle::Quat(q[0], q[1], q[2], q[3]));
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) J16[i * 4 + j] = M(i, j);
}
// measurement model for one point: cam8 = width,height,fku,fkv,u0,v0,kd1,sd
// out: h[2], dh_by_dxp[14] row-major 2x7, dh_by_dy[6] row-major 2x3, R, vis flags (as double)
void orc_measurement_model(const double* cam8, const double* xp, const double* y, const double* xp_org, double* out) {
  oracle::Camera cam;
  cam.width = (int)cam8[0]; cam.height = (int)cam8[1]; cam.fku = cam8[2]; cam.fkv = cam8[3];
  cam.u0 = cam8[4]; cam.v0 = cam8[5]; cam.kd1 = cam8[6]; cam.sd = (int)cam8[7];
  oracle::MotionModel mm;
  oracle::FullFeatureModel ffm;
  ffm.cam = &cam; ffm.mm = &mm;
  ffm.func_hi_and_jacobians(y, xp);
  int k = 0;
  out[k++] = ffm.hi[0]; out[k++] = ffm.hi[1];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 7; ++j) out[k++] = ffm.dhi_by_dxp(i, j);
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) out[k++] = ffm.dhi_by_dyi(i, j);
  out[k++] = cam.MeasurementNoise(ffm.hi);
  const double h[2] = {ffm.hi[0], ffm.hi[1]};
  out[k++] = (double)ffm.visibility_test(xp, y, xp_org, h);
}

// ---- the dense primitives of oracle/dense.hpp (the Eigen semantics the reference's call sites rely on), exported so that
// tests/test_oracle_numpy.py can hold them against NumPy / SciPy - implementations nobody here wrote.  Row-major in / out.
int orc_dense_llt(int n, const double* A, double* L_out) {                 // Eigen::LLT lower factor (kalman.cpp:104)
  Mat M(n, n), L;
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) M(i, j) = A[i * n + j];
  const bool ok = oracle::llt_lower(M, L);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) L_out[i * n + j] = L(i, j);
  return ok ? 1 : 0;
}
void orc_dense_inverse(int n, const double* A, double* out) {              // MatrixXd::inverse() (kalman.cpp:106)
  Mat M(n, n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) M(i, j) = A[i * n + j];
  const Mat X = oracle::general_inverse(M);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) out[i * n + j] = X(i, j);
}
void orc_dense_mul(int m, int k, int n, const double* A, const double* B, double* C) {
  Mat a(m, k), b(k, n);
  for (int i = 0; i < m; ++i) for (int j = 0; j < k; ++j) a(i, j) = A[i * k + j];
  for (int i = 0; i < k; ++i) for (int j = 0; j < n; ++j) b(i, j) = B[i * n + j];
  const Mat c = oracle::mul(a, b);
  for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) C[i * n + j] = c(i, j);
}
// Quaterniond product / inverse() / toRotationMatrix() (motion_model.cpp:102, full_feature_model.cpp:76-80); q = (w, x, y, z)
void orc_quat_ops(const double* qa, const double* qb, double* prod4, double* inv4, double* R9) {
  const oracle::Quat a(qa[0], qa[1], qa[2], qa[3]), b(qb[0], qb[1], qb[2], qb[3]);
  const oracle::Quat p = oracle::qmul(a, b), iv = oracle::qinverse(a);
  prod4[0] = p.w; prod4[1] = p.x; prod4[2] = p.y; prod4[3] = p.z;
  inv4[0] = iv.w; inv4[1] = iv.x; inv4[2] = iv.y; inv4[3] = iv.z;
  const Mat R = oracle::qrot(a);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R9[i * 3 + j] = R(i, j);
}

// ---- CPU baseline: nseq independent sequences, one per worker thread -----------
// Each sequence s has its own oracle handle hs[s]; frames[s] -> nframes consecutive
// frames of frame_bytes each.  Returns wall seconds.  traj (optional) receives
// xv[0:3] after every step: [nseq][nframes][3].
double orc_run_sequences(void** hs, int nseq, const uint8_t* const* frames, int nframes, size_t frame_bytes,
                         int nthreads, double* traj) {
  std::atomic<int> next(0);
  const double t0 = oracle::now_seconds();
  auto worker = [&]() {
    for (;;) {
      const int s = next.fetch_add(1);
      if (s >= nseq) break;
      MonoSLAM* m = (MonoSLAM*)hs[s];
      for (int f = 0; f < nframes; ++f) {
        m->GoOneStep(frames[s] + (size_t)f * frame_bytes, false, false);
        if (traj) for (int i = 0; i < 3; ++i) traj[((size_t)s * nframes + f) * 3 + i] = m->xv(i);
      }
    }
  };
  if (nthreads <= 1) {
    worker();
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(worker);
    for (auto& t : th) t.join();
  }
  return oracle::now_seconds() - t0;
}

}  // extern "C"
