// Front-end kernels of the per-frame step (all FP64, one launch per stage over the
// whole batch; the bodies live in sl2_frontend_dev.hpp, which the fused small-map kernels of sl2_small.hip share):
//   k_predict            Kalman::KalmanFilterPredict          kalman.cpp:50-69
//   k_feature_prediction predict_single_feature_measurements   monoslam.cpp:289-308
//                        + visibility_test + selection_score   full_feature_model.cpp:103-176
//   k_select             auto_select_n_features (ordering)     monoslam.cpp:187-254
//   k_finalize           normalise_state, delete_bad_features, symmetrise,
//                        trajectory_store_                      monoslam.cpp:137-177,616-703
// Data layout: see sl2_common.hpp (dense P[B][ld][ld], x[B][ld]).
#include "sl2_frontend_dev.hpp"

namespace sl2 {

__global__ void __launch_bounds__(256) k_predict(double* __restrict__ x, double* __restrict__ P, const int* __restrict__ n_slots,
                                                 double* __restrict__ prev_r, const int* __restrict__ part_i, int pend, int ld,
                                                 double dt) {
  predict_body(blockIdx.x, x, P, n_slots, prev_r, part_i, pend, ld, dt);
}

__global__ void __launch_bounds__(64) k_feature_prediction(const double* __restrict__ x, const double* __restrict__ P,
                                                           const double* __restrict__ xp_org, int* __restrict__ f_flags,
                                                           const int* __restrict__ n_slots, double* __restrict__ f_h,
                                                           double* __restrict__ f_Hx, double* __restrict__ f_Hy,
                                                           double* __restrict__ f_R, double* __restrict__ f_S,
                                                           double* __restrict__ f_score, int* __restrict__ srch_i, double* __restrict__ srch_d,
                                                           CameraParams cam, int N, int ld) {
  feature_prediction_body(blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x, x, P, xp_org, f_flags, n_slots, f_h, f_Hx, f_Hy, f_R, f_S,
                          f_score, srch_i, srch_d, cam, N, ld);
}

__global__ void __launch_bounds__(256) k_select(const double* __restrict__ f_score, int* __restrict__ f_flags,
                                                const int* __restrict__ n_slots, const double* __restrict__ xp_org,
                                                int* __restrict__ sel_idx, int* __restrict__ n_sel, int* __restrict__ n_vis,
                                                double* __restrict__ last_r, const int* __restrict__ srch_i,
                                                const double* __restrict__ srch_d, int* __restrict__ srch_sel, int N,
                                                int n_want, int* __restrict__ srch_big, int split_bands) {
  extern __shared__ double s_dyn[];
  select_body(blockIdx.x, f_score, f_flags, n_slots, xp_org, sel_idx, n_sel, n_vis, last_r, srch_i, srch_d, srch_sel, N, n_want,
              srch_big, split_bands, s_dyn);
}

__global__ void __launch_bounds__(128) k_finalize(double* __restrict__ x, double* __restrict__ P, int* __restrict__ f_flags,
                                                  const int* __restrict__ n_slots, int* __restrict__ attempted,
                                                  int* __restrict__ successful, const int* __restrict__ m_count,
                                                  const int* __restrict__ n_sel, double* __restrict__ traj,
                                                  int* __restrict__ traj_count, const double* __restrict__ last_r,
                                                  int* __restrict__ status, double* __restrict__ pos_log, int* __restrict__ pos_count, int N, int ld,
                                                  int min_attempts, double match_fraction, int save_trajectory,
                                                  const int* __restrict__ part_i, int pend, int* __restrict__ slots_max,
                                                  unsigned long long* __restrict__ slots_mail, int publish) {
  extern __shared__ int s_del[];
  finalize_body(blockIdx.x, x, P, f_flags, n_slots, attempted, successful, m_count, n_sel, traj, traj_count, last_r, status, pos_log,
                pos_count, N, ld, min_attempts, match_fraction, save_trajectory, part_i, pend, s_del, slots_max, slots_mail, publish);
}

#ifdef SL2_FRONT_TRACE
}  // namespace sl2
extern "C" int sl2_debug_front_trace(long long* dev_buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(sl2::g_front_trace), &dev_buf, sizeof(dev_buf)) == hipSuccess ? 0 : 2;
}
namespace sl2 {
#endif

int launch_predict(sl2_engine* e) {
  LaunchScope ls(e, "k_predict");
  hipLaunchKernelGGL(k_predict, dim3(e->B), dim3(256), 0, e->stream, e->x, e->P, e->n_slots, e->prev_r, e->part_i, e->ppos + 6 * e->kpart, e->ld,
                     e->prm.delta_t);
  SL2_HIP(hipGetLastError());
  return SL2_OK;
}

int launch_feature_prediction(sl2_engine* e) {
  LaunchScope ls(e, "k_feature_prediction");
  dim3 grid((e->N + 63) / 64, e->B);
  hipLaunchKernelGGL(k_feature_prediction, grid, dim3(64), 0, e->stream, e->x, e->P, e->xp_org, e->f_flags, e->n_slots,
                     e->f_h, e->f_Hx, e->f_Hy, e->f_R, e->f_S, e->f_score, e->srch_i, e->srch_d, e->cam, e->N, e->ld);
  SL2_HIP(hipGetLastError());
  return SL2_OK;
}

int launch_select(sl2_engine* e, int n) {
  LaunchScope ls(e, "k_select");
  if (n > e->nsel_max) n = e->nsel_max;
  const size_t shm = (size_t)e->N * (sizeof(double) + 3 * sizeof(int));
  hipLaunchKernelGGL(k_select, dim3(e->B), dim3(256), shm, e->stream, e->f_score, e->f_flags, e->n_slots, e->xp_org,
                     e->sel_idx, e->n_sel, e->n_vis, e->last_r, e->srch_i, e->srch_d, e->srch_sel, e->N, n, e->srch_big,
                     (e->srch_big && e->root->search_variant == 1) ? e->root->search_split : 0);
  SL2_HIP(hipGetLastError());
  return SL2_OK;
}

int launch_finalize(sl2_engine* e, int save_trajectory) {
  LaunchScope ls(e, "k_finalize");
  const size_t shm = (size_t)e->N * 2 * sizeof(int);
  hipLaunchKernelGGL(k_finalize, dim3(e->B), dim3(128), shm, e->stream, e->x, e->P, e->f_flags, e->n_slots, e->attempted,
                     e->successful, e->m_count, e->n_sel, e->traj, e->traj_count, e->last_r, e->status, e->pos_log,
                     e->pos_count, e->N, e->ld, e->prm.minimum_attempted_measurements_of_feature,
                     e->prm.successful_match_fraction, save_trajectory, e->part_i, e->ppos + 6 * e->kpart, e->root->slots_max_dev,
                     e->root->slots_mail_dev, e->group_first == 0 ? 1 : 0);
  SL2_HIP(hipGetLastError());
  return SL2_OK;
}

}  // namespace sl2
