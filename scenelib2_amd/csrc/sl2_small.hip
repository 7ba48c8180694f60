// Small maps: the whole MonoSLAM::GoOneStep (monoslam.cpp:108-180) in THREE launches instead of ten.
//
// The reference's own workload is a map of 4-20 features of which 10 are measured per frame (data/SceneLib2.cfg:60-62).
// There every stage of the step is a few microseconds of arithmetic and the step is the sum of ten dependent launches
// (7-12 us each at one sequence).  For engines whose state fits 128 columns and whose innovation system is ONE 32x32 block
// (at most 16 features measured per frame) the stages on either side of the patch search are fused, a workgroup per sequence:
//
//   k_small_front   KalmanFilterPredict -> predict_single_feature_measurements -> auto_select_n_features
//                   (predict_body, feature_prediction_body, select_body of sl2_frontend_dev.hpp: the same code as the
//                    one-stage kernels, so every bit-exact contract of the model math carries over)
//   k_search_mfma   the patch search, untouched (sl2_search.hip: its large-window sharing needs the whole launch)
//   k_small_back    measurement bookkeeping (search_score_body) -> KalmanFilterUpdate -> normalise / delete / symmetrise /
//                   trajectory (finalize_body).  The update (kalman.cpp:72-119) is written for the one-block case:
//                   A^T = (P H^T)^T and S = H A + R in LDS, S = L L^T and L^-1 by the D-wave routine of the blocked Cholesky
//                   (sl2_chol_diag.hpp) on one wavefront, V = L^-1 A^T in place (a column per thread, the column in
//                   registers), P -= V^T V and x += V^T (L^-1 nu) straight on the covariance in memory - the algebra of
//                   sl2_ekf_update.hip (W S W^T = V^T V), no workspace in HBM at all.
//
// Compiled with -ffp-contract=off like every translation unit that holds model math; the update's FMAs are spelled out.
#include "sl2_frontend_dev.hpp"
#include "sl2_score_dev.hpp"
#include "sl2_chol_diag.hpp"

namespace sl2 {

// development only (SL2_FRONT_TRACE build, scripts/small_trace.py): wall-clock stamps (10 ns units) of the fused kernels' phases
#ifdef SL2_FRONT_TRACE
#define SST(row, slot) do { if (g_front_trace && threadIdx.x == 0) g_front_trace[((size_t)(row) * 4096 + blockIdx.x) * 8 + (slot)] = (long long)wall_clock64(); } while (0)
#else
#define SST(row, slot) do { } while (0)
#endif

constexpr int kSmallThreads = 256;
constexpr int kSmallM = 32;          // rows of the innovation system (one Cholesky block)

__global__ void __launch_bounds__(kSmallThreads) k_small_front(
    double* __restrict__ x, double* __restrict__ P, const int* __restrict__ n_slots, double* __restrict__ prev_r,
    const int* __restrict__ part_i, int pend, int ld, double dt,
    const double* __restrict__ xp_org, int* __restrict__ f_flags, double* __restrict__ f_h, double* __restrict__ f_Hx,
    double* __restrict__ f_Hy, double* __restrict__ f_R, double* __restrict__ f_S, double* __restrict__ f_score,
    int* __restrict__ srch_i, double* __restrict__ srch_d, CameraParams cam, int N,
    int* __restrict__ sel_idx, int* __restrict__ n_sel, int* __restrict__ n_vis, double* __restrict__ last_r,
    int* __restrict__ srch_sel, int n_want, int* __restrict__ srch_big, int split_bands) {
  extern __shared__ double s_dyn[];
  const int b = blockIdx.x;
  SST(2, 0);
  predict_body(b, x, P, n_slots, prev_r, part_i, pend, ld, dt);
  __syncthreads();                                    // x and P of this sequence: written above, read below (same workgroup)
  SST(2, 1);
  for (int i = threadIdx.x; i < N; i += (int)blockDim.x)
    feature_prediction_body(b, i, x, P, xp_org, f_flags, n_slots, f_h, f_Hx, f_Hy, f_R, f_S, f_score, srch_i, srch_d, cam, N, ld);
  __syncthreads();
  SST(2, 2);
  select_body(b, f_score, f_flags, n_slots, xp_org, sel_idx, n_sel, n_vis, last_r, srch_i, srch_d, srch_sel, N, n_want, srch_big,
              split_bands, s_dyn);
  SST(2, 3);
}

// Kalman::KalmanFilterUpdate (kalman.cpp:72-119) for a system of at most 32 measurement rows; all threads of the workgroup.
// The engine's leading dimension ld is whatever the feature CAPACITY asks for (the adapter reserves 128 slots: ld = 448);
// what the update touches is the LIVE part of the state: columns [0, 13 + 3 n_slots), the six states of a partially
// initialised feature at ppos (if one is in flight), and the innovation.  Those are gathered into a compact column index
// c < W (W = 64 or 128, cmap below) for everything that lives in LDS; P and x are addressed in place, at stride ld.
// sAt: [32][W] doubles of dynamic LDS.  H row a = 2 j + r of the j-th successful feature in SLOT order (succ_idx): the seven
// pose coefficients of dh_by_dxv and the three of dh_by_dy at column 13 + 3 slot (monoslam.cpp:548-572; with one partially
// initialised feature per sequence - the only case this kernel is launched for - no recorded position is misplaced, Q28).
constexpr int kSmallW = 128;         // compact columns at most: 13 + 3 * 36 + 6 + 1
constexpr int kSmallBatchMax = 256;  // sequences per group up to which the fused step is the faster one at ANY capacity (scripts/small_latency.py)
__device__ __forceinline__ void small_update_body(const int b, double* __restrict__ x, double* __restrict__ P,
                                                  const double* __restrict__ f_Hx, const double* __restrict__ f_Hy,
                                                  const double* __restrict__ f_nu, const double* __restrict__ f_R,
                                                  const int* __restrict__ succ_idx, const int* __restrict__ m_count,
                                                  const int* __restrict__ n_slots, const int* __restrict__ part_i, int ppos, int pend,
                                                  int N, int ld, int* __restrict__ status, double* sAt) {
  __shared__ double sH[kSmallM][10];
  __shared__ double sR[kSmallM];
  __shared__ int sPos[kSmallM];
  __shared__ double sS[kSmallM][33];
  __shared__ double sLinv[kSmallM * kLinvPitch];
  __shared__ __attribute__((aligned(16))) double sCol[2][64];
  const int tid = threadIdx.x;
  const int cnt = m_count[b];
  if (cnt == 0) return;                               // monoslam.cpp:134: no successful measurement, no update (uniform)
  const int m = 2 * cnt;
  double* xb = x + (size_t)b * ld;
  double* Pb = P + (size_t)b * ld * ld;
  const int nlive = 13 + 3 * n_slots[b];
  const int n_c = nlive + (part_i[(size_t)b * kPartInts + kPartCount] ? pend - ppos : 0);      // compact columns that hold state
  if (n_c + 1 > kSmallW || m > kSmallM) {             // cannot happen: the host launches this kernel only under its bound on n_slots
    if (tid == 0) status[b] |= 8;                     // ... and if it ever did, the sequence says so instead of corrupting memory
    return;
  }
  const int W = (n_c + 1 <= 64) ? 64 : kSmallW;       // row pitch of the LDS panel; column W - 1 carries the innovation
  auto cmap = [&](int c) { return c < nlive ? c : ppos + (c - nlive); };
  if (tid < kSmallM) {
    if (tid < m) {
      const int f = succ_idx[(size_t)b * N + (tid >> 1)];
      const size_t fi = (size_t)b * N + f;
#pragma unroll
      for (int c = 0; c < 7; ++c) sH[tid][c] = f_Hx[fi * 14 + (tid & 1) * 7 + c];
#pragma unroll
      for (int c = 0; c < 3; ++c) sH[tid][7 + c] = f_Hy[fi * 6 + (tid & 1) * 3 + c];
      sR[tid] = f_R[fi];
      sPos[tid] = 13 + 3 * f;
      sAt[tid * W + W - 1] = f_nu[fi * 2 + (tid & 1)];         // the innovation rides along as the last column
    } else {
      sAt[tid * W + W - 1] = 0.0;
    }
  }
  // ---- A^T[a][i] = sum_c H[a][c] P[c][i]: thread (a0, i), rows a0, a0 + 256 / W, ...; the order of every sum is that of
  // k_build_AS
  const int i = tid % W, a0 = tid / W, astep = kSmallThreads / W;
  const int gi = cmap(i);
  double pc[7];
#pragma unroll
  for (int c = 0; c < 7; ++c) pc[c] = (i < n_c) ? ((i < 13) ? Pb[(size_t)i * ld + c] : Pb[(size_t)c * ld + gi]) : 0.0;
  __syncthreads();
  if (i < W - 1) {
    for (int a = a0; a < kSmallM; a += astep) {
      double acc = 0.0;
      if (a < m && i < n_c) {
        const int pos = sPos[a];
#pragma unroll
        for (int c = 0; c < 7; ++c) acc = __builtin_fma(pc[c], sH[a][c], acc);
#pragma unroll
        for (int c = 0; c < 3; ++c) acc = __builtin_fma(Pb[(size_t)(pos + c) * ld + gi], sH[a][7 + c], acc);
      }
      sAt[a * W + i] = acc;
    }
  }
  __syncthreads();
  SST(4, 0);
  // ---- S = H A + R (rows / columns beyond m: the identity), all 32 x 32 entries
  // (lanes run along a - the row of H, whose coefficients and pivot column differ per lane - and share bb, the row of the LDS
  // panel they read: one broadcast address per read.  With bb along the lanes every read was 32 rows of the panel at a pitch of
  // W doubles = the same bank 32 times.)
  for (int e = tid; e < kSmallM * kSmallM; e += kSmallThreads) {
    const int bb = e >> 5, a = e & 31;
    double v = (a == bb) ? 1.0 : 0.0;
    if (a < m && bb < m) {
      const double* arow = sAt + bb * W;
      const int pos = sPos[a];                        // (13 + 3 slot < nlive: the compact index of a feature column is the column)
      double acc = 0.0;
#pragma unroll
      for (int c = 0; c < 7; ++c) acc = __builtin_fma(sH[a][c], arow[c], acc);
#pragma unroll
      for (int c = 0; c < 3; ++c) acc = __builtin_fma(sH[a][7 + c], arow[pos + c], acc);
      if (a == bb) acc += sR[a];
      v = acc;
    }
    sS[a][bb] = v;
  }
  __syncthreads();
  SST(4, 1);
  // ---- S = L L^T and L^-T by one wavefront: [S; I] -> [L; L^-T], a row per lane (sl2_chol_diag.hpp)
  if (tid < 64) {
    const int r = tid & 31;
    const bool low = tid < 32;
    double a[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) a[c] = low ? sS[r][c] : ((c == r) ? 1.0 : 0.0);
    double* rows = low ? &sS[r][0] : &sLinv[r * kLinvPitch];
    DScal t;
    t.s = 0.0;
    d_column<0>(a, t, (unsigned)(size_t)(__attribute__((address_space(3))) double*)&sCol[0][0], &sCol[0][tid], rows);
    rows[31] = a[31];
  }
  __syncthreads();
  SST(4, 2);
  // ---- V = L^-1 A^T in place: a thread owns compact column i (the innovation column W - 1 included), the column in
  // registers; L^-1[k][p] = sLinv[p][k] (row p of L^-T), zero for p > k
  {
    // 256 / W threads per column: thread (h, i) forms the rows k = h, h + 256 / W, ... of column i (h is uniform in a wavefront,
    // so the L^-1 operand of a product stays one broadcast LDS read); the results stay in registers until every thread has
    // read its copy of the column
    const int hs = kSmallThreads / W, h = tid / W, ci = tid % W;
    const bool mine = ci < n_c || ci == W - 1;
    double col[kSmallM], out[kSmallM / 2];
#pragma unroll
    for (int p = 0; p < kSmallM; ++p) col[p] = mine ? sAt[p * W + ci] : 0.0;
#pragma unroll
    for (int k = 0; k < kSmallM; ++k) {
      if (k < m && (k & (hs - 1)) == h) {
        double acc = 0.0;
#pragma unroll
        for (int p = 0; p <= k; ++p) acc = __builtin_fma(sLinv[p * kLinvPitch + k], col[p], acc);
        out[k >> 1] = acc;                              // (hs >= 2: the rows k and k ^ 1 never belong to the same thread)
      }
    }
    __syncthreads();
    if (mine) {
#pragma unroll
      for (int k = 0; k < kSmallM; ++k)
        if (k < m && (k & (hs - 1)) == h) sAt[k * W + ci] = out[k >> 1];
    }
  }
  __syncthreads();
  SST(4, 3);
  // ---- P -= V^T V (both triangles; the same products in the same order on either side of the diagonal, so mirrored entries
  // stay equal bit for bit), 4 x 4 outputs per thread; x += V^T w with w = L^-1 nu = the last column of V
  {
    const int nt = (n_c + 3) >> 2;                    // (4 nt <= W: W is a multiple of four and n_c < W)
    for (int tile = tid; tile < nt * nt; tile += kSmallThreads) {
      const int ti = (tile / nt) * 4, tj = (tile % nt) * 4;
      double acc[4][4], pv[4][4];
#pragma unroll
      for (int q = 0; q < 16; ++q) {                    // the entries of P first: their round trip runs under the products
        const int ii = ti + (q >> 2), jj = tj + (q & 3);  // (requesting them before the factorisation instead was measured: the D wave's
        acc[q >> 2][q & 3] = 0.0;                         // routine 0.8 us longer, this phase 0.2 us shorter)
        pv[q >> 2][q & 3] = (ii < n_c && jj < n_c) ? Pb[(size_t)cmap(ii) * ld + cmap(jj)] : 0.0;
      }
#pragma unroll 4
      for (int k = 0; k < m; ++k) {                     // (unrolled: four k-steps' LDS reads in flight over the products of the step before)
        const double* vk = sAt + k * W;
        double vi[4], vj[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { vi[q] = vk[ti + q]; vj[q] = vk[tj + q]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q >> 2][q & 3] = __builtin_fma(vi[q >> 2], vj[q & 3], acc[q >> 2][q & 3]);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ii = ti + (q >> 2), jj = tj + (q & 3);
        if (ii < n_c && jj < n_c) Pb[(size_t)cmap(ii) * ld + cmap(jj)] = pv[q >> 2][q & 3] - acc[q >> 2][q & 3];
      }
    }
    // (the state's increment on the LAST n_c threads: when the tiles leave threads idle - maps of up to ~45 columns - it runs
    // beside them instead of behind them)
    const int xi = kSmallThreads - 1 - tid;
    if (xi < n_c) {
      double acc = 0.0;
      for (int k = 0; k < m; ++k) acc = __builtin_fma(sAt[k * W + xi], sAt[k * W + W - 1], acc);
      xb[cmap(xi)] += acc;
    }
  }
}

__global__ void __launch_bounds__(kSmallThreads) k_small_back(
    const int* __restrict__ srch_res, const int* __restrict__ srch_i, const uint8_t* __restrict__ patch, const double* __restrict__ f_h,
    const int* __restrict__ sel_idx, const int* __restrict__ n_sel, int* __restrict__ f_flags, double* __restrict__ f_z,
    double* __restrict__ f_nu, int* __restrict__ attempted, int* __restrict__ successful, int* __restrict__ meas_ok,
    double* __restrict__ meas_score, double* __restrict__ work, int* __restrict__ succ_idx, int* __restrict__ f_arow,
    int* __restrict__ m_count, const int* __restrict__ n_slots, const int* __restrict__ pos_err, const int* __restrict__ pos_err_any,
    int* __restrict__ f_hcol, const int* __restrict__ ps_i, int kpart, int ppos0, int N, int* __restrict__ srch_big,
    int* __restrict__ status,
    double* __restrict__ x, double* __restrict__ P, const double* __restrict__ f_Hx, const double* __restrict__ f_Hy,
    const double* __restrict__ f_R, const int* __restrict__ part_i, int pend, int ld,
    double* __restrict__ traj, int* __restrict__ traj_count, const double* __restrict__ last_r, double* __restrict__ pos_log,
    int* __restrict__ pos_count, int min_attempts, double match_fraction, int save_trajectory, int* __restrict__ slots_max,
    unsigned long long* __restrict__ slots_mail, int publish) {
  extern __shared__ double s_dynd[];                  // phase by phase: [N + 8] ints, [32][128] doubles, [2 N] ints
  const int b = blockIdx.x;
  SST(3, 0);
  // the bookkeeping stages' scratch sits behind their index arrays in the dynamic region (launch_small_back sizes it): with it
  // among the statics the kernel needed 56.4 KB at W = 128 - two workgroups per CU; 53.0 KB: three
  double* const s_ext_score = s_dynd + ((size_t)N + 8 + 1) / 2;
  double* const s_ext_final = s_dynd + (size_t)N;
  search_score_body<true>(b, srch_res, srch_i, patch, f_h, sel_idx, n_sel, f_flags, f_z, f_nu, attempted, successful, meas_ok, meas_score,
                          work, succ_idx, f_arow, m_count, n_slots, pos_err, pos_err_any, f_hcol, ps_i, kpart, ppos0, N, srch_big, status,
                          (int*)s_dynd, s_ext_score);
  __syncthreads();
  SST(3, 1);
  small_update_body(b, x, P, f_Hx, f_Hy, f_nu, f_R, succ_idx, m_count, n_slots, part_i, ppos0, pend, N, ld, status, s_dynd);
  __syncthreads();
  SST(3, 2);
  finalize_body<true>(b, x, P, f_flags, n_slots, attempted, successful, m_count, n_sel, traj, traj_count, last_r, status, pos_log,
                      pos_count, N, ld, min_attempts, match_fraction, save_trajectory, part_i, pend, (int*)s_dynd, slots_max, slots_mail,
                      publish, s_ext_final);
  SST(3, 3);
}

// Which stages of a sequence group's step are fused: 0 = none (ten launches), 1 = both sides of the search (three launches),
// 2 = the back side only (scoring + update + finalize in one launch, the front-end stages on their own: six launches).
// Static conditions: at most 16 features measured per frame - the innovation system is one 32 x 32 block - and no recorded
// feature position can be misplaced (Q28 needs two partially initialised features in flight).  Dynamic: the LIVE maps fit
// kSmallW columns - `slots_bound` = the host's upper bound on n_slots of any sequence (sl2_engine.hip: slots_upper_bound, exact
// at synchronised points, from the device's mailbox in between).  Then: everything fused when the group is small enough to be
// latency-bound or the capacity is large (the one-stage kernels work on all ld columns, the fused ones on the live ones:
// scripts/small_latency.py, a dozen features at capacity 128 - ld = 448 - fused is 1.2 x faster at one sequence and 1.8 x at
// 1024); at a small capacity and a large batch only the back side, which holds its own there (0.107 against 0.118 ms for the six
// stages it replaces at 1024 sequences, ld = 128) - k_small_front does not (0.065 against 0.038 ms: 304 registers, one workgroup
// per CU).
int small_step_mode(const sl2_engine* e, int slots_bound) {
  if (!e->root->step_fusion || e->mld != kSmallM || e->kpart != 1 || 13 + 3 * slots_bound + 6 * e->kpart + 1 > kSmallW) return 0;
  return (e->B <= kSmallBatchMax || e->ld >= 256 || e->root->step_fusion == 2) ? 1 : 2;
}

int launch_small_front(sl2_engine* e, int n) {
  LaunchScope ls(e, "k_small_front", true);
  if (n > e->nsel_max) n = e->nsel_max;
  const size_t shm = (size_t)e->N * (sizeof(double) + 3 * sizeof(int));
  hipLaunchKernelGGL(k_small_front, dim3(e->B), dim3(kSmallThreads), shm, e->stream, e->x, e->P, e->n_slots, e->prev_r, e->part_i,
                     e->ppos + 6 * e->kpart, e->ld, e->prm.delta_t, e->xp_org, e->f_flags, e->f_h, e->f_Hx, e->f_Hy, e->f_R, e->f_S,
                     e->f_score, e->srch_i, e->srch_d, e->cam, e->N, e->sel_idx, e->n_sel, e->n_vis, e->last_r, e->srch_sel, n,
                     e->srch_big, (e->srch_big && e->root->search_variant == 1) ? e->root->search_split : 0);
  SL2_HIP(hipGetLastError());
  return SL2_OK;
}

int launch_small_back(sl2_engine* e, int save_trajectory, int slots_bound) {
  LaunchScope ls(e, "k_small_back", true);
  // the LDS panel is [32][W], W = 64 while every live map of the group fits it (the kernel picks W from the sequence's own size,
  // which the bound bounds): a third workgroup per CU at large batches
  size_t shm = sizeof(double) * kSmallM * ((13 + 3 * slots_bound + 6 * e->kpart + 1 <= 64) ? 64 : kSmallW);
  // (the bookkeeping phases: [N + 8] ints + 16 x kWorkDoubles doubles, then [2 N] ints + 354 doubles)
  const size_t ints = sizeof(int) * (2 * (size_t)e->N + 10) + sizeof(double) * (16 + 169 + 169 + 16 * kWorkDoubles);
  if (ints > shm) shm = ints;
  hipLaunchKernelGGL(k_small_back, dim3(e->B), dim3(kSmallThreads), shm, e->stream, e->srch_res, e->srch_i, e->patch, e->f_h, e->sel_idx,
                     e->n_sel, e->f_flags, e->f_z, e->f_nu, e->attempted, e->successful, e->meas_ok, e->meas_score, e->work,
                     e->succ_idx, e->f_arow, e->m_count, e->n_slots, e->pos_err, e->pos_err_any, e->f_hcol, e->ps_i, e->kpart, e->ppos,
                     e->N, e->srch_big, e->status, e->x, e->P, e->f_Hx, e->f_Hy, e->f_R, e->part_i, e->ppos + 6 * e->kpart, e->ld,
                     e->traj, e->traj_count, e->last_r, e->pos_log, e->pos_count, e->prm.minimum_attempted_measurements_of_feature,
                     e->prm.successful_match_fraction, save_trajectory, e->root->slots_max_dev, e->root->slots_mail_dev,
                     e->group_first == 0 ? 1 : 0);
  SL2_HIP(hipGetLastError());
  return SL2_OK;
}

}  // namespace sl2

#ifdef SL2_FRONT_TRACE
extern "C" int sl2_debug_small_trace(long long* dev_buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(sl2::g_front_trace), &dev_buf, sizeof(dev_buf)) == hipSuccess ? 0 : 2;
}
#endif
