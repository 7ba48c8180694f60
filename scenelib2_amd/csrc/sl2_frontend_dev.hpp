// Device bodies of the front-end stages of the per-frame step (all FP64), shared by the one-stage-per-launch kernels of
// sl2_frontend.hip and by the fused small-map kernels of sl2_small.hip:
//   predict_body            Kalman::KalmanFilterPredict          kalman.cpp:50-69
//   feature_prediction_body predict_single_feature_measurements   monoslam.cpp:289-308
//                           + visibility_test + selection_score   full_feature_model.cpp:103-176
//   select_body             auto_select_n_features (ordering)     monoslam.cpp:187-254
//   finalize_body           normalise_state, delete_bad_features, symmetrise,
//                           trajectory_store_                      monoslam.cpp:137-177,616-703
// Every body is called by ALL threads of a workgroup that owns sequence b (they contain barriers); dynamic LDS is passed in.
// Data layout: see sl2_common.hpp (dense P[B][ld][ld], x[B][ld]).
#pragma once
#include "sl2_common.hpp"

namespace sl2 {

#ifdef SL2_FRONT_TRACE
static __device__ long long* g_front_trace = nullptr;        // development only: 8 stamps per workgroup and row; one copy per translation unit, each with its own setter
#define FTR(kern, slot) do { if (g_front_trace && threadIdx.x == 0) g_front_trace[((size_t)(kern) * 4096 + blockIdx.x) * 8 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define FTR(kern, slot) do { } while (0)
#endif

// ---------------------------------------------------------------------------
// k_predict: one workgroup per sequence.  Only the first 13 rows/cols of P change
// (static map): Pxx <- (F Pxx) F^T + Q, strip P[0:13, j] <- F P[0:13, j], mirrored.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void predict_body(const int b, double* __restrict__ x, double* __restrict__ P, const int* __restrict__ n_slots,
                                                 double* __restrict__ prev_r, const int* __restrict__ part_i, int pend, int ld,
                                                 double dt) {
  const int tid = threadIdx.x;
  double* xb = x + (size_t)b * ld;
  double* Pb = P + (size_t)b * ld * ld;
  FTR(0, 0);
  if (tid < 3) prev_r[b * 3 + tid] = xb[tid];   // prev_xp_pos (monoslam.cpp:121-124); xb is rewritten at the very end
  // columns of the map: the 3-D features, and the six states of every partial slot (pend = ppos + 6 kpart) while any is in use
  const int n_used = part_i[(size_t)b * kPartInts + kPartCount] ? pend : 13 + 3 * n_slots[b];
  // the first batch of strip columns is fetched now: its memory latency hides behind the serial motion model
  double v0[13];
  for (int k = 0; k < 13; ++k) v0[k] = (13 + tid < n_used) ? Pb[(size_t)k * ld + 13 + tid] : 0.0;
  __shared__ double s_f[13], s_A[16], s_B[12], s_P[169], s_T[169];
  if (tid == 0) {
    double xv[13];
    for (int i = 0; i < 13; ++i) xv[i] = xb[i];
    double f[13], A44[16], B43[12];
    motion_f_and_blocks(xv, dt, f, A44, B43);
    for (int i = 0; i < 13; ++i) s_f[i] = f[i];
    for (int i = 0; i < 16; ++i) s_A[i] = A44[i];
    for (int i = 0; i < 12; ++i) s_B[i] = B43[i];
  }
  for (int e = tid; e < 169; e += blockDim.x) s_P[e] = Pb[(size_t)(e / 13) * ld + (e % 13)];
  __syncthreads();
  FTR(0, 1);
  for (int e = tid; e < 169; e += blockDim.x) {
    const int i = e / 13, j = e % 13;
    double v[13];
    for (int k = 0; k < 13; ++k) v[k] = s_P[k * 13 + j];
    s_T[e] = frow_dot(i, dt, s_A, s_B, v);
  }
  __syncthreads();
  FTR(0, 2);
  for (int e = tid; e < 169; e += blockDim.x) {
    const int i = e / 13, j = e % 13;
    double v[13];
    for (int k = 0; k < 13; ++k) v[k] = s_T[i * 13 + k];
    Pb[(size_t)i * ld + j] = frow_dot(j, dt, s_A, s_B, v) + process_noise_entry(i, j, dt, s_B);
  }
  FTR(0, 3);
  // The mirrored copy P[j][0..12] goes through LDS so that the 13 entries of a row leave in ONE store instruction
  // (16 lanes per row): written straight from the column owner they were 13 scattered 8-byte stores per row and
  // the kernel spent three quarters of its time on them.
  __shared__ double s_W[256][13];
  for (int j0 = 13; j0 < n_used; j0 += blockDim.x) {
    const int j = j0 + tid;
    if (j < n_used) {
      double v[13], w[13];
      for (int k = 0; k < 13; ++k) v[k] = (j0 == 13) ? v0[k] : Pb[(size_t)k * ld + j];
      for (int i = 0; i < 13; ++i) w[i] = frow_dot(i, dt, s_A, s_B, v);
      for (int i = 0; i < 13; ++i) {
        Pb[(size_t)i * ld + j] = w[i];
        s_W[tid][i] = w[i];
      }
    }
    __syncthreads();
    const int rr = tid >> 4, cc = tid & 15;
    for (int r0 = 0; r0 < (int)blockDim.x; r0 += blockDim.x / 16) {
      const int jr = j0 + r0 + rr;
      if (cc < 13 && jr < n_used) Pb[(size_t)jr * ld + cc] = s_W[r0 + rr][cc];
    }
    __syncthreads();
  }
  if (tid < 13) xb[tid] = s_f[tid];
  FTR(0, 4);
}

// ---------------------------------------------------------------------------
// k_feature_prediction: one thread per (sequence, feature slot).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void feature_prediction_body(const int b, const int i, const double* __restrict__ x, const double* __restrict__ P,
                                                           const double* __restrict__ xp_org, int* __restrict__ f_flags,
                                                           const int* __restrict__ n_slots, double* __restrict__ f_h,
                                                           double* __restrict__ f_Hx, double* __restrict__ f_Hy,
                                                           double* __restrict__ f_R, double* __restrict__ f_S,
                                                           double* __restrict__ f_score, int* __restrict__ srch_i, double* __restrict__ srch_d,
                                                           CameraParams cam, int N, int ld) {
  if (i >= n_slots[b]) return;
  const size_t fi = (size_t)b * N + i;
  int flags = f_flags[fi] & ~(FF_SELECTED | FF_VISIBLE);
  if (!(flags & FF_ACTIVE)) { f_flags[fi] = flags; return; }
  const double* xb = x + (size_t)b * ld;
  const double* Pb = P + (size_t)b * ld * ld;
  double xp[7], y[3], xo[7];
  for (int k = 0; k < 7; ++k) xp[k] = xb[k];
  const int pos = 13 + 3 * i;
  for (int k = 0; k < 3; ++k) y[k] = xb[pos + k];
  for (int k = 0; k < 7; ++k) xo[k] = xp_org[fi * 8 + k];
  double zeroed[3], h[2], Hx[14], Hy[6], Rn;
  measurement_model(cam, xp, y, zeroed, h, Hx, Hy, &Rn);
  double Pxx7[49], Pxy7[21], Pyy[9], S[4];
  for (int r = 0; r < 7; ++r) {
    for (int c = 0; c < 7; ++c) Pxx7[r * 7 + c] = Pb[(size_t)r * ld + c];
    for (int c = 0; c < 3; ++c) Pxy7[r * 3 + c] = Pb[(size_t)r * ld + pos + c];
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Pyy[r * 3 + c] = Pb[(size_t)(pos + r) * ld + pos + c];
  innovation_cov(Hx, Hy, Rn, Pxx7, Pxy7, Pyy, S);
  const int cant_see = visibility_test(cam, xp, y, xo, h);
  f_h[fi * 2 + 0] = h[0]; f_h[fi * 2 + 1] = h[1];
  for (int k = 0; k < 14; ++k) f_Hx[fi * 14 + k] = Hx[k];
  for (int k = 0; k < 6; ++k) f_Hy[fi * 6 + k] = Hy[k];
  f_R[fi] = Rn;
  for (int k = 0; k < 4; ++k) f_S[fi * 4 + k] = S[k];
  f_score[fi] = S[0] + S[3];  // trace (selection_score, full_feature_model.cpp:172-176)
  if (cant_see == 0) flags |= FF_VISIBLE;
  f_flags[fi] = flags;
  // search window of this feature (measure_feature + the head of elliptical_search,
  // monoslam.cpp:371-374, 416-439), so that the search kernel starts from a descriptor
  {
    double a, bq, c;
    sinv_from_S(S, &a, &bq, &c);
    const SearchBounds sb = search_bounds(h, a, bq, c, cam.width, cam.height);
    int* si = srch_i + fi * 8;
    si[0] = sb.ucentre; si[1] = sb.vcentre; si[2] = sb.urelstart; si[3] = sb.urelfinish - sb.urelstart + 1;
    si[4] = sb.vrelstart; si[5] = sb.vrelfinish - sb.vrelstart + 1; si[6] = sb.halfwidth; si[7] = sb.halfheight;
    double* sd = srch_d + fi * 4;
    sd[0] = a; sd[1] = bq; sd[2] = c; sd[3] = 0.0;
  }
}

// ---------------------------------------------------------------------------
// k_select: one workgroup per sequence.  The reference's descending insertion
// (strict '>' => ties keep list order) is a stable sort by score; a feature's
// position is its rank = #{visible j : score_j > score_i or (== and j < i)}.
// Selection stops at the first zero score or after n (monoslam.cpp:241-249).
// Also records rRES_ as left by the last visibility_test (Q12).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void select_body(const int b, const double* __restrict__ f_score, int* __restrict__ f_flags,
                                                const int* __restrict__ n_slots, const double* __restrict__ xp_org,
                                                int* __restrict__ sel_idx, int* __restrict__ n_sel, int* __restrict__ n_vis,
                                                double* __restrict__ last_r, const int* __restrict__ srch_i,
                                                const double* __restrict__ srch_d, int* __restrict__ srch_sel, int N,
                                                int n_want, int* __restrict__ srch_big, int split_bands, double* s_dyn) {
  double* s_score = s_dyn;                 // [N]
  int* s_vis = (int*)(s_dyn + N);          // [N]
  int* s_nu = s_vis + N;                   // [N] rank accumulator
  __shared__ int s_nvis, s_zero_rank, s_last;
  const int tid = threadIdx.x;
  FTR(1, 0);
  const int ns = n_slots[b];
  if (tid == 0) { s_nvis = 0; s_zero_rank = 0x7fffffff; s_last = -1; }
  __syncthreads();
  // (wave-level reductions, one LDS atomic per wavefront: a hundred threads hitting one LDS word one after the other cost
  // 2.5 us each time at batch 1)
  for (int i0 = 0; i0 < ns; i0 += blockDim.x) {
    const int i = i0 + tid;
    int last = -1;
    if (i < ns) {
      const int fl = f_flags[(size_t)b * N + i];
      s_score[i] = f_score[(size_t)b * N + i];
      s_vis[i] = (fl & FF_VISIBLE) ? 1 : 0;
      if (fl & FF_ACTIVE) last = i;
    }
    for (int off = 32; off > 0; off >>= 1) last = max(last, __shfl_xor(last, off, 64));
    if ((tid & 63) == 0 && last >= 0) atomicMax(&s_last, last);
  }
  __syncthreads();
  FTR(1, 1);
  // scores are traces of covariances (>= 0): an invisible feature takes the key -1 and never outranks anyone, so the
  // rank loop has no branch and its LDS reads pipeline
  // (a NaN score - the omega == 0 hazard, Q10 - compares false both ways: it takes the key -0.5, so that the ranks stay a
  // permutation and an all-NaN map keeps list order, which is what the reference's insertion does with it)
  for (int i = tid; i < ns; i += blockDim.x) {
    if (!s_vis[i]) s_score[i] = -1.0;
    else if (s_score[i] != s_score[i]) s_score[i] = -0.5;
    s_nu[i] = 0;                            // rank accumulator
  }
  __syncthreads();
  // rank = number of features that come before i.  With fewer features than threads the j range is split over
  // blockDim.x / roundup(ns, 64) groups of threads (up to four) that add their partial counts: the hundred-step loop was
  // 9 000 of this kernel's 15 000 cycles at 100 features.
  {
    const int G = (ns + 63) / 64 * 64;
    int P = G > 0 ? (int)blockDim.x / G : 1;
    P = P < 1 ? 1 : (P > 4 ? 4 : P);
    const int part = P > 1 ? tid / G : 0;
    for (int i = P > 1 ? tid - part * G : tid; i < ns && part < P; i += P > 1 ? ns : (int)blockDim.x) {
      if (!s_vis[i]) continue;
      const double si = s_score[i];
      const int j0 = part * ns / P, j1 = (part + 1) * ns / P;
      int rank = 0;
#pragma unroll 8
      for (int j = j0; j < j1; ++j) {
        const double sj = s_score[j];
        rank += (sj > si || (sj == si && j < i)) ? 1 : 0;
      }
      if (P > 1) atomicAdd(&s_nu[i], rank); else s_nu[i] = rank;
    }
  }
  __syncthreads();
  for (int i0 = 0; i0 < ns; i0 += blockDim.x) {
    const int i = i0 + tid;
    const bool vis = i < ns && s_vis[i];
    int zero_rank = 0x7fffffff;
    if (vis) {
      const int rank = s_nu[i];
      s_vis[i] = 1 + rank;  // store rank+1
      if (s_score[i] == 0.0) zero_rank = rank;
    }
    const int nv = __popcll(__ballot(vis));
    for (int off = 32; off > 0; off >>= 1) zero_rank = min(zero_rank, __shfl_xor(zero_rank, off, 64));
    if ((tid & 63) == 0) {
      if (nv) atomicAdd(&s_nvis, nv);
      if (zero_rank != 0x7fffffff) atomicMin(&s_zero_rank, zero_rank);
    }
  }
  __syncthreads();
  FTR(1, 2);
  int limit = n_want;
  if (s_zero_rank < limit) limit = s_zero_rank;
  if (s_nvis < limit) limit = s_nvis;
  for (int i = tid; i < ns; i += blockDim.x) {
    if (!s_vis[i]) continue;
    const int rank = s_vis[i] - 1;
    if (rank < limit) {
      sel_idx[(size_t)b * N + rank] = i;
      f_flags[(size_t)b * N + i] |= FF_SELECTED;
      const int* si = srch_i + ((size_t)b * N + i) * 8;
      // the selected position's search record in ONE 64-byte line: slot, window, PuInv (what the search kernel reads)
      int* rec = srch_sel + ((size_t)b * N + rank) * 16;
      rec[0] = i;
#pragma unroll
      for (int q = 0; q < 7; ++q) rec[1 + q] = si[q];
      const double* sd = srch_d + ((size_t)b * N + i) * 4;
      double* recd = (double*)(rec + 8);
      recd[0] = sd[0]; recd[1] = sd[1]; recd[2] = sd[2];
      // A window of many bands (a poorly constrained feature: up to the whole frame) would keep ONE wavefront of the search
      // kernel busy long after every other one has finished: it is cut into units of a few bands that go on the step's list,
      // and the trailing workgroups of the search launch take units (sl2_search.hip, m4_big_windows).  The record then carries
      // nu = kSrchSharedNu - an empty window for the position's own wavefront, which leaves the result alone - and the true
      // width in rec[14].
      int shared = 0;
      if (split_bands > 0 && si[3] > 0 && si[5] > 0) {
        const int TU = (si[3] + 15) >> 4, TV = (si[5] + 15) >> 4;
        const int bands = ((TU + 1) >> 1) * TV;
        if (bands >= split_bands && TU <= 128 && TV <= 64) {            // (kM4MaxTU / kM4MaxTV: beyond them the exact walk)
          const int per = srch_unit_bands(bands), units = (bands + per - 1) / per;
          const int u0 = atomicAdd(srch_big, units);
          if (u0 + units <= kSrchBigUnits) {
            int4 en;
            en.x = b; en.y = rank; en.z = u0; en.w = units;
            for (int q = 0; q < units; ++q) *(int4*)(srch_big + kSrchBigEntries + 4 * (u0 + q)) = en;
            atomicAdd(srch_big + 3, 1);
            shared = 1;
          } else {
            // no room: the window stays with its own wavefront; what was allocated below the capacity reads as "nothing"
            int4 en;
            en.x = -1; en.y = 0; en.z = 0; en.w = 0;
            for (int q = u0; q < min(u0 + units, kSrchBigUnits); ++q) *(int4*)(srch_big + kSrchBigEntries + 4 * q) = en;
          }
        }
      }
      rec[14] = shared ? si[3] : 0;
      if (shared) rec[4] = kSrchSharedNu;
    }
  }
  FTR(1, 3);
  if (tid == 0) {
    n_sel[b] = limit;
    n_vis[b] = s_nvis;
    if (s_last >= 0)
      for (int k = 0; k < 3; ++k) last_r[b * 3 + k] = xp_org[((size_t)b * N + s_last) * 8 + k];
  }
}

// ---------------------------------------------------------------------------
// k_finalize: one workgroup per sequence.
//  (1) if an update happened (m > 0): normalise_state — Pxx <- (Jn Pxx) Jn^T,
//      strip rows 3..6 <- N * strip rows 3..6 (xv itself unchanged, Q9);
//  (2) delete_bad_features with the reference's skip-after-erase iteration (Q27):
//      a deleted feature's rows/cols of P are zeroed and its slot deactivated —
//      arithmetically identical to removing them;
//  (3) symmetrise: only the 13x13 block can be asymmetric in this layout (every
//      other block is stored mirrored), P <- P*0.5 + P^T*0.5;
//  (4) trajectory_store_ push (stale rRES_, Q12); NaN check.
// ---------------------------------------------------------------------------
template <bool kExtScratch = false>
__device__ __forceinline__ void finalize_body(const int b, double* __restrict__ x, double* __restrict__ P, int* __restrict__ f_flags,
                                                  const int* __restrict__ n_slots, int* __restrict__ attempted,
                                                  int* __restrict__ successful, const int* __restrict__ m_count,
                                                  const int* __restrict__ n_sel, double* __restrict__ traj,
                                                  int* __restrict__ traj_count, const double* __restrict__ last_r,
                                                  int* __restrict__ status, double* __restrict__ pos_log, int* __restrict__ pos_count, int N, int ld,
                                                  int min_attempts, double match_fraction, int save_trajectory,
                                                  const int* __restrict__ part_i, int pend, int* s_del,
                                              int* __restrict__ slots_max = nullptr, unsigned long long* __restrict__ slots_mail = nullptr,
                                              int publish = 0, double* s_ext = nullptr) {
  // s_del: [N] slots deleted this frame, then [N] flags
  // (kExtScratch: the 16 + 169 + 169 doubles below live in the caller's LDS, see search_score_body)
  double *s_N, *s_P, *s_T;
  if constexpr (kExtScratch) {
    s_N = s_ext; s_P = s_ext + 16; s_T = s_ext + 16 + 169;
  } else {
    __shared__ double s_N_own[16], s_P_own[169], s_T_own[169];
    s_N = s_N_own; s_P = s_P_own; s_T = s_T_own;
  }
  __shared__ int s_ndel;
  const int tid = threadIdx.x;
  double* xb = x + (size_t)b * ld;
  double* Pb = P + (size_t)b * ld * ld;
  const int ns = n_slots[b];
  const int n_used = part_i[(size_t)b * kPartInts + kPartCount] ? pend : 13 + 3 * ns;
  const bool updated = (n_sel[b] > 0) && (m_count[b] > 0);
  // Everything this kernel reads that does not depend on its own results is requested NOW, in one round trip: the vehicle
  // block, this thread's strip columns (rows 3..6), its feature's flags and counters.  (Phase by phase the kernel was a
  // chain of six dependent round trips: 16 us for a single sequence.)
  constexpr int kStripCols = 6;                       // strip columns per thread: blockDim.x * 6 >= 768 > any n_used - 13 here
  double pre_P[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = tid + q * (int)blockDim.x;
    pre_P[q] = (e < 169) ? Pb[(size_t)(e / 13) * ld + (e % 13)] : 0.0;
  }
  const bool strip_in_regs = updated && (n_used - 13 <= kStripCols * (int)blockDim.x);
  double pre_v[kStripCols][4];
  if (strip_in_regs) {
#pragma unroll
    for (int q = 0; q < kStripCols; ++q) {
      const int j = 13 + tid + q * (int)blockDim.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) pre_v[q][k] = (j < n_used) ? Pb[(size_t)(3 + k) * ld + j] : 0.0;
    }
  }
  int pre_fl = 0, pre_att = 0, pre_suc = 0;
  if (tid < ns) { const size_t fi = (size_t)b * N + tid; pre_fl = f_flags[fi]; pre_att = attempted[fi]; pre_suc = successful[fi]; }
  // ... and what thread 0 needs for the trajectory push and the log at the very end (xv is not changed by this kernel, Q9)
  double pre_x[13], pre_lr[3];
  int pre_tc = 0, pre_pc = 0;
  if (tid == 0) {
    for (int k = 0; k < 13; ++k) pre_x[k] = xb[k];
    for (int k = 0; k < 3; ++k) pre_lr[k] = last_r[b * 3 + k];
    pre_tc = traj_count[b]; pre_pc = pos_count[b];
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = tid + q * (int)blockDim.x;
    if (e < 169) s_P[e] = pre_P[q];
  }
  if (updated) {
    if (tid == 0) {
      double q[4] = {pre_x[3], pre_x[4], pre_x[5], pre_x[6]}, Nn[16];
      dqnorm_by_dq(q, Nn);
      for (int i = 0; i < 16; ++i) s_N[i] = Nn[i];
    }
    __syncthreads();
    // T = Jn * Pxx
    for (int e = tid; e < 169; e += blockDim.x) {
      const int i = e / 13, j = e % 13;
      double v;
      if (i >= 3 && i < 7) {
        v = 0.0;
        for (int k = 0; k < 4; ++k) v += s_N[(i - 3) * 4 + k] * s_P[(3 + k) * 13 + j];
      } else v = s_P[e];
      s_T[e] = v;
    }
    __syncthreads();
    // Pxx = T * Jn^T
    for (int e = tid; e < 169; e += blockDim.x) {
      const int i = e / 13, j = e % 13;
      double v;
      if (j >= 3 && j < 7) {
        v = 0.0;
        for (int k = 0; k < 4; ++k) v += s_T[i * 13 + 3 + k] * s_N[(j - 3) * 4 + k];
      } else v = s_T[e];
      s_P[e] = v;
    }
    // strip
    if (strip_in_regs) {
#pragma unroll
      for (int q = 0; q < kStripCols; ++q) {
        const int j = 13 + tid + q * (int)blockDim.x;
        if (j < n_used) {
          double w[4];
          for (int a = 0; a < 4; ++a) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc += s_N[a * 4 + k] * pre_v[q][k];
            w[a] = acc;
          }
          for (int a = 0; a < 4; ++a) {
            Pb[(size_t)(3 + a) * ld + j] = w[a];
            Pb[(size_t)j * ld + 3 + a] = w[a];
          }
        }
      }
    } else {
      for (int j = 13 + tid; j < n_used; j += blockDim.x) {
        double v[4], w[4];
        for (int k = 0; k < 4; ++k) v[k] = Pb[(size_t)(3 + k) * ld + j];
        for (int a = 0; a < 4; ++a) {
          double acc = 0.0;
          for (int k = 0; k < 4; ++k) acc += s_N[a * 4 + k] * v[k];
          w[a] = acc;
        }
        for (int a = 0; a < 4; ++a) {
          Pb[(size_t)(3 + a) * ld + j] = w[a];
          Pb[(size_t)j * ld + 3 + a] = w[a];
        }
      }
    }
    __syncthreads();
  } else {
    __syncthreads();
  }
  // (2) deletion bookkeeping.  The scheduling test is per feature (all threads); the list walk with its
  // skip-after-erase rule is serial like the reference's, but over LDS (a serial walk over global memory
  // cost one memory round trip per feature: 50 us of this kernel's 60).
  for (int i = tid; i < ns; i += blockDim.x) {
    const size_t fi = (size_t)b * N + i;
    const bool first = i == tid;                                  // this thread's first feature came with the prefetch
    int fl = first ? pre_fl : f_flags[fi];
    if (fl & FF_ACTIVE) {
      const int att = first ? pre_att : attempted[fi], suc = first ? pre_suc : successful[fi];
      if (att >= min_attempts && double(suc) / double(att) < match_fraction) { fl |= FF_SCHEDULED; f_flags[fi] = fl; }
    }
    s_del[N + i] = fl;      // second half of the dynamic LDS: this frame's flags
  }
  __syncthreads();
  // The reference walks feature_list_ and, after erasing a feature, skips the one that follows it (Q27: the iterator is
  // advanced before vector::erase): among the ACTIVE features in list order, del(k) = scheduled(k) && !del(k - 1) - in a
  // run of consecutive scheduled features the 1st, 3rd, 5th ... go.  Every thread decides its own feature from the length
  // of the scheduled run that ends just before it (runs are a few features at most); walked by one thread over LDS the
  // list cost 6 us of a single-sequence step.
  if (tid == 0) s_ndel = 0;
  __syncthreads();
  for (int i = tid; i < ns; i += blockDim.x) {
    const int fl = s_del[N + i];
    if ((fl & FF_ACTIVE) && (fl & FF_SCHEDULED)) {
      int run = 0;                                   // scheduled active features directly before i (inactive slots do not count)
      for (int j = i - 1; j >= 0; --j) {
        const int fj = s_del[N + j];
        if (!(fj & FF_ACTIVE)) continue;
        if (!(fj & FF_SCHEDULED)) break;
        ++run;
      }
      if ((run & 1) == 0) {
        f_flags[(size_t)b * N + i] = FF_USED;        // inactive, deselected
        s_del[atomicAdd(&s_ndel, 1)] = i;            // (the order of the list does not matter: rows / columns are zeroed)
      }
    }
  }
  __syncthreads();
  for (int d = 0; d < s_ndel; ++d) {
    const int pos = 13 + 3 * s_del[d];
    for (int j = tid; j < n_used; j += blockDim.x)
      for (int r = 0; r < 3; ++r) {
        Pb[(size_t)(pos + r) * ld + j] = 0.0;
        Pb[(size_t)j * ld + pos + r] = 0.0;
      }
  }
  // (3) symmetrise the vehicle block
  for (int e = tid; e < 169; e += blockDim.x) {
    const int i = e / 13, j = e % 13;
    Pb[(size_t)i * ld + j] = s_P[i * 13 + j] * 0.5 + s_P[j * 13 + i] * 0.5;
  }
  if (tid == 0) {
    if (save_trajectory) {
      const int c = pre_tc;
      double* t = traj + ((size_t)b * kTrajCapacity + (c % kTrajCapacity)) * 3;
      for (int k = 0; k < 3; ++k) t[k] = pre_lr[k];
      traj_count[b] = c + 1;
    }
    const int log_slot = pre_pc % kTrajCapacity;   // device-side step counter: the launch carries no per-step argument
    pos_count[b] = pre_pc + 1;
    for (int k = 0; k < 3; ++k) pos_log[((size_t)b * kTrajCapacity + log_slot) * 3 + k] = pre_x[k];
    bool bad = false;
    for (int k = 0; k < 13; ++k) bad = bad || !isfinite(pre_x[k]) || !isfinite(s_P[k * 13 + k]);   // (Q10 poisons Pxx first)
    if (bad) status[b] |= 1;
    // How large the maps of the batch are, for the HOST's choice of step kernels (sl2_small.hip) without a synchronisation:
    // slots_max[t & 1] collects the maximum of n_slots over the sequences that finalize step t; the first workgroup of the
    // first sequence group, when it finalizes step t, publishes the COMPLETE maximum of step t - 1 (every finalize of that
    // step has ended: stream order) to pinned host memory as (t << 32 | max) and clears that slot for step t + 1.
    if (slots_max) {
      // (a plain read first: the maps of a batch are mostly the same size, and 1024 atomics on one word were 8 us of this kernel)
      if (ns > __hip_atomic_load(&slots_max[pre_pc & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&slots_max[pre_pc & 1], ns);
      if (publish && b == 0) {
        const int prev = __hip_atomic_load(&slots_max[(pre_pc + 1) & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&slots_max[(pre_pc + 1) & 1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (relaxed: the word is the whole message - a release here is a write-back of the L2, buffer_wbl2, in the middle of the kernel)
        __hip_atomic_store(slots_mail, ((unsigned long long)(unsigned)pre_pc << 32) | (unsigned)prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}


}  // namespace sl2
