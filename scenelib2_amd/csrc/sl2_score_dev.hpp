// Device body of the scoring / bookkeeping stage that follows the patch search (k_search_score of sl2_search.hip; also the
// first phase of the fused small-map kernel k_small_back of sl2_small.hip).  make_measurements / measure_feature bookkeeping:
// monoslam.cpp:336-386, 479-496.  Called by ALL threads of the workgroup that owns sequence b.
#pragma once
#include "sl2_common.hpp"

namespace sl2 {

// One workgroup per sequence, one thread per selected position.  Besides the deferred FP64 scores and the reference's
// bookkeeping it compacts the successful measurements (succ_idx / m_count, what the EKF update reads: see the comment at
// the compaction) and leaves the step's work counters - a launch
// of its own for the compaction and a memset + atomics for the counters were 10 us of a single-sequence step.
template <bool kExtScratch = false>
__device__ __forceinline__ void search_score_body(const int b, const int* __restrict__ srch_res, const int* __restrict__ srch_i,
                                                       const uint8_t* __restrict__ patch, const double* __restrict__ f_h,
                                                       const int* __restrict__ sel_idx, const int* __restrict__ n_sel,
                                                       int* __restrict__ f_flags, double* __restrict__ f_z,
                                                       double* __restrict__ f_nu, int* __restrict__ attempted,
                                                       int* __restrict__ successful, int* __restrict__ meas_ok,
                                                       double* __restrict__ meas_score, double* __restrict__ work,
                                                       int* __restrict__ succ_idx, int* __restrict__ f_arow,
                                                       int* __restrict__ m_count, const int* __restrict__ n_slots,
                                                       const int* __restrict__ pos_err, const int* __restrict__ pos_err_any,
                                                       int* __restrict__ f_hcol, const int* __restrict__ ps_i, int kpart, int ppos0,
                                                       int N, int* __restrict__ srch_big, int* __restrict__ status, int* s_flag,
                                                       double* s_ext = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = (int)blockDim.x >> 6;
  // s_flag: [N + 8] successful measurement of slot i in this frame (dynamic LDS of the caller)
  __shared__ int s_wcnt[16];
  // (kExtScratch: the 16 x kWorkDoubles reduction buffer lives in the caller's LDS - k_small_back keeps its phases' scratch in
  // one place, so that a third workgroup fits a CU)
  double (*s_red)[kWorkDoubles];
  if constexpr (kExtScratch) {
    s_red = reinterpret_cast<double (*)[kWorkDoubles]>(s_ext);
  } else {
    __shared__ double s_red_own[16][kWorkDoubles];
    s_red = s_red_own;
  }
  const int ns = n_sel[b];
  // the step's list of large windows has been worked off by the search kernel: its counters and its length return to zero
  const int nunits_done = (b == 0 && srch_big) ? min(srch_big[0], kSrchBigUnits) : 0;
  const int nshared_done = (b == 0 && srch_big) ? srch_big[3] : 0;
  for (int i = tid; i < nunits_done; i += (int)blockDim.x) srch_big[kSrchBigDone + i] = 0;
  double w_win = 0.0, w_n = 0.0, w_cand = 0.0, w_fb = 0.0, w_tiles = 0.0;
  for (int i = tid; i < N; i += (int)blockDim.x) s_flag[i] = 0;
  __syncthreads();
  if (b == 0 && srch_big && tid == 0) {                         // (every thread has read them: the barrier above)
    srch_big[1] = nshared_done;                                 // sl2_get_step_work
    srch_big[0] = 0; srch_big[2] = 0; srch_big[3] = 0;
  }
  for (int k0 = 0; k0 < ns; k0 += (int)blockDim.x) {
    const int k = k0 + tid;
    if (k < ns) {
      const int f = sel_idx[(size_t)b * N + k];
      const size_t fi = (size_t)b * N + f;
      const int* o = srch_res + ((size_t)b * N + k) * 8;
      const int code = o[0];
      int ok = (o[7] & 2) ? 1 : 0;
      double score = meas_score[(size_t)b * N + k];
      if (code == 1) {   // deferred: the only candidate that can win; reference FP64 score + thresholds
        const unsigned* packed = (const unsigned*)(patch + fi * kPatchStride + kPatchPackedOffset);
        double sd0, sd1;
        score = ncc_score((int)packed[33], o[3], o[5], (int)packed[34], o[4], &sd0, &sd1);
        ok = (!(sd0 < kCorrelationSigmaThreshold) && !(sd1 < kCorrelationSigmaThreshold) && !(score > kCorrThresh2)) ? 1 : 0;
        meas_score[(size_t)b * N + k] = score;
      }
      meas_ok[(size_t)b * N + k] = ok;
      int fl = f_flags[fi];
      attempted[fi] += 1;
      if (ok) {
        successful[fi] += 1;
        const double h0 = f_h[fi * 2], h1 = f_h[fi * 2 + 1];
        f_z[fi * 2] = (double)o[1]; f_z[fi * 2 + 1] = (double)o[2];
        f_nu[fi * 2] = (double)o[1] - h0; f_nu[fi * 2 + 1] = (double)o[2] - h1;   // func_nui
        fl |= FF_SUCCESS;
        s_flag[f] = 1;
      } else {
        fl &= ~FF_SUCCESS;
      }
      f_flags[fi] = fl;
      const int* si = srch_i + fi * 8;
      w_win += (double)(2 * si[6] + 11) * (double)(2 * si[7] + 11);
      w_n += 1.0; w_cand += (double)o[6]; w_fb += (o[7] & 4) ? 1.0 : 0.0;
      w_tiles += (double)(((si[3] > 0 ? si[3] : 0) + 15) >> 4) * (double)(((si[5] > 0 ? si[5] : 0) + 15) >> 4);
    }
  }
  __syncthreads();
  // The successful measurements, compacted in SLOT order (= feature_list_ order).  The reference stacks them in
  // selected_feature_list_ order (construct_total_measurement_stuff, monoslam.cpp:548-572); any order of the rows of H gives
  // the same update, and with this one the features of a 64-column tile of P are consecutive rows of A^T and consecutive
  // rows / columns of S, which is what lets k_build_AS_tiles write whole blocks (sl2_ekf_update.hip).
  int base = 0;
  for (int i0 = 0; i0 < N; i0 += (int)blockDim.x) {
    const int i = i0 + tid;
    const bool ok = i < N && s_flag[i] != 0;
    const unsigned long long mask = __ballot(ok);
    if (lane == 0) s_wcnt[wave] = __popcll(mask);
    __syncthreads();
    int off = base, total = 0;
    for (int w = 0; w < nwave; ++w) {
      const int c = s_wcnt[w];
      if (w < wave) off += c;
      total += c;
    }
    const int rank = off + __popcll(mask & ((1ull << lane) - 1ull));
    if (ok) succ_idx[(size_t)b * N + rank] = i;
    if (i < N) f_arow[(size_t)b * N + i] = ok ? 2 * rank : -1;       // (what k_build_AS_tiles indexes by slot)
    base += total;
    __syncthreads();
  }
  if (tid == 0) m_count[b] = base;
  for (int off = 32; off > 0; off >>= 1) {
    w_win += __shfl_xor(w_win, off, 64); w_n += __shfl_xor(w_n, off, 64);
    w_cand += __shfl_xor(w_cand, off, 64); w_fb += __shfl_xor(w_fb, off, 64); w_tiles += __shfl_xor(w_tiles, off, 64);
  }
  if (lane == 0) { s_red[wave][0] = w_win; s_red[wave][1] = w_n; s_red[wave][2] = w_cand; s_red[wave][3] = w_fb; s_red[wave][4] = w_tiles; }
  __syncthreads();
  if (tid < kWorkDoubles) {
    double acc = 0.0;
    for (int w = 0; w < nwave; ++w) acc += s_red[w][tid];
    work[b * kWorkDoubles + tid] = acc;
  }
  // Q28 (feature.cpp:254, monoslam.cpp:564): a sequence in which a feature's recorded position_in_total_state_vector_ lies
  // below its true one gets, per slot, the engine column its dh_by_dy block therefore lands on - the reference's position
  // minus the error, looked up in the reference's state order (feature_list_ order, partial features with six states)
  if (pos_err_any[b] && tid == 0) {
    int* chunk_col = s_flag;                   // (s_flag has done its work) [N + 8]: the three-state chunks of the reference's state
    const int ns = n_slots[b];
    const int* psb = ps_i + (size_t)b * kpart * kPsInts;
    int c = 0;
    for (int f = 0; f < ns && c < N + 6; ++f) {
      const int fl = f_flags[(size_t)b * N + f];
      if (fl & FF_ACTIVE) chunk_col[c++] = 13 + 3 * f;
      else if (fl & FF_PARTIAL) {
        int ks = 0;
        for (int k = 0; k < kpart; ++k) if (psb[k * kPsInts + kPsActive] && psb[k * kPsInts + kPsLabel] == f) ks = k;
        chunk_col[c++] = ppos0 + 6 * ks;
        chunk_col[c++] = ppos0 + 6 * ks + 3;
      }
    }
    c = 0;
    for (int f = 0; f < ns; ++f) {
      const int fl = f_flags[(size_t)b * N + f];
      if (!(fl & (FF_ACTIVE | FF_PARTIAL))) continue;
      const int t = c - pos_err[(size_t)b * N + f] / 3;
      // t < 0: inside the vehicle state, same columns here (k_build_AS reproduces the overwrite of dh_by_dxv below column 7);
      // t <= -5 is a NEGATIVE column - the reference's block() write is out of bounds there (undefined behaviour): the
      // column is held at 0 for memory safety and the sequence is flagged
      const int hc = 13 + 3 * t;
      f_hcol[(size_t)b * N + f] = t >= 0 ? chunk_col[t < N + 7 ? t : N + 7] : (hc > 0 ? hc : 0);
      if (t < 0 && hc < 0) status[b] |= 4;
      c += (fl & FF_PARTIAL) ? 2 : 1;
    }
  }
}

}  // namespace sl2
