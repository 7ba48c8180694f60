// EKF update — Kalman::KalmanFilterUpdate (kalman.cpp:72-119) for the whole batch.
//
// Reference algebra:  S = H P H^T + R ; W = P H^T S^-1 ; x += W nu ; P -= W S W^T.
// Executed here as (mathematically identical, FP64 throughout):
//   A  = P H^T                     sparse rows of H: 7 pose + 3 feature columns (a6/a14)
//   S  = H A + R = L L^T           blocked Cholesky, 32x32 blocks
//   V  = A L^-T                    blocked forward substitution (one pass over A)
//   P -= V V^T ; x += V (L^-1 nu)  SYRK; nu rides along as an extra column of A
// because W S W^T = A S^-1 A^T = V V^T and W nu = V L^-1 nu.
//
// All dense operands are stored "k-major" (XT[k][i]: the contraction index is the
// slow one) so that every FP64 MFMA fragment (v_mfma_f64_16x16x4_f64: A[i=l&15]
// [k=l>>4], B[k=l>>4][j=l&15], D[row=(l>>4)+4r][col=l&15]) is a 128-byte
// coalesced row segment per 16 lanes, and the D fragment of one product is
// directly the B fragment (k-step s == register s) of the next.
#include "sl2_common.hpp"
#include "sl2_chol_diag.hpp"

namespace sl2 {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4d mfma_f64(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// k_build_A: At[a][i] = (P H^T)[i][a], a = 2j+r for the j-th successful feature.
// Column ld-1 carries the innovation nu (so that L^-1 nu and W nu fall out of the
// same substitution / SYRK).  Rows of padding up to a multiple of 32 are zeroed.
// One workgroup per sequence; a thread keeps the 7 pose entries of its column in
// registers and loops over the features (3 coalesced row reads of P + 2 coalesced row
// writes of At per feature, each a full contiguous row for the workgroup).
// ---------------------------------------------------------------------------
#ifdef SL2_TESTING   // section 1 of sl2_ekf_update_testing.inc: maps beyond what k_build_AS takes are not built by the product (sl2_create rejects them): TEST build only (SL2_BUILD_VARIANT=0)
#define SL2_EKF_TEST_SECTION 1
#include "sl2_ekf_update_testing.inc"
#undef SL2_EKF_TEST_SECTION
#endif  // SL2_TESTING

// ---------------------------------------------------------------------------
// k_build_AS: k_build_A and k_build_S in one pass (state sizes up to 1024 columns): the rows of
// At are produced four features (eight rows) at a time, parked in LDS, and S = H A + R for those
// eight columns of S is formed from LDS before they are overwritten -- At is not read back from
// memory at all (k_build_S re-reads all of it: 0.8 GB per launch at batch 1024, 100 features).
// Thread i plays two roles: column i of At, and row a = i of H (its 10 non-zeros in registers).
// The summation order of every entry is that of the two separate kernels.
// (Measured slower: requesting the next batch's rows of P before the S phase of the current one - 118 registers instead
// of 78, four waves per SIMD instead of six: 0.54 vs 0.33 ms.  The plain loop at full occupancy streams at 5.2 TB/s.)
// ---------------------------------------------------------------------------
constexpr int kASBatch = 4;   // features per LDS batch (state sizes up to 1024 columns)

// NQ: columns per thread (thread t owns columns t, t + 1024, ...: NQ = 1 for ld <= 1024, 2 up to 2048 - the 1280x720 /
// 500-feature configuration, ld = 1536); BATCH: features per LDS batch (2 * BATCH * ld doubles of LDS).
template <int NQ, int BATCH>
__global__ void __launch_bounds__(1024) k_build_AS(const double* __restrict__ P, const double* __restrict__ f_Hx,
                                                  const double* __restrict__ f_Hy, const double* __restrict__ f_nu,
                                                  const double* __restrict__ f_R, const int* __restrict__ succ_idx,
                                                  const int* __restrict__ m_count, double* __restrict__ At,
                                                  double* __restrict__ St, const int* __restrict__ f_hcol,
                                                  const int* __restrict__ pos_err_any, int N, int ld, int mld) {
  extern __shared__ double sAt[];   // [2 * BATCH][ld]
  const int b = blockIdx.x;
  const int cnt = m_count[b];
  if (cnt == 0) return;
  const int m = 2 * cnt, mp = (m + 31) / 32 * 32;
  const int t = threadIdx.x;        // blockDim.x * NQ >= ld; the S role below needs blockDim.x >= mp
  double* Ab = At + (size_t)b * mld * ld;
  double* Sb = St + (size_t)b * mld * mld;
  const double* Pb = P + (size_t)b * ld * ld;
  const int* sidx = succ_idx + (size_t)b * N;
  // the state columns feature f's dh_by_dy multiplies: its own, 13 + 3 f - unless the sequence carries the reference's
  // misplaced position_in_total_state_vector_ (Q28, feature.cpp:254 / monoslam.cpp:564): then what k_search_score looked up
  const int* hcol = (pos_err_any[b] != 0) ? f_hcol + (size_t)b * N : nullptr;
  // role "columns i of At"
  double pc[NQ][7];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int i = t + q * (int)blockDim.x;
#pragma unroll
    for (int c = 0; c < 7; ++c) pc[q][c] = (i >= ld) ? 0.0 : ((i < 13) ? Pb[(size_t)i * ld + c] : Pb[(size_t)c * ld + i]);
  }
  // role "row a = t of H"
  double hx[7], hy[3], Rn = 0.0;
  int posa = 0;
  if (t < m) {
    const int fa = sidx[t >> 1];
    const size_t fia = (size_t)b * N + fa;
    posa = hcol ? hcol[fa] : 13 + 3 * fa;
    // (Q28 with the block landing INSIDE the vehicle state, posa < 7: the reference's dh_by_dx_tot.block(...) = dh_by_dy
    // OVERWRITES those entries of dh_by_dxv, monoslam.cpp:562-565 - the overlapped pose coefficients do not count)
#pragma unroll
    for (int c = 0; c < 7; ++c) hx[c] = (hcol && (unsigned)(c - posa) < 3u) ? 0.0 : f_Hx[fia * 14 + (t & 1) * 7 + c];
#pragma unroll
    for (int c = 0; c < 3; ++c) hy[c] = f_Hy[fia * 6 + (t & 1) * 3 + c];
    Rn = f_R[fia];
  }
  // gridDim.y workgroups share a sequence (small batches: one workgroup walking all the features is a chain of ~25 HBM
  // round trips, 86 us at batch 1): each takes a contiguous range of feature batches = rows of At and columns of S
  const int nbatch = (cnt + BATCH - 1) / BATCH, per = (nbatch + (int)gridDim.y - 1) / (int)gridDim.y;
  const int jbeg = (int)blockIdx.y * per * BATCH;
  const int jend = (jbeg + per * BATCH < cnt) ? jbeg + per * BATCH : cnt;
  for (int j0 = jbeg; j0 < jend; j0 += BATCH) {
    const int nb = (jend - j0 < BATCH) ? jend - j0 : BATCH;
#pragma unroll
    for (int jj = 0; jj < BATCH; ++jj) {
      if (jj < nb) {
        const int f = sidx[j0 + jj];
        const size_t fi = (size_t)b * N + f;
        const int pos = hcol ? hcol[f] : 13 + 3 * f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int i = t + q * (int)blockDim.x;
          if (i < ld) {
            double py[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
              py[c] = Pb[(size_t)(pos + c) * ld + i];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              double acc = 0.0;
#pragma unroll
              for (int c = 0; c < 7; ++c) acc += pc[q][c] * ((hcol && (unsigned)(c - pos) < 3u) ? 0.0 : f_Hx[fi * 14 + r * 7 + c]);
#pragma unroll
              for (int c = 0; c < 3; ++c) acc += py[c] * f_Hy[fi * 6 + r * 3 + c];
              if (i == ld - 1) acc = f_nu[fi * 2 + r];
              Ab[(size_t)(2 * (j0 + jj) + r) * ld + i] = acc;
              sAt[(2 * jj + r) * ld + i] = acc;
            }
          }
        }
      }
    }
    __syncthreads();
    if (t < mp) {
#pragma unroll
      for (int kk = 0; kk < 2 * BATCH; ++kk) {
        const int k = 2 * j0 + kk;
        if (kk < 2 * nb && (t | 31) >= k) {     // blocks on and below the block diagonal
          double v = 0.0;
          if (t < m) {
            const double* arow = sAt + kk * ld;
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < 7; ++c) acc += hx[c] * arow[c];
#pragma unroll
            for (int c = 0; c < 3; ++c) acc += hy[c] * arow[posa + c];
            if (t == k) acc += Rn;
            v = acc;
          }
          Sb[(size_t)k * mld + t] = v;
        }
      }
    }
    __syncthreads();
  }
  // padding: rows of At up to the 32-multiple are zero, S is the identity there
  if (blockIdx.y != 0) return;
  for (int k = m; k < mp; ++k) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = t + q * (int)blockDim.x;
      if (i < ld) Ab[(size_t)k * ld + i] = 0.0;
    }
    if (t < mp && (t | 31) >= k) Sb[(size_t)k * mld + t] = (t == k) ? 1.0 : 0.0;
  }
}


// ---------------------------------------------------------------------------
// k_build_S: S = H A + R, stored St[c][r] = S[r][c] (32x32 blocks on and below the block
// diagonal; the factorisation reads r >= c plus the full diagonal blocks).  Padding: identity.  A thread owns one row
// a of H (its 10 non-zeros in registers) and loops over 32 columns bb: per column
// 7 wave-uniform loads (pose part of At row bb) + 3 gathered loads within that row.
// ---------------------------------------------------------------------------
#ifdef SL2_TESTING   // section 3 of sl2_ekf_update_testing.inc: maps beyond what k_build_AS takes are not built by the product (sl2_create rejects them): TEST build only (SL2_BUILD_VARIANT=0)
#define SL2_EKF_TEST_SECTION 3
#include "sl2_ekf_update_testing.inc"
#undef SL2_EKF_TEST_SECTION
#endif  // SL2_TESTING

// ---------------------------------------------------------------------------
// Blocked Cholesky of S (right-looking over 32x32 blocks, three launches per
// block column J): diag -> panel -> trailing update.
// ---------------------------------------------------------------------------

__device__ __forceinline__ double fast_rcp(double p) {
  double r = __builtin_amdgcn_rcp(p);
  r = __builtin_fma(__builtin_fma(-p, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-p, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double fast_rsqrt(double p) {
  double y = __builtin_amdgcn_rsq(p);
  y = y * __builtin_fma(-0.5 * p * y, y, 1.5);
  y = y * __builtin_fma(-0.5 * p * y, y, 1.5);
  return y;
}

// k_chol_diag: one wave per sequence factors the (already updated) 32x32 diagonal
// block entirely in registers: lane r holds row r; pivots and column entries are
// broadcast with v_readlane (scalar operands of the FP64 FMAs), so there is no LDS
// round trip in the 32-step dependency chain.  Then the triangle is inverted row-wise
// (lane k solves X[k][:] L = e_k) and both L_JJ (in place) and LinvT are written
// with coalesced stores.
// (Measured alternatives on MI355X, per launch at batch 1024: LDS-resident block with one
// row per lane 67 us; column broadcast through LDS 43 us; 256-thread cooperative version with
// two barriers per column 36 us; single wave, element-parallel in LDS 47 us; this one 35 us —
// every variant is bound by the ~1 us dependent chain per column, not by throughput.)

#ifdef SL2_TESTING   // section 4 of sl2_ekf_update_testing.inc: superseded variant: TEST build only (sl2_set_update_variant)
#define SL2_EKF_TEST_SECTION 4
#include "sl2_ekf_update_testing.inc"
#undef SL2_EKF_TEST_SECTION
#endif  // SL2_TESTING

// k_chol_panel: L[I][J] = S[I][J] * L_JJ^-T for every block row I > J; one wave
// per 32x32 tile.  In k-major storage: new[k][i] = sum_p Linv[k][p] * St[J+p][I+i].
#ifdef SL2_TESTING   // section 5 of sl2_ekf_update_testing.inc: superseded variant: TEST build only (sl2_set_update_variant)
#define SL2_EKF_TEST_SECTION 5
#include "sl2_ekf_update_testing.inc"
#undef SL2_EKF_TEST_SECTION
#endif  // SL2_TESTING

// k_chol_trail: S[I][K] -= L[I][J] L[K][J]^T for J < K <= I; one wave per tile.
#ifdef SL2_TESTING   // section 6 of sl2_ekf_update_testing.inc: superseded variant: TEST build only (sl2_set_update_variant)
#define SL2_EKF_TEST_SECTION 6
#include "sl2_ekf_update_testing.inc"
#undef SL2_EKF_TEST_SECTION
#endif  // SL2_TESTING


// ---------------------------------------------------------------------------
// 32 x 32 tile helpers of the one-launch Cholesky (k_chol_left below) and its panel kernels.  (The RIGHT-looking one-launch
// kernel of rounds 1-2, k_chol_fused4, is profiles/r06_retired_variants.patch.)
// ---------------------------------------------------------------------------
struct Tile32 {
  v4d f[2][2];   // f[kt][it][r] = element (k = 16 kt + 4 r + hi, i = 16 it + lo) of a k-major 32x32 tile
};

__device__ __forceinline__ Tile32 tile_load(const double* __restrict__ Sb, int mld, int k0, int i0, int lo, int hi) {
  Tile32 t;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) t.f[kt][it][r] = Sb[(size_t)(k0 + 16 * kt + hi + 4 * r) * mld + i0 + 16 * it + lo];
  return t;
}
__device__ __forceinline__ void tile_store(double* __restrict__ Sb, int mld, int k0, int i0, int lo, int hi, const Tile32& t) {
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) Sb[(size_t)(k0 + 16 * kt + hi + 4 * r) * mld + i0 + 16 * it + lo] = t.f[kt][it][r];
}
// panel tile: out[k][i] = sum_p Linv[k][p] S[o+p][i0+i]
// `need` (wave-uniform) says which 16 x 16 quarters f[kt][it] of a tile are wanted, bit 2 kt + it: the tiles of the LAST block row
// of a system whose last block holds at most 16 real rows (m mod 32 in 1 .. 16: m = 200 is one) have their rows 16 .. 31 = the
// identity padding, whose products are zeros (kTileRowsLo); of a DIAGONAL tile the D wave reads the lower triangle only, so the
// quarter above the diagonal (kt = 1, it = 0) is never looked at (kTileDiag), and with a padded last block only the first quarter
// is (kTileDiagLo).  A 7-block system of 200 rows runs 58 instead of 77 tile operations' worth of MFMAs.
constexpr int kTileAll = 15, kTileRowsLo = 5, kTileDiag = 11, kTileDiagLo = 1;
__device__ __forceinline__ Tile32 tile_panel(const double* sLinv, const double* __restrict__ Sb, int mld, int o, int i0, int lo,
                                             int hi, int need) {
  double b0[8], b1[8];
#pragma unroll
  for (int s8 = 0; s8 < 8; ++s8) {
    b0[s8] = Sb[(size_t)(o + 4 * s8 + hi) * mld + i0 + lo];
    b1[s8] = (need & 10) ? Sb[(size_t)(o + 4 * s8 + hi) * mld + i0 + 16 + lo] : 0.0;
  }
  Tile32 t;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int it = 0; it < 2; ++it) t.f[kt][it] = (v4d){0, 0, 0, 0};
#pragma unroll
  for (int s8 = 0; s8 < 8; ++s8) {
    const int p = 4 * s8 + hi;
    const double a0 = sLinv[p * kLinvPitch + lo], a1 = sLinv[p * kLinvPitch + 16 + lo];
    t.f[0][0] = mfma_f64(a0, b0[s8], t.f[0][0]);
    if (need & 2) t.f[0][1] = mfma_f64(a0, b1[s8], t.f[0][1]);
    t.f[1][0] = mfma_f64(a1, b0[s8], t.f[1][0]);
    if (need & 8) t.f[1][1] = mfma_f64(a1, b1[s8], t.f[1][1]);
  }
  return t;
}
// the same with the tile in registers: the fragment (k = 4 s8 + hi, i = 16 it + lo) of an accumulator IS the B operand of k-step s8
__device__ __forceinline__ Tile32 tile_panel_regs(const double* sLinv, const Tile32& in, int lo, int hi, int need) {
  Tile32 t;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int it = 0; it < 2; ++it) t.f[kt][it] = (v4d){0, 0, 0, 0};
#pragma unroll
  for (int s8 = 0; s8 < 8; ++s8) {
    const int p = 4 * s8 + hi;
    const double a0 = sLinv[p * kLinvPitch + lo], a1 = sLinv[p * kLinvPitch + 16 + lo];
    const double b0 = in.f[s8 >> 2][0][s8 & 3], b1 = in.f[s8 >> 2][1][s8 & 3];
    t.f[0][0] = mfma_f64(a0, b0, t.f[0][0]);
    if (need & 2) t.f[0][1] = mfma_f64(a0, b1, t.f[0][1]);
    t.f[1][0] = mfma_f64(a1, b0, t.f[1][0]);
    if (need & 8) t.f[1][1] = mfma_f64(a1, b1, t.f[1][1]);
  }
  return t;
}
// trailing tile (K, I): acc[kk][ii] -= sum_p L[K*32+kk][o+p] L[I*32+ii][o+p], operands from memory
__device__ __forceinline__ void tile_trail(double* __restrict__ Sb, int mld, int o, int K, int I, int lo, int hi) {
  Tile32 acc = tile_load(Sb, mld, K * 32, I * 32, lo, hi);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    double a0[4], a1[4], b0[4], b1[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const size_t row = (size_t)(o + 16 * h + 4 * s4 + hi) * mld;
      a0[s4] = Sb[row + K * 32 + lo];
      a1[s4] = Sb[row + K * 32 + 16 + lo];
      b0[s4] = Sb[row + I * 32 + lo];
      b1[s4] = Sb[row + I * 32 + 16 + lo];
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      acc.f[0][0] = mfma_f64(-a0[s4], b0[s4], acc.f[0][0]);
      acc.f[0][1] = mfma_f64(-a0[s4], b1[s4], acc.f[0][1]);
      acc.f[1][0] = mfma_f64(-a1[s4], b0[s4], acc.f[1][0]);
      acc.f[1][1] = mfma_f64(-a1[s4], b1[s4], acc.f[1][1]);
    }
  }
  tile_store(Sb, mld, K * 32, I * 32, lo, hi, acc);
}


#undef TR

// ---------------------------------------------------------------------------
// k_chol_left: the LEFT-looking form of the same one-launch factorisation.  The right-looking
// kernel above reads and re-writes every trailing tile once per block column (1.6 GB of traffic per
// launch at batch 1024, m = 200, against 0.23 GB of matrix): here tile (I, J) is read once, updated
// with all its J products L[I][K] L[J][K]^T from finished columns, solved and written once.
// Per block column J (waves: D = 0, M0..M2 = 1..3), two barriers:
//   P2  D factors + inverts the diagonal block (from LDS) while the M waves update the tiles below it; M0 takes tile
//       (J+1, J) and also forms the next diagonal tile (J+1, J+1) up to its products K < J, parked in LDS             X2
//   P3  panel solves L[I][J] = T[I][J] L_JJ^-T; M0 solves (J+1, J) first, subtracts its square (still in registers)
//       from the next diagonal tile and leaves that in LDS in the layout D reads                                     X3
// (The first version updated the whole diagonal tile in a phase of its own at the top of the column, M0 alone with
// three waves waiting: 0.8 + 2.6 J kcycles of a 20-40 kcycle column, scripts/chol_trace.py.)
// ---------------------------------------------------------------------------
// (the D wave's routine - d_column: [A; I] -> [L; L^-T] by column Cholesky, a row per lane - lives in sl2_chol_diag.hpp, which the
// fused small-map kernel of sl2_small.hip shares)
// acc(I, J) -= sum_{K < kend} L[J][K-block] L[I][K-block]^T (kend = J: the complete left-looking update)
__device__ __forceinline__ Tile32 tile_left_update(double* __restrict__ Sb, int mld, int J, int I, int kend, int lo, int hi, int need) {
  Tile32 acc = tile_load(Sb, mld, J * 32, I * 32, lo, hi);
  // (measured slower: an explicit register prefetch of the next half-step's operands, 0.275 vs 0.243 ms; all 32 operand
  // loads of a product in one batch, 0.271 vs 0.266)
  for (int K = 0; K < kend; ++K) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double a0[4], a1[4], b0[4], b1[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const size_t row = (size_t)(K * 32 + 16 * h + 4 * s4 + hi) * mld;
        a0[s4] = Sb[row + J * 32 + lo];
        a1[s4] = (need & 12) ? Sb[row + J * 32 + 16 + lo] : 0.0;
        b0[s4] = Sb[row + I * 32 + lo];
        b1[s4] = (need & 10) ? Sb[row + I * 32 + 16 + lo] : 0.0;
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        acc.f[0][0] = mfma_f64(-a0[s4], b0[s4], acc.f[0][0]);
        if (need & 2) acc.f[0][1] = mfma_f64(-a0[s4], b1[s4], acc.f[0][1]);
        if (need & 4) acc.f[1][0] = mfma_f64(-a1[s4], b0[s4], acc.f[1][0]);
        if (need & 8) acc.f[1][1] = mfma_f64(-a1[s4], b1[s4], acc.f[1][1]);
      }
    }
  }
  return acc;
}
// diagonal tile: acc -= l^T l for the k-major panel tile l = L[I][J-block] held in registers (A and B fragments coincide)
__device__ __forceinline__ void tile_diag_sub_regs(Tile32& acc, const Tile32& l, int need) {
#pragma unroll
  for (int s8 = 0; s8 < 8; ++s8) {
    const double v0 = l.f[s8 >> 2][0][s8 & 3], v1 = l.f[s8 >> 2][1][s8 & 3];
    acc.f[0][0] = mfma_f64(-v0, v0, acc.f[0][0]);
    if (need & 2) acc.f[0][1] = mfma_f64(-v0, v1, acc.f[0][1]);
    if (need & 4) acc.f[1][0] = mfma_f64(-v1, v0, acc.f[1][0]);
    if (need & 8) acc.f[1][1] = mfma_f64(-v1, v1, acc.f[1][1]);
  }
}

__global__ void __launch_bounds__(256, 4) k_chol_left(double* __restrict__ St, double* __restrict__ LinvT,
                                                      const int* __restrict__ m_count, int mld, int nblk_max, int J0, int nb_cap,
                                                      long long* trace) {
  // J0 / nb_cap: factor the diagonal sub-matrix of blocks J0 .. J0 + nb_cap - 1 only (panel-wise factorisation of
  // large systems, launch_chol_panels); block indices below are relative to J0.  (0, any) = the whole matrix.
  const int b = blockIdx.x;
  const int cnt = m_count[b];
  if (cnt == 0) return;
  int nblk = (2 * cnt + 31) / 32 - J0;
  if (nblk <= 0) return;
  if (nblk > nb_cap) nblk = nb_cap;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool isD = wave == 0;
  const int mw = wave - 1;
  // the last block of THIS launch holds at most 16 real rows (its rows 16 .. 31 are the identity padding): see kTileRowsLo
  const bool half_last = 2 * cnt - 32 * (J0 + nblk - 1) <= 16;
  auto off_need = [&](int I) { return (half_last && I == nblk - 1) ? kTileRowsLo : kTileAll; };
  auto diag_need = [&](int I) { return (half_last && I == nblk - 1) ? kTileDiagLo : kTileDiag; };
  __shared__ double sTile[32][33];
  __shared__ double sLinv[32 * kLinvPitch];
  __shared__ double sNext[1024];      // the next diagonal tile (partial), MFMA fragment order
  __shared__ __attribute__((aligned(16))) double sCol[2][64];   // the D wave's last two columns of L (scalar operands of its products)
  __shared__ double sIdent[32][33];   // the identity the D wave starts its inverse from
  double* Sb = St + (size_t)b * mld * mld + (size_t)J0 * 32 * mld + J0 * 32;
  LinvT += (size_t)J0 * 1024;
#ifdef SL2_CHOL_TRACE
#define TRL(slot) do { if (trace && lane == 0 && J < 8) trace[(((size_t)b * 4 + wave) * 8 + J) * 4 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TRL(slot) do { } while (0)
#endif
#ifdef SL2_CHOL_TRACE
  if (trace && lane == 0) trace[(size_t)gridDim.x * 4 * 8 * 4 + b * 4 + wave] = __builtin_amdgcn_s_getreg(63492);   // HW_ID
#endif
  // (Wave 0 = D of consecutive workgroups already lands on rotating SIMDs - 250 / 255 / 257 / 262 of 1024 per SIMD - ; an
  // explicit per-CU ticket that picks the D wave by SIMD id made the launch 6 % slower.)
  auto diag_to_lds = [&](const Tile32& t, int lo, int hi) {
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) sTile[16 * it + lo][16 * jt + hi + 4 * r4] = t.f[jt][it][r4];
  };
  if (mw == 0) diag_to_lds(tile_load(Sb, mld, 0, 0, lane & 15, lane >> 4), lane & 15, lane >> 4);
  for (int i = threadIdx.x; i < 32 * 33; i += 256) (&sIdent[0][0])[i] = (i / 33 == i % 33) ? 1.0 : 0.0;
  __syncthreads();
  for (int J = 0; J < nblk; ++J) {
    const int o = J * 32;
    int lane_j = lane;
    asm volatile("" : "+v"(lane_j));      // (a copy the compiler cannot hoist: as a loop-carried value of the J loop it cost six spilled registers, round 2)
    const int lo = lane_j & 15, hi = lane_j >> 4;
    const bool more = J + 1 < nblk;
    TRL(0);
    TRL(1);
    // P3 for the tiles that went through memory (tasks 3, 5, 6, 7, ...): the D wave (idle in that phase), M1 and M2 in turn
    auto solve_stored = [&](int turn) {
      for (int sidx = turn; ; sidx += 3) {
        const int t = sidx == 0 ? 3 : sidx + 4;
        if (J + t >= nblk) break;
        const Tile32 u = tile_panel(sLinv, Sb, mld, o, (J + t) * 32, lo, hi, off_need(J + t));
        tile_store(Sb, mld, o, (J + t) * 32, lo, hi, u);
      }
    };
    // ---- P2 ----
    if (isD) {
      // the diagonal factorisation is the critical path of the whole kernel (its wave shares a SIMD with MFMA waves of
      // other sequences and was measured 2x slower under that contention): let the scheduler prefer it
      __builtin_amdgcn_s_setprio(3);
      const int r = lane_j & 31;
      const bool low = lane_j < 32;
      double a[32];
      const double* src = low ? &sTile[r][0] : &sIdent[r][0];
#pragma unroll
      for (int c = 0; c < 32; ++c) a[c] = src[c];
      // rows of L^-T (lanes 32..63) go to sLinv as their columns finish; the rows of L, which nobody reads, back into sTile
      double* rows = low ? &sTile[r][0] : &sLinv[r * kLinvPitch];
      DScal t;
      t.s = 0.0;
      d_column<0>(a, t, (unsigned)(size_t)(__attribute__((address_space(3))) double*)&sCol[0][0], &sCol[0][lane_j], rows);
      rows[31] = a[31];
      __builtin_amdgcn_s_setprio(0);
      TRL(2);
      __syncthreads();                     // X2: L_JJ^-1 is in LDS, every tile of column J is updated
      solve_stored(0);
    } else {
      // The column's update tasks - 0: tile (J+1, J); 1: the next diagonal tile (J+1, J+1) as far as finished columns go;
      // t >= 2: tile (J+t, J) - are dealt M0, M1, M2, M0, ...: every task costs J tile products, so this is what balances the
      // three waves.  (M0 used to take (J+1, J) AND the next diagonal tile on top of its share of the others: 3 / 1 / 1 tasks
      // at J = 2, 2 / 0 / 0 at J = 5, and its 40-60 k cycles were the column's critical path next to 20-30 k for M1 / M2.)
      // The next diagonal tile is parked in LDS in fragment order (held in registers across the barrier and the panel solve
      // it spilled); M0 picks it up in P3.
      // Round 3: a wave's FIRST tile (tasks 0 / 4 / 2 of M0 / M1 / M2) is updated last and stays in its registers across X2 -
      // the accumulator fragment of the update is the B fragment of the panel solve - instead of going out to memory and
      // coming back: 14 of a 7-block system's 21 tiles, 0.24 MB of a sequence's 1.1 MB of traffic in a bandwidth-bound
      // kernel, and one store -> barrier -> load round trip less on M0's chain.
      Tile32 kept;                         // the wave's first tile of the column, from its update to its solve
#pragma unroll
      for (int q = 0; q < 4; ++q) kept.f[q >> 1][q & 1] = (v4d){0, 0, 0, 0};   // (left undefined it becomes a value carried around the J loop: 86 spilled registers)
      int keptI = nblk;
      if (more) {                                   // (the last column has no tile below it)
        if (mw == 1) {                              // task 1
          const Tile32 dn = tile_left_update(Sb, mld, J + 1, J + 1, J, lo, hi, diag_need(J + 1));
#pragma unroll
          for (int k = 0; k < 16; ++k) sNext[k * 64 + lane_j] = dn.f[k >> 3][(k >> 2) & 1][k & 3];
        }
        const int t_keep = mw == 0 ? 0 : (mw == 1 ? 4 : 2);
        for (int t = t_keep + 3; J + t < nblk; t += 3) {          // the wave's later tiles: updated, stored
          const Tile32 u = tile_left_update(Sb, mld, J, J + t, J, lo, hi, off_need(J + t));
          tile_store(Sb, mld, o, (J + t) * 32, lo, hi, u);
        }
        keptI = mw == 0 ? J + 1 : J + t_keep;
        if (keptI < nblk) kept = tile_left_update(Sb, mld, J, keptI, J, lo, hi, off_need(keptI));
      }
      TRL(2);
      __syncthreads();                     // X2 (the D wave meets it in its own branch)
      // ---- P3 ----
      if (keptI < nblk) {
        const Tile32 l = tile_panel_regs(sLinv, kept, lo, hi, off_need(keptI));
        tile_store(Sb, mld, o, keptI * 32, lo, hi, l);
        if (mw == 0) {                     // the chain: the next diagonal tile minus the square of the tile just solved
          Tile32 dn;
#pragma unroll
          for (int q = 0; q < 16; ++q) dn.f[q >> 3][(q >> 2) & 1][q & 3] = sNext[q * 64 + lane_j];
          tile_diag_sub_regs(dn, l, diag_need(J + 1));
          diag_to_lds(dn, lo, hi);
        }
      }
      if (mw == 2) {   // LinvT block to memory for the forward substitution, coalesced
        double* Lb = LinvT + ((size_t)b * nblk_max + J) * 1024;
#pragma unroll
        for (int q = 0; q < 16; ++q) Lb[q * 64 + lane_j] = sLinv[(q * 2 + (lane_j >> 5)) * kLinvPitch + (lane_j & 31)];
      }
      if (mw >= 1) solve_stored(mw);       // M0 keeps to the chain above
    }
    TRL(3);
    if (more) __syncthreads();             // X3: column J of L is complete, the next diagonal tile is in LDS
  }
}
#undef TRL

// ---------------------------------------------------------------------------
// k_fwdsub: Vt = L^-1 At by blocked forward substitution.  Columns are
// independent: each wave owns 16 columns and walks the block rows J in order,
//   acc = At[J] - sum_{K<J} L[J][K] Vt[K] ;  Vt[J] = L_JJ^-1 acc.
// A lane re-reads only Vt elements it stored itself (D fragment == B fragment).
// The chain is latency-bound (every k-step needs 3 L2-resident operands); what hides
// it is occupancy (64 VGPRs -> 7 waves per SIMD) plus the unrolled k-loop, which puts
// 8 k-steps of loads in flight ahead of their MFMAs.  A last block whose upper 16
// rows are all padding skips them.  32 columns per wave give four independent
// accumulator chains and 4 MFMAs per 3 fragment loads.
// (Tried and measured slower on MI355X, batch 1024: L blocks through LDS with a 4-wave
// barrier per block, 0.72 ms; V^T resident in LDS with one wave per workgroup, 2.1 ms;
// an explicit two-stage register pipeline, 1.04 ms at 256 VGPRs — each loses the
// occupancy that hides the chain's latency.  This version: see profiles/.)
// ---------------------------------------------------------------------------
#ifdef SL2_TESTING   // section 8 of sl2_ekf_update_testing.inc: superseded variant: TEST build only (sl2_set_update_variant)
#define SL2_EKF_TEST_SECTION 8
#include "sl2_ekf_update_testing.inc"
#undef SL2_EKF_TEST_SECTION
#endif  // SL2_TESTING

// ---------------------------------------------------------------------------
// k_fwdsub_lds<NB>: forward substitution with BOTH operands on chip.  A workgroup of four
// waves owns a 64-column tile of the state (16 columns per wave):
//   * the solved block rows of V^T stay in registers: the MFMA D fragment of block K is, unchanged,
//     the B fragment of every later product with it (NB = mld/32 block rows of a 16-column strip
//     = NB*16 VGPRs);
//   * the 32x32 tiles of L (and the inverted diagonal blocks) are streamed once per
//     workgroup through LDS, double-buffered: the global loads of tile t+1 fly under the
//     MFMAs of tile t, one barrier per tile (the scheme of k_syrk).
// Per k-step a wave issues two ds_read_b64 and two MFMAs and nothing else: no global load
// sits on the accumulator chain, which is what bounded k_fwdsub (every k-step there waits
// on three L2-resident operands).
// ---------------------------------------------------------------------------
constexpr int kFwdPitch = 48;   // LDS row pitch (doubles): the two k-rows read by a 32-lane group land on disjoint banks

template <int NB>
// (At, Vt and St are NOT __restrict__: the panel solve of the large-system Cholesky calls this kernel in place, with all three
// on the factor's own storage - every element is read by the lane that later overwrites it, before it does.)
__global__ void __launch_bounds__(256, (NB <= 8 ? 3 : 2)) k_fwdsub_lds(const double* At, double* Vt,
                                                    const double* St, const double* __restrict__ LinvT,
                                                    const int* __restrict__ m_count, int ld, int mld, int nblk_max, int B,
                                                    int J0, int col0, int ntile, int nb_cap) {
  // J0: first block row of this launch.  For maps of more than 13 blocks the substitution runs in groups of
  // NB = 8 block rows: k_fwd_gemm first subtracts the contribution of all earlier groups from the group's
  // rows of At, then this kernel solves within the group (block indices below are relative to J0).
  // col0 / ntile: the 64-column tiles col0 + 64 t, t < ntile, of the right-hand side (0, ld / 64 for the EKF substitution;
  // the panel solve of the large-system Cholesky passes the columns below its panel).  nb_cap caps the block rows solved.
  int b, ct;
  if (!xcd_map(ntile, B, &b, &ct)) return;
  const int cnt = m_count[b];
  if (cnt == 0) return;
  int m = 2 * cnt - 32 * J0;                // rows left from block J0 on
  if (m <= 0) return;
  if (m > 32 * nb_cap) m = 32 * nb_cap;
  const int nblk = (m + 31) / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 15, hi = lane >> 4;
  const int i0 = col0 + ct * 64 + wave * 16;
  const double* Ab = At + (size_t)b * mld * ld + (size_t)J0 * 32 * ld;
  double* Vb = Vt + (size_t)b * mld * ld + (size_t)J0 * 32 * ld;
  const double* Sb = St + (size_t)b * mld * mld + (size_t)J0 * 32 * mld + J0 * 32;
  const double* Lb = LinvT + ((size_t)b * nblk_max + J0) * 1024;
  __shared__ double sL[2][32 * kFwdPitch];
  // staging role: row `srow` of the tile (contraction index), 4 consecutive doubles at column `sc4`.  (Lanes 0 and 4 of a
  // row's eight share their banks in the ds_write_b128: SQ_LDS_BANK_CONFLICT 8.6e6 per launch.  Round 4 tried the cure that
  // worked in k_syrk - two pieces of two doubles per thread, so that a store's eight lanes write 128 consecutive bytes - and
  // measured it SLOWER on the same box, 0.331 against 0.322-0.324 ms (profiles/r04_fwdsub_staging_ab.txt): twice the global
  // load instructions in front of every barrier cost more than the conflict cycles, which sit under the MFMAs here.)
  const int srow = tid >> 3, sc4 = (tid & 7) * 4;
  const int soff = srow * kFwdPitch + sc4;
  // tile (J, K): K < J -> L[J][K] from St (k-major), K == J -> LinvT block J
#define SL2_TILE_PTR(Jv, Kv) \
  (((Kv) < (Jv)) ? (Sb + (size_t)((Kv) * 32 + srow) * mld + (Jv) * 32 + sc4) : (Lb + (size_t)(Jv) * 1024 + srow * 32 + sc4))
  double4 pre = *(const double4*)SL2_TILE_PTR(0, 0);
  v4d V[NB][2];
  v4d at[2];
  const bool half0 = (16 >= m);
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) at[jt][r] = (jt == 1 && half0) ? 0.0 : Ab[(size_t)(16 * jt + hi + 4 * r) * ld + i0 + lo];
  int t = 0;   // running tile counter (buffer parity)
#pragma unroll
  for (int J = 0; J < NB; ++J) {
    if (J < nblk) {
      const bool half = (J * 32 + 16 >= m);    // rows J*32+16.. are padding
      v4d acc[2];
      acc[0] = (v4d){0, 0, 0, 0};
      acc[1] = (v4d){0, 0, 0, 0};
      v4d rhs[2];
#pragma unroll
      for (int K = 0; K <= J; ++K) {
        double* buf = sL[t & 1];
        *(double4*)&buf[soff] = pre;
        __syncthreads();
        // next tile of the stream (uniform control flow: nblk is per sequence = per workgroup)
        if (K < J) pre = *(const double4*)SL2_TILE_PTR(J, K + 1);
        else if (J + 1 < nblk && J + 1 < NB) pre = *(const double4*)SL2_TILE_PTR(J + 1, 0);
        const double* pa = buf + hi * kFwdPitch + lo;
        if (K < J) {
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            const double bv = V[K][s >> 2][s & 3];
            acc[0] = mfma_f64(pa[4 * s * kFwdPitch], bv, acc[0]);
            if (!half) acc[1] = mfma_f64(pa[4 * s * kFwdPitch + 16], bv, acc[1]);
          }
        } else {
          // rhs = At[J] - sum_K L[J][K] V[K];  V[J] = Linv_JJ rhs  (Linv lower triangular: rows 0..15 need p < 16 only)
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) rhs[jt] = at[jt] - acc[jt];
          if (J + 1 < nblk && J + 1 < NB) {   // prefetch the next block row of At under the diagonal product
            const bool halfn = ((J + 1) * 32 + 16 >= m);
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                at[jt][r] = (jt == 1 && halfn) ? 0.0 : Ab[(size_t)((J + 1) * 32 + 16 * jt + hi + 4 * r) * ld + i0 + lo];
          }
          v4d out[2];
          out[0] = (v4d){0, 0, 0, 0};
          out[1] = (v4d){0, 0, 0, 0};
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            const double bv = rhs[s >> 2][s & 3];
            if (s < 4) out[0] = mfma_f64(pa[4 * s * kFwdPitch], bv, out[0]);
            if (!half) out[1] = mfma_f64(pa[4 * s * kFwdPitch + 16], bv, out[1]);
          }
          V[J][0] = out[0];
          V[J][1] = out[1];
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) {
            if (jt == 1 && half) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) Vb[(size_t)(J * 32 + 16 * jt + hi + 4 * r) * ld + i0 + lo] = out[jt][r];
          }
        }
        ++t;
      }
    }
  }
#undef SL2_TILE_PTR
}

// ---------------------------------------------------------------------------
// k_fwdsub_ksplit: the forward substitution for SMALL batches, where k_fwdsub_lds is a chain, not a throughput problem:
// there one wavefront walks all (J + 1) tile products of every block row of its 16 columns one after the other - 23
// dependent 16-MFMA steps, 26 us for a single sequence.  Here a four-wave workgroup owns ONE 16-column strip: the solved
// block rows live in LDS (the B fragments of every wave), the J products of block row J are dealt to the four waves,
// their partial sums meet in LDS, and wave 0 applies the inverted diagonal block.  Same sums in a different association
// (four partial sums): tolerance parity like every other variant.  Up to kKsMaxBlocks 32-blocks.
// ---------------------------------------------------------------------------
constexpr int kKsMaxBlocks = 8;
__global__ void __launch_bounds__(256) k_fwdsub_ksplit(const double* __restrict__ At, double* __restrict__ Vt,
                                                       const double* __restrict__ St, const double* __restrict__ LinvT,
                                                       const int* __restrict__ m_count, int ld, int mld, int nblk_max, int B) {
  int b, ct;
  if (!xcd_map(ld / 16, B, &b, &ct)) return;
  const int cnt = m_count[b];
  if (cnt == 0) return;
  const int nblk = (2 * cnt + 31) / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 15, hi = lane >> 4;
  const int i0 = ct * 16;
  const double* Ab = At + (size_t)b * mld * ld;
  double* Vb = Vt + (size_t)b * mld * ld;
  const double* Sb = St + (size_t)b * mld * mld;
  const double* Lb = LinvT + (size_t)b * nblk_max * 1024;
  __shared__ double sV[kKsMaxBlocks][32 * 16];      // solved block rows: [row][column]
  __shared__ double sPart[4][32 * 16];              // the waves' partial sums of the current block row
  // Operand prefetch (registers): the L tile of the wave's NEXT product - its (J, K) pairs are known in advance: K = wave,
  // wave + 4, ... < J for J = wave + 1, wave + 2, ... -, and for wave 0 the next block row of At and the inverted diagonal
  // block of the current one: nothing on the chain waits for a memory round trip it could have started earlier.
  auto load_l = [&](int J, int K, double (&a0)[8], double (&a1)[8]) {
    const double* lt = Sb + (size_t)(K * 32 + hi) * mld + J * 32 + lo;       // L[J][K], k-major: [contraction][row of J]
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) { a0[s8] = lt[(size_t)(4 * s8) * mld]; a1[s8] = lt[(size_t)(4 * s8) * mld + 16]; }
  };
  auto load_linv = [&](int J, double (&l0)[4], double (&l1)[8]) {
    const double* li = Lb + (size_t)J * 1024 + hi * 32 + lo;                 // LinvT[p][k]: [contraction][row]
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) { if (s8 < 4) l0[s8] = li[4 * s8 * 32]; l1[s8] = li[4 * s8 * 32 + 16]; }
  };
  v4d at[2];
  double li0[4], li1[8];
  double pa0[8], pa1[8];
  int pJ = wave + 1, pK = wave;                       // the wave's next (J, K)
  if (pJ < nblk) load_l(pJ, pK, pa0, pa1);
  if (wave == 0) {
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) at[jt][r] = Ab[(size_t)(16 * jt + hi + 4 * r) * ld + i0 + lo];
    load_linv(0, li0, li1);
  }
  for (int J = 0; J < nblk; ++J) {
    // ---- the products of this block row: K = wave, wave + 4, ...
    v4d acc[2];
    acc[0] = (v4d){0, 0, 0, 0};
    acc[1] = (v4d){0, 0, 0, 0};
    for (int K = wave; K < J; K += 4) {
      double a0[8], a1[8];
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) { a0[s8] = pa0[s8]; a1[s8] = pa1[s8]; }
      // next pair: same block row while K + 4 < J, else the first product of the next block row that has one for this wave
      if (K + 4 < J) { pJ = J; pK = K + 4; } else { pJ = J + 1; pK = wave; }
      if (pJ < nblk) load_l(pJ, pK, pa0, pa1);
      const double* vk = &sV[K][hi * 16 + lo];
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const double bv = vk[4 * s8 * 16];
        acc[0] = mfma_f64(a0[s8], bv, acc[0]);
        acc[1] = mfma_f64(a1[s8], bv, acc[1]);
      }
    }
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sPart[wave][(16 * jt + hi + 4 * r) * 16 + lo] = acc[jt][r];
    __syncthreads();
    // ---- wave 0: rhs = At[J] - sum of the partial sums; V[J] = Linv_JJ rhs
    if (wave == 0) {
      v4d rhs[2];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = (16 * jt + hi + 4 * r) * 16 + lo;
          rhs[jt][r] = at[jt][r] - (((sPart[0][e] + sPart[1][e]) + sPart[2][e]) + sPart[3][e]);
        }
      double l0[4], l1[8];
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) { if (s8 < 4) l0[s8] = li0[s8]; l1[s8] = li1[s8]; }
      if (J + 1 < nblk) {          // the next block row of At and its inverted diagonal block travel under this product
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) at[jt][r] = Ab[(size_t)((J + 1) * 32 + 16 * jt + hi + 4 * r) * ld + i0 + lo];
        load_linv(J + 1, li0, li1);
      }
      v4d out[2];
      out[0] = (v4d){0, 0, 0, 0};
      out[1] = (v4d){0, 0, 0, 0};
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const double bv = rhs[s8 >> 2][s8 & 3];
        if (s8 < 4) out[0] = mfma_f64(l0[s8], bv, out[0]);          // Linv lower triangular: rows 0..15 need p < 16
        out[1] = mfma_f64(l1[s8], bv, out[1]);
      }
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * jt + hi + 4 * r;
          sV[J][row * 16 + lo] = out[jt][r];
          Vb[(size_t)(J * 32 + row) * ld + i0 + lo] = out[jt][r];
        }
    }
    __syncthreads();
  }
}

static int launch_fwdsub_lds(sl2_engine* e, int B, bool* done) {
  *done = true;
  // small batches: the chain-shortening kernel (one 16-column strip per workgroup, products dealt to its four waves)
  if (e->nblk_max <= kKsMaxBlocks && (long long)B * (e->ld / 16) <= 160 && !e->root->no_ksplit) {
    LaunchScope ls(e, "k_fwdsub_ksplit", true);
    hipLaunchKernelGGL(k_fwdsub_ksplit, dim3(xcd_grid(e->ld / 16, B)), dim3(256), 0, e->stream, e->At, e->Vt, e->St, e->LinvT,
                       e->m_count, e->ld, e->mld, e->nblk_max, B);
    SL2_HIP(hipGetLastError());
    return SL2_OK;
  }
  if (e->nblk_max > e->root->group_from) { *done = false; return SL2_OK; }     // beyond the register-resident strip (13 blocks): the grouped form
  const dim3 grid(xcd_grid(e->ld / 64, B)), block(256);
  LaunchScope ls(e, "k_fwdsub_lds", true);
#define SL2_FWD_CASE(NBV)                                                                                           \
  case NBV:                                                                                                         \
    hipLaunchKernelGGL((k_fwdsub_lds<NBV>), grid, block, 0, e->stream, e->At, e->Vt, e->St, e->LinvT, e->m_count, \
                       e->ld, e->mld, e->nblk_max, B, 0, 0, e->ld / 64, NBV);                                       \
    break;
  switch (e->nblk_max) {
    SL2_FWD_CASE(1) SL2_FWD_CASE(2) SL2_FWD_CASE(3) SL2_FWD_CASE(4) SL2_FWD_CASE(5) SL2_FWD_CASE(6) SL2_FWD_CASE(7)
    SL2_FWD_CASE(8) SL2_FWD_CASE(9) SL2_FWD_CASE(10) SL2_FWD_CASE(11) SL2_FWD_CASE(12) SL2_FWD_CASE(13)
    default: break;
  }
#undef SL2_FWD_CASE
  SL2_HIP(hipGetLastError());
  return SL2_OK;
}

constexpr int kSyrkKC = 16;       // K rows per chunk (k_syrk, k_fwd_gemm, k_chol_syrk)

// ---------------------------------------------------------------------------
// The K loop of k_syrk as a building block for the large-map GEMMs (k_fwd_gemm, k_chol_syrk): a 64x64 tile, four waves of
// 32x32, acc[it][jt][r] += sum_k A[k][wi + 16 it + 4 r + hi] * B[k][wj + 16 jt + lo] over `nchunk` chunks of 16 k-rows,
// both operands k-major in memory (row segments), staged through the 32 KB swapped-half-row LDS layout with the register
// prefetch two chunks ahead.  (Rounds 1-2 ran these two kernels on the first SYRK pipeline - 40 KB at pitch 80, one
// chunk ahead: 49.5 and 39 TFLOP/s at configs[4] against k_syrk's 59.)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void kpanel_product(const double* __restrict__ A, size_t lda, const double* __restrict__ Bm, size_t ldb,
                                               int nchunk, bool idle, int wi, int wj, double (&sAB)[2][2][kSyrkKC * 64],
                                               v4d (&acc)[2][2]) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int lo = lane & 15, hi = lane >> 4;
  const int kr = tid >> 4, c2 = (tid & 15) * 2;
  const double* gA = A + (size_t)kr * lda + c2;
  const double* gB = Bm + (size_t)kr * ldb + c2;
  struct Stage { double2 a0, a1, b0, b1; };
  auto stage_load = [&](int chunk) {
    Stage r;
    r.a0 = *(const double2*)(gA + (size_t)chunk * kSyrkKC * lda); r.a1 = *(const double2*)(gA + (size_t)chunk * kSyrkKC * lda + 32);
    r.b0 = *(const double2*)(gB + (size_t)chunk * kSyrkKC * ldb); r.b1 = *(const double2*)(gB + (size_t)chunk * kSyrkKC * ldb + 32);
    return r;
  };
  auto stage_store = [&](int buf, const Stage& r) {
    const int cs = c2 ^ ((kr & 1) << 4);
    *(double2*)&sAB[buf][0][kr * 64 + cs] = r.a0;
    *(double2*)&sAB[buf][0][kr * 64 + 32 + cs] = r.a1;
    *(double2*)&sAB[buf][1][kr * 64 + cs] = r.b0;
    *(double2*)&sAB[buf][1][kr * 64 + 32 + cs] = r.b1;
  };
  auto chunk_mfma = [&](int buf) {
    const int sw = (hi & 1) << 4;
    const double* pa = &sAB[buf][0][hi * 64 + wi + lo];
    const double* pb = &sAB[buf][1][hi * 64 + wj + lo];
#pragma unroll
    for (int ks = 0; ks < kSyrkKC / 4; ++ks) {
      const double a0 = pa[ks * 256 + sw], a1 = pa[ks * 256 + (sw ^ 16)];
      const double b0 = pb[ks * 256 + sw], b1 = pb[ks * 256 + (sw ^ 16)];
      acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
      acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
      acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
      acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
    }
  };
  if (nchunk <= 0) return;
  Stage r0 = stage_load(0);
  Stage r1 = r0;
  if (nchunk > 1) r1 = stage_load(1);
  for (int ch = 0; ch < nchunk; ch += 2) {
    stage_store(0, r0);
    __syncthreads();
    if (ch + 2 < nchunk) r0 = stage_load(ch + 2);
    if (!idle) chunk_mfma(0);
    if (ch + 1 >= nchunk) break;
    stage_store(1, r1);
    __syncthreads();
    if (ch + 3 < nchunk) r1 = stage_load(ch + 3);
    if (!idle) chunk_mfma(1);
  }
}

// ---------------------------------------------------------------------------
// k_fwd_gemm: At[R0 .. R0+256) -= L[R0 .. R0+256)[0 .. R0) * Vt[0 .. R0), the contribution of the already
// solved block rows to one group of eight block rows (R0 = 32 J0).  64x64 output tiles, four waves of 32x32,
// both operands streamed through double-buffered LDS in chunks of 16 k-rows like k_syrk (L is stored
// k-major in St, so both staging reads are row segments).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 4) k_fwd_gemm(double* __restrict__ At, const double* __restrict__ Vt,
                                                  const double* __restrict__ St, const int* __restrict__ m_count, int ld, int mld,
                                                  int B, int J0, int nrt) {
  // nrt: 64-row tiles of the group (4 = eight block rows, 2 = four)
  int b, t;
  const int ntc = ld / 64;
  if (!xcd_map(nrt * ntc, B, &b, &t)) return;
  const int cnt = m_count[b];
  if (cnt == 0) return;
  const int mp = (2 * cnt + 31) / 32 * 32;
  const int R0 = J0 * 32;
  const int r0 = R0 + (t / ntc) * 64, c0 = (t % ntc) * 64;
  if (r0 >= mp) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 15, hi = lane >> 4;
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
  double* Ab = At + (size_t)b * mld * ld;
  const double* Vb = Vt + (size_t)b * mld * ld;
  const double* Sb = St + (size_t)b * mld * mld;
  __shared__ double sAB[2][2][kSyrkKC * 64];
  v4d acc[2][2];
  for (int it = 0; it < 2; ++it) for (int jt = 0; jt < 2; ++jt) acc[it][jt] = (v4d){0, 0, 0, 0};
  // L[r0 + c][k] = St[k][r0 + c]: both operands are row segments
  kpanel_product(Sb + r0, (size_t)mld, Vb + c0, (size_t)ld, R0 / kSyrkKC, false, wi, wj, sAB, acc);
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* q = Ab + (size_t)(r0 + wi + 16 * it + hi + 4 * r) * ld + c0 + wj + 16 * jt + lo;
        *q -= acc[it][jt][r];
      }
}

// ---------------------------------------------------------------------------
// Cholesky of systems with more than 16 blocks (launch_chol_panels): right-looking over PANELS of eight block columns
// (256 columns; four in rounds 1-2), so that the trailing matrix is read and written once per panel instead of once per 32
// columns (the launch-per-block kernels are bound by exactly that traffic: 22 GB per factorisation at m = 1000, batch 256):
//   k_chol_left (J0, 8 blocks)      factor the 256x256 diagonal sub-matrix in one launch
//   k_fwdsub_lds<8> on S itself     panel solve X = S[below][panel] L_dd^-T  (S is k-major: the panel's columns are rows
//                                   of St, i.e. exactly the right-hand-side layout of the EKF substitution)
//   k_chol_syrk                     S[below][below] -= X X^T, 64x64 tiles of the lower block triangle, K = 256
// (with K = 128 a tile of the trailing update had eight K-chunks to amortise its read-modify-write over: 38 TFLOP/s)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_chol_syrk(double* __restrict__ St, const int* __restrict__ m_count, int mld, int B,
                                                   int k0, int kp, int c0) {
  int b, t;
  const int nt = (mld - c0) / 64;
  if (!xcd_map(nt * (nt + 1) / 2, B, &b, &t)) return;
  const int cnt = m_count[b];
  if (cnt == 0) return;
  const int mp = (2 * cnt + 31) / 32 * 32;
  int ti = 0;
  while (t > ti) { t -= ti + 1; ++ti; }
  const int tj = t;                       // tj <= ti : column block j <= row block i (lower triangle, stored St[j][i])
  const int j0 = c0 + tj * 64, i0 = c0 + ti * 64;
  if (j0 >= mp || i0 >= mp) return;       // (j0 <= i0; tiles of the padding are never read)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 15, hi = lane >> 4;
  const int wj = (wave >> 1) * 32, wi = (wave & 1) * 32;
  double* Sb = St + (size_t)b * mld * mld;
  __shared__ double sAB[2][2][kSyrkKC * 64];
  v4d acc[2][2];
  for (int a = 0; a < 2; ++a) for (int c = 0; c < 2; ++c) acc[a][c] = (v4d){0, 0, 0, 0};
  // rows of the panel beyond this sequence's own (padded) system are never initialised: stop at mp
  const int kvalid = (mp - k0 < kp) ? mp - k0 : kp;
  // in a diagonal tile the block (rows j0 + 32 .., columns i0 ..) lies above the diagonal: nobody reads it
  const bool idle = (ti == tj) && (wj > wi);
  // X[j0 + c][k] = St[k][j0 + c]
  kpanel_product(Sb + (size_t)k0 * mld + j0, (size_t)mld, Sb + (size_t)k0 * mld + i0, (size_t)mld, kvalid / kSyrkKC, idle, wj, wi, sAB,
                 acc);
  if (idle) return;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* q = Sb + (size_t)(j0 + wj + 16 * a + hi + 4 * r) * mld + i0 + wi + 16 * c + lo;
        *q -= acc[a][c][r];
      }
}

constexpr int kCholPanelBlocks = 4;

static int launch_chol_panels(sl2_engine* e, int B) {
  // panel width in 32-blocks: 4 (128 columns) or 8 (256 columns: half as many trailing updates, each with twice the K)
  const int pb = e->root->chol_panel == 8 ? 8 : kCholPanelBlocks;
  for (int p0 = 0; p0 < e->nblk_max; p0 += pb) {
    const int nb = e->nblk_max - p0 < pb ? e->nblk_max - p0 : pb;
    {
      LaunchScope ls(e, "k_chol_left", true);
      hipLaunchKernelGGL(k_chol_left, dim3(B), dim3(256), 0, e->stream, e->St, e->LinvT, e->m_count, e->mld, e->nblk_max, p0, nb,
                         (long long*)nullptr);
      SL2_HIP(hipGetLastError());
    }
    const int c0 = (p0 + nb) * 32;
    if (c0 >= e->mld) break;
    const int ntile = (e->mld - c0) / 64;      // mld is a multiple of 64 for these sizes (sl2_create)
    {
      LaunchScope ls(e, "k_fwdsub_lds@chol", true);
      if (pb == 8)
        hipLaunchKernelGGL((k_fwdsub_lds<8>), dim3(xcd_grid(ntile, B)), dim3(256), 0, e->stream, e->St, e->St, e->St,
                           e->LinvT, e->m_count, e->mld, e->mld, e->nblk_max, B, p0, c0, ntile, nb);
      else
        hipLaunchKernelGGL((k_fwdsub_lds<kCholPanelBlocks>), dim3(xcd_grid(ntile, B)), dim3(256), 0, e->stream, e->St, e->St, e->St,
                           e->LinvT, e->m_count, e->mld, e->mld, e->nblk_max, B, p0, c0, ntile, nb);
      SL2_HIP(hipGetLastError());
    }
    {
      LaunchScope ls(e, "k_chol_syrk", true);
      hipLaunchKernelGGL(k_chol_syrk, dim3(xcd_grid(ntile * (ntile + 1) / 2, B)), dim3(256), 0, e->stream, e->St, e->m_count, e->mld,
                         B, p0 * 32, nb * 32, c0);
      SL2_HIP(hipGetLastError());
    }
  }
  return SL2_OK;
}

// substitution in groups of eight (or four: fwd_group) block rows, for systems of more than 13 blocks
static int launch_fwdsub_grouped(sl2_engine* e, int B) {
  const int g = e->root->fwd_group == 4 ? 4 : 8;
  for (int J0 = 0; J0 < e->nblk_max; J0 += g) {
    if (J0 > 0) {
      LaunchScope ls(e, "k_fwd_gemm", true);
      hipLaunchKernelGGL(k_fwd_gemm, dim3(xcd_grid((g / 2) * (e->ld / 64), B)), dim3(256), 0, e->stream, e->At, e->Vt, e->St,
                         e->m_count, e->ld, e->mld, B, J0, g / 2);
      SL2_HIP(hipGetLastError());
    }
    LaunchScope ls(e, "k_fwdsub_lds", true);
    if (g == 4)
      hipLaunchKernelGGL((k_fwdsub_lds<4>), dim3(xcd_grid(e->ld / 64, B)), dim3(256), 0, e->stream, e->At, e->Vt, e->St, e->LinvT,
                         e->m_count, e->ld, e->mld, e->nblk_max, B, J0, 0, e->ld / 64, 4);
    else
      hipLaunchKernelGGL((k_fwdsub_lds<8>), dim3(xcd_grid(e->ld / 64, B)), dim3(256), 0, e->stream, e->At, e->Vt, e->St, e->LinvT,
                         e->m_count, e->ld, e->mld, e->nblk_max, B, J0, 0, e->ld / 64, 8);
    SL2_HIP(hipGetLastError());
  }
  return SL2_OK;
}

// ---------------------------------------------------------------------------
// k_syrk: P -= V V^T on 64x64 tiles of the upper block triangle (ti <= tj),
// mirrored to the lower one; 4 waves per tile, 32x32 per wave.  The column
// ld-1 of Vt is w = L^-1 nu, so the same product yields x += V w there.
//
// The two 64-column panels of V^T are streamed through LDS in K-chunks of 16 rows,
// double-buffered: the global loads of chunk c+1 are in flight while the 16 MFMAs
// of chunk c run (the kernel was memory-latency-bound with direct fragment loads:
// SQ_WAIT_ANY 55 %, MFMA pipe 38 % busy).  LDS: 32 KB, rows of 64 doubles without padding; the odd k-rows swap their
// two 16-column halves within each 32-column group (column ^ 16), so that the two k-rows read by one 32-lane group of a
// ds_read_b64 land on disjoint banks (the first version padded the rows to 80 doubles: 40 KB, three workgroups per CU).
// ---------------------------------------------------------------------------

__global__ void __launch_bounds__(256, 4) k_syrk(const double* __restrict__ Vt, double* __restrict__ P, double* __restrict__ x,
                                              const int* __restrict__ m_count, int ld, int mld, int B,
                                              const int* __restrict__ n_slots, int ppos
#ifdef SL2_CHOL_TRACE
                                              , long long* trace
#endif
                                              ) {
#ifdef SL2_CHOL_TRACE   // development builds: entry / exit cycle stamps of wave 0 (scripts/syrk_clock.py)
  struct Stamp {
    long long* p;
    __device__ Stamp(long long* q) : p(q) { if (p && threadIdx.x == 0) p[2 * blockIdx.x] = (long long)__builtin_readcyclecounter(); }
    __device__ ~Stamp() { if (p && threadIdx.x == 0) p[2 * blockIdx.x + 1] = (long long)__builtin_readcyclecounter(); }
  } stamp(trace);
#endif
  int b, t;
  const int ntl = ld / 64;
  if (!xcd_map(ntl * (ntl + 1) / 2, B, &b, &t)) return;
  const int cnt = m_count[b];
  if (cnt == 0) return;
  const int mp = (2 * cnt + kSyrkKC - 1) / kSyrkKC * kSyrkKC;   // rows >= 2 cnt of Vt are zero up to the 16-multiple
  // k-steps (of 4 rows) of the LAST chunk that hold measurements: with m = 198 the 13th chunk has 6 live rows of 16,
  // and the two k-steps of zeros were 3 % of the launch's matrix instructions
  const int nks_last = (2 * cnt - (mp - kSyrkKC) + 3) >> 2;
  int tj = 0;
  while (t > tj) { t -= tj + 1; ++tj; }
  const int ti = t;  // ti <= tj
  // The engine's P has room for max_features slots; the columns of slots a sequence has never used - between its last slot
  // and the partially initialised features at ppos - are zero in P and in V^T, and a tile that lies in them has nothing to
  // subtract: with mapping on, a map that has grown to a third of its capacity pays for a ninth of the tiles (at 84-100
  // live features of 200: 21 tiles of 55, k_syrk 1.04 -> 0.43 ms, profiles/r04_live_tiles.txt).
  const int n_live = 13 + 3 * n_slots[b];
  {
    const int t_lo = (n_live + 63) >> 6, t_hi = ppos >> 6;
    if ((ti >= t_lo && ti < t_hi) || (tj >= t_lo && tj < t_hi)) return;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 15, hi = lane >> 4;
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
  const double* Vb = Vt + (size_t)b * mld * ld;
  double* Pb = P + (size_t)b * ld * ld;
  // LDS: [buffer][A | B][16 k-rows x 64 columns], 32 KB (the 80-double row pitch of the first version cost 40 KB = three
  // workgroups per CU: with m = 200 a tile has only 13 chunks, and a third of its life is prologue and epilogue, which
  // only other resident workgroups can cover).  Registers: 76 + 32 accumulator = 108 -> FOUR workgroups per CU; forcing
  // five (__launch_bounds__(256, 5): 94 registers, no spills) measured 0.507 against 0.498 ms, so four it stays.  Instead of padding, the odd k-rows swap their two
  // 16-column halves within each 32-column group (column ^ 16): the two k-rows read by one 32-lane group still land on
  // disjoint banks.
  __shared__ double sAB[2][2][kSyrkKC * 64];
  // staging role of this thread: row kr of the chunk, columns c2, c2+1 and 32+c2, 33+c2: sixteen lanes write 256
  // CONTIGUOUS bytes per ds_write_b128 (four consecutive doubles per lane put lanes 0 and 8 of a row on the same banks:
  // 2.6e7 conflict cycles per launch, three per LDS instruction)
  const int kr = tid >> 4, c2 = (tid & 15) * 2;
  const double* gA = Vb + (size_t)kr * ld + ti * 64 + c2;
  const double* gB = Vb + (size_t)kr * ld + tj * 64 + c2;
  struct Stage { double2 a0, a1, b0, b1; };
  auto stage_load = [&](int chunk) {
    Stage r;
    const size_t off = (size_t)chunk * kSyrkKC * ld;
    r.a0 = *(const double2*)(gA + off); r.a1 = *(const double2*)(gA + off + 32);
    r.b0 = *(const double2*)(gB + off); r.b1 = *(const double2*)(gB + off + 32);
    return r;
  };
  auto stage_store = [&](int buf, const Stage& r) {
    const int cs = c2 ^ ((kr & 1) << 4);
    *(double2*)&sAB[buf][0][kr * 64 + cs] = r.a0;
    *(double2*)&sAB[buf][0][kr * 64 + 32 + cs] = r.a1;
    *(double2*)&sAB[buf][1][kr * 64 + cs] = r.b0;
    *(double2*)&sAB[buf][1][kr * 64 + 32 + cs] = r.b1;
  };
  Stage r0 = stage_load(0);
  // In a diagonal tile the sub-block (wi = 32, wj = 0) is the mirror of (0, 32): that wave only stages.  So does a wave whose
  // 32 rows or 32 columns lie in never-used slots (the same rule as for whole tiles above, at the wavefront's granularity:
  // in the mapping workload - 128 columns of capacity, ~55 live, the partially initialised feature at 109 - nine of the
  // sixteen 32 x 32 blocks are live).
  const int i0 = ti * 64 + wi, j0 = tj * 64 + wj;
  const int b_lo = (n_live + 31) >> 5, b_hi = ppos >> 5;
  const bool idle = ((ti == tj) && (wi > wj)) || ((i0 >> 5) >= b_lo && (i0 >> 5) < b_hi) || ((j0 >> 5) >= b_lo && (j0 >> 5) < b_hi);
  v4d acc[2][2];
  for (int it = 0; it < 2; ++it) for (int jt = 0; jt < 2; ++jt) acc[it][jt] = (v4d){0, 0, 0, 0};
  const int nchunk = mp / kSyrkKC;
  // register prefetch runs TWO chunks ahead of the MFMAs (one chunk of work does not cover an
  // HBM round trip under load); chunks alternate between the register sets r0 / r1.
  Stage r1 = r0;
  if (nchunk > 1) r1 = stage_load(1);
  // The two diagonal 32x32 blocks of a diagonal tile are symmetric: their lower-left 16x16 quarter is not computed, the
  // epilogue writes it as the transpose of the upper-right one (interior blocks only: the innovation row / column of the
  // last block keeps the general path).
  const bool interior = (i0 + 32 < ld && j0 + 32 < ld);
  const bool diagw = (ti == tj) && (wi == wj) && interior;
  auto chunk_mfma = [&](int buf, int nks) {
    // k-row 4 ks + hi has the parity of hi: its first / second 16 columns sit at sw / sw ^ 16
    const int sw = (hi & 1) << 4;
    const double* pa = &sAB[buf][0][hi * 64 + wi + lo];
    const double* pb = &sAB[buf][1][hi * 64 + wj + lo];
#pragma unroll
    for (int ks = 0; ks < kSyrkKC / 4; ++ks) {
      if (ks < nks) {
        const double a0 = pa[ks * 256 + sw], a1 = pa[ks * 256 + (sw ^ 16)];
        const double b0 = pb[ks * 256 + sw], b1 = pb[ks * 256 + (sw ^ 16)];
        acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
        acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
        if (!diagw) acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
        acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
      }
    }
  };
  // The P tile of an interior block is fetched in one burst right after the K loop.  (Fetching it under the MFMAs of the
  // last chunk gained 2 % at three workgroups per CU, but holds 32 more registers through the loop: with the 32 KB LDS
  // layout the kernel is better off without the 32 extra registers, 0.535 vs 0.552 ms.  Peeling the last
  // chunk out of the loop made the compiler copy the prefetch registers and wait for every load on the spot: 1.40 ms.)
  double pold[2][2][4];
  auto fetch_tile = [&]() {
    const double* prow = Pb + (size_t)(i0 + hi) * ld + j0 + lo;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) pold[it][jt][r] = prow[(size_t)(16 * it + 4 * r) * ld + 16 * jt];
  };
  const bool want_tile = !idle;
  for (int ch = 0; ch < nchunk; ch += 2) {
    // even chunk: registers r0 -> buffer 0
    stage_store(0, r0);
    __syncthreads();
    if (ch + 2 < nchunk) r0 = stage_load(ch + 2);
    if (!idle) chunk_mfma(0, (ch + 1 >= nchunk) ? nks_last : kSyrkKC / 4);
    if (ch + 1 >= nchunk) break;
    // odd chunk: registers r1 -> buffer 1
    stage_store(1, r1);
    __syncthreads();
    if (ch + 3 < nchunk) r1 = stage_load(ch + 3);
    if (!idle) chunk_mfma(1, (ch + 2 >= nchunk) ? nks_last : kSyrkKC / 4);
    // a buffer is rewritten two chunks after it was read, with a barrier in between: one barrier per chunk suffices
  }
  const int lastbuf = (nchunk - 1) & 1;
  if (want_tile) fetch_tile();
  if (idle) return;
  const bool mirror = (ti != tj) || (wi != wj);
  // Row / column ld-1 (the innovation column riding along) only exists in the last tile row / column, and there only in the
  // last 16 x 16 quarter of a wave's block: every other quarter takes the branch-free path.  (Rounds 1-2 sent the whole
  // 32x32 block of such a wave - 40 of a sequence's 210 quarters, 20 of which hold the column - down the element-wise path
  // with its uncoalesced mirror stores.)
  auto plain = [&](int it, int jt) { return interior || ((i0 + 16 * it + 16 < ld) && (j0 + 16 * jt + 16 < ld)); };
  {
    double* prow = Pb + (size_t)(i0 + hi) * ld + j0 + lo;       // element (i0 + hi, j0 + lo)
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        if (!plain(it, jt)) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (diagw && it == 1 && jt == 0) continue;            // written below as the transpose of quarter (0, 1)
          const double pn = pold[it][jt][r] - acc[it][jt][r];
          prow[(size_t)(16 * it + 4 * r) * ld + 16 * jt] = pn;
          acc[it][jt][r] = pn;
        }
      }
    if (mirror || diagw) {
      // The mirror block goes through LDS so that its stores are row segments of 128 bytes like the direct ones (written
      // straight from the accumulator layout every store instruction touched 16 rows with 32 bytes each: the mirror cost
      // 0.07 ms of the 0.58 ms launch).  The staging buffer that the LAST chunk did not use is free: every wave is past
      // the barrier that followed its last read.  Wave-private region, LDS operations of a wave execute in order.
      const int other = lastbuf ^ 1;
      double* sM = &sAB[other][0][0] + wave * (16 * 17);          // 16 x 16 block at pitch 17, one per wave
      double* pm = Pb + (size_t)(j0 + hi) * ld + i0 + lo;       // element (j0 + hi, i0 + lo)
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          if (diagw && !(it == 0 && jt == 1)) continue;
          if (!plain(it, jt)) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) sM[lo * 17 + 4 * r + hi] = acc[it][jt][r];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int k = 0; k < 4; ++k) pm[(size_t)(16 * jt + 4 * k) * ld + 16 * it] = sM[(4 * k + hi) * 17 + lo];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (interior) return;
  }
  double* xb = x + (size_t)b * ld;
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      if (plain(it, jt)) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i0 + 16 * it + hi + 4 * r, col = j0 + 16 * jt + lo;
        const double v = acc[it][jt][r];
        if (col == ld - 1) {
          if (row != ld - 1) xb[row] += v;  // x += V (L^-1 nu)
        } else if (row != ld - 1) {
          const double pn = pold[it][jt][r] - v;
          Pb[(size_t)row * ld + col] = pn;
          if (mirror) Pb[(size_t)col * ld + row] = pn;
        }
      }
    }
}

#ifdef SL2_TESTING   // section 9 of sl2_ekf_update_testing.inc
#define SL2_EKF_TEST_SECTION 9
#include "sl2_ekf_update_testing.inc"
#undef SL2_EKF_TEST_SECTION
#endif  // SL2_TESTING


// Profiling scopes carry the SYMBOL of the kernel they bracket (so that the bench's per-kernel times and rocprofv3's
// kernel_stats.csv join on the name); "@phase" tells two uses of one kernel apart (k_fwdsub_lds is also the panel solve of
// the large-map Cholesky).
static int launch_update_range(sl2_engine* e) {
  const int B = e->B;     // (succ_idx / m_count, the successful measurements in slot order, come from k_search_score)
#ifdef SL2_TESTING
  const int build_variant = e->root->build_variant, chol_variant = e->root->chol_variant, fwd_variant = e->root->fwd_variant;
#else
  const int chol_variant = 1, fwd_variant = 1;
#endif
#ifdef SL2_TESTING
  if (build_variant == 0) {
    {
      LaunchScope ls(e, "k_build_A", true);
      const int threads = e->ld <= 512 ? e->ld : 512;
      hipLaunchKernelGGL(k_build_A, dim3(B), dim3(threads), 0, e->stream, e->P, e->f_Hx, e->f_Hy, e->f_nu, e->succ_idx,
                         e->m_count, e->At, e->N, e->ld, e->mld);
      SL2_HIP(hipGetLastError());
    }
    {
      LaunchScope ls(e, "k_build_S");
      dim3 grid(e->mld / 32, (e->mld + 255) / 256, B);
      hipLaunchKernelGGL(k_build_S, grid, dim3(256), 0, e->stream, e->At, e->f_Hx, e->f_Hy, e->f_R, e->succ_idx, e->m_count,
                         e->St, e->N, e->ld, e->mld);
      SL2_HIP(hipGetLastError());
    }
  } else
#endif
  {
    // sl2_create admits maps of up to 2048 state columns / 512 measured features
    LaunchScope ls(e, "k_build_AS", true);
    // One workgroup per sequence from batch 1024 on (one exact round of four per CU at 1024; it needs the launch in front
    // of it - k_search_score - to consist of single-wave workgroups, see launch_search).  Smaller batches: ~3000 workgroups
    // in all, so that the chip is full and a sequence's 25 feature batches are not one chain.
    int nsplit = B >= 1024 ? 1 : (3072 + B - 1) / B;
    if (e->root->build_split > 0) nsplit = e->root->build_split;        // experiments (TEST build: SL2_BUILD_SPLIT)
    if (nsplit < 1) nsplit = 1;
    const int nbatch_max = (e->N + kASBatch - 1) / kASBatch;
    if (nsplit > nbatch_max) nsplit = nbatch_max;
    if (e->ld <= 1024) {
      size_t shm = sizeof(double) * 2 * kASBatch * e->ld;
      if ((size_t)e->root->build_lds_min > shm) shm = (size_t)e->root->build_lds_min;
      hipLaunchKernelGGL((k_build_AS<1, kASBatch>), dim3(B, nsplit), dim3(e->ld), shm, e->stream, e->P, e->f_Hx, e->f_Hy, e->f_nu, e->f_R,
                         e->succ_idx, e->m_count, e->At, e->St, e->f_hcol, e->pos_err_any, e->N, e->ld, e->mld);
    } else {
      const size_t shm = sizeof(double) * 2 * 2 * e->ld;      // <= 64 KB
      hipLaunchKernelGGL((k_build_AS<2, 2>), dim3(B, nsplit), dim3(1024), shm, e->stream, e->P, e->f_Hx, e->f_Hy, e->f_nu, e->f_R,
                         e->succ_idx, e->m_count, e->At, e->St, e->f_hcol, e->pos_err_any, e->N, e->ld, e->mld);
    }
    SL2_HIP(hipGetLastError());
  }
  // sl2_create sizes the innovation system so that one of the first two branches always applies (mld a multiple of 128
  // beyond 16 blocks)
  if (e->nblk_max > e->root->panel_from && chol_variant >= 1 && e->mld % 64 == 0 && e->nblk_max % kCholPanelBlocks == 0) {
    int rc = launch_chol_panels(e, B);
    if (rc != SL2_OK) return rc;
  } else if (e->nblk_max <= e->root->panel_from && chol_variant == 1) {
    LaunchScope ls(e, "k_chol_left", true);
    hipLaunchKernelGGL(k_chol_left, dim3(B), dim3(256), 0, e->stream, e->St, e->LinvT, e->m_count, e->mld, e->nblk_max, 0,
                       e->nblk_max, (long long*)e->root->chol_trace);
    SL2_HIP(hipGetLastError());
  } else {
#ifdef SL2_TESTING
    {
      for (int J = 0; J < e->nblk_max; ++J) {
        {
          LaunchScope ls(e, "k_chol_diag");
          hipLaunchKernelGGL(k_chol_diag, dim3(B), dim3(64), 0, e->stream, e->St, e->LinvT, e->m_count, e->mld, e->nblk_max, J);
          SL2_HIP(hipGetLastError());
        }
        const int rem = e->nblk_max - 1 - J;
        if (rem > 0) {
          {
            LaunchScope ls(e, "k_chol_panel");
            hipLaunchKernelGGL(k_chol_panel, dim3(rem, B), dim3(64), 0, e->stream, e->St, e->LinvT, e->m_count, e->mld,
                               e->nblk_max, J);
            SL2_HIP(hipGetLastError());
          }
          {
            LaunchScope ls(e, "k_chol_trail");
            hipLaunchKernelGGL(k_chol_trail, dim3(rem * (rem + 1) / 2, B), dim3(64), 0, e->stream, e->St, e->m_count, e->mld, J);
            SL2_HIP(hipGetLastError());
          }
        }
      }
    }
#else
    set_error("launch_update: no factorisation for this system size (sl2_create should have padded it)");
    return SL2_ERR_INVALID;
#endif
  }
  {
    bool done = false;
    if (fwd_variant == 1) {
      int rc = launch_fwdsub_lds(e, B, &done);
      if (rc != SL2_OK) return rc;
      // the grouped form works on 64-row tiles: it needs mld to be a multiple of 64 (sl2_create pads systems of more
      // than 13 blocks to 64, of more than 16 to 128)
      if (!done && e->mld % 64 == 0) {
        rc = launch_fwdsub_grouped(e, B);
        if (rc != SL2_OK) return rc;
        done = true;
      }
    }
    if (!done) {
#ifdef SL2_TESTING
      LaunchScope ls(e, "k_fwdsub", true);
      hipLaunchKernelGGL(k_fwdsub, dim3(xcd_grid(e->ld / 64, B)), dim3(128), 0, e->stream, e->At, e->Vt, e->St, e->LinvT,
                         e->m_count, e->ld, e->mld, e->nblk_max, B);
      SL2_HIP(hipGetLastError());
#else
      set_error("launch_update: no substitution kernel for this system size (sl2_create should have padded it)");
      return SL2_ERR_INVALID;
#endif
    }
  }
  {
    LaunchScope ls(e, "k_syrk", true);
    const int nt = e->ld / 64;
#ifdef SL2_CHOL_TRACE   // the stamp buffer is the Cholesky's unless SL2_TRACE_SYRK is set (scripts/syrk_clock.py sets it)
    static const bool trace_syrk = getenv("SL2_TRACE_SYRK") != nullptr;
    hipLaunchKernelGGL(k_syrk, dim3(xcd_grid(nt * (nt + 1) / 2, B)), dim3(256), 0, e->stream, e->Vt, e->P, e->x, e->m_count,
                       e->ld, e->mld, B, e->n_slots, e->ppos, trace_syrk ? (long long*)e->root->chol_trace : nullptr);
#else
    hipLaunchKernelGGL(k_syrk, dim3(xcd_grid(nt * (nt + 1) / 2, B)), dim3(256), 0, e->stream, e->Vt, e->P, e->x, e->m_count,
                       e->ld, e->mld, B, e->n_slots, e->ppos);
#endif
    SL2_HIP(hipGetLastError());
  }
  return SL2_OK;
}

int launch_update(sl2_engine* e) { return launch_update_range(e); }

// k_syrk on the given covariance and V^T, everything else the engine's: sl2_create's placement probe (sl2_engine.hip:
// place_large_matrices), which fills m_count / n_slots for the occasion - on an engine that holds nothing but zeros the launch
// does its full work and changes nothing.
int launch_syrk_on(sl2_engine* e, const double* Vt, double* P) {
  const int nt = e->ld / 64;
#ifdef SL2_CHOL_TRACE
  hipLaunchKernelGGL(k_syrk, dim3(xcd_grid(nt * (nt + 1) / 2, e->B)), dim3(256), 0, e->stream, Vt, P, e->x, e->m_count, e->ld, e->mld, e->B,
                     e->n_slots, e->ppos, (long long*)nullptr);
#else
  hipLaunchKernelGGL(k_syrk, dim3(xcd_grid(nt * (nt + 1) / 2, e->B)), dim3(256), 0, e->stream, Vt, P, e->x, e->m_count, e->ld, e->mld, e->B,
                     e->n_slots, e->ppos);
#endif
  SL2_HIP(hipGetLastError());
  return SL2_OK;
}

}  // namespace sl2

#ifdef SL2_TESTING   // section 10 of sl2_ekf_update_testing.inc: everything below is test / calibration code: libscenelib2_amd_test.so only (include/scenelib2_amd_testing.h)
#define SL2_EKF_TEST_SECTION 10
#include "sl2_ekf_update_testing.inc"
#undef SL2_EKF_TEST_SECTION
#endif  // SL2_TESTING
