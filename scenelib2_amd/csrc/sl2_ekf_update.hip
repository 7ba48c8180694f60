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

namespace sl2 {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4d mfma_f64(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// k_compact: successful measurements in selected_feature_list_ order
// (construct_total_measurement_stuff, monoslam.cpp:548-572).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_compact(const int* __restrict__ sel_idx, const int* __restrict__ n_sel,
                                                const int* __restrict__ meas_ok, int* __restrict__ succ_idx,
                                                int* __restrict__ m_count, int N) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int ns = n_sel[b];
  int base = 0;
  for (int k0 = 0; k0 < ns; k0 += 64) {
    const int k = k0 + lane;
    const int flag = (k < ns) ? (meas_ok[(size_t)b * N + k] != 0) : 0;
    const unsigned long long mask = __ballot(flag);
    if (flag) {
      const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
      succ_idx[(size_t)b * N + pos] = sel_idx[(size_t)b * N + k];
    }
    base += __popcll(mask);
  }
  if (lane == 0) m_count[b] = base;
}

// ---------------------------------------------------------------------------
// k_build_A: At[a][i] = (P H^T)[i][a], a = 2j+r for the j-th successful feature.
// Column ld-1 carries the innovation nu (so that L^-1 nu and W nu fall out of the
// same substitution / SYRK).  Rows of padding up to a multiple of 32 are zeroed.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_build_A(const double* __restrict__ P, const double* __restrict__ f_Hx,
                                                 const double* __restrict__ f_Hy, const double* __restrict__ f_nu,
                                                 const int* __restrict__ succ_idx, const int* __restrict__ m_count,
                                                 double* __restrict__ At, int N, int ld, int mld) {
  const int b = blockIdx.z;
  const int cnt = m_count[b];
  if (cnt == 0) return;
  const int cnt_pad = (cnt + 15) / 16 * 16;
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  if (j >= cnt_pad) return;
  double* Ab = At + (size_t)b * mld * ld;
  if (j >= cnt) {
    Ab[(size_t)(2 * j) * ld + i] = 0.0;
    Ab[(size_t)(2 * j + 1) * ld + i] = 0.0;
    return;
  }
  const int f = succ_idx[(size_t)b * N + j];
  const size_t fi = (size_t)b * N + f;
  const int pos = 13 + 3 * f;
  const double* Pb = P + (size_t)b * ld * ld;
  double pc[7], py[3];
  for (int c = 0; c < 7; ++c) pc[c] = (i < 13) ? Pb[(size_t)i * ld + c] : Pb[(size_t)c * ld + i];
  for (int c = 0; c < 3; ++c) py[c] = Pb[(size_t)(pos + c) * ld + i];
  for (int r = 0; r < 2; ++r) {
    double acc = 0.0;
    for (int c = 0; c < 7; ++c) acc += pc[c] * f_Hx[fi * 14 + r * 7 + c];
    for (int c = 0; c < 3; ++c) acc += py[c] * f_Hy[fi * 6 + r * 3 + c];
    if (i == ld - 1) acc = f_nu[fi * 2 + r];
    Ab[(size_t)(2 * j + r) * ld + i] = acc;
  }
}

// ---------------------------------------------------------------------------
// k_build_S: S = H A + R, stored St[c][r] = S[r][c] (both triangles written;
// the factorisation reads r >= c).  Padding: identity.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_build_S(const double* __restrict__ At, const double* __restrict__ f_Hx,
                                                 const double* __restrict__ f_Hy, const double* __restrict__ f_R,
                                                 const int* __restrict__ succ_idx, const int* __restrict__ m_count,
                                                 double* __restrict__ St, int N, int ld, int mld) {
  const int b = blockIdx.z;
  const int cnt = m_count[b];
  if (cnt == 0) return;
  const int cnt_pad = (cnt + 15) / 16 * 16;
  const int j = blockIdx.x * 16 + threadIdx.x;  // row feature
  const int k = blockIdx.y * 16 + threadIdx.y;  // column feature
  if (j >= cnt_pad || k >= cnt_pad || j < k) return;
  double* Sb = St + (size_t)b * mld * mld;
  if (j >= cnt || k >= cnt) {
    for (int r = 0; r < 2; ++r)
      for (int s = 0; s < 2; ++s) {
        const double v = (j == k && r == s) ? 1.0 : 0.0;
        const int a = 2 * j + r, bb = 2 * k + s;
        Sb[(size_t)bb * mld + a] = v;
        Sb[(size_t)a * mld + bb] = v;
      }
    return;
  }
  const int fj = succ_idx[(size_t)b * N + j];
  const size_t fi = (size_t)b * N + fj;
  const int posj = 13 + 3 * fj;
  const double* Ab = At + (size_t)b * mld * ld;
  for (int s = 0; s < 2; ++s) {
    const int bb = 2 * k + s;
    const double* arow = Ab + (size_t)bb * ld;
    double ac[7], ay[3];
    for (int c = 0; c < 7; ++c) ac[c] = arow[c];
    for (int c = 0; c < 3; ++c) ay[c] = arow[posj + c];
    for (int r = 0; r < 2; ++r) {
      double acc = 0.0;
      for (int c = 0; c < 7; ++c) acc += f_Hx[fi * 14 + r * 7 + c] * ac[c];
      for (int c = 0; c < 3; ++c) acc += f_Hy[fi * 6 + r * 3 + c] * ay[c];
      if (j == k && r == s) acc += f_R[fi];
      const int a = 2 * j + r;
      Sb[(size_t)bb * mld + a] = acc;
      if (j != k) Sb[(size_t)a * mld + bb] = acc;
    }
  }
}

// ---------------------------------------------------------------------------
// Blocked Cholesky of S (right-looking over 32x32 blocks, three launches per
// block column J): diag -> panel -> trailing update.
// ---------------------------------------------------------------------------

// k_chol_diag: one wave per sequence factors the (already updated) diagonal block
// in LDS, inverts the 32x32 triangle, writes L_JJ back in place and LinvT.
__global__ void __launch_bounds__(64) k_chol_diag(double* __restrict__ St, double* __restrict__ LinvT,
                                                  const int* __restrict__ m_count, int mld, int nblk_max, int J) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int cnt = m_count[b];
  if (cnt == 0) return;
  const int nblk = (2 * cnt + 31) / 32;
  if (J >= nblk) return;
  __shared__ double D[32][33];
  __shared__ double X[32][33];
  double* Sb = St + (size_t)b * mld * mld;
  const int o = J * 32;
  // D[r][c] = S[o+r][o+c] = St[(o+c)*mld + o+r]
  for (int e = lane; e < 1024; e += 64) {
    const int c = e >> 5, r = e & 31;
    D[r][c] = Sb[(size_t)(o + c) * mld + o + r];
  }
  __syncthreads();
  for (int c = 0; c < 32; ++c) {
    const double d = sqrt(D[c][c]);
    __syncthreads();
    if (lane == c) D[c][c] = d;
    if (lane > c && lane < 32) D[lane][c] = D[lane][c] / d;
    __syncthreads();
    if (lane > c && lane < 32) {
      const double lrc = D[lane][c];
      for (int cc = c + 1; cc <= lane; ++cc) D[lane][cc] -= lrc * D[cc][c];
    }
    __syncthreads();
  }
  // inverse of the lower triangle, one column per lane
  if (lane < 32) {
    const int j = lane;
    for (int i = 0; i < j; ++i) X[i][j] = 0.0;
    X[j][j] = 1.0 / D[j][j];
    for (int i = j + 1; i < 32; ++i) {
      double s = 0.0;
      for (int p = j; p < i; ++p) s -= D[i][p] * X[p][j];
      X[i][j] = s / D[i][i];
    }
  }
  __syncthreads();
  double* Lb = LinvT + ((size_t)b * nblk_max + J) * 1024;
  for (int e = lane; e < 1024; e += 64) {
    const int c = e >> 5, r = e & 31;
    // L block in place (zero above the diagonal), LinvT[p][k] = Linv[k][p]
    Sb[(size_t)(o + c) * mld + o + r] = (r >= c) ? D[r][c] : 0.0;
    Lb[c * 32 + r] = X[r][c];
  }
}

// k_chol_panel: L[I][J] = S[I][J] * L_JJ^-T for every block row I > J; one wave
// per 32x32 tile.  In k-major storage: new[k][i] = sum_p Linv[k][p] * St[J+p][I+i].
__global__ void __launch_bounds__(64) k_chol_panel(double* __restrict__ St, const double* __restrict__ LinvT,
                                                   const int* __restrict__ m_count, int mld, int nblk_max, int J) {
  const int b = blockIdx.y, lane = threadIdx.x;
  const int cnt = m_count[b];
  if (cnt == 0) return;
  const int nblk = (2 * cnt + 31) / 32;
  const int I = J + 1 + blockIdx.x;
  if (I >= nblk) return;
  const int lo = lane & 15, hi = lane >> 4;
  double* Sb = St + (size_t)b * mld * mld;
  const double* Lb = LinvT + ((size_t)b * nblk_max + J) * 1024;
  v4d acc[2][2];
  for (int kt = 0; kt < 2; ++kt) for (int it = 0; it < 2; ++it) acc[kt][it] = (v4d){0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int p = 4 * s + hi;
    const double a0 = Lb[p * 32 + lo], a1 = Lb[p * 32 + 16 + lo];
    const double b0 = Sb[(size_t)(J * 32 + p) * mld + I * 32 + lo];
    const double b1 = Sb[(size_t)(J * 32 + p) * mld + I * 32 + 16 + lo];
    acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
    acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
    acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
    acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
  }
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Sb[(size_t)(J * 32 + 16 * kt + hi + 4 * r) * mld + I * 32 + 16 * it + lo] = acc[kt][it][r];
}

// k_chol_trail: S[I][K] -= L[I][J] L[K][J]^T for J < K <= I; one wave per tile.
__global__ void __launch_bounds__(64) k_chol_trail(double* __restrict__ St, const int* __restrict__ m_count, int mld, int J) {
  const int b = blockIdx.y, lane = threadIdx.x;
  const int cnt = m_count[b];
  if (cnt == 0) return;
  const int nblk = (2 * cnt + 31) / 32;
  // tile index -> (I, K), lower triangle of the trailing matrix, row-major
  int t = blockIdx.x, ri = 0;
  while (t > ri) { t -= ri + 1; ++ri; }
  const int I = J + 1 + ri, K = J + 1 + t;
  if (I >= nblk) return;
  const int lo = lane & 15, hi = lane >> 4;
  double* Sb = St + (size_t)b * mld * mld;
  v4d acc[2][2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[jt][it][r] = Sb[(size_t)(K * 32 + 16 * jt + hi + 4 * r) * mld + I * 32 + 16 * it + lo];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const size_t row = (size_t)(J * 32 + 4 * s + hi) * mld;
    const double a0 = -Sb[row + K * 32 + lo], a1 = -Sb[row + K * 32 + 16 + lo];
    const double b0 = Sb[row + I * 32 + lo], b1 = Sb[row + I * 32 + 16 + lo];
    acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
    acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
    acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
    acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
  }
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Sb[(size_t)(K * 32 + 16 * jt + hi + 4 * r) * mld + I * 32 + 16 * it + lo] = acc[jt][it][r];
}

// ---------------------------------------------------------------------------
// k_fwdsub: Vt = L^-1 At by blocked forward substitution.  Columns are
// independent: each wave owns 16 columns and walks the block rows J in order,
//   acc = At[J] - sum_{K<J} L[J][K] Vt[K] ;  Vt[J] = L_JJ^-1 acc.
// A lane re-reads only Vt elements it stored itself (D fragment == B fragment).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fwdsub(const double* __restrict__ At, double* Vt, const double* __restrict__ St,
                                                const double* __restrict__ LinvT, const int* __restrict__ m_count, int ld,
                                                int mld, int nblk_max) {
  const int b = blockIdx.y;
  const int cnt = m_count[b];
  if (cnt == 0) return;
  const int nblk = (2 * cnt + 31) / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lo = lane & 15, hi = lane >> 4;
  const int i0 = blockIdx.x * 64 + wave * 16;
  const double* Ab = At + (size_t)b * mld * ld;
  double* Vb = Vt + (size_t)b * mld * ld;
  const double* Sb = St + (size_t)b * mld * mld;
  for (int J = 0; J < nblk; ++J) {
    v4d acc[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[jt][r] = Ab[(size_t)(J * 32 + 16 * jt + hi + 4 * r) * ld + i0 + lo];
    for (int kk = 0; kk < J * 32; kk += 4) {
      const size_t row = (size_t)(kk + hi);
      const double a0 = -Sb[row * mld + J * 32 + lo], a1 = -Sb[row * mld + J * 32 + 16 + lo];
      const double bv = Vb[row * ld + i0 + lo];
      acc[0] = mfma_f64(a0, bv, acc[0]);
      acc[1] = mfma_f64(a1, bv, acc[1]);
    }
    const double* Lb = LinvT + ((size_t)b * nblk_max + J) * 1024;
    v4d out[2] = {(v4d){0, 0, 0, 0}, (v4d){0, 0, 0, 0}};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int p = 4 * s + hi;
      const double a0 = Lb[p * 32 + lo], a1 = Lb[p * 32 + 16 + lo];
      const double bs = acc[s >> 2][s & 3];
      out[0] = mfma_f64(a0, bs, out[0]);
      out[1] = mfma_f64(a1, bs, out[1]);
    }
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) Vb[(size_t)(J * 32 + 16 * jt + hi + 4 * r) * ld + i0 + lo] = out[jt][r];
  }
}

// ---------------------------------------------------------------------------
// k_syrk: P -= V V^T on 64x64 tiles of the upper block triangle (ti <= tj),
// mirrored to the lower one; 4 waves per tile, 32x32 per wave.  The column
// ld-1 of Vt is w = L^-1 nu, so the same product yields x += V w there.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_syrk(const double* __restrict__ Vt, double* __restrict__ P, double* __restrict__ x,
                                              const int* __restrict__ m_count, int ld, int mld) {
  const int b = blockIdx.y;
  const int cnt = m_count[b];
  if (cnt == 0) return;
  const int mp = (2 * cnt + 3) / 4 * 4;
  int t = blockIdx.x, tj = 0;
  while (t > tj) { t -= tj + 1; ++tj; }
  const int ti = t;  // ti <= tj
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lo = lane & 15, hi = lane >> 4;
  const int i0 = ti * 64 + (wave >> 1) * 32, j0 = tj * 64 + (wave & 1) * 32;
  const double* Vb = Vt + (size_t)b * mld * ld;
  double* Pb = P + (size_t)b * ld * ld;
  v4d acc[2][2];
  for (int it = 0; it < 2; ++it) for (int jt = 0; jt < 2; ++jt) acc[it][jt] = (v4d){0, 0, 0, 0};
  const double* xp = Vb + (size_t)hi * ld + i0 + lo;
  const double* yp = Vb + (size_t)hi * ld + j0 + lo;
#pragma unroll 4
  for (int k = 0; k < mp; k += 4) {
    const double a0 = xp[0], a1 = xp[16], b0 = yp[0], b1 = yp[16];
    acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
    acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
    acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
    acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
    xp += (size_t)4 * ld;
    yp += (size_t)4 * ld;
  }
  double* xb = x + (size_t)b * ld;
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i0 + 16 * it + hi + 4 * r, col = j0 + 16 * jt + lo;
        const double v = acc[it][jt][r];
        if (col == ld - 1) {
          if (row != ld - 1) xb[row] += v;  // x += V (L^-1 nu)
        } else if (row != ld - 1) {
          const double pn = Pb[(size_t)row * ld + col] - v;
          Pb[(size_t)row * ld + col] = pn;
          if (ti != tj) Pb[(size_t)col * ld + row] = pn;
        }
      }
}

// ---------------------------------------------------------------------------
// Debug GEMM on the same fragment conventions (tests the MFMA layout):
// C[m][n] = sum_k XT[k][m] YT[k][n], one wave per 32x32 tile.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_gemm_kt(const double* __restrict__ XT, int ldx, const double* __restrict__ YT, int ldy,
                                                int K, double* __restrict__ C, int ldc) {
  const int lane = threadIdx.x;
  const int lo = lane & 15, hi = lane >> 4;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  v4d acc[2][2];
  for (int it = 0; it < 2; ++it) for (int jt = 0; jt < 2; ++jt) acc[it][jt] = (v4d){0, 0, 0, 0};
  for (int k = 0; k < K; k += 4) {
    const double a0 = XT[(size_t)(k + hi) * ldx + i0 + lo], a1 = XT[(size_t)(k + hi) * ldx + i0 + 16 + lo];
    const double b0 = YT[(size_t)(k + hi) * ldy + j0 + lo], b1 = YT[(size_t)(k + hi) * ldy + j0 + 16 + lo];
    acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
    acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
    acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
    acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
  }
  for (int it = 0; it < 2; ++it)
    for (int jt = 0; jt < 2; ++jt)
      for (int r = 0; r < 4; ++r) C[(size_t)(i0 + 16 * it + hi + 4 * r) * ldc + j0 + 16 * jt + lo] = acc[it][jt][r];
}

int launch_update(sl2_engine* e) {
  const int B = e->B;
  {
    LaunchScope ls(e, "k_compact");
    hipLaunchKernelGGL(k_compact, dim3(B), dim3(64), 0, e->stream, e->sel_idx, e->n_sel, e->meas_ok, e->succ_idx, e->m_count, e->N);
    SL2_HIP(hipGetLastError());
  }
  const int cnt_pad_max = e->mld / 2;  // feature pairs incl. padding
  {
    LaunchScope ls(e, "k_build_A");
    dim3 grid(e->ld / 64, (cnt_pad_max + 3) / 4, B);
    hipLaunchKernelGGL(k_build_A, grid, dim3(64, 4), 0, e->stream, e->P, e->f_Hx, e->f_Hy, e->f_nu, e->succ_idx, e->m_count,
                       e->At, e->N, e->ld, e->mld);
    SL2_HIP(hipGetLastError());
  }
  {
    LaunchScope ls(e, "k_build_S");
    dim3 grid((cnt_pad_max + 15) / 16, (cnt_pad_max + 15) / 16, B);
    hipLaunchKernelGGL(k_build_S, grid, dim3(16, 16), 0, e->stream, e->At, e->f_Hx, e->f_Hy, e->f_R, e->succ_idx, e->m_count,
                       e->St, e->N, e->ld, e->mld);
    SL2_HIP(hipGetLastError());
  }
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
  {
    LaunchScope ls(e, "k_fwdsub");
    hipLaunchKernelGGL(k_fwdsub, dim3(e->ld / 64, B), dim3(256), 0, e->stream, e->At, e->Vt, e->St, e->LinvT, e->m_count,
                       e->ld, e->mld, e->nblk_max);
    SL2_HIP(hipGetLastError());
  }
  {
    LaunchScope ls(e, "k_syrk");
    const int nt = e->ld / 64;
    hipLaunchKernelGGL(k_syrk, dim3(nt * (nt + 1) / 2, B), dim3(256), 0, e->stream, e->Vt, e->P, e->x, e->m_count, e->ld, e->mld);
    SL2_HIP(hipGetLastError());
  }
  return SL2_OK;
}

}  // namespace sl2

extern "C" int sl2_debug_gemm_kt(int device, const double* XT, int ldx, const double* YT, int ldy, int M, int N, int K,
                                 double* C, int ldc) {
  using namespace sl2;
  if (!XT || !YT || !C || M % 32 || N % 32 || K % 4 || M <= 0 || N <= 0 || K <= 0) return SL2_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device"); return SL2_ERR_NO_DEVICE; }
  SL2_HIP(hipSetDevice(device));
  double *dX = nullptr, *dY = nullptr, *dC = nullptr;
  SL2_HIP(hipMalloc(&dX, sizeof(double) * (size_t)K * ldx));
  SL2_HIP(hipMalloc(&dY, sizeof(double) * (size_t)K * ldy));
  SL2_HIP(hipMalloc(&dC, sizeof(double) * (size_t)M * ldc));
  SL2_HIP(hipMemcpy(dX, XT, sizeof(double) * (size_t)K * ldx, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(dY, YT, sizeof(double) * (size_t)K * ldy, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_gemm_kt, dim3(N / 32, M / 32), dim3(64), 0, 0, dX, ldx, dY, ldy, K, dC, ldc);
  SL2_HIP(hipGetLastError());
  SL2_HIP(hipDeviceSynchronize());
  SL2_HIP(hipMemcpy(C, dC, sizeof(double) * (size_t)M * ldc, hipMemcpyDeviceToHost));
  hipFree(dX); hipFree(dY); hipFree(dC);
  return SL2_OK;
}
