// Development aid: the 32x32 "factor + invert in registers" routine of k_chol_left's D wave in isolation, in several
// forms, timed per block (one wave alone = latency; 1024 workgroups of one wave = one per SIMD, throughput).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast scripts/dwave_bench.hip -o scripts/_dwave_bench
//   gpurun -- scripts/_dwave_bench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}
template <int NEWTON>
__device__ __forceinline__ double rsqrt_n(double p) {
  double y = __builtin_amdgcn_rsq(p);
  if (NEWTON >= 1) y = y * __builtin_fma(-0.5 * p * y, y, 1.5);
  if (NEWTON >= 2) y = y * __builtin_fma(-0.5 * p * y, y, 1.5);
  return y;
}

constexpr int kPitch = 33;

// ---- V6: off-chain scalars through LDS with the reads, their waits and the chain placed by hand -------------------------
// A single wave issues one instruction per four cycles whatever its kind (s_waitcnt and s_nop included): the routine is
// bound by its instruction count.
typedef double v2d __attribute__((ext_vector_type(2)));
template <int K> struct DCol {                         // column K: products cc = K+1 .. K+F through v_readlane, the rest through LDS
  static constexpr int F = (11 - K) > 1 ? (11 - K) : 1;  // at most 20 scalars of a column in registers at a time (24: spills at the 128-register budget)
  static constexpr int B0 = K + F + 1;
  static constexpr int NB = (32 - B0) > 0 ? (32 - B0) : 0;
  static constexpr bool ODD = (B0 & 1) != 0;
  static constexpr int P0 = B0 + (ODD ? 1 : 0);        // first index of the aligned pairs
  static constexpr int NP = (32 - P0) > 0 ? (32 - P0) / 2 : 0;
  static constexpr int H = (NP + 1) / 2;               // pairs guarded by the first wait
};
struct DScal { double s; v2d p[12]; };
template <int K, int I, int END> __device__ __forceinline__ void d_read_pairs(DScal& t, unsigned colbase) {
  if constexpr (I < END) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t.p[I]) : "v"(colbase), "i"(((K & 1) * 64 + DCol<K>::P0 + 2 * I) * 8) : "memory");
    d_read_pairs<K, I + 1, END>(t, colbase);
  }
}
// The scalars of column K are fetched in two halves, each as soon as the registers of the same half of column K - 1 are
// free, so that a half is in flight over the other half's products AND the next column's chain (the chain alone, ~16
// instructions, does not cover a burst of eight ds_read_b128).
template <int K> struct DHalfA { static constexpr int N = DCol<K>::NB > 0 ? (DCol<K>::ODD ? 1 : 0) + DCol<K>::H : 0; };
template <int K> __device__ __forceinline__ void d_read_a(DScal& t, unsigned colbase, double& token) {
  if constexpr (DCol<K>::NB > 0) {
    asm volatile("" : "+v"(token) :: "memory");
    if constexpr (DCol<K>::ODD)
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t.s) : "v"(colbase), "i"(((K & 1) * 64 + DCol<K>::B0) * 8) : "memory");
    d_read_pairs<K, 0, DCol<K>::H>(t, colbase);
  }
}
template <int K> __device__ __forceinline__ void d_read_b(DScal& t, unsigned colbase, double& token) {
  if constexpr (DCol<K>::NP > DCol<K>::H) {
    asm volatile("" : "+v"(token) :: "memory");
    d_read_pairs<K, DCol<K>::H, DCol<K>::NP>(t, colbase);
  }
}
template <int K, int I, int END> __device__ __forceinline__ void d_guard(DScal& t) {   // the products with p[I..END) stay behind the wait
  if constexpr (I < END) {
    asm volatile("" : "+v"(t.p[I]));
    d_guard<K, I + 1, END>(t);
  }
}
template <int K, int I, int END> __device__ __forceinline__ void d_fma_pairs(DScal& t, double (&a)[32]) {
  if constexpr (I < END) {
    constexpr int cc = DCol<K>::P0 + 2 * I;
    a[cc] = __builtin_fma(-a[K], t.p[I].x, a[cc]);
    a[cc + 1] = __builtin_fma(-a[K], t.p[I].y, a[cc + 1]);
    d_fma_pairs<K, I + 1, END>(t, a);
  }
}
template <int K> __device__ __forceinline__ void d_apply_a(DScal& t, double (&a)[32], double after) {
  using D = DCol<K>;
  if constexpr (D::NB > 0) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(t.s) : "v"(after), "i"(D::NP - D::H));     // behind it: the second half
    d_guard<K, 0, D::H>(t);
    if constexpr (D::ODD) a[D::B0] = __builtin_fma(-a[K], t.s, a[D::B0]);
    d_fma_pairs<K, 0, D::H>(t, a);
  }
}
template <int K> __device__ __forceinline__ void d_apply_b(DScal& t, double (&a)[32]) {
  using D = DCol<K>;
  if constexpr (D::NP > D::H) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(t.p[D::H]) : "i"(DHalfA<K + 1>::N));      // behind it: the next column's first half
    d_guard<K, D::H + 1, D::NP>(t);
    d_fma_pairs<K, D::H, D::NP>(t, a);
  }
}
// 1 / sqrt(p): v_rsq_f64 (5e-8) and one cubically convergent step, five dependent instructions (two Newton steps: seven)
__device__ __forceinline__ double rsqrt_halley(double p) {
  const double y = __builtin_amdgcn_rsq(p);
  const double r = __builtin_fma(-(p * y), y, 1.0);
  return __builtin_fma(y * r, __builtin_fma(r, 0.375, 0.5), y);
}
template <int C> __device__ __forceinline__ void d_column(double (&a)[32], DScal& t, unsigned colbase, double* col_store,
                                                          double* row_store) {
  const double piv = readlane_f64(a[C], C);
  const double l = a[C] * rsqrt_halley(piv);
  a[C] = l;
  if constexpr (DCol<C>::NB > 0) col_store[(C & 1) * 64] = l;
#pragma unroll
  for (int f = 1; f <= DCol<C>::F; ++f)
    if (C + f < 32) a[C + f] = __builtin_fma(-l, readlane_f64(l, C + f), a[C + f]);
  if constexpr (C > 0) d_apply_a<C - 1>(t, a, C + 1 < 32 ? a[C + 1] : l);
  d_read_a<C>(t, colbase, a[DCol<C>::P0 + 2 * DCol<C>::H - 1 < 32 ? DCol<C>::P0 + 2 * DCol<C>::H - 1 : 31]);
  if constexpr (C > 0) d_apply_b<C - 1>(t, a);
  d_read_b<C>(t, colbase, a[31]);
  if constexpr (C > 0) row_store[C - 1] = a[C - 1];    // column C - 1 is done with
  if constexpr (C + 1 < 32) d_column<C + 1>(a, t, colbase, col_store, row_store);
}

// V = 0: every broadcast through v_readlane (round 2).   1: without the Newton steps (timing probe; wrong digits).
//     2: only the on-chain product of every column (timing probe; wrong result).
//     3: on-chain product through v_readlane, the others through LDS one column later (compiler-scheduled).
//     4: like 3, the column's scalars fetched in one burst at the top of the iteration and finished columns retired to LDS.
//     5: pivots ahead: the next pivot from the previous one without waiting for the column (s = a_cc - a_c^2 / piv).
template <int V>
__global__ void __launch_bounds__(64) k_dwave(const double* __restrict__ in, double* __restrict__ out, int iters) {
  __shared__ double sTile[32][kPitch];
  __shared__ double sLinv[32 * kPitch];
  __shared__ __attribute__((aligned(16))) double sCol[2][64];
  __shared__ double sIdent[32][kPitch];
  __shared__ double sDump[32][kPitch];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) { sTile[i >> 5][i & 31] = in[i]; sIdent[i >> 5][i & 31] = ((i >> 5) == (i & 31)) ? 1.0 : 0.0; }
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    int lane_j = lane;
    asm volatile("" : "+v"(lane_j));
    const int r = lane_j & 31;
    const bool low = lane_j < 32;
    double a[32];
    if (V == 6) {
      const double* src = low ? &sTile[r][0] : &sIdent[r][0];
#pragma unroll
      for (int c = 0; c < 32; ++c) a[c] = src[c];
      DScal t;
      t.s = 0.0;
      const unsigned colbase = (unsigned)(size_t)(__attribute__((address_space(3))) double*)&sCol[0][0];
      double* rows = low ? &sDump[r][0] : &sLinv[r * kPitch];   // (in k_chol_left the rows of L go back into sTile)
      d_column<0>(a, t, colbase, &sCol[0][lane_j], rows);
      rows[31] = a[31];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      continue;
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const double v = sTile[r][c];
      a[c] = low ? v : ((r == c) ? 1.0 : 0.0);
    }
    if (V == 0 || V == 1 || V == 2) {
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const double piv = readlane_f64(a[c], c);
        const double dinv = (V == 1) ? rsqrt_n<0>(piv) : rsqrt_n<2>(piv);
        const double l = (!low || r >= c) ? a[c] * dinv : 0.0;
        a[c] = l;
#pragma unroll
        for (int cc = c + 1; cc < (V == 2 ? (c + 2 < 32 ? c + 2 : 32) : 32); ++cc) a[cc] -= l * readlane_f64(l, cc);
      }
    } else if (V == 3) {
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const double piv = readlane_f64(a[c], c);
        const double dinv = rsqrt_n<2>(piv);
        const double l = (!low || r >= c) ? a[c] * dinv : 0.0;
        a[c] = l;
        if (c + 1 < 32) {
          if (c > 0) a[c + 1] -= a[c - 1] * sCol[(c - 1) & 1][c + 1];
          a[c + 1] -= l * readlane_f64(l, c + 1);
          if (c + 2 < 32) sCol[c & 1][lane_j] = l;
          __builtin_amdgcn_wave_barrier();
          if (c > 0) {
#pragma unroll
            for (int cc = c + 2; cc < 32; ++cc) a[cc] -= a[c - 1] * sCol[(c - 1) & 1][cc];
          }
        }
      }
    } else if (V == 4) {
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        // scalars of column c - 1 (stored one iteration ago): all requested now, consumed after this column's chain
        double t[32];
        if (c > 0) {
#pragma unroll
          for (int cc = c + 1; cc < 32; ++cc) t[cc] = sCol[(c - 1) & 1][cc];
        }
        __builtin_amdgcn_sched_barrier(0);
        const double piv = readlane_f64(a[c], c);
        const double dinv = rsqrt_n<2>(piv);
        const double l = (!low || r >= c) ? a[c] * dinv : 0.0;
        if (c + 2 < 32) sCol[c & 1][lane_j] = l;
        if (c + 1 < 32) {
          if (c > 0) a[c + 1] -= a[c - 1] * t[c + 1];
          a[c + 1] -= l * readlane_f64(l, c + 1);
        }
        if (c > 0) {
#pragma unroll
          for (int cc = c + 2; cc < 32; ++cc) a[cc] -= a[c - 1] * t[cc];
          if (!low) sLinv[r * kPitch + c - 1] = a[c - 1];     // column c - 1 is done with
        }
        a[c] = l;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!low) sLinv[r * kPitch + 31] = a[31];
    } else if (V == 5) {
      // The pivot recurrence on its own: piv_{c+1} = a[c+1][c+1] - sum_k l[c+1][k]^2 is known as soon as row c + 1 of the
      // partial factor is: s = a_{c+1} (lane c + 1, all updates but column c's) - (a_{c+1,c})^2 / piv_c, one division-free
      // step from the previous pivot when 1 / piv_c = dinv_c^2.
      double piv = readlane_f64(a[0], 0);
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const double dinv = rsqrt_n<2>(piv);
        const double l = (!low || r >= c) ? a[c] * dinv : 0.0;
        a[c] = l;
        if (c + 1 < 32) {
          if (c > 0) a[c + 1] -= a[c - 1] * sCol[(c - 1) & 1][c + 1];
          // next pivot, from registers that do not wait for l: lane c + 1 holds a[c][c+1] (before scaling) in a[c]
          a[c + 1] -= l * readlane_f64(l, c + 1);
          piv = readlane_f64(a[c + 1], c + 1);
          if (c + 2 < 32) sCol[c & 1][lane_j] = l;
          __builtin_amdgcn_wave_barrier();
          if (c > 0) {
#pragma unroll
            for (int cc = c + 2; cc < 32; ++cc) a[cc] -= a[c - 1] * sCol[(c - 1) & 1][cc];
          }
        }
      }
    }
    if (V != 4) {
      if (!low) {
#pragma unroll
        for (int c = 0; c < 32; ++c) sLinv[r * kPitch + c] = a[c];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (blockIdx.x == 0)
    for (int i = lane; i < 1024; i += 64) out[i] = sLinv[(i >> 5) * kPitch + (i & 31)];   // out[p][k] = Linv[k][p]
}

template <int V>
static void run(const double* d_in, double* d_out, const std::vector<double>& A, const char* what) {
  const int iters = 200;
  std::vector<double> got(1024);
  for (int grid : {1, 1024}) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_dwave<V>, dim3(grid), dim3(64), 0, 0, d_in, d_out, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_dwave<V>, dim3(grid), dim3(64), 0, 0, d_in, d_out, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(got.data(), d_out, 1024 * 8, hipMemcpyDeviceToHost));
    // check: Linv * A * Linv^T = I
    double worst = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double s = 0;
        for (int p = 0; p < 32; ++p)
          for (int q = 0; q < 32; ++q) s += got[p * 32 + i] * A[p * 32 + q] * got[q * 32 + j];
        worst = fmax(worst, fabs(s - (i == j)));
      }
    printf("V%d %-58s grid %4d: %7.3f us per block   |Linv A Linv^T - I| = %.2e\n", V, what, grid, ms * 1e3 / iters, worst);
  }
}

int main() {
  std::vector<double> M(1024), A(1024);
  srand(7);
  for (auto& v : M) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double s = (i == j) ? 4.0 : 0.0;
      for (int k = 0; k < 32; ++k) s += M[i * 32 + k] * M[j * 32 + k];
      A[i * 32 + j] = s;
    }
  double *d_in, *d_out;
  CK(hipMalloc(&d_in, 8192)); CK(hipMalloc(&d_out, 8192));
  CK(hipMemcpy(d_in, A.data(), 8192, hipMemcpyHostToDevice));
  run<0>(d_in, d_out, A, "all v_readlane (round 2)");
  run<1>(d_in, d_out, A, "probe: no Newton steps");
  run<2>(d_in, d_out, A, "probe: on-chain products only");
  run<3>(d_in, d_out, A, "off-chain scalars through LDS");
  run<4>(d_in, d_out, A, "LDS, burst at the top, columns retired");
  run<5>(d_in, d_out, A, "LDS, pivot read before the column store");
  run<6>(d_in, d_out, A, "LDS by hand: reads | chain | waits + products");
  return 0;
}
