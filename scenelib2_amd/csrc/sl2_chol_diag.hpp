// The diagonal-block routine of the blocked Cholesky (Eigen::LLT + matrixL().inverse() of kalman.cpp:104-107 for one 32x32
// block): d_column<0>(...) turns [A; I] into [L; L^-T] by column Cholesky in the registers of ONE wavefront, a row per lane
// (lanes 0..31: the diagonal tile, lanes 32..63: the identity).  Shared by k_chol_left (sl2_ekf_update.hip) and by the fused
// small-map kernel k_small_back (sl2_small.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace sl2 {

constexpr int kLinvPitch = 33;   // LDS row pitch of the inverted diagonal block (conflict-free row-per-lane writes)

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}

// ---- the D wave's routine: [A; I] -> [L; L^-T] by column Cholesky, a row per lane (lanes 0..31: the diagonal tile, lanes
// 32..63: the identity), 32 columns unrolled ------------------------------------------------------------------------------
// A wave on its own issues one instruction every ~2.5 ns whatever its kind (s_waitcnt and s_nop included), dependent or not
// (scripts/dwave_bench.hip: the routine's time follows its instruction count, 2300 -> 5.9 us): the routine is bound by the
// number of instructions, not by the "dependent chain" the first rounds blamed.  Column c needs the scalars l[cc], cc > c,
// in every lane.  Through v_readlane that is three instructions a product (two v_readlane_b32, one v_fma_f64 with an SGPR
// pair): 1488 of the 2300.  Here only the product the next pivot waits for goes that way; the others are stored to LDS as a
// column and come back as wave-uniform ds_read_b128 (two scalars an instruction) one column later: 1.5 instructions a
// product.  The compiler places a ds_read right in front of its use and waits on the spot (that form measured no gain at
// all), so the reads, their s_waitcnt and the order of the pieces are pinned with asm statements: the scalars of a column
// are fetched in two halves, each as soon as the registers of the same half of the previous column are free, and are in
// flight over the other half's products and the next column's pivot arithmetic.  No masking of the upper triangle (those
// lanes' values are never read by another lane and never stored), 1 / sqrt by v_rsq_f64 and one cubic step.
// 1436 instructions, 3.65 us a block alone (5.9 before), 4.3 with a wave on every SIMD (6.5).
typedef double DPair __attribute__((ext_vector_type(2)));
template <int K> struct DCol {                         // column K: products cc = K+1 .. K+F through v_readlane, the rest through LDS
  static constexpr int F = (11 - K) > 1 ? (11 - K) : 1;  // at most 20 scalars of a column in registers at a time (24: spills at the 128-register budget)
  static constexpr int B0 = K + F + 1;
  static constexpr int NB = (32 - B0) > 0 ? (32 - B0) : 0;
  static constexpr bool ODD = (B0 & 1) != 0;
  static constexpr int P0 = B0 + (ODD ? 1 : 0);        // first index of the aligned pairs
  static constexpr int NP = (32 - P0) > 0 ? (32 - P0) / 2 : 0;
  static constexpr int H = (NP + 1) / 2;               // pairs guarded by the first wait
};
struct DScal { double s; DPair p[12]; };
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
  if constexpr (C > 0) d_apply_a<C - 1>(t, a, C + 1 < 32 ? a[C + 1 < 32 ? C + 1 : 31] : l);
  d_read_a<C>(t, colbase, a[DCol<C>::P0 + 2 * DCol<C>::H - 1 < 32 ? DCol<C>::P0 + 2 * DCol<C>::H - 1 : 31]);
  if constexpr (C > 0) d_apply_b<C - 1>(t, a);
  d_read_b<C>(t, colbase, a[31]);
  if constexpr (C > 0) row_store[C - 1] = a[C - 1];    // column C - 1 is done with
  if constexpr (C + 1 < 32) d_column<C + 1>(a, t, colbase, col_store, row_store);
}

}  // namespace sl2
