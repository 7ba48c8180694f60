// Elliptical normalised-SSD patch search — MonoSLAM::elliptical_search
// (monoslam.cpp:401-477) + correlate2_warning (improc/improc.cpp:55-134).
//
// One 64-lane wavefront per (sequence, selected feature).  Integer sums are exact
// int32; the score epilogue is the reference's FP64 expression (ncc_score in
// sl2_math.hpp, -ffp-contract=off) so the arg-min, the "<=" last-candidate-wins
// tie rule (Q2), the sigma >= 10 tests (Q3) and the 0.40 threshold are decided on
// bit-identical numbers.
//
// Variant 0 ("baseline"): lane = candidate, 121-pixel loop straight from
// global/L2, full FP64 epilogue per candidate.  Kept as the simple, obviously
// faithful kernel that the faster variants are cross-checked against.
#include "sl2_common.hpp"

namespace sl2 {

struct SearchResult {
  int ok, found, u, v, ncand;
  double score;
};

// Wave-wide arg-min with the reference's sequential semantics: scanning in
// candidate order with "corr <= corrmax" means the winner is the candidate with
// the smallest score and, among equals, the LARGEST order index.
__device__ __forceinline__ void wave_argmin(double& best, int& best_order) {
  for (int off = 32; off > 0; off >>= 1) {
    const double ob = __shfl_xor(best, off, 64);
    const int oo = __shfl_xor(best_order, off, 64);
    if (ob < best || (ob == best && oo > best_order)) { best = ob; best_order = oo; }
  }
}

__device__ __forceinline__ SearchResult search_core_v0(const uint8_t* __restrict__ image, int width, int height,
                                       const uint8_t* __restrict__ patch, const double centre[2], double a, double b,
                                       double c) {
  const int lane = threadIdx.x & 63;
  const SearchBounds sb = search_bounds(centre, a, b, c, width, height);
  const int nu = sb.urelfinish - sb.urelstart + 1;
  const int nv = sb.vrelfinish - sb.vrelstart + 1;
  // template sums (wave-uniform; every lane computes them redundantly)
  int Sg0 = 0, Sg0sq = 0;
  for (int p = 0; p < 121; ++p) { const int g = patch[p]; Sg0 += g; Sg0sq += g * g; }
  double best = 1000000.0;  // corrmax initial value, monoslam.cpp:444
  int best_order = -1;
  int ncand = 0;
  if (nu > 0 && nv > 0) {
    const int total = nu * nv;
    for (int idx = lane; idx < total; idx += 64) {
      const int urel = sb.urelstart + idx / nv;
      const int vrel = sb.vrelstart + idx % nv;
      if (!in_ellipse(a, b, c, urel, vrel)) continue;
      ++ncand;
      const int x1 = sb.ucentre + urel - 5, y1 = sb.vcentre + vrel - 5;
      const uint8_t* p1 = image + (size_t)y1 * width + x1;
      int Sg1 = 0, Sg0g1 = 0, Sg1sq = 0;
      for (int r = 0; r < 11; ++r)
        for (int cc = 0; cc < 11; ++cc) {
          const int g0 = patch[r * 11 + cc];
          const int g1 = p1[r * width + cc];
          Sg1 += g1; Sg0g1 += g0 * g1; Sg1sq += g1 * g1;
        }
      double sd0, sd1;
      const double corr = ncc_score(Sg0, Sg1, Sg0g1, Sg0sq, Sg1sq, &sd0, &sd1);
      // accept rule of monoslam.cpp:457-466; within a lane the scan is in order
      if (corr <= best && !(sd0 < kCorrelationSigmaThreshold) && !(sd1 < kCorrelationSigmaThreshold)) {
        best = corr;
        best_order = idx;
      }
    }
  }
  wave_argmin(best, best_order);
  for (int off = 32; off > 0; off >>= 1) ncand += __shfl_xor(ncand, off, 64);
  SearchResult r;
  r.ncand = ncand;
  r.score = best;
  r.u = 0; r.v = 0;
  r.found = best_order >= 0;
  if (best_order >= 0) {
    r.u = sb.ucentre + sb.urelstart + best_order / nv;
    r.v = sb.vcentre + sb.vrelstart + best_order % nv;
  }
  r.ok = (best_order >= 0 && !(best > kCorrThresh2)) ? 1 : 0;
  return r;
}

// ---------------------------------------------------------------------------
// Variant 1 ("column walk"): the search window is staged once in LDS with coalesced
// row loads; the 11x11 template lives in 33 SGPRs; a lane owns one candidate column u
// (several lanes share a column, each a segment of v) and walks down the rows keeping
// the last 11 image rows of its 11-byte strip in registers, so each new candidate
// costs ONE new row fetch, sliding row sums for sum(g1), sum(g1^2) and 33
// v_dot4_u32_u8 for the cross term.  All sums are exact int32.
//
// Candidate ranking is done on rho_f = cov/sqrt(var0 var1) in FP32 from the exact
// integers (error < 1e-6); only candidates within 4e-6 of the best are then scored
// with the reference's FP64 expression and its accept/tie rules, which decides the
// result exactly as the sequential scan would.  Anything the fast path cannot
// decide exactly (two near-best candidates in one lane, the sigma == 10 boundary,
// windows larger than the LDS tile) falls back to search_core_v0 — same results,
// just slower.
// ---------------------------------------------------------------------------
constexpr int kWinPitchDw = 20;   // LDS row pitch in dwords (80 B)
constexpr int kWinRows = 64;
constexpr int kMaxNu = 51, kMaxNv = 54;

__device__ __forceinline__ unsigned udot4(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_udot4(a, b, c, false); }

__device__ __forceinline__ SearchResult search_core_v1(const uint8_t* __restrict__ image, int width, int height,
                                       const uint8_t* __restrict__ patch, const double centre[2], double a, double b,
                                       double c, unsigned* s_win, unsigned long long* s_mask) {
  const int lane = threadIdx.x & 63;
  const SearchBounds sb = search_bounds(centre, a, b, c, width, height);
  const int nu = sb.urelfinish - sb.urelstart + 1;
  const int nv = sb.vrelfinish - sb.vrelstart + 1;
  SearchResult res;
  res.ok = 0; res.found = 0; res.u = 0; res.v = 0; res.ncand = 0; res.score = 1000000.0;
  if (nu <= 0 || nv <= 0) return res;
  if (nu > kMaxNu || nv > kMaxNv) { res.found = -1; return res; }   // caller falls back to search_core_v0

  // ---- template -> 33 wave-uniform dwords (row r: bytes 0..10, byte 11 = 0) ----
  unsigned tv = 0;
  if (lane < 33) {
    const int r = lane / 3, d = lane - 3 * r;
    for (int k = 0; k < 4; ++k) {
      const int col = 4 * d + k;
      const unsigned byte = (col < 11) ? patch[r * 11 + col] : 0u;
      tv |= byte << (8 * k);
    }
  }
  unsigned T[33];
#pragma unroll
  for (int i = 0; i < 33; ++i) T[i] = __builtin_amdgcn_readlane(tv, i);
  unsigned uSg0 = 0, uSg0sq = 0;
#pragma unroll
  for (int i = 0; i < 33; ++i) { uSg0 = udot4(T[i], 0x01010101u, uSg0); uSg0sq = udot4(T[i], T[i], uSg0sq); }
  const int Sg0 = (int)uSg0, Sg0sq = (int)uSg0sq;
  {  // patch sigma test, exactly as correlate2_warning + elliptical_search evaluate it
    const double g0bar = (double)Sg0 / 121.0;
    const double varg0 = (double)Sg0sq / 121.0 - (g0bar * g0bar);
    const double sigmag0 = sqrt(varg0);
    if (sigmag0 < kCorrelationSigmaThreshold) {
      // every candidate is skipped; still report the candidate count
      int n = 0;
      for (int idx = lane; idx < nu * nv; idx += 64)
        n += in_ellipse(a, b, c, sb.urelstart + idx / nv, sb.vrelstart + idx % nv) ? 1 : 0;
      for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
      res.ncand = n;
      return res;
    }
  }
  const int D0 = 121 * Sg0sq - Sg0 * Sg0;   // n^2 var0 > 0 here

  // ---- stage the window: rows y0 .. y0+nv+9, bytes x0 .. x0+nu+9, dword-aligned loads ----
  const int x0 = sb.ucentre + sb.urelstart - 5, y0 = sb.vcentre + sb.vrelstart - 5;
  const int Hw = nv + 10;
  const size_t base_addr = (size_t)image + (size_t)y0 * width + x0;
  {
    const int k = lane & 15, rsub = lane >> 4;   // 16 dwords per row, 4 rows per pass
    for (int r0 = 0; r0 < Hw; r0 += 4) {
      const int r = r0 + rsub;
      if (r < Hw) {
        const size_t addr = base_addr + (size_t)r * width;
        const size_t al = addr & ~(size_t)3;
        const int o = (int)(addr & 3);
        const int need = (o + nu + 10 + 3) >> 2;
        unsigned v = 0;
        if (k < need) v = *(const unsigned*)(al + 4 * (size_t)k);
        s_win[r * kWinPitchDw + k] = v;
        if (k == 0) { s_win[r * kWinPitchDw + 16] = (need > 16) ? *(const unsigned*)(al + 64) : 0u; }
      }
    }
  }
  if (lane < kWinRows) s_mask[lane] = 0ull;
  __syncthreads();
  // ---- ellipse membership bitmasks (exact FP64 test), candidate count ----
  int ncand = 0;
  for (int idx = lane; idx < nu * nv; idx += 64) {
    const int ui = idx / nv, vi = idx - ui * nv;
    if (in_ellipse(a, b, c, sb.urelstart + ui, sb.vrelstart + vi)) {
      atomicOr(&s_mask[ui], 1ull << vi);
      ++ncand;
    }
  }
  for (int off = 32; off > 0; off >>= 1) ncand += __shfl_xor(ncand, off, 64);
  res.ncand = ncand;
  __syncthreads();

  // ---- column walk ----
  const int nseg = 64 / nu;                       // >= 1
  const int vs = (nv + nseg - 1) / nseg;          // rows of candidates per segment
  const int seg = lane / nu, ui = lane - seg * nu;
  const int vstart = seg * vs;
  const bool active = (seg < nseg) && (vstart < nv);
  const int vlen = active ? min(vs, nv - vstart) : 0;
  const unsigned long long mymask = active ? s_mask[ui] : 0ull;
  const int wmod = width & 3;
  const int o_first = (int)(base_addr & 3);
  const int tmax = vs + 10;

  unsigned ring[11][3];
  int rs1[11], rs2[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) { rs1[i] = 0; rs2[i] = 0; ring[i][0] = ring[i][1] = ring[i][2] = 0; }
  int S1 = 0, S2 = 0;
  float best_q = -3.0e38f, second_q = -3.0e38f;
  int best_idx = -1, best_S1 = 0, best_S2 = 0, best_X = 0;
  int need_exact = 0;
  const float d0f = (float)D0;

  for (int tb = 0; tb < tmax; tb += 11) {
#pragma unroll
    for (int s = 0; s < 11; ++s) {
      const int t = tb + s;
      if (t < tmax) {
        const bool row_ok = active && (t < vlen + 10);
        // fetch window row (vstart + t), bytes ui .. ui+11
        unsigned r0 = 0, r1 = 0, r2 = 0;
        if (row_ok) {
          const int wr = vstart + t;
          const int bo = ((o_first + wr * wmod) & 3) + ui;
          const int k0 = bo >> 2, sh = bo & 3;
          const unsigned* rowp = s_win + wr * kWinPitchDw + k0;
          const unsigned d0 = rowp[0], d1 = rowp[1], d2 = rowp[2], d3 = rowp[3];
          r0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
          r1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
          r2 = __builtin_amdgcn_alignbyte(d3, d2, sh) & 0x00ffffffu;
        }
        const int n1 = (int)(udot4(r0, 0x01010101u, 0u) + udot4(r1, 0x01010101u, 0u) + udot4(r2, 0x01010101u, 0u));
        const int n2 = (int)(udot4(r0, r0, 0u) + udot4(r1, r1, 0u) + udot4(r2, r2, 0u));
        S1 += n1 - rs1[s];
        S2 += n2 - rs2[s];
        rs1[s] = n1; rs2[s] = n2;
        ring[s][0] = r0; ring[s][1] = r1; ring[s][2] = r2;
        if (t >= 10) {
          const int vi = vstart + t - 10;
          const bool cand = row_ok && ((mymask >> vi) & 1ull);
          if (cand) {
            unsigned X0 = 0, X1 = 0, X2 = 0;      // three independent accumulation chains
#pragma unroll
            for (int j = 0; j < 11; ++j) {
              const int slot = (s + 1 + j) % 11;
              X0 = udot4(ring[slot][0], T[3 * j + 0], X0);
              X1 = udot4(ring[slot][1], T[3 * j + 1], X1);
              X2 = udot4(ring[slot][2], T[3 * j + 2], X2);
            }
            const unsigned X = X0 + X1 + X2;
            const int D1 = 121 * S2 - S1 * S1;           // n^2 var1, exact
            if (D1 == 1464100) need_exact = 1;            // sigma1 == 10 boundary: decided in FP64 only
            if (D1 > 1464100) {                           // sigma1 >= 10 for certain
              const int Nc = 121 * (int)X - Sg0 * S1;     // n^2 cov, exact
              const float q = (float)Nc * __builtin_amdgcn_rsqf((float)D1 * d0f);   // ~ rho
              const int idx = ui * nv + vi;
              if (q > best_q) {
                second_q = best_q;
                best_q = q; best_idx = idx; best_S1 = S1; best_S2 = S2; best_X = (int)X;
              } else if (q > second_q) {
                second_q = q;
              }
            }
          }
        }
      }
    }
  }
  // ---- decide ----
  float gmax = best_q;
  for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
  const float thr = gmax - 4.0e-6f;
  const bool lane_amb = (second_q >= thr) && (best_idx >= 0);
  if (__any(need_exact) || __any(lane_amb)) { res.found = -1; return res; }
  double best = 1000000.0;
  int best_order = -1;
  if (best_idx >= 0 && best_q >= thr) {
    double sd0, sd1;
    const double corr = ncc_score(Sg0, best_S1, best_X, Sg0sq, best_S2, &sd0, &sd1);
    if (corr <= best && !(sd0 < kCorrelationSigmaThreshold) && !(sd1 < kCorrelationSigmaThreshold)) { best = corr; best_order = best_idx; }
  }
  wave_argmin(best, best_order);
  res.score = best;
  res.found = best_order >= 0;
  if (best_order >= 0) {
    res.u = sb.ucentre + sb.urelstart + best_order / nv;
    res.v = sb.vcentre + sb.vrelstart + best_order % nv;
  }
  res.ok = (best_order >= 0 && !(best > kCorrThresh2)) ? 1 : 0;
  return res;
}

// Engine kernel: grid (nsel_max, B), one wave per block.
template <int VARIANT>
__global__ void __launch_bounds__(64) k_search(const uint8_t* __restrict__ frames, size_t seq_stride, int width, int height,
                                                  const uint8_t* __restrict__ patch, const double* __restrict__ f_h,
                                                  const double* __restrict__ f_S, const int* __restrict__ sel_idx,
                                                  const int* __restrict__ n_sel, int* __restrict__ f_flags,
                                                  double* __restrict__ f_z, double* __restrict__ f_nu,
                                                  int* __restrict__ attempted, int* __restrict__ successful,
                                                  int* __restrict__ meas_ok, double* __restrict__ meas_score,
                                                  double* __restrict__ work, int N) {
  const int b = blockIdx.y, k = blockIdx.x;
  if (k >= n_sel[b]) return;
  const int f = sel_idx[(size_t)b * N + k];
  const size_t fi = (size_t)b * N + f;
  const double h[2] = {f_h[fi * 2], f_h[fi * 2 + 1]};
  const double S[4] = {f_S[fi * 4], f_S[fi * 4 + 1], f_S[fi * 4 + 2], f_S[fi * 4 + 3]};
  double a, bb, c;
  sinv_from_S(S, &a, &bb, &c);
  __shared__ unsigned s_win[kWinRows * kWinPitchDw];
  __shared__ unsigned long long s_mask[kWinRows];
  SearchResult r;
  r.found = -1;
  if (VARIANT == 1) r = search_core_v1(frames + (size_t)b * seq_stride, width, height, patch + fi * kPatchStride, h, a, bb, c, s_win, s_mask);
  const bool fell_back = r.found < 0;
  if (fell_back) r = search_core_v0(frames + (size_t)b * seq_stride, width, height, patch + fi * kPatchStride, h, a, bb, c);
  if ((threadIdx.x & 63) == 0) {
    meas_ok[(size_t)b * N + k] = r.ok;
    meas_score[(size_t)b * N + k] = r.score;
    int fl = f_flags[fi];
    attempted[fi] += 1;  // failed_/successful_measurement_of_feature, monoslam.cpp:479-496
    if (r.ok) {
      successful[fi] += 1;
      f_z[fi * 2] = (double)r.u; f_z[fi * 2 + 1] = (double)r.v;
      f_nu[fi * 2] = (double)r.u - h[0]; f_nu[fi * 2 + 1] = (double)r.v - h[1];
      fl |= FF_SUCCESS;
    } else {
      fl &= ~FF_SUCCESS;
    }
    f_flags[fi] = fl;
    const SearchBounds sb = search_bounds(h, a, bb, c, width, height);
    atomicAdd(&work[b * 4 + 0], (double)(2 * sb.halfwidth + 11) * (double)(2 * sb.halfheight + 11));
    atomicAdd(&work[b * 4 + 1], 1.0);
    atomicAdd(&work[b * 4 + 2], (double)r.ncand);
    if (VARIANT == 1 && fell_back) atomicAdd(&work[b * 4 + 3], 1.0);
  }
}

// Stateless batch kernel (C-ABI seam S1): grid (count), one wave per search.
template <int VARIANT>
__global__ void __launch_bounds__(64) k_search_batch(const uint8_t* __restrict__ images, int width, int height,
                                                        const int* __restrict__ image_index, const uint8_t* __restrict__ patches,
                                                        const double* __restrict__ centre, const double* __restrict__ puinv,
                                                        int* __restrict__ ok, int* __restrict__ uv, double* __restrict__ score) {
  const int i = blockIdx.x;
  const double ce[2] = {centre[i * 2], centre[i * 2 + 1]};
  __shared__ unsigned s_win[kWinRows * kWinPitchDw];
  __shared__ unsigned long long s_mask[kWinRows];
  const uint8_t* img = images + (size_t)image_index[i] * width * height;
  SearchResult r;
  r.found = -1;
  if (VARIANT == 1) r = search_core_v1(img, width, height, patches + (size_t)i * 121, ce, puinv[i * 3], puinv[i * 3 + 1], puinv[i * 3 + 2], s_win, s_mask);
  if (r.found < 0) r = search_core_v0(img, width, height, patches + (size_t)i * 121, ce, puinv[i * 3], puinv[i * 3 + 1], puinv[i * 3 + 2]);
  if ((threadIdx.x & 63) == 0) {
    ok[i] = r.ok;
    score[i] = r.score;
    if (r.found) { uv[i * 2] = r.u; uv[i * 2 + 1] = r.v; }  // untouched if nothing qualified (Q4)
  }
}

int launch_search(sl2_engine* e) {
  LaunchScope ls(e, "k_search");
  SL2_HIP(hipMemsetAsync(e->work, 0, sizeof(double) * 4 * e->B, e->stream));
  dim3 grid(e->nsel_max, e->B);
  if (e->search_variant == 0)
    hipLaunchKernelGGL(k_search<0>, grid, dim3(64), 0, e->stream, e->cur_frames, e->cur_stride, e->cam.width, e->cam.height,
                       e->patch, e->f_h, e->f_S, e->sel_idx, e->n_sel, e->f_flags, e->f_z, e->f_nu, e->attempted,
                       e->successful, e->meas_ok, e->meas_score, e->work, e->N);
  else
    hipLaunchKernelGGL(k_search<1>, grid, dim3(64), 0, e->stream, e->cur_frames, e->cur_stride, e->cam.width, e->cam.height,
                       e->patch, e->f_h, e->f_S, e->sel_idx, e->n_sel, e->f_flags, e->f_z, e->f_nu, e->attempted,
                       e->successful, e->meas_ok, e->meas_score, e->work, e->N);
  SL2_HIP(hipGetLastError());
  return SL2_OK;
}

}  // namespace sl2

extern "C" int sl2_elliptical_search_batch(int device, const uint8_t* images, int nimages, int width, int height,
                                           const int32_t* image_index, const uint8_t* patches, const double* centre,
                                           const double* puinv, int count, int32_t* ok, int32_t* uv, double* score,
                                           int variant) {
  using namespace sl2;
  if (!images || !patches || !centre || !puinv || !ok || !uv || !score || count < 0 || nimages <= 0) return SL2_ERR_INVALID;
  if (variant != 0 && variant != 1) return SL2_ERR_INVALID;
  if (count == 0) return SL2_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device (the engine has no CPU fallback)"); return SL2_ERR_NO_DEVICE; }
  SL2_HIP(hipSetDevice(device));
  uint8_t *d_img = nullptr, *d_pat = nullptr;
  int *d_idx = nullptr, *d_ok = nullptr, *d_uv = nullptr;
  double *d_ce = nullptr, *d_pu = nullptr, *d_sc = nullptr;
  const size_t img_bytes = (size_t)nimages * width * height;
  SL2_HIP(hipMalloc(&d_img, img_bytes));
  SL2_HIP(hipMalloc(&d_pat, (size_t)count * 121));
  SL2_HIP(hipMalloc(&d_idx, sizeof(int) * count));
  SL2_HIP(hipMalloc(&d_ok, sizeof(int) * count));
  SL2_HIP(hipMalloc(&d_uv, sizeof(int) * 2 * count));
  SL2_HIP(hipMalloc(&d_ce, sizeof(double) * 2 * count));
  SL2_HIP(hipMalloc(&d_pu, sizeof(double) * 3 * count));
  SL2_HIP(hipMalloc(&d_sc, sizeof(double) * count));
  SL2_HIP(hipMemcpy(d_img, images, img_bytes, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_pat, patches, (size_t)count * 121, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_idx, image_index, sizeof(int) * count, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_uv, uv, sizeof(int) * 2 * count, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_ce, centre, sizeof(double) * 2 * count, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_pu, puinv, sizeof(double) * 3 * count, hipMemcpyHostToDevice));
  if (variant == 0)
    hipLaunchKernelGGL(k_search_batch<0>, dim3(count), dim3(64), 0, 0, d_img, width, height, d_idx, d_pat, d_ce, d_pu, d_ok, d_uv, d_sc);
  else
    hipLaunchKernelGGL(k_search_batch<1>, dim3(count), dim3(64), 0, 0, d_img, width, height, d_idx, d_pat, d_ce, d_pu, d_ok, d_uv, d_sc);
  SL2_HIP(hipGetLastError());
  SL2_HIP(hipDeviceSynchronize());
  SL2_HIP(hipMemcpy(ok, d_ok, sizeof(int) * count, hipMemcpyDeviceToHost));
  SL2_HIP(hipMemcpy(uv, d_uv, sizeof(int) * 2 * count, hipMemcpyDeviceToHost));
  SL2_HIP(hipMemcpy(score, d_sc, sizeof(double) * count, hipMemcpyDeviceToHost));
  hipFree(d_img); hipFree(d_pat); hipFree(d_idx); hipFree(d_ok); hipFree(d_uv); hipFree(d_ce); hipFree(d_pu); hipFree(d_sc);
  return SL2_OK;
}
