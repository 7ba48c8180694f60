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

__device__ SearchResult search_core_v0(const uint8_t* __restrict__ image, int width, int height,
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

// Engine kernel: grid (nsel_max, B), one wave per block.
__global__ void __launch_bounds__(64) k_search_v0(const uint8_t* __restrict__ frames, size_t seq_stride, int width, int height,
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
  const SearchResult r = search_core_v0(frames + (size_t)b * seq_stride, width, height, patch + fi * kPatchStride, h, a, bb, c);
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
  }
}

// Stateless batch kernel (C-ABI seam S1): grid (count), one wave per search.
__global__ void __launch_bounds__(64) k_search_batch_v0(const uint8_t* __restrict__ images, int width, int height,
                                                        const int* __restrict__ image_index, const uint8_t* __restrict__ patches,
                                                        const double* __restrict__ centre, const double* __restrict__ puinv,
                                                        int* __restrict__ ok, int* __restrict__ uv, double* __restrict__ score) {
  const int i = blockIdx.x;
  const double ce[2] = {centre[i * 2], centre[i * 2 + 1]};
  const SearchResult r = search_core_v0(images + (size_t)image_index[i] * width * height, width, height,
                                        patches + (size_t)i * 121, ce, puinv[i * 3], puinv[i * 3 + 1], puinv[i * 3 + 2]);
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
  hipLaunchKernelGGL(k_search_v0, grid, dim3(64), 0, e->stream, e->cur_frames, e->cur_stride, e->cam.width, e->cam.height,
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
  if (variant != 0) return SL2_ERR_INVALID;
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
  hipLaunchKernelGGL(k_search_batch_v0, dim3(count), dim3(64), 0, 0, d_img, width, height, d_idx, d_pat, d_ce, d_pu, d_ok,
                     d_uv, d_sc);
  SL2_HIP(hipGetLastError());
  SL2_HIP(hipDeviceSynchronize());
  SL2_HIP(hipMemcpy(ok, d_ok, sizeof(int) * count, hipMemcpyDeviceToHost));
  SL2_HIP(hipMemcpy(uv, d_uv, sizeof(int) * 2 * count, hipMemcpyDeviceToHost));
  SL2_HIP(hipMemcpy(score, d_sc, sizeof(double) * count, hipMemcpyDeviceToHost));
  hipFree(d_img); hipFree(d_pat); hipFree(d_idx); hipFree(d_ok); hipFree(d_uv); hipFree(d_ce); hipFree(d_pu); hipFree(d_sc);
  return SL2_OK;
}
