// Image operators of the feature-initialisation path, as stateless
// batched device operators behind the C ABI:
//
//   sl2_find_best_patch_batch                      MonoSLAM::find_best_patch_inside_region
//                                                  (Shi-Tomasi detector, monoslam.cpp:1070-1205)
//   sl2_search_multiple_overlapping_ellipses_batch SearchMultipleOverlappingEllipses::search
//                                                  (improc/search_multiple_overlapping_ellipses.cpp:106-196)
//
// Both produce the reference's results bit for bit:
//   * the detector's FP64 sums are exact in the reference (every gradient product is a multiple
//     of 1/4 below 2^24), so they are formed here as int32 box sums of (2 gx)^2, (2 gy)^2,
//     (2 gx)(2 gy) and divided by 4 once; only the square root and the final halving round, and
//     they are evaluated with the reference's expression (no FMA contraction in this file);
//   * the multi-ellipse search's score cache cannot change a result (a position's score does
//     not depend on the ellipse that asks for it): every position of the union is scored once,
//     by the first ellipse that visits it, into a score map; every ellipse then takes its
//     arg-min over its own positions with the reference's "<=" (last wins) rule.
#include <vector>

#include "sl2_common.hpp"
#include "sl2_math.hpp"

namespace sl2 {

// ---------------------------------------------------------------------------
// Shi-Tomasi detector.  One workgroup per job; a thread owns positions idx, idx+256, ... of the
// clamped region in the reference's scan order (v outer, u inner) and keeps the first maximum
// (strict '>'); the workgroup reduction prefers the smaller scan index among equal maxima.
// ---------------------------------------------------------------------------
constexpr int kDetThreads = 256;

__global__ void __launch_bounds__(kDetThreads) k_find_best_patch(const uint8_t* __restrict__ images, int width, int height,
                                                                 const int* __restrict__ image_index,
                                                                 const int* __restrict__ region, int* __restrict__ uv,
                                                                 double* __restrict__ evbest) {
  const int job = blockIdx.x, tid = threadIdx.x;
  const uint8_t* img = images + (size_t)image_index[job] * width * height;
  int ustart = region[4 * job + 0], vstart = region[4 * job + 1], ufinish = region[4 * job + 2], vfinish = region[4 * job + 3];
  const int half = (kBoxSize - 1) / 2;
  if (ustart < half + 1) ustart = half + 1;                    // monoslam.cpp:1080-1091
  if (ufinish > width - half - 1) ufinish = width - half - 1;
  if (vstart < half + 1) vstart = half + 1;
  if (vfinish > height - half - 1) vfinish = height - half - 1;
  if (vstart >= vfinish || ustart >= ufinish) {                // :1094-1099
    if (tid == 0) { uv[2 * job] = ustart; uv[2 * job + 1] = vstart; evbest[job] = 0.0; }
    return;
  }
  const int nu = ufinish - ustart, nv = vfinish - vstart;
  double best = 0.0;   // *evbest = 0 (:1136): only a strictly positive eigenvalue can win
  int best_idx = -1;
  for (int idx = tid; idx < nu * nv; idx += kDetThreads) {
    const int v = vstart + idx / nu, u = ustart + idx % nu;
    int sxx = 0, syy = 0, sxy = 0;
    for (int r = v - half; r <= v + half; ++r) {
      const uint8_t* up = img + (size_t)(r - 1) * width;
      const uint8_t* mid = img + (size_t)r * width;
      const uint8_t* dn = img + (size_t)(r + 1) * width;
#pragma unroll
      for (int c = -5; c <= 5; ++c) {
        const int gx2 = (int)mid[u + c + 1] - (int)mid[u + c - 1];   // 2 gx
        const int gy2 = (int)dn[u + c] - (int)up[u + c];             // 2 gy
        sxx += gx2 * gx2; syy += gy2 * gy2; sxy += gx2 * gy2;
      }
    }
    const double A = sxx / 4.0, Bq = sxy / 4.0, C = syy / 4.0;       // exact
    const double BB = sqrt((A + C) * (A + C) - 4 * (A * C - Bq * Bq));  // find_eigenvalues, :1194-1205
    const double e2 = (A + C - BB) / 2.0;
    if (e2 > best) { best = e2; best_idx = idx; }
  }
  __shared__ double s_best[kDetThreads];
  __shared__ int s_idx[kDetThreads];
  s_best[tid] = best;
  s_idx[tid] = best_idx;
  __syncthreads();
  for (int off = kDetThreads / 2; off > 0; off >>= 1) {
    if (tid < off) {
      const double ob = s_best[tid + off];
      const int oi = s_idx[tid + off];
      const double mb = s_best[tid];
      const int mi = s_idx[tid];
      // larger eigenvalue wins; among equals the earlier scan position (a lane without a candidate has idx -1)
      if (oi >= 0 && (mi < 0 || ob > mb || (ob == mb && oi < mi))) { s_best[tid] = ob; s_idx[tid] = oi; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    evbest[job] = s_idx[0] >= 0 ? s_best[0] : 0.0;
    if (s_idx[0] >= 0) {                 // otherwise *ubest / *vbest keep the caller's values
      uv[2 * job] = ustart + s_idx[0] % nu;
      uv[2 * job + 1] = vstart + s_idx[0] / nu;
    }
  }
}

// ---------------------------------------------------------------------------
// Multi-ellipse search.
// ---------------------------------------------------------------------------
// per ellipse: uc, vc, urelstart, nu, vrelstart, nv, halfwidth, halfheight   (SearchDatum + the clipping of search())
__global__ void __launch_bounds__(64) k_me_describe(const double* __restrict__ puinv, const double* __restrict__ centre, int total,
                                                    int width, int height, int* __restrict__ desc) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= total) return;
  const double a = puinv[3 * e], b = puinv[3 * e + 1], c = puinv[3 * e + 2];
  const int hw = (int)(kNoSigma / sqrt(a - b * b / c));   // cpp:49-50
  const int hh = (int)(kNoSigma / sqrt(c - b * b / a));
  const int uc = int(centre[2 * e]), vc = int(centre[2 * e + 1]);   // truncation, no +0.5 (cpp:127-128)
  const int half = (kBoxSize - 1) / 2;
  int us = -hw, uf = hw, vs = -hh, vf = hh;
  if (uc + us - half < 0) us = half - uc;                               // cpp:131-149
  if (uc + uf - half > width - kBoxSize) uf = width - kBoxSize - uc + half;
  if (vc + vs - half < 0) vs = half - vc;
  if (vc + vf - half > height - kBoxSize) vf = height - kBoxSize - vc + half;
  int* d = desc + 8 * (size_t)e;
  d[0] = uc; d[1] = vc; d[2] = us; d[3] = uf - us + 1; d[4] = vs; d[5] = vf - vs + 1; d[6] = hw; d[7] = hh;
}

__device__ __forceinline__ bool me_visits(const int* __restrict__ d, const double* __restrict__ pu, int x, int y) {
  const int urel = x - d[0], vrel = y - d[1];
  if (urel < d[2] || urel >= d[2] + d[3] || vrel < d[4] || vrel >= d[4] + d[5]) return false;
  return in_ellipse(pu[0], pu[1], pu[2], urel, vrel);
}

// One workgroup per ellipse: score every position this ellipse visits that no EARLIER ellipse of the
// same job visits (the position's owner), with correlate2_warning's exact integer sums + FP64 epilogue,
// plus the low-image-sigma penalty (cpp:169-175).
__global__ void __launch_bounds__(256) k_me_scores(const uint8_t* __restrict__ images, int width, int height,
                                                   const int* __restrict__ image_index, const uint8_t* __restrict__ patches,
                                                   const int* __restrict__ ell_job, const int* __restrict__ job_first,
                                                   const int* __restrict__ desc, const double* __restrict__ puinv,
                                                   double* __restrict__ score_map) {
  const int e = blockIdx.x, tid = threadIdx.x;
  const int job = ell_job[e], first = job_first[job];
  const int* d = desc + 8 * (size_t)e;
  const int nu = d[3], nv = d[5];
  if (nu <= 0 || nv <= 0) return;
  __shared__ int s_patch[121];
  __shared__ int s_sums[2];
  if (tid < 121) s_patch[tid] = patches[(size_t)job * 121 + tid];
  __syncthreads();
  if (tid == 0) {
    int s0 = 0, s0q = 0;
    for (int p = 0; p < 121; ++p) { s0 += s_patch[p]; s0q += s_patch[p] * s_patch[p]; }
    s_sums[0] = s0; s_sums[1] = s0q;
  }
  __syncthreads();
  const int Sg0 = s_sums[0], Sg0sq = s_sums[1];
  const uint8_t* img = images + (size_t)image_index[job] * width * height;
  double* map = score_map + (size_t)job * width * height;
  const double* pu = puinv + 3 * (size_t)e;
  for (int idx = tid; idx < nu * nv; idx += 256) {
    const int urel = d[2] + idx / nv, vrel = d[4] + idx % nv;
    if (!in_ellipse(pu[0], pu[1], pu[2], urel, vrel)) continue;
    const int x = d[0] + urel, y = d[1] + vrel;
    bool owned = true;
    for (int q = first; q < e; ++q)
      if (me_visits(desc + 8 * (size_t)q, puinv + 3 * (size_t)q, x, y)) { owned = false; break; }
    if (!owned) continue;
    const uint8_t* p1 = img + (size_t)(y - 5) * width + (x - 5);
    int Sg1 = 0, Sg0g1 = 0, Sg1sq = 0;
    for (int r = 0; r < 11; ++r)
#pragma unroll
      for (int cc = 0; cc < 11; ++cc) {
        const int g0 = s_patch[r * 11 + cc];
        const int g1 = p1[r * width + cc];
        Sg1 += g1; Sg0g1 += g0 * g1; Sg1sq += g1 * g1;
      }
    double sd0, sd1;
    double corr = ncc_score(Sg0, Sg1, Sg0g1, Sg0sq, Sg1sq, &sd0, &sd1);
    if (sd1 < kCorrelationSigmaThreshold) corr += 5.0;     // LOW_SIGMA_PENALTY, h:56 / cpp:173-175
    map[(size_t)y * width + x] = corr;
  }
}

// One wave per ellipse: arg-min over its positions in the reference's scan order (u outer, v inner),
// "corr <= corrmax" => the last minimum wins.
__global__ void __launch_bounds__(64) k_me_argmin(int width, int height, const int* __restrict__ ell_job,
                                                  const int* __restrict__ desc, const double* __restrict__ puinv,
                                                  const double* __restrict__ score_map, int* __restrict__ result,
                                                  double* __restrict__ corrmax) {
  const int e = blockIdx.x, lane = threadIdx.x;
  const int* d = desc + 8 * (size_t)e;
  const int nu = d[3], nv = d[5];
  const double* map = score_map + (size_t)ell_job[e] * width * height;
  const double* pu = puinv + 3 * (size_t)e;
  double best = 1000000.0;   // cpp:156
  int order = -1;
  if (nu > 0 && nv > 0) {
    for (int idx = lane; idx < nu * nv; idx += 64) {
      const int urel = d[2] + idx / nv, vrel = d[4] + idx % nv;
      if (!in_ellipse(pu[0], pu[1], pu[2], urel, vrel)) continue;
      const double corr = map[(size_t)(d[1] + vrel) * width + (d[0] + urel)];
      if (corr <= best) { best = corr; order = idx; }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double ob = __shfl_xor(best, off, 64);
    const int oo = __shfl_xor(order, off, 64);
    if (oo >= 0 && (order < 0 || ob < best || (ob == best && oo > order))) { best = ob; order = oo; }
  }
  if (lane == 0) {
    int* r = result + 3 * (size_t)e;
    r[0] = (order >= 0 && !(best > kCorrThresh2)) ? 1 : 0;           // cpp:187-191
    r[1] = order >= 0 ? d[0] + d[2] + order / nv : 0;                  // result_u_ / result_v_ start at 0 (cpp:45-46)
    r[2] = order >= 0 ? d[1] + d[4] + order % nv : 0;
    if (corrmax) corrmax[e] = best;
  }
}

static int check_device(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device (the engine has no CPU fallback)"); return SL2_ERR_NO_DEVICE; }
  SL2_HIP(hipSetDevice(device));
  return SL2_OK;
}

struct DevBuf {   // scope-bound device allocation
  void* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess ? SL2_OK : SL2_ERR_HIP; }
  template <typename T> T* as() { return (T*)p; }
};

}  // namespace sl2

extern "C" int sl2_find_best_patch_batch(int device, const uint8_t* images, int nimages, int width, int height, int njobs,
                                         const int32_t* image_index, const int32_t* region, int32_t* uv, double* evbest,
                                         double* kernel_ms) {
  using namespace sl2;
  if (!images || !image_index || !region || !uv || !evbest || njobs < 0 || nimages <= 0 || width < 13 || height < 13)
    return SL2_ERR_INVALID;
  for (int j = 0; j < njobs; ++j)
    if (image_index[j] < 0 || image_index[j] >= nimages) return SL2_ERR_INVALID;
  if (njobs == 0) return SL2_OK;
  int rc = check_device(device);
  if (rc != SL2_OK) return rc;
  DevBuf d_img, d_idx, d_reg, d_uv, d_ev;
  const size_t img_bytes = (size_t)nimages * width * height;
  if (d_img.alloc(img_bytes) || d_idx.alloc(sizeof(int) * njobs) || d_reg.alloc(sizeof(int) * 4 * njobs) ||
      d_uv.alloc(sizeof(int) * 2 * njobs) || d_ev.alloc(sizeof(double) * njobs)) {
    set_error("hipMalloc failed");
    return SL2_ERR_HIP;
  }
  SL2_HIP(hipMemcpy(d_img.p, images, img_bytes, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_idx.p, image_index, sizeof(int) * njobs, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_reg.p, region, sizeof(int) * 4 * njobs, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_uv.p, uv, sizeof(int) * 2 * njobs, hipMemcpyHostToDevice));
  hipEvent_t ev0, ev1;
  SL2_HIP(hipEventCreate(&ev0));
  SL2_HIP(hipEventCreate(&ev1));
  SL2_HIP(hipEventRecord(ev0, 0));
  hipLaunchKernelGGL(k_find_best_patch, dim3(njobs), dim3(kDetThreads), 0, 0, d_img.as<uint8_t>(), width, height, d_idx.as<int>(),
                     d_reg.as<int>(), d_uv.as<int>(), d_ev.as<double>());
  SL2_HIP(hipGetLastError());
  SL2_HIP(hipEventRecord(ev1, 0));
  SL2_HIP(hipDeviceSynchronize());
  if (kernel_ms) { float ms = 0.f; hipEventElapsedTime(&ms, ev0, ev1); *kernel_ms = ms; }
  hipEventDestroy(ev0);
  hipEventDestroy(ev1);
  SL2_HIP(hipMemcpy(uv, d_uv.p, sizeof(int) * 2 * njobs, hipMemcpyDeviceToHost));
  SL2_HIP(hipMemcpy(evbest, d_ev.p, sizeof(double) * njobs, hipMemcpyDeviceToHost));
  return SL2_OK;
}

extern "C" int sl2_search_multiple_overlapping_ellipses_batch(int device, const uint8_t* images, int nimages, int width, int height,
                                                              int njobs, const int32_t* image_index, const uint8_t* patches,
                                                              const int32_t* ellipse_count, const double* puinv,
                                                              const double* centre, int32_t* result, double* corrmax,
                                                              double* kernel_ms) {
  using namespace sl2;
  if (!images || !image_index || !patches || !ellipse_count || !result || njobs < 0 || nimages <= 0 || width < 11 || height < 11)
    return SL2_ERR_INVALID;
  std::vector<int> first(njobs + 1, 0);
  for (int j = 0; j < njobs; ++j) {
    if (ellipse_count[j] < 0 || image_index[j] < 0 || image_index[j] >= nimages) return SL2_ERR_INVALID;
    first[j + 1] = first[j] + ellipse_count[j];
  }
  const int total = njobs ? first[njobs] : 0;
  if (total == 0) return SL2_OK;
  if (!puinv || !centre) return SL2_ERR_INVALID;
  int rc = check_device(device);
  if (rc != SL2_OK) return rc;
  std::vector<int> ell_job(total);
  for (int j = 0; j < njobs; ++j)
    for (int e = first[j]; e < first[j + 1]; ++e) ell_job[e] = j;
  DevBuf d_img, d_idx, d_pat, d_job, d_first, d_pu, d_ce, d_desc, d_map, d_res, d_corr;
  const size_t img_bytes = (size_t)nimages * width * height;
  if (d_img.alloc(img_bytes) || d_idx.alloc(sizeof(int) * njobs) || d_pat.alloc((size_t)njobs * 121) ||
      d_job.alloc(sizeof(int) * total) || d_first.alloc(sizeof(int) * (njobs + 1)) || d_pu.alloc(sizeof(double) * 3 * total) ||
      d_ce.alloc(sizeof(double) * 2 * total) || d_desc.alloc(sizeof(int) * 8 * total) ||
      d_map.alloc(sizeof(double) * (size_t)njobs * width * height) || d_res.alloc(sizeof(int) * 3 * total) ||
      d_corr.alloc(sizeof(double) * total)) {
    set_error("hipMalloc failed");
    return SL2_ERR_HIP;
  }
  SL2_HIP(hipMemcpy(d_img.p, images, img_bytes, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_idx.p, image_index, sizeof(int) * njobs, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_pat.p, patches, (size_t)njobs * 121, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_job.p, ell_job.data(), sizeof(int) * total, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_first.p, first.data(), sizeof(int) * (njobs + 1), hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_pu.p, puinv, sizeof(double) * 3 * total, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_ce.p, centre, sizeof(double) * 2 * total, hipMemcpyHostToDevice));
  hipEvent_t ev0, ev1;
  SL2_HIP(hipEventCreate(&ev0));
  SL2_HIP(hipEventCreate(&ev1));
  SL2_HIP(hipEventRecord(ev0, 0));
  hipLaunchKernelGGL(k_me_describe, dim3((total + 63) / 64), dim3(64), 0, 0, d_pu.as<double>(), d_ce.as<double>(), total, width,
                     height, d_desc.as<int>());
  hipLaunchKernelGGL(k_me_scores, dim3(total), dim3(256), 0, 0, d_img.as<uint8_t>(), width, height, d_idx.as<int>(),
                     d_pat.as<uint8_t>(), d_job.as<int>(), d_first.as<int>(), d_desc.as<int>(), d_pu.as<double>(),
                     d_map.as<double>());
  hipLaunchKernelGGL(k_me_argmin, dim3(total), dim3(64), 0, 0, width, height, d_job.as<int>(), d_desc.as<int>(), d_pu.as<double>(),
                     d_map.as<double>(), d_res.as<int>(), d_corr.as<double>());
  SL2_HIP(hipGetLastError());
  SL2_HIP(hipEventRecord(ev1, 0));
  SL2_HIP(hipDeviceSynchronize());
  if (kernel_ms) { float ms = 0.f; hipEventElapsedTime(&ms, ev0, ev1); *kernel_ms = ms; }
  hipEventDestroy(ev0);
  hipEventDestroy(ev1);
  SL2_HIP(hipMemcpy(result, d_res.p, sizeof(int) * 3 * total, hipMemcpyDeviceToHost));
  if (corrmax) SL2_HIP(hipMemcpy(corrmax, d_corr.p, sizeof(double) * total, hipMemcpyDeviceToHost));
  return SL2_OK;
}
