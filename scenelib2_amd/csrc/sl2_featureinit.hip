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

#include "sl2_improc_dev.hpp"

namespace sl2 {

// ---------------------------------------------------------------------------
// Shi-Tomasi detector.  One workgroup per job; a thread owns positions idx, idx+256, ... of the
// clamped region in the reference's scan order (v outer, u inner) and keeps the first maximum
// (strict '>'); the workgroup reduction prefers the smaller scan index among equal maxima.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kDetThreads) k_find_best_patch(const uint8_t* __restrict__ images, int width, int height,
                                                                 const int* __restrict__ image_index,
                                                                 const int* __restrict__ region, int* __restrict__ uv,
                                                                 double* __restrict__ evbest) {
  const int job = blockIdx.x;
  detect_region_wg(images + (size_t)image_index[job] * width * height, width, height, region[4 * job + 0], region[4 * job + 1],
                   region[4 * job + 2], region[4 * job + 3], uv + 2 * job, evbest + job);
}

// ---------------------------------------------------------------------------
// Multi-ellipse search: describe, then one workgroup per job (stamp the union, score it once, per-ellipse arg-min).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_me_describe(const double* __restrict__ puinv, const double* __restrict__ centre, int total,
                                                    int width, int height, int* __restrict__ desc) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= total) return;
  me_describe(puinv[3 * e], puinv[3 * e + 1], puinv[3 * e + 2], centre[2 * e], centre[2 * e + 1], width, height, desc + 8 * (size_t)e);
}

// one workgroup per job: me_search_fused_wg (stamps and scores of the union's bounding box in LDS); a job whose union exceeds
// kMeCap positions goes on the list of k_me_big_* (the image-sized score maps are touched by those only)
struct MeJobsBatch {
  const uint8_t* images; const int* image_index; const uint8_t* patches; const int* first; const int* desc_base; const double* puinv;
  double* map_base; int* result; double* corrmax; int width, height;
  __device__ const uint8_t* img(int j) const { return images + (size_t)image_index[j] * width * height; }
  __device__ const uint8_t* patch(int j) const { return patches + (size_t)j * 121; }
  __device__ const int* desc(int j) const { return desc_base + 8 * (size_t)first[j]; }
  __device__ int n_ell(int j) const { return first[j + 1] - first[j]; }
  __device__ const double* pu(int j, int e) const { return puinv + 3 * ((size_t)first[j] + e); }
  __device__ double* map(int j) const { return map_base + (size_t)j * width * height; }
  __device__ void emit(int j, int e, int flag, int u, int v, double best) const {
    const size_t k = (size_t)first[j] + e;
    result[3 * k] = flag; result[3 * k + 1] = u; result[3 * k + 2] = v;
    if (corrmax) corrmax[k] = best;
  }
};
__global__ void __launch_bounds__(1024) k_me_search(MeJobsBatch J, int* __restrict__ big_list, int* __restrict__ big_count) {
  const int job = blockIdx.x;
  const bool done = me_search_fused_wg(J.img(job), J.width, J.patch(job), J.desc(job), J.n_ell(job), [&](int e) { return J.pu(job, e); },
                                       [&](int e, int flag, int u, int v, double best) { J.emit(job, e, flag, u, v, best); });
  if (!done && threadIdx.x == 0) big_list[atomicAdd(big_count, 1)] = job;
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
  DevBuf d_img, d_idx, d_pat, d_first, d_pu, d_ce, d_desc, d_map, d_res, d_corr, d_big;
  const size_t img_bytes = (size_t)nimages * width * height;
  if (d_img.alloc(img_bytes) || d_idx.alloc(sizeof(int) * njobs) || d_pat.alloc((size_t)njobs * 121) ||
      d_big.alloc(sizeof(int) * (njobs + 1)) || d_first.alloc(sizeof(int) * (njobs + 1)) || d_pu.alloc(sizeof(double) * 3 * total) ||
      d_ce.alloc(sizeof(double) * 2 * total) || d_desc.alloc(sizeof(int) * 8 * total) ||
      d_map.alloc(sizeof(double) * (size_t)njobs * width * height) ||
      d_res.alloc(sizeof(int) * 3 * total) ||
      d_corr.alloc(sizeof(double) * total)) {
    set_error("hipMalloc failed");
    return SL2_ERR_HIP;
  }
  SL2_HIP(hipMemcpy(d_img.p, images, img_bytes, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_idx.p, image_index, sizeof(int) * njobs, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_pat.p, patches, (size_t)njobs * 121, hipMemcpyHostToDevice));
  SL2_HIP(hipMemset(d_big.p, 0, sizeof(int) * (njobs + 1)));
  SL2_HIP(hipMemcpy(d_first.p, first.data(), sizeof(int) * (njobs + 1), hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_pu.p, puinv, sizeof(double) * 3 * total, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_ce.p, centre, sizeof(double) * 2 * total, hipMemcpyHostToDevice));
  hipEvent_t ev0, ev1;
  SL2_HIP(hipEventCreate(&ev0));
  SL2_HIP(hipEventCreate(&ev1));
  SL2_HIP(hipEventRecord(ev0, 0));
  hipLaunchKernelGGL(k_me_describe, dim3((total + 63) / 64), dim3(64), 0, 0, d_pu.as<double>(), d_ce.as<double>(), total, width,
                     height, d_desc.as<int>());
  MeJobsBatch J;
  J.images = d_img.as<uint8_t>(); J.image_index = d_idx.as<int>(); J.patches = d_pat.as<uint8_t>(); J.first = d_first.as<int>();
  J.desc_base = d_desc.as<int>(); J.puinv = d_pu.as<double>(); J.map_base = d_map.as<double>();
  J.result = d_res.as<int>(); J.corrmax = corrmax ? d_corr.as<double>() : nullptr; J.width = width; J.height = height;
  hipLaunchKernelGGL(k_me_search, dim3(njobs), dim3(1024), 0, 0, J, d_big.as<int>(), d_big.as<int>() + njobs);
  me_big_launch(J, d_big.as<int>(), d_big.as<int>() + njobs, width, 0);
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
