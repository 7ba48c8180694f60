// Development probe (round 6): the streaming probe of place_probe.hip on memory obtained through the virtual-memory-management calls
// (hipMemCreate / hipMemAddressReserve / hipMemMap) instead of hipMalloc - does a physically contiguous handle land "fast" every time?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(1024) probe(double* buf, size_t per_seq) {
  double* p = buf + (size_t)blockIdx.x * per_seq;
  for (size_t i = threadIdx.x; i < per_seq; i += blockDim.x) p[i] = p[i] + 0.0;
}
static float time_probe(double* p, int B, size_t per_seq) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    (void)hipEventRecord(a, 0); hipLaunchKernelGGL(probe, dim3(B), dim3(1024), 0, 0, p, per_seq); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); if (r > 0 && ms < best) best = ms;
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  return best;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const int B = 1024; const size_t per_seq = 320 * 320; const size_t need = sizeof(double) * B * per_seq;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gmin = 0, grec = 0;
  CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity min %zu recommended %zu\n", gmin, grec);
  const size_t size = (need + grec - 1) / grec * grec;
  struct H { hipMemGenericAllocationHandle_t h; void* va; };
  std::vector<H> held;
  printf("one handle per buffer:");
  for (int k = 0; k < 12; ++k) {
    H x;
    CK(hipMemCreate(&x.h, size, &prop, 0));
    CK(hipMemAddressReserve(&x.va, size, 0, nullptr, 0));
    CK(hipMemMap(x.va, size, 0, x.h, 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(x.va, size, &acc, 1));
    CK(hipMemset(x.va, 0, size));
    held.push_back(x);
    printf(" %.3f", time_probe((double*)x.va, B, per_seq));
  }
  printf("\n");
  for (H& x : held) { (void)hipMemUnmap(x.va, size); (void)hipMemAddressFree(x.va, size); (void)hipMemRelease(x.h); }
  // hipMalloc for comparison in the same process
  std::vector<void*> m;
  printf("hipMalloc:            ");
  for (int k = 0; k < 12; ++k) { void* p = nullptr; if (hipMalloc(&p, need) != hipSuccess) break; (void)hipMemset(p, 0, need); m.push_back(p); printf(" %.3f", time_probe((double*)p, B, per_seq)); }
  printf("\n");
  for (void* p : m) (void)hipFree(p);
  return 0;
}
