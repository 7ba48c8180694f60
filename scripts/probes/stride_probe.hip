// Development probe (round 6): does the distance between the sequences' blocks matter to a launch in which every workgroup streams its own
// block?  1024 workgroups, 102400 doubles each (a 320 x 320 covariance), read + write; the blocks `stride` doubles apart.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void __launch_bounds__(1024) probe(double* buf, size_t stride, size_t len) {
  double* p = buf + (size_t)blockIdx.x * stride;
  for (size_t i = threadIdx.x; i < len; i += blockDim.x) p[i] = p[i] + 0.0;
}
static float time_probe(double* p, int B, size_t stride, size_t len) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    (void)hipEventRecord(a, 0); hipLaunchKernelGGL(probe, dim3(B), dim3(1024), 0, 0, p, stride, len); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); if (r > 0 && ms < best) best = ms;
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  return best;
}
int main() {
  const int B = 1024; const size_t len = 320 * 320;
  const size_t pads[] = {0, 12288, 28672};   // doubles
  std::vector<void*> held;
  for (int k = 0; k < 28; ++k) {
    void* p = nullptr;
    if (hipMalloc(&p, sizeof(double) * B * (len + 32768)) != hipSuccess) break;
    (void)hipMemset(p, 0, sizeof(double) * B * (len + 32768));
    held.push_back(p);
    printf("alloc %2d:", k);
    for (size_t pad : pads) printf(" %5zu:%.3f", pad, time_probe((double*)p, B, len + pad, len));
    printf("\n");
  }
  for (void* p : held) (void)hipFree(p);
  return 0;
}
