// Development probe (round 6): how the speed of a streaming kernel over an allocation depends on how the allocation was made.
// Each of B workgroups streams its own per_seq doubles, read + write, like sl2_engine.hip: k_place_probe.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/place_probe.hip -o gpurun_out/place_probe && gpurun_out/place_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void __launch_bounds__(1024) probe(double* buf, size_t per_seq) {
  double* p = buf + (size_t)blockIdx.x * per_seq;
  for (size_t i = threadIdx.x; i < per_seq; i += blockDim.x) p[i] = p[i] + 0.0;
}
static float time_probe(double* p, int B, size_t per_seq) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(a, 0); hipLaunchKernelGGL(probe, dim3(B), dim3(1024), 0, 0, p, per_seq); hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (r > 0 && ms < best) best = ms;
  }
  hipEventDestroy(a); hipEventDestroy(b);
  return best;
}
int main() {
  const int B = 1024; const size_t per_seq = 320 * 320; const size_t need = sizeof(double) * B * per_seq;   // 800 MiB
  struct Case { const char* name; size_t alloc; size_t offset; } cases[] = {
    {"800 MiB exact", need, 0}, {"1 GiB", (size_t)1 << 30, 0}, {"2 GiB", (size_t)2 << 30, 0}, {"2 GiB, use from +1 GiB", (size_t)2 << 30, (size_t)1 << 30},
    {"800 MiB + 64 KiB", need + 65536, 0}, {"4 GiB", (size_t)4 << 30, 0}};
  for (const Case& c : cases) {
    std::vector<void*> held; std::vector<float> ms;
    for (int k = 0; k < 12; ++k) {
      void* p = nullptr;
      if (hipMalloc(&p, c.alloc) != hipSuccess) break;
      hipMemset(p, 0, c.alloc);
      held.push_back(p);
      ms.push_back(time_probe((double*)((char*)p + c.offset), B, per_seq));
    }
    printf("%-26s", c.name);
    for (float v : ms) printf(" %.3f", v);
    std::sort(ms.begin(), ms.end());
    printf("   | min %.3f median %.3f max %.3f\n", ms.front(), ms[ms.size() / 2], ms.back());
    for (void* p : held) hipFree(p);
  }
  // one arena, twelve slices of it
  {
    void* arena = nullptr;
    if (hipMalloc(&arena, need * 12) == hipSuccess) {
      hipMemset(arena, 0, need * 12);
      printf("%-26s", "slices of one 9.4 GiB");
      for (int k = 0; k < 12; ++k) printf(" %.3f", time_probe((double*)((char*)arena + need * k), B, per_seq));
      printf("\n");
      hipFree(arena);
    }
  }
  return 0;
}
