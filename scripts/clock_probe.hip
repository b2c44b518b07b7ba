// Probe: effective shader clock seen by short kernels vs a long kernel (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(unsigned long long* out, int iters, float seed) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float x = seed + threadIdx.x;
  for (int i = 0; i < iters; ++i) x = x * 1.0000001f + 0.5f;
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
  if (x == 12345.f) out[2] = 1;
}
__global__ void tiny(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
int main() {
  unsigned long long* d; hipMalloc(&d, 64); float* f; hipMalloc(&f, 4096 * 4); hipMemset(f, 0, 4096 * 4);
  unsigned long long h[3];
  auto report = [&](const char* tag) { hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("%-28s ticks %llu real %llu -> %.0f MHz\n", tag, h[0], h[1], 100.0 * h[0] / (double)h[1]); };
  probe<<<1, 64>>>(d, 20000, 1.f); hipDeviceSynchronize(); report("cold single");
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 3000; ++i) tiny<<<128, 256>>>(f);
    probe<<<256, 256>>>(d, 20000, 1.f); hipDeviceSynchronize(); report("after 3000 tiny launches");
  }
  probe<<<1024, 256>>>(d, 4000000, 1.f); hipDeviceSynchronize(); report("long busy kernel");
  probe<<<256, 256>>>(d, 20000, 1.f); hipDeviceSynchronize(); report("right after long kernel");
  // launch-to-launch cadence of dependent tiny kernels
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a); for (int i = 0; i < 2000; ++i) tiny<<<128, 256>>>(f); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); printf("2000 tiny kernels: %.2f us each (GPU timeline)\n", ms * 1e3 / 2000);
  return 0;
}
