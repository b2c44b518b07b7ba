// gemm_probe: the product's skinny-M GEMM kernels timed one launch at a time on the decoder's shapes, with the
// activations rewritten by another kernel before every launch (as in the real chain) -- and with parts of the kernel
// compiled out (-DVLN_PROBE_NO_MFMA) to see what a K-step is made of.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude [-DVLN_PROBE_NO_MFMA] scripts/gemm_probe.hip -o scripts/gemm_probe
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../curriculum-learning-for-vln_amd/csrc/vln_internal.h"
namespace vln {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
const char* get_error() { return ""; }
int check_hip(hipError_t e, const char* what) { if (e == hipSuccess) return 0; fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return 2; }
int g_tunable[8] = {384, 1, 1, 512, 0, 0, 0, 0};
// the probe launches every kernel on its own: no chained-step recorder (chain.h)
int chain_flush() { return 0; }
bool chain_add(hipStream_t, int, int, int, int, const void*, int, double, int) { return false; }
unsigned g_prof_mask = 1u;            // time gemm_nt
static hipEvent_t g_a, g_b;
bool prof_slot(int, double, hipEvent_t* a, hipEvent_t* b) { *a = g_a; *b = g_b; return true; }
void prof_begin(hipStream_t, int, double) {}
void prof_end(hipStream_t, int) {}
}
#include "../curriculum-learning-for-vln_amd/csrc/gemm.hip"
using namespace vln;

__global__ void fill_f32_k(float* p, long n, float v) { for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v + (float)(i & 15) * 0.01f; }
__global__ void fill_u16_k(unsigned short* p, long n, unsigned short v) { for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v; }

int main() {
  hipEventCreate(&g_a); hipEventCreate(&g_b);
  struct Shape { int M, N, K; const char* what; };
  const Shape shapes[] = {{64, 2176, 512, "H->F projection (visual / candidate query)"}, {64, 2048, 2752, "LSTM gates"},
                          {64, 2752, 2048, "d xcat"}, {64, 512, 2176, "F->H (dX of the projections)"}, {64, 512, 512, "H->H"},
                          {64, 512, 1024, "text linear_out"}, {64, 1024, 512, "d tcat"}};
  float *X, *Y, *ws; unsigned short* W;
  const long wsf = 48L * 64 * 2752;
  hipMalloc(&X, 64L * 2752 * 4); hipMalloc(&Y, 64L * 2752 * 4); hipMalloc(&ws, wsf * 4); hipMalloc(&W, 2752L * 2176 * 2);
  hipLaunchKernelGGL(fill_u16_k, dim3(1024), dim3(256), 0, 0, W, 2752L * 2176, (unsigned short)0x3c00);
  const int targets[] = {256, 512, 1024, 2048};
  for (int variant = 0; variant < 4; ++variant) {
    g_tunable[5] = 1;                 // depth-2 prefetch
    g_tunable[0] = targets[variant];  // workgroups wanted in flight (split-K)
    g_tunable[1] = 0;                 // shallow products may split too
    printf("== split-K target %d workgroups ==\n", targets[variant]);
    for (const Shape& s : shapes) {
      for (int n16 = 0; n16 < 1; ++n16) {
        g_tunable[2] = n16;
        int nsplit = 0;
        std::vector<float> t;
        for (int r = 0; r < 15; ++r) {
          hipLaunchKernelGGL(fill_f32_k, dim3(256), dim3(256), 0, 0, X, 64L * s.K, 0.5f + r);
          int rc = gemm_nt(0, X, s.K, W, W_BF16, s.K, Y, s.N, s.M, s.N, s.K, nullptr, ACT_NONE, ws, wsf, &nsplit);
          if (rc) return 1;
          hipDeviceSynchronize();
          float ms; hipEventElapsedTime(&ms, g_a, g_b); t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        printf("  M=%d N=%4d K=%4d  nsplit=%2d wgs=%4d  median %6.2f us  min %6.2f   %s\n", s.M, s.N, s.K, nsplit, nsplit * ((s.N + 63) / 64), t[7], t[0], s.what);
      }
    }
  }
  return 0;
}
