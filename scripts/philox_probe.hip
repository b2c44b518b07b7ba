// What a keep-mask costs a wave: the Philox call + the 8 thresholds of dropout_scale8, in the forms that were candidates for
// common.h.  One 256-thread workgroup per CU (one wave per SIMD, as a gather passenger has it), each thread draws CALLS masks
// of 8 and folds them into a sink.  Prints ns per call per wave and the sink (all variants of the same rounds must agree).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o scripts/philox_probe scripts/philox_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>

struct P4 { uint32_t x, y, z, w; };

template <int ROUNDS, bool WIDE>
__device__ __forceinline__ P4 philox(uint64_t seed, uint64_t offset, uint32_t idx) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = idx, c1 = 0u, c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    if constexpr (WIDE) {
      const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
      hi0 = (uint32_t)(p0 >> 32); lo0 = (uint32_t)p0; hi1 = (uint32_t)(p1 >> 32); lo1 = (uint32_t)p1;
    } else {
      hi0 = __umulhi(0xD2511F53u, c0); lo0 = 0xD2511F53u * c0;
      hi1 = __umulhi(0xCD9E8D57u, c2); lo1 = 0xCD9E8D57u * c2;
    }
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return P4{c0, c1, c2, c3};
}

template <int ROUNDS, bool WIDE, bool ITHR>
__device__ __forceinline__ void scale8(uint64_t seed, uint64_t offset, uint32_t idx8, float p, float inv, uint32_t thr, float (&m)[8]) {
  const P4 r = philox<ROUNDS, WIDE>(seed, offset, idx8);
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if constexpr (ITHR) {
      m[2 * k] = ((w[k] & 0xFFFFu) >= thr) ? inv : 0.0f;
      m[2 * k + 1] = ((w[k] >> 16) >= thr) ? inv : 0.0f;
    } else {
      const float u = 1.0f / 65536.0f;
      m[2 * k] = ((float)(w[k] & 0xFFFFu) * u >= p) ? inv : 0.0f;
      m[2 * k + 1] = ((float)(w[k] >> 16) * u >= p) ? inv : 0.0f;
    }
  }
}

template <int ROUNDS, bool WIDE, bool ITHR>
__global__ __launch_bounds__(256) void probe(float* sink, uint64_t seed, float p, int calls) {
  const float inv = 1.0f / (1.0f - p);
  const uint32_t thr = (uint32_t)ceilf(p * 65536.0f);
  float acc = 0.f;
  uint32_t idx = (blockIdx.x * 256 + threadIdx.x) * (uint32_t)calls;
  for (int i = 0; i < calls; ++i) {
    float m[8];
    scale8<ROUNDS, WIDE, ITHR>(seed, 77, idx + i, p, inv, thr, m);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += m[j] * (float)(j + 1);
  }
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int ROUNDS, bool WIDE, bool ITHR>
static void run(const char* name, float* sink, int blocks, int calls) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  probe<ROUNDS, WIDE, ITHR><<<blocks, 256>>>(sink, 0x1234567887654321ull, 0.3f, calls);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < 10; ++i) probe<ROUNDS, WIDE, ITHR><<<blocks, 256>>>(sink, 0x1234567887654321ull, 0.3f, calls);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  static float host[256 * 256];
  (void)hipMemcpy(host, sink, sizeof(float) * blocks * 256, hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < blocks * 256; ++i) s += host[i];
  printf("%-28s %8.2f ns per call per wave   sink %.6e\n", name, ms / 10 * 1e6 / calls, s);
}

int main() {
  const int blocks = 256, calls = 4096;
  float* sink;
  (void)hipMalloc(&sink, sizeof(float) * blocks * 256);
  run<10, false, false>("10 rounds, hi+lo, float thr", sink, blocks, calls);
  run<10, true, false>("10 rounds, 64-bit, float thr", sink, blocks, calls);
  run<10, false, true>("10 rounds, hi+lo, int thr", sink, blocks, calls);
  run<10, true, true>("10 rounds, 64-bit, int thr", sink, blocks, calls);
  run<7, false, false>("7 rounds, hi+lo, float thr", sink, blocks, calls);
  run<7, true, true>("7 rounds, 64-bit, int thr", sink, blocks, calls);
  (void)hipFree(sink);
  return 0;
}
