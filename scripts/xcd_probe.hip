// xcd_probe (round 4): the two numbers a decoder step as ONE persistent launch confined to ONE XCD would live on:
//   (1) the streaming bandwidth ONE XCD sustains on a weight-sized buffer (17.7 MB, re-read every repetition like the LSTM gate
//       product's weights are every decoder step), by workgroups per CU;
//   (2) the latency of a barrier among the workgroups of one XCD (one agent-scope counter, release / acquire), against a barrier
//       over workgroups on all eight XCDs.
// Workgroups are dealt to the XCDs round-robin by id: a launch of 8 * n workgroups in which only ids with id % 8 == 0 work puts n
// workers on XCD 0.     bash scripts/build_xcd_probe.sh && scripts/xcd_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// every worker streams its contiguous share of `n16` 16-byte words, `depth` independent loads in flight per thread
template <int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ src, long n16, int xcd_only, int workers, unsigned* sink) {
  int w = blockIdx.x;
  if (xcd_only) { if (w & 7) return; w >>= 3; }
  const long per = (n16 + workers - 1) / workers;
  const long beg = (long)w * per, end = min(n16, beg + per);
  unsigned acc = 0;
  long i = beg + threadIdx.x;
  for (; i + (DEPTH - 1) * 256 < end; i += DEPTH * 256) {
    uint4 v[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) v[k] = src[i + k * 256];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  }
  for (; i < end; i += 256) { const uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345u) sink[0] = acc;
}

// NG XCDs each stream the WHOLE buffer (an episode-sharded decoder: every group reads all of the weights), 32 * per_cu workers per XCD
template <int DEPTH>
__global__ __launch_bounds__(256) void stream_replicated_kernel(const uint4* __restrict__ src, long n16, int ng, int workers, unsigned* sink) {
  const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
  if (xcd >= ng) return;
  const long per = (n16 + workers - 1) / workers;
  const long beg = (long)w * per, end = min(n16, beg + per);
  unsigned acc = 0;
  long i = beg + threadIdx.x;
  for (; i + (DEPTH - 1) * 256 < end; i += DEPTH * 256) {
    uint4 v[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) v[k] = src[i + k * 256];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  }
  for (; i < end; i += 256) { const uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345u) sink[0] = acc;
}

// the same stream with `sc1` (agent-scope, past the L1) 16-byte buffer loads: how a consumer would read producer-written data inside
// one launch without an invalidate
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <int DEPTH>
__global__ __launch_bounds__(256) void stream_sc1_kernel(const uint4* __restrict__ src, long n16, int xcd_only, int workers, unsigned* sink) {
  int w = blockIdx.x;
  if (xcd_only) { if (w & 7) return; w >>= 3; }
  const long per = (n16 + workers - 1) / workers;
  const long beg = (long)w * per, end = min(n16, beg + per);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(src), 0, (unsigned)(n16 * 16), 0x00020000);
  unsigned acc = 0;
  long i = beg + threadIdx.x;
  for (; i + (DEPTH - 1) * 256 < end; i += DEPTH * 256) {
    u32x4_t v[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((i + k * 256) * 16), 0, 16);
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  }
  if (acc == 0x12345u) sink[0] = acc;
}

// `rounds` barriers among the `workers` participating workgroups; round r waits for the counter to reach (r + 1) * workers
__global__ __launch_bounds__(256) void barrier_kernel(unsigned* counter, int rounds, int xcd_only, int workers, float* scratch, int mode) {
  int w = blockIdx.x;
  if (xcd_only) { if (w & 7) return; w >>= 3; }
  __shared__ int dummy;
  for (int r = 0; r < rounds; ++r) {
    scratch[(long)w * 256 + threadIdx.x] = (float)r;                       // a store the barrier has to publish
    if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // mode 1: no L2 write-back (one XCD = one L2)
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(r + 1) * (unsigned)workers;
      unsigned spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {      // relaxed polls: no invalidate per poll
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 24)) break;                                     // never hang the box
      }
      dummy = 1;
    }
    __syncthreads();
    if (mode == 0 || mode == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // mode 2: acquire only (L1 + L2 invalidate), no write-back
  }
  if (dummy == 12345) scratch[0] = 1.f;
}

static float timed(std::function<void()> f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  std::vector<float> ms;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

int main() {
  const long bytes = 2048L * 2752 * 2 + 64L * 2752 * 4;      // the LSTM gate product's weights (bf16) + its activations
  const long n16 = bytes / 16;
  uint4* buf; CK(hipMalloc(&buf, n16 * 16)); CK(hipMemset(buf, 1, n16 * 16));
  unsigned* sink; CK(hipMalloc(&sink, 64)); CK(hipMemset(sink, 0, 64));
  unsigned* counter; CK(hipMalloc(&counter, 256)); 
  float* scratch; CK(hipMalloc(&scratch, 2048L * 256 * 4));
  printf("streaming %.1f MB (median of 20 launches, launch floor included):\n", bytes / 1e6);
  for (int xcd_only = 1; xcd_only >= 0; --xcd_only)
    for (int per_cu : {1, 2, 4, 8}) {
      const int workers = (xcd_only ? 32 : 256) * per_cu;
      const int grid = xcd_only ? workers * 8 : workers;
      float t4 = timed([&] { hipLaunchKernelGGL(stream_kernel<4>, dim3(grid), dim3(256), 0, 0, buf, n16, xcd_only, workers, sink); }, 20);
      float t8 = timed([&] { hipLaunchKernelGGL(stream_kernel<8>, dim3(grid), dim3(256), 0, 0, buf, n16, xcd_only, workers, sink); }, 20);
      printf("  %s  %4d workgroups (%d per CU): depth 4 %7.2f us = %5.2f TB/s | depth 8 %7.2f us = %5.2f TB/s\n", xcd_only ? "ONE XCD " : "all XCDs", workers,
             per_cu, t4 * 1e3, bytes / (t4 * 1e-3) / 1e12, t8 * 1e3, bytes / (t8 * 1e-3) / 1e12);
    }
  printf("NG XCDs EACH stream the whole %.1f MB (128 workgroups per XCD, depth 8):\n", bytes / 1e6);
  for (int ng : {1, 2, 4, 8}) {
    const int workers = 128;
    float t = timed([&] { hipLaunchKernelGGL(stream_replicated_kernel<8>, dim3(workers * 8), dim3(256), 0, 0, buf, n16, ng, workers, sink); }, 20);
    printf("  %d XCD(s): %7.2f us = %5.2f TB/s aggregate, %5.2f TB/s per XCD\n", ng, t * 1e3, ng * bytes / (t * 1e-3) / 1e12, bytes / (t * 1e-3) / 1e12);
  }
  printf("the same stream with sc1 (L1-bypassing) 16-byte loads:\n");
  for (int xcd_only = 1; xcd_only >= 0; --xcd_only)
    for (int per_cu : {1, 4}) {
      const int workers = (xcd_only ? 32 : 256) * per_cu;
      const int grid = xcd_only ? workers * 8 : workers;
      float t4 = timed([&] { hipLaunchKernelGGL(stream_sc1_kernel<4>, dim3(grid), dim3(256), 0, 0, buf, n16, xcd_only, workers, sink); }, 20);
      float t8 = timed([&] { hipLaunchKernelGGL(stream_sc1_kernel<8>, dim3(grid), dim3(256), 0, 0, buf, n16, xcd_only, workers, sink); }, 20);
      printf("  %s  %4d workgroups (%d per CU): depth 4 %7.2f us = %5.2f TB/s | depth 8 %7.2f us = %5.2f TB/s\n", xcd_only ? "ONE XCD " : "all XCDs", workers,
             per_cu, t4 * 1e3, bytes / (t4 * 1e-3) / 1e12, t8 * 1e3, bytes / (t8 * 1e-3) / 1e12);
    }
  printf("barrier (100 rounds per launch, launch floor subtracted with a 0-round launch):\n");
  for (int xcd_only = 1; xcd_only >= 0; --xcd_only)
    for (int workers : {32, 64, 128}) {
      if (!xcd_only && workers == 64) continue;
      const int grid = xcd_only ? workers * 8 : workers;
      for (int mode = 0; mode < 3; ++mode) {
        auto run = [&](int rounds) {
          return timed([&] { CK(hipMemsetAsync(counter, 0, 4)); hipLaunchKernelGGL(barrier_kernel, dim3(grid), dim3(256), 0, 0, counter, rounds, xcd_only, workers, scratch, mode); }, 20);
        };
        const float t0 = run(0), t1 = run(100);
        printf("  %s  %4d workgroups, %s: %6.2f us per barrier\n", xcd_only ? "ONE XCD " : "all XCDs", workers,
               mode == 0 ? "agent-scope release / acquire fences" : mode == 1 ? "no fences (drained stores only)      " : "drained stores + acquire fence only ", (t1 - t0) * 1e3 / 100);
      }
    }
  return 0;
}
