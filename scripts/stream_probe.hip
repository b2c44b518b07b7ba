// stream_probe: how fast can W workgroups pull bytes that another kernel just wrote (the situation of every launch in
// the decoder chain)?  Sweeps workgroups x bytes-per-workgroup x loads-in-flight-per-thread and prints the kernel time
// (hipEvent pair riding on the dispatch).  Build: hipcc -O3 --offload-arch=gfx950 scripts/stream_probe.hip -o scripts/stream_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void writer(uint4* p, long n, unsigned v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = make_uint4(v, v + 1, v + 2, v + 3);
}

template <int DEPTH>
__global__ __launch_bounds__(256) void reader(const uint4* src, unsigned* out, int vec_per_wg, unsigned long long* stamps) {
  const uint4* p = src + (long)blockIdx.x * vec_per_wg;
  unsigned acc = 0;
  const unsigned long long t0 = wall_clock64();
  for (int i = threadIdx.x; i < vec_per_wg; i += 256 * DEPTH) {
    uint4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = (i + d * 256 < vec_per_wg) ? p[i + d * 256] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
  }
  if (acc == 0x12345u) out[blockIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = wall_clock64(); }
}
// The access shape of a direct global->VGPR MFMA B-operand load from a row-major [N,K] bf16 weight: a workgroup owns
// 64 rows of `rowb` bytes; per K-step of 128 B wave w / lane (fi = l%16, fq = l/16) reads 2 x 16 B at
// row (16w+fi), byte step*128 + fq*32: every 16-lane group touches 16 different cache lines.
template <int DEPTH>
__global__ __launch_bounds__(256) void reader_frag(const unsigned char* src, unsigned* out, int rowb, unsigned long long* stamps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fi = lane & 15, fq = lane >> 4;
  const unsigned char* p = src + (long)blockIdx.x * 64 * rowb + (long)(wave * 16 + fi) * rowb + fq * 32;
  unsigned acc = 0;
  const unsigned long long t0 = wall_clock64();
  const int nsteps = rowb / 128;
  for (int s = 0; s < nsteps; s += DEPTH) {
    uint4 v[DEPTH][2];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int ss = (s + d < nsteps) ? s + d : nsteps - 1;
      v[d][0] = *reinterpret_cast<const uint4*>(p + ss * 128);
      v[d][1] = *reinterpret_cast<const uint4*>(p + ss * 128 + 16);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d][0].x ^ v[d][0].w ^ v[d][1].y ^ v[d][1].z;
  }
  if (acc == 0x12345u) out[blockIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = wall_clock64(); }
}
__global__ void empty_kernel() {}

static unsigned long long* g_stamps; static unsigned long long* g_hst;
template <int DEPTH>
float run(const uint4* src, unsigned* out, int wgs, int vec_per_wg, uint4* all, long nall, bool rewrite, hipEvent_t a, hipEvent_t b) {
  std::vector<float> t;
  for (int r = 0; r < 9; ++r) {
    if (rewrite) hipLaunchKernelGGL(writer, dim3(1024), dim3(256), 0, 0, all, std::min(nall, (long)wgs * vec_per_wg), (unsigned)r);
    hipExtLaunchKernelGGL((reader<DEPTH>), dim3(wgs), dim3(256), 0, 0, a, b, 0, src, out, vec_per_wg, g_stamps);
    CK(hipEventSynchronize(b));
    CK(hipMemcpy(g_hst, g_stamps, 16L * wgs, hipMemcpyDeviceToHost));
    unsigned long long lo = ~0ull, hi = 0;
    for (int i = 0; i < wgs; ++i) { lo = std::min(lo, g_hst[2 * i]); hi = std::max(hi, g_hst[2 * i + 1]); }
    t.push_back((float)(hi - lo) * 0.01f);        // 100 MHz constant clock -> us: first workgroup start .. last workgroup end
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

template <int DEPTH>
float run_frag(const uint4* src, unsigned* out, int wgs, int rowb, uint4* all, long nall, bool rewrite) {
  std::vector<float> t;
  for (int r = 0; r < 9; ++r) {
    if (rewrite) hipLaunchKernelGGL(writer, dim3(1024), dim3(256), 0, 0, all, std::min(nall, (long)wgs * 64 * rowb / 16), (unsigned)r);
    hipLaunchKernelGGL((reader_frag<DEPTH>), dim3(wgs), dim3(256), 0, 0, (const unsigned char*)src, out, rowb, g_stamps);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(g_hst, g_stamps, 16L * wgs, hipMemcpyDeviceToHost));
    unsigned long long lo = ~0ull, hi = 0;
    for (int i = 0; i < wgs; ++i) { lo = std::min(lo, g_hst[2 * i]); hi = std::max(hi, g_hst[2 * i + 1]); }
    t.push_back((float)(hi - lo) * 0.01f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  const long bytes = 128L << 20;
  uint4* buf; unsigned* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 1 << 16));
  CK(hipMalloc(&g_stamps, 16 * 4096)); g_hst = (unsigned long long*)malloc(16 * 4096);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  {   // empty-kernel duration
    std::vector<float> t;
    for (int r = 0; r < 20; ++r) {
      hipExtLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, 0, a, b, 0);
      CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    printf("empty kernel (64 wgs): median %.2f us\n", t[10]);
  }
  const int wgs_list[] = {32, 64, 128, 256, 512, 1024};
  const int kb_list[] = {8, 24, 48, 96, 160};
  for (int rewrite = 1; rewrite >= 0; --rewrite) {
    printf("== source %s ==\n", rewrite ? "just rewritten by another kernel (dirty in remote L2 / MALL)" : "untouched since the last read (L2 / MALL warm)");
    printf("%6s %8s | %8s %8s %8s %8s %8s   (us from the first workgroup's start to the last one's end, in-kernel 100 MHz clock; GB/s at depth 8)\n", "wgs", "KB/wg", "d1", "d2", "d4", "d8", "d16");
    for (int wgs : wgs_list)
      for (int kb : kb_list) {
        if ((long)wgs * kb * 1024 > bytes) continue;
        fflush(stdout);
        const int vec = kb * 1024 / 16;
        float t1 = run<1>(buf, out, wgs, vec, buf, bytes / 16, rewrite, a, b);
        float t2 = run<2>(buf, out, wgs, vec, buf, bytes / 16, rewrite, a, b);
        float t4 = run<4>(buf, out, wgs, vec, buf, bytes / 16, rewrite, a, b);
        float t8 = run<8>(buf, out, wgs, vec, buf, bytes / 16, rewrite, a, b);
        float t16 = run<16>(buf, out, wgs, vec, buf, bytes / 16, rewrite, a, b);
        printf("%6d %8d | %8.2f %8.2f %8.2f %8.2f %8.2f   %7.0f GB/s total, %6.1f GB/s per wg\n", wgs, kb, t1, t2, t4, t8, t16,
               (double)wgs * kb * 1024 / (t8 * 1e-6) / 1e9, (double)kb * 1024 / (t8 * 1e-6) / 1e9);
      }
  }
  printf("== MFMA-fragment access shape (16 rows x 64 B per wave instruction pair) vs the same bytes read as 1 KiB per instruction ==\n");
  printf("%6s %8s | %8s %8s %8s | %8s %8s   (in-kernel us; source rewritten before every launch)\n", "wgs", "row B", "frag d1", "frag d4", "frag d8", "lin d4", "lin d8");
  for (int wgs : {34, 136, 256})
    for (int rowb : {1024, 4352, 5504}) {
      const int vec = 64 * rowb / 16;
      if ((long)wgs * 64 * rowb > bytes) { printf("skip %d x %d: exceeds the buffer\n", wgs, rowb); continue; }
      float f1 = run_frag<1>(buf, out, wgs, rowb, buf, bytes / 16, true);
      float f4 = run_frag<4>(buf, out, wgs, rowb, buf, bytes / 16, true);
      float f8 = run_frag<8>(buf, out, wgs, rowb, buf, bytes / 16, true);
      float l4 = run<4>(buf, out, wgs, vec, buf, bytes / 16, true, a, b);
      float l8 = run<8>(buf, out, wgs, vec, buf, bytes / 16, true, a, b);
      printf("%6d %8d | %8.2f %8.2f %8.2f | %8.2f %8.2f   (%d KB per workgroup)\n", wgs, rowb, f1, f4, f8, l4, l8, 64 * rowb / 1024);
    }
  return 0;
}
