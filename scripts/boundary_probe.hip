// boundary_probe: what one DEPENDENT trivial kernel costs on this chip, by launch path.
// A chain of 12 dependent launches of a trivial stage (out[i] = f(in[i]) over [64, 512] floats, each stage reads what
// the one before it wrote), timed over many repetitions:  us per launch = wall / (reps * 12).
//   api   : hipLaunchKernelGGL | hipExtLaunchKernelGGL (no events) | one hipGraph of the 12 launches
//   stream: created non-blocking | created blocking | the null (legacy default) stream
//   args  : 3 scalars/pointers | the same plus a 320-byte by-value struct (the library's argument blocks are 100-400 bytes)
//   grid  : 128 or 256 workgroups of 256 threads
//   slabs : the stage reads S partial slabs [S][64,512] and writes S slabs (the split-K hand-over shape)
// Run under HIP_FORCE_DEV_KERNARG=0 / 1 (kernel arguments in host-coherent vs device memory).
// build: hipcc -O3 --offload-arch=gfx950 scripts/boundary_probe.hip -o scripts/boundary_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Big { float pad[80]; };

__global__ __launch_bounds__(256) void stage_small(const float* __restrict__ in, float* __restrict__ out, int n, int S) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int stride = gridDim.x * 256;
  for (int j = i; j < n; j += stride) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += in[(long)s * n + j];
    for (int s = 0; s < S; ++s) out[(long)s * n + j] = a * 0.5f + (float)s;
  }
}
__global__ __launch_bounds__(256) void stage_big(const float* __restrict__ in, float* __restrict__ out, int n, int S, Big b) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int stride = gridDim.x * 256;
  for (int j = i; j < n; j += stride) {
    float a = b.pad[S & 63];
    for (int s = 0; s < S; ++s) a += in[(long)s * n + j];
    for (int s = 0; s < S; ++s) out[(long)s * n + j] = a * 0.5f + (float)s;
  }
}
// float4 form: one 16-byte load per slab per thread, all loads of a thread issued before any is used
__global__ __launch_bounds__(256) void stage_vec(const float4* __restrict__ in, float4* __restrict__ out, int n4, int S) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n4) return;
  float4 v[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) if (s < S) v[s] = in[(long)s * n4 + j];
  float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int s = 0; s < 8; ++s) if (s < S) { a.x += v[s].x; a.y += v[s].y; a.z += v[s].z; a.w += v[s].w; }
#pragma unroll
  for (int s = 0; s < 8; ++s) if (s < S) out[(long)s * n4 + j] = make_float4(a.x * .5f, a.y * .5f, a.z * .5f, a.w * .5f + s);
}

// cross-XCD form: thread j reads element (j + shift) mod n4 of every slab -- with shift = 256 (one block) the bytes were
// written by the NEXT block of the previous launch, which round-robin placement puts on another XCD: the load cannot hit
// this XCD's L2 and comes from the Infinity Cache / HBM after the producer's write-back
__global__ __launch_bounds__(256) void stage_vec_x(const float4* __restrict__ in, float4* __restrict__ out, int n4, int S, int shift) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n4) return;
  int jr = j + shift; if (jr >= n4) jr -= n4;
  float4 v[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) if (s < S) v[s] = in[(long)s * n4 + jr];
  float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int s = 0; s < 8; ++s) if (s < S) { a.x += v[s].x; a.y += v[s].y; a.z += v[s].z; a.w += v[s].w; }
#pragma unroll
  for (int s = 0; s < 8; ++s) if (s < S) out[(long)s * n4 + j] = make_float4(a.x * .5f, a.y * .5f, a.z * .5f, a.w * .5f + s);
}
// two DEPENDENT round trips: an index word read from the previous launch's output selects the row that is read next
__global__ __launch_bounds__(256) void stage_vec_xx(const float4* __restrict__ in, float4* __restrict__ out, int n4, int S, int shift) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n4) return;
  int jr = j + shift; if (jr >= n4) jr -= n4;
  const float4 first = in[jr];
  int j2 = jr + ((int)first.w & 1) + shift; while (j2 >= n4) j2 -= n4;
  float4 v[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) if (s < S) v[s] = in[(long)s * n4 + j2];
  float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int s = 0; s < 8; ++s) if (s < S) { a.x += v[s].x; a.y += v[s].y; a.z += v[s].z; a.w += v[s].w; }
#pragma unroll
  for (int s = 0; s < 8; ++s) if (s < S) out[(long)s * n4 + j] = make_float4(a.x * .5f, a.y * .5f, a.z * .5f, 0.f);
}

enum Api { API_GGL, API_EXT, API_GRAPH };
enum Str { STR_NB, STR_BLOCKING, STR_NULL };

static double run(Api api, Str str, int kind /*0 small 1 big 2 vec 3 vec cross-XCD 4 two dependent round trips*/, int wgs, int S, int chain, int reps, int shift = 256) {
  const int n = 64 * 512;
  float *a, *b;
  CK(hipMalloc(&a, sizeof(float) * n * 8));
  CK(hipMalloc(&b, sizeof(float) * n * 8));
  CK(hipMemset(a, 0, sizeof(float) * n * 8));
  CK(hipMemset(b, 0, sizeof(float) * n * 8));
  hipStream_t st = nullptr;
  if (str == STR_NB) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  if (str == STR_BLOCKING) CK(hipStreamCreate(&st));
  Big big{};
  auto issue = [&](hipStream_t s) {
    for (int k = 0; k < chain; ++k) {
      const float* in = (k & 1) ? b : a;
      float* out = (k & 1) ? a : b;
      if (kind == 3 || kind == 4) {
        const int g = (n / 4 + 255) / 256;
        if (kind == 3) hipLaunchKernelGGL(stage_vec_x, dim3(g), dim3(256), 0, s, (const float4*)in, (float4*)out, n / 4, S, shift);
        else hipLaunchKernelGGL(stage_vec_xx, dim3(g), dim3(256), 0, s, (const float4*)in, (float4*)out, n / 4, S, shift);
      } else if (kind == 2) {
        const int g = (n / 4 + 255) / 256;
        if (api == API_EXT) hipExtLaunchKernelGGL(stage_vec, dim3(g), dim3(256), 0, s, nullptr, nullptr, 0u, (const float4*)in, (float4*)out, n / 4, S);
        else hipLaunchKernelGGL(stage_vec, dim3(g), dim3(256), 0, s, (const float4*)in, (float4*)out, n / 4, S);
      } else if (kind == 1) {
        if (api == API_EXT) hipExtLaunchKernelGGL(stage_big, dim3(wgs), dim3(256), 0, s, nullptr, nullptr, 0u, in, out, n, S, big);
        else hipLaunchKernelGGL(stage_big, dim3(wgs), dim3(256), 0, s, in, out, n, S, big);
      } else {
        if (api == API_EXT) hipExtLaunchKernelGGL(stage_small, dim3(wgs), dim3(256), 0, s, nullptr, nullptr, 0u, in, out, n, S);
        else hipLaunchKernelGGL(stage_small, dim3(wgs), dim3(256), 0, s, in, out, n, S);
      }
    }
  };
  hipGraphExec_t exec = nullptr;
  if (api == API_GRAPH) {
    hipStream_t cs;
    CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipGraph_t g;
    CK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    issue(cs);
    CK(hipStreamEndCapture(cs, &g));
    CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(cs));
  }
  auto once = [&]() { if (exec) CK(hipGraphLaunch(exec, st)); else issue(st); };
  for (int r = 0; r < 20; ++r) once();
  CK(hipStreamSynchronize(st));
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) once();
  CK(hipStreamSynchronize(st));
  auto t1 = std::chrono::steady_clock::now();
  CK(hipGetLastError());
  if (exec) CK(hipGraphExecDestroy(exec));
  if (st) CK(hipStreamDestroy(st));
  CK(hipFree(a)); CK(hipFree(b));
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / ((double)reps * chain);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 400;
  const bool only_graph = argc > 2;      // any second argument: only the graph-of-48 table (short run for rocprofv3)
  // third argument: process conditions of a framework process -- h = a host-mapped pinned allocation exists, s = 40 more streams
  // exist, m = 4 GiB of device memory allocated
  if (argc > 3) {
    for (const char* c = argv[3]; *c; ++c) {
      if (*c == 'h') { void* hp; CK(hipHostMalloc(&hp, 1 << 20, hipHostMallocMapped)); printf("# host-mapped pinned allocation made\n"); }
      if (*c == 's') { for (int i = 0; i < 40; ++i) { hipStream_t t; CK(hipStreamCreateWithFlags(&t, hipStreamNonBlocking)); hipLaunchKernelGGL(stage_small, dim3(1), dim3(256), 0, t, (const float*)nullptr, (float*)nullptr, 0, 0); } CK(hipDeviceSynchronize()); printf("# 40 extra streams created and used\n"); }
      if (*c == 'm') { void* dp; CK(hipMalloc(&dp, 4ull << 30)); CK(hipMemset(dp, 1, 4ull << 30)); printf("# 4 GiB allocated\n"); }
    }
  }
  const char* env = getenv("HIP_FORCE_DEV_KERNARG");
  printf("# boundary_probe: us per dependent launch (wall / launches), chain of 12, %d repetitions; HIP_FORCE_DEV_KERNARG=%s\n", reps, env ? env : "(unset)");
  const char* apin[] = {"hipLaunchKernelGGL", "hipExtLaunchKernelGGL", "hipGraph(12)"};
  const char* strn[] = {"non-blocking", "blocking", "null"};
  const char* kindn[] = {"small-args", "320B-args", "vec16"};
  printf("%-22s %-13s %-10s %4s %2s %8s\n", "api", "stream", "args", "wgs", "S", "us");
  for (int pass = 0; pass < 2 && !only_graph; ++pass)
    for (int api = 0; api < 3; ++api)
      for (int str = 0; str < 3; ++str) {
        if (api == API_GRAPH && str == STR_BLOCKING) continue;
        for (int kind = 0; kind < 3; ++kind)
          for (int wgs : {128, 256})
            for (int S : {1, 4}) {
              if (kind == 2 && wgs == 256) continue;
              if (pass == 0 && !(kind == 0 && wgs == 128 && S == 1)) continue;   // pass 0: one warm row per path
              double us = run((Api)api, (Str)str, kind, wgs, S, 12, reps);
              if (pass) printf("%-22s %-13s %-10s %4d %2d %8.2f\n", apin[api], strn[str], kindn[kind], kind == 2 ? 32 : wgs, S, us);
            }
      }
  // chain length: is the per-launch price flat in the chain length (host-bound paths are not)?
  for (int chain : {1, 4, 12, 48})
    if (!only_graph) printf("chain %2d  eager non-blocking small S=1: %.2f us   graph: %.2f us\n", chain,
           run(API_GGL, STR_NB, 0, 128, 1, chain, reps), run(API_GRAPH, STR_NB, 0, 128, 1, chain, reps));
  printf("# graph of 48, non-blocking stream: where the dependent bytes come from\n");
  for (int S : {1, 4, 8}) {
    printf("S=%d  same block (XCD-local L2): %.2f   next block (another XCD): %.2f   4 blocks on: %.2f   two dependent round trips, other XCD: %.2f\n", S,
           run(API_GRAPH, STR_NB, 3, 32, S, 48, reps, 0), run(API_GRAPH, STR_NB, 3, 32, S, 48, reps, 256),
           run(API_GRAPH, STR_NB, 3, 32, S, 48, reps, 1024), run(API_GRAPH, STR_NB, 3, 32, S, 48, reps, 256 + 0 * S) * 0 + run(API_GRAPH, STR_NB, 4, 32, S, 48, reps, 256));
  }
  return 0;
}
