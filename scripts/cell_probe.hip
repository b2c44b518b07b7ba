// cell_probe (round 4): the decoder's LSTM cell as ONE launch per direction of time, as north_star words it --
// "a fused 4-gate LSTM cell": the gate contraction [B, K] x [4H, K]^T (K = AE + F + H = 2752, 4H = 2048, B = 64) and the
// i, f, g, o pointwise update in the SAME kernel -- against what the product ships: gemm_nt (split-K over workgroups into
// partial slabs) + lstm_pw (sums the slabs while loading).  Reference: nn.LSTMCell at policy.py:192,237-238.
//
// A fused cell needs FINISHED gate sums in the workgroup that applies the nonlinearities, so K cannot be split over workgroups:
// it is split over the WAVES of one workgroup and reduced through LDS.  Two tilings, both with every (row, unit) of the cell
// owned by exactly one thread of the epilogue:
//   A  64 rows x (4 units x 4 gates = 16 columns) per workgroup: 128 workgroups, each streams 16 x 2752 weights (88 KB bf16)
//      and the WHOLE activation block 64 x 2752 fp32 (704 KB, from L2 after the first toucher of the XCD)
//   B  32 rows x 16 columns: 256 workgroups, 88 KB of weights + 352 KB of activations each
// X is split hi + lo in registers exactly like gemm_nt does in LDS (two bf16 MFMAs per product; same arithmetic).
// The activations are rewritten by another kernel before every launch, as in the real chain.
//   bash scripts/build_cell_probe.sh && scripts/cell_probe
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../curriculum-learning-for-vln_amd/csrc/vln_internal.h"
#include "../include/vln_hip.h"
#include <type_traits>
#include "../curriculum-learning-for-vln_amd/csrc/step_bodies.h"
namespace vln {
#include "../curriculum-learning-for-vln_amd/csrc/gemm_nt_body.h"
}
using namespace vln;

// SHARDED cell (round 4, the measured basis of DESIGN section 10's episode-sharded persistent decoder): ONE launch of 8 groups x
// `wpg` workgroups; group g = blockIdx % 8 sits on XCD g and owns the episodes [g * B / 8, (g + 1) * B / 8): its workgroups walk ALL
// tiles of the gate product for those rows (gemm_nt_body, the product's own K split and slab layout), meet at a barrier among the
// group's workgroups only -- drained stores + a relaxed agent-scope counter, NO fences (one XCD = one L2) -- and apply the
// pointwise stage to the group's rows (lstm_pw_fwd_body reading the slabs).  Same arithmetic per row as the two launches.
__global__ __launch_bounds__(256) void sharded_cell_kernel(GemmNTArgs ga, int gx, int gy, LstmPwFwd pw, unsigned* bar, int rpg, int wpg) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[gemm_nt_smem_bytes(2)];
  __shared__ float sg[4][64];
  const int group = blockIdx.x & 7, local = blockIdx.x >> 3, tid = threadIdx.x;
  const int r0 = group * rpg;
  GemmNTArgs a = ga;
  a.X += (long)r0 * a.ldx; a.Y += (long)r0 * a.ldy; a.M = rpg;
  const int nvb = gx * gy, iters = (nvb + wpg - 1) / wpg;
  for (int it = 0; it < iters; ++it) {
    const int v = local + it * wpg;
    const bool active = v < nvb;
    const int vc = active ? v : nvb - 1;
    const VBlock vb{vc % gx, vc / gx, 0, tid, smem};
    gemm_nt_body<bf16_raw, 2, true, 1>(a, vb, active, gemm_nt_nsteps(a, vb.by, 64), [] {});
    __syncthreads();
  }
  // the group's barrier (single use per launch; the last workgroup through leaves both words zero)
  unsigned* cnt = bar + group * 64;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)wpg) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 24)) break;
    }
    const unsigned through = __hip_atomic_fetch_add(cnt + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (through + 1u == (unsigned)wpg) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(cnt + 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  LstmPwFwd p = pw;
  const long ro = (long)r0;
  p.gates += ro * 4 * p.H; p.c0 += ro * p.ldc0; p.h1 += ro * p.ldh1; p.c1 += ro * p.ldc1; p.act += ro * 4 * p.H; p.tanh_c1 += ro * p.H; p.B = rpg;
  const int groups64 = (rpg * p.H + 63) / 64;
  lstm_pw_fwd_body(p, local, wpg, (groups64 + wpg - 1) / wpg, tid, sg);
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ROWS = rows per workgroup (64 or 32); 256 threads = 4 waves; wave w contracts K-steps w, w + 4, ... (64 columns of K each)
template <int ROWS>
__global__ __launch_bounds__(256) void fused_cell_kernel(const float* __restrict__ X, long ldx, const unsigned short* __restrict__ W, long ldw,
                                                         const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                         const float* __restrict__ c0, float* __restrict__ h1, float* __restrict__ c1,
                                                         float* __restrict__ act, float* __restrict__ tanh_c, int B, int H, int K) {
  constexpr int RB = ROWS / 16;
  __shared__ float red[4][ROWS][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fi = lane & 15, fq = lane >> 4;
  const int u0 = blockIdx.x * 4;                   // first of this workgroup's 4 units
  const int m0 = blockIdx.y * ROWS;
  // this lane's weight row: column c = gate * 4 + unit_local  ->  weight row gate * H + u0 + unit_local
  const int gate = fi >> 2, ul = fi & 3;
  const unsigned short* wrow = W + (long)(gate * H + u0 + ul) * ldw;
  const float* xrow[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) xrow[r] = X + (long)min(m0 + r * 16 + fi, B - 1) * ldx;
  f32x4_t acc[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) acc[r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nsteps = K / 64;
  for (int s = wave; s < nsteps; s += 4) {
    const int k = s * 64 + fq * 16;                // this lane's 16 consecutive k (two MFMA k-groups of 8)
    const bf16x8_t w0 = *reinterpret_cast<const bf16x8_t*>(wrow + k), w1 = *reinterpret_cast<const bf16x8_t*>(wrow + k + 8);
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      float x[16];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float4 t = *reinterpret_cast<const float4*>(xrow[r] + k + v * 4);
        x[v * 4] = t.x; x[v * 4 + 1] = t.y; x[v * 4 + 2] = t.z; x[v * 4 + 3] = t.w;
      }
      bf16x8_t h0, h1v, l0, l1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        h0[j] = (__bf16)x[j]; l0[j] = (__bf16)(x[j] - (float)h0[j]);
        h1v[j] = (__bf16)x[8 + j]; l1[j] = (__bf16)(x[8 + j] - (float)h1v[j]);
      }
      acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0, w0, acc[r], 0, 0, 0);
      acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, w1, acc[r], 0, 0, 0);
      acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, w0, acc[r], 0, 0, 0);
      acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1v, w1, acc[r], 0, 0, 0);
    }
  }
  // C layout of the 16x16 MFMA: column = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][r * 16 + fq * 4 + e][fi] = acc[r][e];
  __syncthreads();
  // epilogue: one thread per (row, unit)
  for (int t = threadIdx.x; t < ROWS * 4; t += 256) {
    const int row = t >> 2, u = t & 3;
    const int b = m0 + row;
    if (b >= B) continue;
    float g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = q * 4 + u;
      g[q] = ((red[0][row][c] + red[1][row][c]) + (red[2][row][c] + red[3][row][c])) + b_ih[q * H + u0 + u] + b_hh[q * H + u0 + u];
    }
    const float si = 1.f / (1.f + __expf(-g[0])), sf = 1.f / (1.f + __expf(-g[1])), tg = tanhf(g[2]), so = 1.f / (1.f + __expf(-g[3]));
    const long e = (long)b * H + u0 + u;
    const float cn = sf * c0[e] + si * tg, tc = tanhf(cn);
    h1[e] = so * tc; c1[e] = cn; tanh_c[e] = tc;
    float* a = act + (long)b * 4 * H + u0 + u;
    a[0] = si; a[H] = sf; a[2 * H] = tg; a[3 * H] = so;
  }
}

// SHARDED cell, streaming body (what an episode-sharded decoder's GEMM stage has to look like): group g = blockIdx % 8 (one XCD) owns
// R = B / 8 episodes.  Its activation rows [R, K] are split hi + lo into two bf16 LDS planes ONCE (88 KB at K = 2752: one workgroup per
// CU); every wave then owns 16 gate columns x a K range and STREAMS its 16 weight rows with kDepth x 64 k of loads in flight per
// lane -- no LDS staging per K-step, no barrier in the loop.  K is split over `ksplit` waves of the SAME workgroup (partials meet in
// LDS), so the pre-activations leave the workgroup finished: no slabs.  Then the group's fence-free barrier and the pointwise stage.
constexpr int kSkDepth = 6;
__global__ __launch_bounds__(512) void sharded_stream_cell_kernel(const float* __restrict__ X, long ldx, const unsigned short* __restrict__ W, long ldw,
                                                                  const float* b_ih, const float* b_hh, const float* c0, float* h1, float* c1, float* act,
                                                                  float* tanh_c, float* pre /*[KH][B, 4H] scratch*/, unsigned* bar, int B, int H, int K, int wpg,
                                                                  int NG /*groups: 8 or 4 (XCDs used)*/, int KH /*K parts, each on wpg / KH workgroups*/,
                                                                  int xmode /*0: the rows staged into LDS planes first; 1: X fragments straight from L2, split per step*/,
                                                                  long long* stamps /*nullable: wall_clock64 at the phase ends of group 0's first workgroup*/) {
#define SK_STAMP(i) do { if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[i] = wall_clock64(); } while (0)
  SK_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  const int R = B / NG;                                  // rows of the group (<= 16)
  const int group = blockIdx.x & 7, tid = threadIdx.x;
  if (group >= NG) return;
  const int kh = (int)(blockIdx.x >> 3) % KH, local = (int)(blockIdx.x >> 3) / KH;      // K part, workgroup index inside the part
  const int wph = wpg / KH;                                                               // workgroups per K part
  const int lane = tid & 63, wave = tid >> 6, fi = lane & 15, fq = lane >> 4;
  const int r0 = group * R;
  const int ksteps_all = K / 64;
  const int hs0 = (int)((long)ksteps_all * kh / KH), hs1 = (int)((long)ksteps_all * (kh + 1) / KH);      // this part's K-steps
  const int Kp = (hs1 - hs0) * 64, kof = hs0 * 64;
  const int LDP = Kp + 8;                                // plane row stride in bf16 (16-byte pad)
  __bf16* ph = reinterpret_cast<__bf16*>(dyn);
  __bf16* pl = ph + (long)R * LDP;
  float* red = reinterpret_cast<float*>(pl + (long)R * LDP);          // [8 waves][16 rows][17]
  // (1) activations -> LDS planes
  if (xmode == 0) {
    const long total4 = (long)R * (Kp / 4);
    constexpr int kU = 12;                                // float4 loads in flight per thread (R = 16, Kp = 1408: 11 per thread)
    for (long e0 = tid; e0 < total4; e0 += 512L * kU) {
      float4 v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const long e = min(e0 + 512L * u, total4 - 1);
        const int r = (int)(e / (Kp / 4)), k4 = (int)(e % (Kp / 4)) * 4;
        v[u] = *reinterpret_cast<const float4*>(X + (long)(r0 + r) * ldx + kof + k4);
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const long e = e0 + 512L * u;
        if (e < total4) {
          const int r = (int)(e / (Kp / 4)), k4 = (int)(e % (Kp / 4)) * 4;
          const float x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
          typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
          bf16x4_t h, l;
#pragma unroll
          for (int j = 0; j < 4; ++j) { h[j] = (__bf16)x[j]; l[j] = (__bf16)(x[j] - (float)h[j]); }
          *reinterpret_cast<bf16x4_t*>(ph + (long)r * LDP + k4) = h;
          *reinterpret_cast<bf16x4_t*>(pl + (long)r * LDP + k4) = l;
        }
      }
    }
  }
  __syncthreads();
  SK_STAMP(1);
  // (2) every wave: column tile ct (16 gate columns), K part kp of ksplit
  const int N = 4 * H, ntiles = N / 16;
  const int waves_g = wph * 8;                           // waves of this K part
  const int ksplit = max(1, waves_g / ntiles);           // waves per column tile (all in ONE workgroup: ksplit divides 8)
  const int gw = local * 8 + wave;                       // wave index in the part
  // (the groups read the SAME weights: each starts at another column tile so that the XCDs do not walk the same lines -- the same
  // memory channels -- in lockstep)
  const int ct0 = gw / ksplit, kp = gw % ksplit;
  const int ct = ct0 < ntiles ? (ct0 + group * (ntiles / 8)) % ntiles : ct0;
  const int ksteps = hs1 - hs0;                          // 64 k per step (two MFMA k-blocks), relative to the part's first
  const int sbeg = (int)((long)ksteps * kp / ksplit), send = (int)((long)ksteps * (kp + 1) / ksplit);
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  if (ct < ntiles) {
    const unsigned short* wrow = W + (long)(ct * 16 + fi) * ldw + kof + fq * 16;      // 32 contiguous bytes per lane and step: whole 128-byte lines per row
    bf16x8_t wq[kSkDepth][2];
    auto ldw_ = [&](int buf, int s) {
      wq[buf][0] = *reinterpret_cast<const bf16x8_t*>(wrow + (long)s * 64);
      wq[buf][1] = *reinterpret_cast<const bf16x8_t*>(wrow + (long)s * 64 + 8);
    };
    // every load is UNCONDITIONAL (clamped step index): a branch around a load closes the loads in flight with vmcnt(0)
#pragma unroll
    for (int d = 0; d < kSkDepth; ++d) ldw_(d, min(sbeg + d, send - 1));
    const bool row_ok = fi < R;
    if (xmode == 1) {
      // X fragments from global memory (the group's rows are L2-resident), prefetched like the weights, split hi + lo in registers
      const float* xrow = X + (long)(r0 + (row_ok ? fi : 0)) * ldx + kof + fq * 16;
      float4 xq[kSkDepth][4];
      auto ldx_ = [&](int buf, int s) {
#pragma unroll
        for (int v = 0; v < 4; ++v) xq[buf][v] = *reinterpret_cast<const float4*>(xrow + (long)s * 64 + v * 4);
      };
#pragma unroll
      for (int d = 0; d < kSkDepth; ++d) ldx_(d, min(sbeg + d, send - 1));
      for (int s0 = sbeg; s0 < send; s0 += kSkDepth) {
#pragma unroll
        for (int d = 0; d < kSkDepth; ++d) {
          const int s = s0 + d;
          const bf16x8_t w0 = wq[d][0], w1 = wq[d][1];
          float x[16];
#pragma unroll
          for (int v = 0; v < 4; ++v) { x[v * 4] = xq[d][v].x; x[v * 4 + 1] = xq[d][v].y; x[v * 4 + 2] = xq[d][v].z; x[v * 4 + 3] = xq[d][v].w; }
          ldw_(d, min(s + kSkDepth, send - 1));
          ldx_(d, min(s + kSkDepth, send - 1));
          bf16x8_t a0, a1, l0, l1;
          const bool on = row_ok && s < send;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float u0 = on ? x[j] : 0.f, u1 = on ? x[8 + j] : 0.f;
            a0[j] = (__bf16)u0; l0[j] = (__bf16)(u0 - (float)a0[j]);
            a1[j] = (__bf16)u1; l1[j] = (__bf16)(u1 - (float)a1[j]);
          }
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0, w0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, w1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, w1, acc, 0, 0, 0);
        }
      }
    } else {
    const __bf16* xh = ph + (long)(row_ok ? fi : 0) * LDP + fq * 16;
    const __bf16* xl = pl + (long)(row_ok ? fi : 0) * LDP + fq * 16;
    for (int s0 = sbeg; s0 < send; s0 += kSkDepth) {
#pragma unroll
      for (int d = 0; d < kSkDepth; ++d) {
        const int s = s0 + d;
        {
          const bf16x8_t w0 = wq[d][0], w1 = wq[d][1];
          ldw_(d, min(s + kSkDepth, send - 1));
          const int sx = min(s, send - 1);
          bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(xh + (long)sx * 64), a1 = *reinterpret_cast<const bf16x8_t*>(xh + (long)sx * 64 + 8);
          bf16x8_t l0 = *reinterpret_cast<const bf16x8_t*>(xl + (long)sx * 64), l1 = *reinterpret_cast<const bf16x8_t*>(xl + (long)sx * 64 + 8);
          if (!row_ok || s >= send) { a0 = bf16x8_t{}; a1 = bf16x8_t{}; l0 = bf16x8_t{}; l1 = bf16x8_t{}; }      // (steps past the end multiply zeros)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0, w0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, w1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, w1, acc, 0, 0, 0);
        }
      }
    }
    }
  }
  SK_STAMP(2);
  // partials of the ksplit waves of a tile meet in LDS (they are consecutive waves of this workgroup)
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 16 + fq * 4 + r) * 17 + fi] = acc[r];
  __syncthreads();
  if (kp == 0 && ct < ntiles) {
    for (int e = lane; e < R * 16; e += 64) {
      const int r = e >> 4, c = e & 15;
      float v = 0.f;
      for (int q = 0; q < ksplit; ++q) v += red[((wave + q) * 16 + r) * 17 + c];
      pre[(long)kh * B * N + (long)(r0 + r) * N + ct * 16 + c] = v;
    }
  }
  SK_STAMP(3);
  // (3) the group's barrier
  unsigned* cnt = bar + group * 64;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)wpg) { __builtin_amdgcn_s_sleep(2); if (++spins > (1u << 24)) break; }
    const unsigned through = __hip_atomic_fetch_add(cnt + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (through + 1u == (unsigned)wpg) { __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(cnt + 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  }
  __syncthreads();
  SK_STAMP(4);
  // (4) pointwise on the group's rows (pre-activations written by the group's other workgroups: same L2, not in this CU's L1)
  for (int e = (int)(blockIdx.x >> 3) * 512 + tid; e < R * H; e += wpg * 512) {
    const int r = r0 + e / H, j = e % H;
    const float* g = pre + (long)r * N + j;
    float gi = b_ih[j] + b_hh[j], gf = b_ih[H + j] + b_hh[H + j], gg = b_ih[2 * H + j] + b_hh[2 * H + j], go = b_ih[3 * H + j] + b_hh[3 * H + j];
    for (int q = 0; q < KH; ++q) { const float* gq = g + (long)q * B * N; gi += gq[0]; gf += gq[H]; gg += gq[2 * H]; go += gq[3 * H]; }
    const float si = 1.f / (1.f + __expf(-gi)), sf = 1.f / (1.f + __expf(-gf)), tg = tanhf(gg), so = 1.f / (1.f + __expf(-go));
    const long o = (long)r * H + j;
    const float cn = sf * c0[o] + si * tg, tcv = tanhf(cn);
    h1[o] = so * tcv; c1[o] = cn; tanh_c[o] = tcv;
    float* a = act + (long)r * N + j;
    a[0] = si; a[H] = sf; a[2 * H] = tg; a[3 * H] = so;
  }
  SK_STAMP(5);
}

__global__ void fill_f32_k(float* p, long n, float v) { for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v * 0.01f + (float)(i % 97) * 0.003f - 0.1f; }
__global__ void fill_w_k(unsigned short* p, long n) { for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { float f = ((float)((i * 2654435761u) % 1000) - 500.f) * 4e-5f; p[i] = (unsigned short)(__float_as_uint(f) >> 16); } }

static float median(std::vector<float>& t) { std::sort(t.begin(), t.end()); return t[t.size() / 2]; }

int main() {
  const int B = 64, H = 512, K = 2752, N = 4 * H;
  float *X, *c0, *h1, *c1, *act, *tc, *bi, *bh, *ws, *h1b, *c1b, *actb, *tcb;
  unsigned short* W;
  const long wsf = 16L * B * N + 1024;
  hipMalloc(&X, (long)B * K * 4); hipMalloc(&W, (long)N * K * 2); hipMalloc(&c0, (long)B * H * 4);
  hipMalloc(&h1, (long)B * H * 4); hipMalloc(&c1, (long)B * H * 4); hipMalloc(&act, (long)B * N * 4); hipMalloc(&tc, (long)B * H * 4);
  hipMalloc(&h1b, (long)B * H * 4); hipMalloc(&c1b, (long)B * H * 4); hipMalloc(&actb, (long)B * N * 4); hipMalloc(&tcb, (long)B * H * 4);
  hipMalloc(&bi, N * 4); hipMalloc(&bh, N * 4); hipMalloc(&ws, wsf * 4);
  hipLaunchKernelGGL(fill_w_k, dim3(1024), dim3(256), 0, 0, W, (long)N * K);
  hipLaunchKernelGGL(fill_f32_k, dim3(64), dim3(256), 0, 0, c0, (long)B * H, 1.f);
  hipLaunchKernelGGL(fill_f32_k, dim3(8), dim3(256), 0, 0, bi, (long)N, 2.f);
  hipLaunchKernelGGL(fill_f32_k, dim3(8), dim3(256), 0, 0, bh, (long)N, 3.f);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* what, auto&& launch) {
    std::vector<float> t;
    for (int r = 0; r < 25; ++r) {
      hipLaunchKernelGGL(fill_f32_k, dim3(256), dim3(256), 0, 0, X, (long)B * K, 0.5f + r);     // the producer of xcat
      hipEventRecord(e0, 0);
      launch();
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
    }
    const float m = median(t);
    printf("%-78s median %6.2f us   min %6.2f us\n", what, m, t[0]);
    return m;
  };
  // ... and inside a hipGraph (what the whole-iteration graph pays): 20 x (refill X, the launches) captured, replayed, per repetition,
  // minus the same graph with the refill only
  hipStream_t cs; hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
  auto graph_us = [&](auto&& launch) {
    hipGraph_t g; hipGraphExec_t ex;
    hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    for (int r = 0; r < 20; ++r) { hipLaunchKernelGGL(fill_f32_k, dim3(256), dim3(256), 0, cs, X, (long)B * K, 0.5f + r); launch(cs); }
    hipStreamEndCapture(cs, &g); hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    std::vector<float> t;
    for (int i = 0; i < 9; ++i) {
      hipEventRecord(e0, cs); hipGraphLaunch(ex, cs); hipEventRecord(e1, cs); hipStreamSynchronize(cs);
      float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f / 20.f);
    }
    hipGraphExecDestroy(ex); hipGraphDestroy(g);
    return median(t);
  };
  // the product's two launches: split-K gemm_nt into slabs + the pointwise kernel that sums them while loading
  int nsplit = 0;
  timeit("product: gemm_nt (split-K slabs) + lstm_pw (2 launches)", [&] {
    gemm_nt(0, X, K, W, W_BF16, K, nullptr, 0, B, N, K, nullptr, ACT_NONE, ws, wsf, &nsplit);
    LstmPwFwd pw{};
    pw.gates = ws; pw.nsplit = nsplit; pw.slab_stride = (long)B * N; pw.bias_a = bi; pw.bias_b = bh; pw.c0 = c0; pw.ldc0 = H;
    pw.h1 = h1b; pw.ldh1 = H; pw.c1 = c1b; pw.ldc1 = H; pw.act = actb; pw.tanh_c1 = tcb; pw.h1_drop = nullptr; pw.B = B; pw.H = H;
    lstm_pointwise_fwd(0, pw);
  });
  printf("   (gemm_nt split K over %d workgroup rows -> %d workgroups)\n", nsplit, nsplit * (N / 64));
  timeit("   gemm_nt alone", [&] { gemm_nt(0, X, K, W, W_BF16, K, nullptr, 0, B, N, K, nullptr, ACT_NONE, ws, wsf, &nsplit); });
  timeit("fused cell A: 64 rows x 16 gate columns per workgroup, 128 workgroups (1 launch)", [&] {
    hipLaunchKernelGGL(fused_cell_kernel<64>, dim3(H / 4, 1), dim3(256), 0, 0, X, (long)K, W, (long)K, bi, bh, c0, h1, c1, act, tc, B, H, K);
  });
  timeit("fused cell B: 32 rows x 16 gate columns per workgroup, 256 workgroups (1 launch)", [&] {
    hipLaunchKernelGGL(fused_cell_kernel<32>, dim3(H / 4, 2), dim3(256), 0, 0, X, (long)K, W, (long)K, bi, bh, c0, h1, c1, act, tc, B, H, K);
  });
  // the sharded cell: the product's split (kchunk, slab count) taken from gemm_nt itself
  unsigned* bar; hipMalloc(&bar, 8 * 64 * 4); hipMemset(bar, 0, 8 * 64 * 4);
  const int steps_per = (K / 64 + nsplit - 1) / nsplit;
  GemmNTArgs sga{}; sga.X = X; sga.ldx = K; sga.W = W; sga.ldw = K; sga.Y = ws; sga.ldy = N; sga.slab_stride = (long)B * N; sga.bias = nullptr; sga.act = ACT_NONE;
  sga.M = B; sga.N = N; sga.K = K; sga.kchunk = steps_per * 64; sga.xvec = 1; sga.wvec = 1; sga.xcd = 0;
  auto spw = [&]() { LstmPwFwd pw{}; pw.gates = ws; pw.nsplit = nsplit; pw.slab_stride = (long)B * N; pw.bias_a = bi; pw.bias_b = bh; pw.c0 = c0; pw.ldc0 = H;
                     pw.h1 = h1; pw.ldh1 = H; pw.c1 = c1; pw.ldc1 = H; pw.act = act; pw.tanh_c1 = tc; pw.h1_drop = nullptr; pw.B = B; pw.H = H; return pw; };
  for (int wpg : {32, 64})
    timeit(wpg == 32 ? "SHARDED cell: 8 XCD groups x 32 workgroups, 8 episodes each (1 launch, group barrier)" :
                       "SHARDED cell: 8 XCD groups x 64 workgroups, 8 episodes each (1 launch, group barrier)", [&] {
      hipLaunchKernelGGL(sharded_cell_kernel, dim3(8 * wpg), dim3(256), 0, 0, sga, N / 64, nsplit, spw(), bar, B / 8, wpg);
    });
  // the streaming form: one 512-thread workgroup per CU (88 KB of LDS planes), 32 per XCD group
  float* pre; hipMalloc(&pre, 2L * B * N * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(sharded_stream_cell_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  int sNG = 8, sKH = 1, sXM = 0;
  long long* sStamps = nullptr;
  auto stream_launch = [&](hipStream_t st) {
    const int R_ = B / sNG, Kp = ((K / 64 + sKH - 1) / sKH) * 64;
    const size_t sk_lds = (size_t)2 * R_ * (Kp + 8) * 2 + 8 * 16 * 17 * 4;
    hipLaunchKernelGGL(sharded_stream_cell_kernel, dim3(8 * 32), dim3(512), sk_lds, st, X, (long)K, W, (long)K, bi, bh, c0, h1, c1, act, tc, pre, bar, B, H, K, 32, sNG, sKH, sXM, sStamps);
  };
  for (int v = 0; v < 4; ++v) {
    sNG = (v == 0 || v == 3) ? 8 : 4; sKH = (v == 0 || v == 3) ? 1 : 2; sXM = v >= 3 ? 1 : 0;
    if (v == 2) { sNG = 8; sKH = 2; }
    char what[200]; snprintf(what, sizeof what, "SHARDED cell, streaming body: %d XCD groups x 32 workgroups x 8 waves, %d episodes each, K in %d part(s), X %s", sNG, B / sNG, sKH,
                             sXM ? "from L2 per step" : "staged in LDS");
    timeit(what, [&] { stream_launch(0); });
    const float base = graph_us([&](hipStream_t) {});
    printf("   inside a hipGraph: %6.2f us\n", graph_us([&](hipStream_t st) { stream_launch(st); }) - base);
  }
  sNG = 8; sKH = 1; sXM = 0;
  {   // where the time goes inside one launch (group 0's first workgroup; wall_clock64 ticks at 100 MHz)
    long long* st; hipMalloc(&st, 8 * 8); hipMemset(st, 0, 64);
    sStamps = st;
    for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL(fill_f32_k, dim3(256), dim3(256), 0, 0, X, (long)B * K, 1.5f + i); stream_launch(0); }
    hipDeviceSynchronize();
    long long h[6]; hipMemcpy(h, st, 48, hipMemcpyDeviceToHost);
    printf("   phases of the streaming sharded cell (us): staging %.2f | weight stream + MFMA %.2f | cross-wave reduction %.2f | drain + barrier %.2f | pointwise %.2f | total %.2f\n",
           (h[1] - h[0]) * 0.01, (h[2] - h[1]) * 0.01, (h[3] - h[2]) * 0.01, (h[4] - h[3]) * 0.01, (h[5] - h[4]) * 0.01, (h[5] - h[0]) * 0.01);
    sStamps = nullptr;
  }
  {
    const float base = graph_us([&](hipStream_t) {});
    for (int wpg : {32, 64}) {
      const float tsh = graph_us([&](hipStream_t st) { hipLaunchKernelGGL(sharded_cell_kernel, dim3(8 * wpg), dim3(256), 0, st, sga, N / 64, nsplit, spw(), bar, B / 8, wpg); });
      printf("inside a hipGraph: sharded cell, %d workgroups per XCD group: %6.2f us\n", wpg, tsh - base);
    }
  }
  {
    const float base = graph_us([&](hipStream_t) {});
    auto pw_of = [&]() { LstmPwFwd pw{}; pw.gates = ws; pw.nsplit = nsplit; pw.slab_stride = (long)B * N; pw.bias_a = bi; pw.bias_b = bh; pw.c0 = c0; pw.ldc0 = H;
                         pw.h1 = h1b; pw.ldh1 = H; pw.c1 = c1b; pw.ldc1 = H; pw.act = actb; pw.tanh_c1 = tcb; pw.h1_drop = nullptr; pw.B = B; pw.H = H; return pw; };
    const float tp = graph_us([&](hipStream_t st) { gemm_nt(st, X, K, W, W_BF16, K, nullptr, 0, B, N, K, nullptr, ACT_NONE, ws, wsf, &nsplit); lstm_pointwise_fwd(st, pw_of()); });
    const float ta = graph_us([&](hipStream_t st) { hipLaunchKernelGGL(fused_cell_kernel<64>, dim3(H / 4, 1), dim3(256), 0, st, X, (long)K, W, (long)K, bi, bh, c0, h1, c1, act, tc, B, H, K); });
    const float tb = graph_us([&](hipStream_t st) { hipLaunchKernelGGL(fused_cell_kernel<32>, dim3(H / 4, 2), dim3(256), 0, st, X, (long)K, W, (long)K, bi, bh, c0, h1, c1, act, tc, B, H, K); });
    printf("inside a hipGraph, per repetition, refill-only graph (%.2f us) subtracted:\n  product (gemm_nt + lstm_pw) %6.2f us | fused cell A %6.2f us | fused cell B %6.2f us\n",
           base, tp - base, ta - base, tb - base);
  }
  // same numbers?  one more matched pair on the SAME activations
  hipLaunchKernelGGL(fill_f32_k, dim3(256), dim3(256), 0, 0, X, (long)B * K, 7.f);
  gemm_nt(0, X, K, W, W_BF16, K, nullptr, 0, B, N, K, nullptr, ACT_NONE, ws, wsf, &nsplit);
  {
    LstmPwFwd pw{};
    pw.gates = ws; pw.nsplit = nsplit; pw.slab_stride = (long)B * N; pw.bias_a = bi; pw.bias_b = bh; pw.c0 = c0; pw.ldc0 = H;
    pw.h1 = h1b; pw.ldh1 = H; pw.c1 = c1b; pw.ldc1 = H; pw.act = actb; pw.tanh_c1 = tcb; pw.h1_drop = nullptr; pw.B = B; pw.H = H;
    lstm_pointwise_fwd(0, pw);
  }
  hipLaunchKernelGGL(sharded_cell_kernel, dim3(8 * 32), dim3(256), 0, 0, sga, N / 64, nsplit, spw(), bar, B / 8, 32);
  hipDeviceSynchronize();
  {
    std::vector<float> a((long)B * H), b((long)B * H);
    hipMemcpy(a.data(), h1, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), h1b, b.size() * 4, hipMemcpyDeviceToHost);
    double md = 0;
    for (size_t i = 0; i < a.size(); ++i) md = std::max(md, (double)fabsf(a[i] - b[i]));
    printf("h1: max |sharded - product| = %.3e (0 = bit-identical)\n", md);
  }
  stream_launch(0);
  hipDeviceSynchronize();
  {
    std::vector<float> a((long)B * H), b((long)B * H);
    hipMemcpy(a.data(), h1, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), h1b, b.size() * 4, hipMemcpyDeviceToHost);
    double md = 0;
    for (size_t i = 0; i < a.size(); ++i) md = std::max(md, (double)fabsf(a[i] - b[i]));
    printf("h1: max |sharded streaming - product| = %.3e\n", md);
  }
  hipLaunchKernelGGL(fused_cell_kernel<32>, dim3(H / 4, 2), dim3(256), 0, 0, X, (long)K, W, (long)K, bi, bh, c0, h1, c1, act, tc, B, H, K);
  hipDeviceSynchronize();
  std::vector<float> a((long)B * H), b((long)B * H);
  hipMemcpy(a.data(), h1, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), h1b, b.size() * 4, hipMemcpyDeviceToHost);
  double md = 0, mx = 0;
  for (size_t i = 0; i < a.size(); ++i) { md = std::max(md, (double)fabsf(a[i] - b[i])); mx = std::max(mx, (double)fabsf(b[i])); }
  printf("h1: max |fused - product| = %.3e (max |h1| = %.3e)\n", md, mx);
  return 0;
}
