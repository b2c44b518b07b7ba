// One-launch attention rows (included by attention.hip inside namespace vln).
//
// The two-kernel path (attn_dot, then attn_softmax_wsum / attn_bwd) touches the [S,D] context of a batch row twice
// from two dependent launches, and its weighted-sum loop keeps ONE 16-byte load in flight per lane (S/4 dependent
// round trips to the fabric: 9 us for the 80 x 512 instruction context, 17 us with the in-place dctx update).  Here one
// workgroup of 8 waves owns a batch row: every lane issues ALL of its 16-byte loads up front (the whole [S,D] block
// lives in the workgroup's registers: 80 KB for the instruction context, 157 KB for the 36 x 2176 panorama in bf16),
// so the launch pays one memory latency, and the context is read exactly once for both passes.
//
//   forward  (units.py:106-118): dots = ctx . q -> w = softmax(mask(dots)) -> out = sum_s w[s] ctx[s,:]
//   backward                   : dots = ctx . dwc (= d alpha) -> w = alpha * (dots + ext - sum alpha (dots + ext))
//                                 -> out = sum_s w[s] ctx[s,:] (= d query);  w is also written out (d logits) so that
//                                 the context gradient can be formed once per rollout (attn_dctx_deferred).
//
// Geometry: wave w owns rows s = w, w+8, ... (RW per wave), lane l owns 16-byte segments l, l+64, ... (SL per row).
// Cross-wave sum of the partial weighted sums: fixed-order tree through LDS (deterministic).
#pragma once

constexpr int kFusedWaves = 8;

struct AttnFusedArgs {
  const void* ctx;        // [B,S,D] streamed (fp32 or bf16)
  SlabVec vec;            // [B,D] query (fwd) / d(weighted context) (bwd), possibly still in split-K slabs
  float* vec_out; long ldvo;   // nullable: the summed vector is written back (kept for backward / the deferred dctx)
  const uint8_t* mask;    // fwd: [B,S] 1 = masked, nullable
  float* alpha;           // fwd: out [B,S] (nullable); bwd: in [B,S]
  const float* ext;       // bwd: external gradient on alpha [B,S], nullable
  float* wout;            // bwd: d logits out [B,S], nullable
  float* out;             // [B,D] weighted sum
  long ldo;
  int S, D;
};

template <typename TC, int SL> constexpr int attn_fused_dp() { return 64 * SL * Elt<TC>::kVec; }
template <typename TC, int RW, int SL> constexpr int attn_fused_smem_bytes() {
  return (attn_fused_dp<TC, SL>() * (1 + kFusedWaves / 2) + kFusedWaves * RW) * (int)sizeof(float);
}

// One workgroup (8 waves) = one batch row `b`; `smem` = attn_fused_smem_bytes() of 16-byte aligned LDS.
// `pre()` runs after the block's context loads have been issued and before the query vector (the producer's output) is
// read: the chained step kernel (chain.hip) waits for the previous stage there.  `ctx_ready()` runs before the context
// loads (the chained kernel waits there for the stage that produced the context, if one did).
template <typename TC, int RW, int SL, bool kBwd, typename CtxReady, typename Pre>
__device__ __forceinline__ void attn_fused_body(const AttnFusedArgs& a, int b, unsigned char* smem, CtxReady ctx_ready, Pre pre) {
  constexpr int V = Elt<TC>::kVec;
  constexpr int NW = kFusedWaves;
  constexpr int DP = 64 * SL * V;                   // padded row length covered by the lanes
  float* sq = reinterpret_cast<float*>(smem);                                  // [DP]
  float (*red)[DP] = reinterpret_cast<float (*)[DP]>(smem + DP * sizeof(float));   // [NW / 2][DP]
  float* sdots = reinterpret_cast<float*>(smem + (size_t)DP * (1 + NW / 2) * sizeof(float));   // [NW * RW]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = a.S, D = a.D, nseg = D / V;
  const TC* base = reinterpret_cast<const TC*>(a.ctx) + (long)b * S * D;

  ctx_ready();
  // (1) every load of the block in flight at once
  uint4 data[RW][SL];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int s = wave + r * NW;
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      const int seg = lane + j * 64;
      if (s < S && seg < nseg) data[r][j] = *reinterpret_cast<const uint4*>(base + (long)s * D + (long)seg * V);
      else data[r][j] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  pre();
  for (int i = threadIdx.x * 4; i < DP; i += NW * 64 * 4) {      // D % 4 == 0 (host check)
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < D) {
      t = a.vec.at4(b, i);
      if (a.vec_out) *reinterpret_cast<float4*>(a.vec_out + (long)b * a.ldvo + i) = t;
    }
    *reinterpret_cast<float4*>(&sq[i]) = t;
  }
  __syncthreads();

  auto unpack = [](const uint4& u, float (&x)[V]) {
    if constexpr (V == 8) {
      x[0] = __uint_as_float(u.x << 16); x[1] = __uint_as_float(u.x & 0xffff0000u);
      x[2] = __uint_as_float(u.y << 16); x[3] = __uint_as_float(u.y & 0xffff0000u);
      x[4] = __uint_as_float(u.z << 16); x[5] = __uint_as_float(u.z & 0xffff0000u);
      x[6] = __uint_as_float(u.w << 16); x[7] = __uint_as_float(u.w & 0xffff0000u);
    } else {
      x[0] = __uint_as_float(u.x); x[1] = __uint_as_float(u.y); x[2] = __uint_as_float(u.z); x[3] = __uint_as_float(u.w);
    }
  };

  // (2) row dots
  float dot[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) dot[r] = 0.f;
#pragma unroll
  for (int j = 0; j < SL; ++j) {
    float qv[V];
#pragma unroll
    for (int e = 0; e < V; e += 4) {
      const float4 t = *reinterpret_cast<const float4*>(&sq[(lane + j * 64) * V + e]);
      qv[e] = t.x; qv[e + 1] = t.y; qv[e + 2] = t.z; qv[e + 3] = t.w;
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      float x[V];
      unpack(data[r][j], x);
#pragma unroll
      for (int e = 0; e < V; ++e) dot[r] += x[e] * qv[e];
    }
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const float t = wave_sum(dot[r]);
    if (lane == 0) sdots[wave + r * NW] = t;
  }
  __syncthreads();

  // (3) row weights: every wave derives all of them (S <= 128: two per lane), then picks its own rows
  float w0, w1;
  {
    const int s0 = lane, s1 = lane + 64;
    const long ro = (long)b * S;
    if constexpr (!kBwd) {
      float v0 = -INFINITY, v1 = -INFINITY;
      if (s0 < S && !(a.mask && a.mask[ro + s0])) v0 = sdots[s0];
      if (s1 < S && !(a.mask && a.mask[ro + s1])) v1 = sdots[s1];
      const float mx = wave_max(fmaxf(v0, v1));
      const float e0 = (s0 < S) ? __expf(v0 - mx) : 0.f, e1 = (s1 < S) ? __expf(v1 - mx) : 0.f;
      const float inv = 1.0f / wave_sum(e0 + e1);
      w0 = e0 * inv; w1 = e1 * inv;
      if (wave == 0 && a.alpha) {
        if (s0 < S) a.alpha[ro + s0] = w0;
        if (s1 < S) a.alpha[ro + s1] = w1;
      }
    } else {
      float a0 = 0.f, a1 = 0.f, g0 = 0.f, g1 = 0.f;
      if (s0 < S) { a0 = a.alpha[ro + s0]; g0 = sdots[s0] + (a.ext ? a.ext[ro + s0] : 0.f); }
      if (s1 < S) { a1 = a.alpha[ro + s1]; g1 = sdots[s1] + (a.ext ? a.ext[ro + s1] : 0.f); }
      const float tot = wave_sum(a0 * g0 + a1 * g1);
      w0 = a0 * (g0 - tot); w1 = a1 * (g1 - tot);
      if (wave == 0 && a.wout) {
        if (s0 < S) a.wout[ro + s0] = w0;
        if (s1 < S) a.wout[ro + s1] = w1;
      }
    }
  }

  // (4) partial weighted sums of this wave's rows
  float acc[SL][V];
#pragma unroll
  for (int j = 0; j < SL; ++j)
#pragma unroll
    for (int e = 0; e < V; ++e) acc[j][e] = 0.f;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int s = wave + r * NW;                    // wave-uniform
    const float wlo = __shfl(w0, s & 63, 64), whi = __shfl(w1, s & 63, 64);
    const float w = (s < 64) ? wlo : whi;           // rows past S carry zero data
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      float x[V];
      unpack(data[r][j], x);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[j][e] += w * x[e];
    }
  }

  // (5) cross-wave tree (fixed order): 8 -> 4 -> 2 -> 1
#pragma unroll
  for (int half = NW / 2; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half) {
#pragma unroll
      for (int j = 0; j < SL; ++j)
#pragma unroll
        for (int e = 0; e < V; e += 4)
          *reinterpret_cast<float4*>(&red[wave - half][(lane + j * 64) * V + e]) =
              make_float4(acc[j][e], acc[j][e + 1], acc[j][e + 2], acc[j][e + 3]);
    }
    __syncthreads();
    if (wave < half) {
#pragma unroll
      for (int j = 0; j < SL; ++j)
#pragma unroll
        for (int e = 0; e < V; e += 4) {
          const float4 t = *reinterpret_cast<const float4*>(&red[wave][(lane + j * 64) * V + e]);
          acc[j][e] += t.x; acc[j][e + 1] += t.y; acc[j][e + 2] += t.z; acc[j][e + 3] += t.w;
        }
    }
    __syncthreads();
  }
  if (wave == 0) {
    float* o = a.out + (long)b * a.ldo;
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      const int d = (lane + j * 64) * V;
      if (d < D) {
#pragma unroll
        for (int e = 0; e < V; e += 4)
          *reinterpret_cast<float4*>(o + d + e) = make_float4(acc[j][e], acc[j][e + 1], acc[j][e + 2], acc[j][e + 3]);
      }
    }
  }
}

template <typename TC, int RW, int SL, bool kBwd>
__global__ __launch_bounds__(kFusedWaves * 64) void attn_fused_kernel(AttnFusedArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[attn_fused_smem_bytes<TC, RW, SL>()];
  attn_fused_body<TC, RW, SL, kBwd>(a, (int)blockIdx.x, smem, [] {}, [] {});
}

#ifndef VLN_ATTN_FUSED_BODY_ONLY
template <typename TC, int RW, int SL>
static void attn_fused_launch(hipStream_t st, const AttnFusedArgs& a, int B, bool bwd, double bytes) {
  dim3 grid(B), block(kFusedWaves * 64);
  if (bwd) launch_timed(K_ATTN_BWD, bytes, attn_fused_kernel<TC, RW, SL, true>, grid, block, 0, st, a);
  else launch_timed(K_ATTN_WSUM, bytes, attn_fused_kernel<TC, RW, SL, false>, grid, block, 0, st, a);
}

// Returns true when a fused configuration covers (ctype, S, D) and the pointers are aligned; the launch is then issued.
static bool attn_fused_try(hipStream_t st, int ctype, const AttnFusedArgs& a, int B, bool bwd) {
  if (g_tunable[4]) return false;                               // tunable[4] = 1 forces the two-kernel path (A/B, tests)
  const int V = (ctype == W_BF16) ? 8 : 4;
  const int S = a.S, D = a.D;
  if (D % V != 0 || !aligned16(a.ctx) || !aligned16(a.vec.p) || !aligned16(a.out) || (a.vec.ld & 3) || (a.vec.stride & 3) || !aligned16(a.vec.bias) ||
      (a.ldo & 3) || (D & 3) || (a.vec_out && (!aligned16(a.vec_out) || (a.ldvo & 3)))) return false;
  const int nseg = D / V;
  const int sl = (nseg + 63) / 64, rw = (S + kFusedWaves - 1) / kFusedWaves;
  const double bytes = (double)B * S * D * (ctype == W_BF16 ? 2 : 4) + 8.0 * B * D + 8.0 * B * S;
#define VLN_FUSED_CASE(TC, RWv, SLv, k) \
  if (rw <= RWv && sl <= SLv) { \
    attn_fused_launch<TC, RWv, SLv>(st, a, B, bwd, bytes); return true; }
  if (ctype == W_BF16) {
    VLN_FUSED_CASE(bf16_raw, 10, 1, 0)       // instruction context: S <= 80, D <= 512
    VLN_FUSED_CASE(bf16_raw, 2, 2, 1)        // projected candidates (Self-Monitor): S <= 16, D <= 1024
    VLN_FUSED_CASE(bf16_raw, 2, 5, 2)        // candidate features: S <= 16, D <= 2560
    VLN_FUSED_CASE(bf16_raw, 5, 5, 3)        // panorama: S <= 40, D <= 2560
  } else {
    VLN_FUSED_CASE(float, 10, 2, 0)          // S <= 80, D <= 512
    VLN_FUSED_CASE(float, 2, 4, 1)           // S <= 16, D <= 1024
    VLN_FUSED_CASE(float, 2, 9, 2)           // S <= 16, D <= 2304
    VLN_FUSED_CASE(float, 5, 9, 3)           // S <= 40, D <= 2304
  }
#undef VLN_FUSED_CASE
  return false;
}

// ---------------------------------------------------------------------------------------------------------------
// Context gradient of a whole rollout in one pass (replaces T read-modify-write sweeps over [B,L,H]):
//   dctx[b,s,:] (+)= sum_t  alpha_t[b,s] * g_t[b,:]  +  dl_t[b,s] * q_t[b,:]
// with g_t = d(weighted context) and q_t = the attention query of step t (units.py:106-118 differentiated).
// grid (B, ceil(S/16)); the 2T vectors of the batch row are staged in LDS once per workgroup.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kDctxMaxSteps = 24;
struct DctxArgs {
  const float* alpha[kDctxMaxSteps];   // [B,S]
  const float* dl[kDctxMaxSteps];      // [B,S]
  const float* g[kDctxMaxSteps];       // [B,*] row stride ldg
  const float* q[kDctxMaxSteps];       // [B,*] row stride ldq
  long ldg, ldq;
  float* dctx;                         // [B,S,D]
  float* dk;                           // nullable [B,S,D]: the (dl, q) half goes HERE (overwritten) instead of into dctx
  int T, S, D, accumulate, dk_accumulate, vec_ok;
  // optional: step t's contribution passed through a dropout mask of the attended tensor (the Self-Monitor agent attends
  // dropout(ctx + pe) with a fresh mask every step: units.py:188-207); p == 0: none.  Mask index = flat [B,S,D] index.
  unsigned long long drop_seed[kDctxMaxSteps], drop_off[kDctxMaxSteps]; float drop_p[kDctxMaxSteps];
  const unsigned long long* drop_base;   // nullable: offsets relative to a device word, (*drop_base) * 8 + drop_off[t]
};
__global__ __launch_bounds__(256) void attn_dctx_deferred_kernel(DctxArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [2T][D] vectors, then [2T][16] row weights
  const int b = blockIdx.x, s0 = blockIdx.y * 16;
  const int T = a.T, D = a.D, S = a.S;
  float* sv = sm;
  float* swt = sm + (long)2 * T * D;
  for (int i = threadIdx.x; i < 2 * T * (D / 4); i += 256) {
    const int v = i / (D / 4), d4 = i % (D / 4);
    const float* base = (v < T) ? a.g[v] : a.q[v - T];
    const float* src = base ? base + (long)b * ((v < T) ? a.ldg : a.ldq) + d4 * 4 : nullptr;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (src) {
      if (a.vec_ok) t = *reinterpret_cast<const float4*>(src);
      else t = make_float4(src[0], src[1], src[2], src[3]);
    }
    *reinterpret_cast<float4*>(&sv[(long)v * D + d4 * 4]) = t;
  }
  for (int i = threadIdx.x; i < 2 * T * 16; i += 256) {
    const int v = i / 16, r = i % 16;
    const int s = s0 + r;
    float w = 0.f;
    const float* wp = (v < T) ? a.alpha[v] : a.dl[v - T];
    if (s < S && wp) w = wp[(long)b * S + s];
    swt[i] = w;
  }
  __syncthreads();
  // thread -> (row r = tid / 16, float4 column group c = tid % 16 + 16 k)
  const int r = threadIdx.x >> 4, c0 = threadIdx.x & 15;
  const int s = s0 + r;
  if (s >= S) return;
  float* out = a.dctx + ((long)b * S + s) * D;
  bool any_drop = false;
  for (int t = 0; t < T; ++t) any_drop |= a.drop_p[t] > 0.f;
  for (int c = c0; c < D / 4; c += 16) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.dk) {                               // two outputs: sum_t alpha_t g_t -> dctx, sum_t dl_t q_t -> dk
      float4 acc2 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int v = 0; v < T; ++v) {
        const float w = swt[v * 16 + r], w2 = swt[(T + v) * 16 + r];
        const float4 x = *reinterpret_cast<const float4*>(&sv[(long)v * D + c * 4]);
        const float4 x2 = *reinterpret_cast<const float4*>(&sv[(long)(T + v) * D + c * 4]);
        acc.x += w * x.x; acc.y += w * x.y; acc.z += w * x.z; acc.w += w * x.w;
        acc2.x += w2 * x2.x; acc2.y += w2 * x2.y; acc2.z += w2 * x2.z; acc2.w += w2 * x2.w;
      }
      float4* o2 = reinterpret_cast<float4*>(a.dk + ((long)b * S + s) * D + c * 4);
      if (a.dk_accumulate) { const float4 p = *o2; acc2.x += p.x; acc2.y += p.y; acc2.z += p.z; acc2.w += p.w; }
      *o2 = acc2;
    } else if (!any_drop) {
      for (int v = 0; v < 2 * T; ++v) {
        const float w = swt[v * 16 + r];
        const float4 x = *reinterpret_cast<const float4*>(&sv[(long)v * D + c * 4]);
        acc.x += w * x.x; acc.y += w * x.y; acc.z += w * x.z; acc.w += w * x.w;
      }
    } else {
      for (int t = 0; t < T; ++t) {          // per step: (alpha_t g_t + dl_t q_t) * mask_t
        const float w0 = swt[t * 16 + r], w1 = swt[(T + t) * 16 + r];
        const float4 x0 = *reinterpret_cast<const float4*>(&sv[(long)t * D + c * 4]);
        const float4 x1 = *reinterpret_cast<const float4*>(&sv[(long)(T + t) * D + c * 4]);
        float m[4] = {1.f, 1.f, 1.f, 1.f};
        if (a.drop_p[t] > 0.f) dropout_scale4(a.drop_seed[t], (a.drop_base ? *a.drop_base * 8ull : 0ull) + a.drop_off[t], (uint32_t)((((long)b * S + s) * D + c * 4) >> 2), a.drop_p[t], m);
        acc.x += (w0 * x0.x + w1 * x1.x) * m[0]; acc.y += (w0 * x0.y + w1 * x1.y) * m[1];
        acc.z += (w0 * x0.z + w1 * x1.z) * m[2]; acc.w += (w0 * x0.w + w1 * x1.w) * m[3];
      }
    }
    float4* o = reinterpret_cast<float4*>(out + c * 4);
    if (a.accumulate) { const float4 p = *o; acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
    *o = acc;
  }
}
#endif  // VLN_ATTN_FUSED_BODY_ONLY
