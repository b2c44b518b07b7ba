// Batched dot-product attention over short sequences (36 panoramic views, <=80 instruction tokens,
// <=16 candidates) of wide rows (2176 / 512 floats).  HBM-bound streaming of the [B,S,D] context:
//   attn_dot          : one WAVE per (b,s) row, 16-byte loads, wave-shuffle reduction
//   attn_softmax_wsum : grid (B, D-chunks); softmax over S recomputed per workgroup in one wave (S is tiny),
//                       the 4 waves split the S rows, lanes own 16 bytes of the D-chunk, LDS cross-wave sum
//   attn_bwd          : same geometry; softmax backward + d(query vector) and (text attention only) the
//                       in-place dctx accumulation
// Reference semantics: units.py:100-122 (SoftDotAttention), units.py:138-160 (VisualSoftDotAttention).
#include "vln_internal.h"
#include "../../include/vln_hip.h"

namespace vln {

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------------------
// vmul (nullable, [D]): the vector is multiplied column-wise by it before the dots (and before the write-back) -- ActionScoring's
// q = target (.) w_out (units.py:181-183); add0 (nullable, [1]): a scalar added to every dot (its b_out)
template <typename TC>
__global__ __launch_bounds__(256) void attn_dot_kernel(const TC* ctx, SlabVec vec, float* dots,
                                                       long rows, int S, int D, int vec_ok, float* vec_out, long ldvo,
                                                       const float* vmul, const float* add0) {
  constexpr int V = Elt<TC>::kVec;   // elements per 16-byte access
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float addv = add0 ? add0[0] : 0.f;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const int b = (int)(r / S);
    const TC* c = ctx + r * (long)D;
    const bool wb = vec_out && (r - (long)b * S) == 0;     // the row's first wave writes the summed vector back (nullable)
    float acc = 0.f;
    if (vec_ok) {
      for (int d = lane * V; d < D; d += 64 * V) {
        float x[V];
        Elt<TC>::ld16(c + d, x);
#pragma unroll
        for (int j = 0; j < V; j += 4) {
          float4 t = vec.at4(b, d + j);
          if (vmul) { const float4 m = *reinterpret_cast<const float4*>(vmul + d + j); t.x *= m.x; t.y *= m.y; t.z *= m.z; t.w *= m.w; }
          if (wb) *reinterpret_cast<float4*>(vec_out + (long)b * ldvo + d + j) = t;
          acc += dot4(&x[j], t);
        }
      }
    } else {
      for (int d = lane; d < D; d += 64) {
        float t = vec.at(b, d);
        if (vmul) t *= vmul[d];
        if (wb) vec_out[(long)b * ldvo + d] = t;
        acc += Elt<TC>::ld(c + d) * t;
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) dots[r] = acc + addv;
  }
}

int attn_dot(hipStream_t st, const void* ctx, int ctype, const float* vec, long ldv, float* dots, int B, int S,
             int D) {
  return attn_dot_sv(st, ctx, ctype, plain_vec(vec, ldv), dots, B, S, D);
}
int attn_dot_sv(hipStream_t st, const void* ctx, int ctype, SlabVec vec, float* dots, int B, int S, int D, float* vec_out, long ldvo,
                const float* vmul, const float* add0) {
  if (B <= 0 || S <= 0 || D <= 0) { set_error("attn_dot: bad dims"); return VLN_ERR_ARG; }
  const long ldv = vec.ld;
  long rows = (long)B * S;
  int blocks = (int)((rows + 3) / 4);
  if (blocks > 8192) blocks = 8192;
  const int V = (ctype == W_BF16) ? 8 : 4;
  int vec_ok = aligned16(ctx) && aligned16(vec.p) && (D % V == 0) && (ldv % 4 == 0) && (vec.stride % 4 == 0) && aligned16(vec.bias) &&
               (!vec_out || (aligned16(vec_out) && (ldvo % 4 == 0))) && aligned16(vmul);
  const double bytes = (double)rows * D * (ctype == W_BF16 ? 2 : 4) + 4.0 * B * D + 4.0 * rows;
  if (ctype == W_BF16)
    launch_timed(K_ATTN_DOT, bytes, attn_dot_kernel<bf16_raw>, dim3(blocks), dim3(256), 0, st, (const bf16_raw*)ctx, vec, dots, rows, S, D, vec_ok,
                 vec_out, ldvo, vmul, add0);
  else
    launch_timed(K_ATTN_DOT, bytes, attn_dot_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)ctx, vec, dots, rows, S, D, vec_ok, vec_out,
                 ldvo, vmul, add0);
  VLN_CHECK_LAUNCH("attn_dot");
  return VLN_OK;
}

// ---------------------------------------------------------------------------
constexpr int kMaxS = 512;   // longest attended sequence supported (reference: <= 80)

// softmax over S (<= kMaxS) with optional mask -> LDS, computed by wave 0
__device__ __forceinline__ void softmax_to_lds(const float* logits, const uint8_t* mask, int S, float* sw) {
  const int lane = threadIdx.x & 63;
  if ((threadIdx.x >> 6) == 0) {
    float mx = -INFINITY;
    for (int s = lane; s < S; s += 64) {
      float v = (mask && mask[s]) ? -INFINITY : logits[s];
      sw[s] = v;
      mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int s = lane; s < S; s += 64) {
      float e = __expf(sw[s] - mx);
      sw[s] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int s = lane; s < S; s += 64) sw[s] *= inv;
  }
}

template <typename TC, bool kSoftmax>
__global__ __launch_bounds__(256) void attn_wsum_kernel(const TC* ctx, const float* logits, const uint8_t* mask,
                                                        float* attn, float* out, long ldo, int S, int D,
                                                        int vec_ok) {
  constexpr int V = Elt<TC>::kVec;
  constexpr int DC = 64 * V;
  __shared__ float sw[kMaxS];
  __shared__ float red[4][DC];
  const int b = blockIdx.x, chunk = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* lg = logits + (long)b * S;
  if constexpr (kSoftmax) {
    softmax_to_lds(lg, mask ? mask + (long)b * S : nullptr, S, sw);
  } else {
    for (int s = threadIdx.x; s < S; s += 256) sw[s] = lg[s];
  }
  __syncthreads();
  if (kSoftmax && attn && chunk == 0)
    for (int s = threadIdx.x; s < S; s += 256) attn[(long)b * S + s] = sw[s];

  const int d0 = chunk * DC + lane * V;
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  const TC* base = ctx + (long)b * S * D;
  if (vec_ok) {
    if (d0 < D) {
      for (int s = wave; s < S; s += 4) {
        const float w = sw[s];
        const TC* p = base + (long)s * D + d0;
        float x[V];
        Elt<TC>::ld16(p, x);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += w * x[j];
      }
    }
  } else {
    for (int s = wave; s < S; s += 4) {
      const float w = sw[s];
#pragma unroll
      for (int j = 0; j < V; ++j)
        if (d0 + j < D) acc[j] += w * Elt<TC>::ld(base + (long)s * D + d0 + j);
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) red[wave][lane * V + j] = acc[j];
  __syncthreads();
  for (int i = threadIdx.x; i < DC; i += 256) {
    const int d = chunk * DC + i;
    if (d < D) out[(long)b * ldo + d] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
  }
}

static int launch_wsum(hipStream_t st, const void* ctx, int ctype, const float* logits, const uint8_t* mask,
                       float* attn, float* out, long ldo, int B, int S, int D, bool softmax) {
  if (B <= 0 || S <= 0 || D <= 0 || S > kMaxS) { set_error("attn wsum: bad dims B=%d S=%d D=%d", B, S, D); return VLN_ERR_ARG; }
  const int V = (ctype == W_BF16) ? 8 : 4;
  const int DC = 64 * V;
  int vec_ok = aligned16(ctx) && (D % V == 0);
  dim3 grid(B, (D + DC - 1) / DC), block(256);
  const double bytes = (double)B * S * D * (ctype == W_BF16 ? 2 : 4) + 4.0 * B * D + 8.0 * B * S;
  if (ctype == W_BF16) {
    if (softmax) launch_timed(K_ATTN_WSUM, bytes, attn_wsum_kernel<bf16_raw, true>, grid, block, 0, st, (const bf16_raw*)ctx, logits, mask, attn, out, ldo, S, D, vec_ok);
    else launch_timed(K_ATTN_WSUM, bytes, attn_wsum_kernel<bf16_raw, false>, grid, block, 0, st, (const bf16_raw*)ctx, logits, mask, attn, out, ldo, S, D, vec_ok);
  } else {
    if (softmax) launch_timed(K_ATTN_WSUM, bytes, attn_wsum_kernel<float, true>, grid, block, 0, st, (const float*)ctx, logits, mask, attn, out, ldo, S, D, vec_ok);
    else launch_timed(K_ATTN_WSUM, bytes, attn_wsum_kernel<float, false>, grid, block, 0, st, (const float*)ctx, logits, mask, attn, out, ldo, S, D, vec_ok);
  }
  VLN_CHECK_LAUNCH("attn_wsum");
  return VLN_OK;
}

int attn_softmax_wsum(hipStream_t st, const void* ctx, int ctype, const float* logits, const uint8_t* mask,
                      float* attn, float* out, long ldo, int B, int S, int D) {
  return launch_wsum(st, ctx, ctype, logits, mask, attn, out, ldo, B, S, D, true);
}
int rows_wsum(hipStream_t st, const void* ctx, int ctype, const float* w, float* out, long ldo, int B, int S,
              int D) {
  return launch_wsum(st, ctx, ctype, w, nullptr, nullptr, out, ldo, B, S, D, false);
}

// The candidate-logit branch of a WHOLE rollout's backward: out_t[b,:] = sum_c w_t[b,c] ctx_t[b,c,:] for every step t in one
// launch (steps differ in S = candidate count).  Same arithmetic and summation order as attn_wsum_kernel<TC,false>.
struct WsumMulti {
  const void* ctx[VLN_CE_MAX_STEPS]; const float* w[VLN_CE_MAX_STEPS]; float* out[VLN_CE_MAX_STEPS]; int S[VLN_CE_MAX_STEPS];
  const float* probs[VLN_CE_MAX_STEPS]; const long long* target[VLN_CE_MAX_STEPS];
  int T, B, D; long ldo; float ce_scale; const float* ce_dloss; long ignore_index;
};
template <typename TC>
__global__ __launch_bounds__(256) void rows_wsum_multi_kernel(WsumMulti m, int vec_ok) {
  constexpr int V = Elt<TC>::kVec;
  constexpr int DC = 64 * V;
  __shared__ float sw[kMaxS];
  __shared__ float red[4][DC];
  const int t = blockIdx.x / m.B, b = blockIdx.x - t * m.B, chunk = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = m.S[t], D = m.D;
  if (m.w[t]) {
    const float* lg = m.w[t] + (long)b * S;
    for (int s = threadIdx.x; s < S; s += 256) sw[s] = lg[s];
  } else {                                   // d logits of the rollout's cross-entropy, formed here (masked_ce_multi_bwd_kernel's formula)
    const long tg = m.target[t][b];
    const float g = m.ce_dloss[0] * m.ce_scale;
    const float* pr = m.probs[t] + (long)b * S;
    for (int s = threadIdx.x; s < S; s += 256) sw[s] = (tg == m.ignore_index) ? 0.f : g * (pr[s] - (s == tg ? 1.f : 0.f));
  }
  __syncthreads();
  const int d0 = chunk * DC + lane * V;
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  const TC* base = reinterpret_cast<const TC*>(m.ctx[t]) + (long)b * S * D;
  if (vec_ok) {
    if (d0 < D) {
      for (int s = wave; s < S; s += 4) {
        const float w = sw[s];
        float x[V];
        Elt<TC>::ld16(base + (long)s * D + d0, x);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += w * x[j];
      }
    }
  } else {
    for (int s = wave; s < S; s += 4) {
      const float w = sw[s];
#pragma unroll
      for (int j = 0; j < V; ++j)
        if (d0 + j < D) acc[j] += w * Elt<TC>::ld(base + (long)s * D + d0 + j);
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) red[wave][lane * V + j] = acc[j];
  __syncthreads();
  float* out = m.out[t];
  for (int i = threadIdx.x; i < DC; i += 256) {
    const int d = chunk * DC + i;
    if (d < D) out[(long)b * m.ldo + d] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
  }
}
int rows_wsum_multi(hipStream_t st, const vln_wsum_step* steps, int T, int ctype, int B, int D, long ldo, float ce_scale,
                    const float* ce_dloss, long ignore_index) {
  if (!steps || T <= 0 || T > VLN_CE_MAX_STEPS || B <= 0 || D <= 0) { set_error("rows_wsum_multi: bad dims"); return VLN_ERR_ARG; }
  WsumMulti m{};
  m.T = T; m.B = B; m.D = D; m.ldo = ldo; m.ce_scale = ce_scale; m.ce_dloss = ce_dloss; m.ignore_index = ignore_index;
  const int V = (ctype == W_BF16) ? 8 : 4;
  int vec_ok = (D % V == 0);
  for (int t = 0; t < T; ++t) {
    const bool ce = !steps[t].w && steps[t].probs && steps[t].target && ce_dloss;
    if (!steps[t].ctx || (!steps[t].w && !ce) || !steps[t].out || steps[t].S <= 0 || steps[t].S > kMaxS) { set_error("rows_wsum_multi: bad step %d", t); return VLN_ERR_ARG; }
    m.ctx[t] = steps[t].ctx; m.w[t] = steps[t].w; m.out[t] = steps[t].out; m.S[t] = steps[t].S;
    m.probs[t] = steps[t].probs; m.target[t] = (const long long*)steps[t].target;
    vec_ok = vec_ok && aligned16(steps[t].ctx);
  }
  const int DC = 64 * V;
  dim3 grid(T * B, (D + DC - 1) / DC), block(256);
  if (ctype == W_BF16) VLN_LAUNCH(rows_wsum_multi_kernel<bf16_raw>, grid, block, 0, st, m, vec_ok);
  else VLN_LAUNCH(rows_wsum_multi_kernel<float>, grid, block, 0, st, m, vec_ok);
  VLN_CHECK_LAUNCH("rows_wsum_multi");
  return VLN_OK;
}

// The candidate logits of a WHOLE rollout in one launch: dots_t[b,s] = ctx_t[b,s,:] . vec_t[b,:] (steps differ in S).  One wave
// per (step, episode, candidate) row; same arithmetic as attn_dot_kernel with a one-slab vector.
struct DotMulti {
  const void* ctx[VLN_CE_MAX_STEPS]; const float* vec[VLN_CE_MAX_STEPS]; float* dots[VLN_CE_MAX_STEPS]; int S[VLN_CE_MAX_STEPS];
  int row0[VLN_CE_MAX_STEPS + 1];      // prefix sums of B * S_t
  int T, B, D; long ldv;
};
template <typename TC>
__global__ __launch_bounds__(256) void attn_dot_multi_kernel(DotMulti m, int vec_ok) {
  constexpr int V = Elt<TC>::kVec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = m.D, total = m.row0[m.T];
  for (int r = (int)blockIdx.x * 4 + wave; r < total; r += (int)gridDim.x * 4) {
    int t = 0;
    while (t + 1 < m.T && r >= m.row0[t + 1]) ++t;
    const int q = r - m.row0[t], b = q / m.S[t];
    const TC* c = reinterpret_cast<const TC*>(m.ctx[t]) + (long)q * D;
    const float* v = m.vec[t] + (long)b * m.ldv;
    float acc = 0.f;
    if (vec_ok) {
      for (int d = lane * V; d < D; d += 64 * V) {
        float x[V];
        Elt<TC>::ld16(c + d, x);
#pragma unroll
        for (int j = 0; j < V; j += 4) {
          const float4 w = *reinterpret_cast<const float4*>(v + d + j);
          acc += dot4(&x[j], w);
        }
      }
    } else {
      for (int d = lane; d < D; d += 64) acc += Elt<TC>::ld(c + d) * v[d];
    }
    acc = wave_sum(acc);
    if (lane == 0) m.dots[t][q] = acc;
  }
}
int attn_dot_multi(hipStream_t st, const vln_dot_step* steps, int T, int ctype, int B, int D, long ldv) {
  if (!steps || T <= 0 || T > VLN_CE_MAX_STEPS || B <= 0 || D <= 0) { set_error("attn_dot_multi: bad dims"); return VLN_ERR_ARG; }
  DotMulti m{};
  m.T = T; m.B = B; m.D = D; m.ldv = ldv;
  const int V = (ctype == W_BF16) ? 8 : 4;
  int vec_ok = (D % V == 0) && (ldv % 4 == 0);
  int rows = 0;
  for (int t = 0; t < T; ++t) {
    if (!steps[t].ctx || !steps[t].vec || !steps[t].dots || steps[t].S <= 0) { set_error("attn_dot_multi: bad step %d", t); return VLN_ERR_ARG; }
    m.ctx[t] = steps[t].ctx; m.vec[t] = steps[t].vec; m.dots[t] = steps[t].dots; m.S[t] = steps[t].S;
    m.row0[t] = rows; rows += B * steps[t].S;
    vec_ok = vec_ok && aligned16(steps[t].ctx) && aligned16(steps[t].vec);
  }
  m.row0[T] = rows;
  int blocks = (rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  if (ctype == W_BF16) VLN_LAUNCH(attn_dot_multi_kernel<bf16_raw>, dim3(blocks), dim3(256), 0, st, m, vec_ok);
  else VLN_LAUNCH(attn_dot_multi_kernel<float>, dim3(blocks), dim3(256), 0, st, m, vec_ok);
  VLN_CHECK_LAUNCH("attn_dot_multi");
  return VLN_OK;
}

// ---------------------------------------------------------------------------
// The candidate logits of one step AND the sampled-action branch on them in one launch (policy.py:199-206 + envdrop.py:173,
// 186-195): logits[b,c] = cand[b,c,:] . q[b,:]; probs = softmax(mask(logits)); a ~ Categorical(probs) (the kernels' Philox
// stream, or a given action); log pi(a), the entropy (torch.distributions' clamp).  One workgroup per episode: the four waves
// split the candidate rows (16-byte loads, wave-shuffle reduction), wave 0 then does categorical_fwd_kernel's row (pointwise.hip:
// same arithmetic).  The action also goes straight to a host-mapped word when the caller gives one: the host that steps the
// simulator polls it -- no copy launch behind the draw.  A sampled rollout had three launches here (dot, draw, D2H copy).
// ---------------------------------------------------------------------------
struct CandSample {
  const void* cand; SlabVec q; float* logits;
  const unsigned char* mask; const long long* action_in; long long* action_out; long long* action_host;
  float* probs; float* logp; float* ent;
  uint64_t seed, offset; const unsigned long long* offset_base_dev;
  int B, C, D, vec_ok;
};
// 16 waves per workgroup (round 6): up to 16 candidate rows in flight at once, the query summed from its split-K slabs ONCE into
// LDS (every row's dot re-read the slabs: C x nsplit loads of the same values on the step's dependent chain).  Same per-lane
// accumulation order as the four-wave form: identical logits.
constexpr int kCandWaves = 16;
constexpr int kCandQMax = 4096;             // query columns the LDS copy holds (wider: the slabs are read in place)
template <typename TC>
__global__ __launch_bounds__(kCandWaves * 64) void cand_sample_kernel(CandSample a) {
  constexpr int V = Elt<TC>::kVec;
  __shared__ float sdot[64];
  __shared__ __attribute__((aligned(16))) float sq[kCandQMax];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = a.C, D = a.D;
  const TC* base = reinterpret_cast<const TC*>(a.cand) + (long)b * C * D;
  const bool staged = a.vec_ok && D <= kCandQMax;
  // the rows' first loads go out before the query is staged
  float x0[V];
  const bool own = wave < C && a.vec_ok && lane * V < D;
  if (own) Elt<TC>::ld16(base + (long)wave * D + lane * V, x0);
  if (staged) {
    for (int d = threadIdx.x * 4; d < D; d += kCandWaves * 64 * 4) *reinterpret_cast<float4*>(&sq[d]) = a.q.at4(b, d);
    __syncthreads();
  }
  for (int c = wave; c < C; c += kCandWaves) {
    const TC* row = base + (long)c * D;
    float acc = 0.f;
    if (a.vec_ok) {
      for (int d = lane * V; d < D; d += 64 * V) {
        float x[V];
        if (c == wave && d == lane * V) {
#pragma unroll
          for (int j = 0; j < V; ++j) x[j] = x0[j];
        } else {
          Elt<TC>::ld16(row + d, x);
        }
#pragma unroll
        for (int j = 0; j < V; j += 4) {
          const float4 t = staged ? *reinterpret_cast<const float4*>(&sq[d + j]) : a.q.at4(b, d + j);
          acc += dot4(&x[j], t);
        }
      }
    } else {
      for (int d = lane; d < D; d += 64) acc += Elt<TC>::ld(row + d) * a.q.at(b, d);
    }
    acc = wave_sum(acc);
    if (lane == 0) sdot[c] = acc;
  }
  __syncthreads();
  if (wave != 0) return;
  uint64_t offset = a.offset;
  if (a.offset_base_dev) offset += *a.offset_base_dev * 8ull;
  const float eps = 1.1920928955078125e-07f;
  const bool in = lane < C;
  const float raw = in ? sdot[lane] : 0.f;
  if (in) a.logits[(long)b * C + lane] = raw;
  const bool masked = in && a.mask && a.mask[(long)b * C + lane];
  const float l = (in && !masked) ? raw : -INFINITY;
  const float mx = wave_max(l);
  const float e = (in && !masked) ? __expf(l - mx) : 0.f;
  const float inv = 1.f / wave_sum(e);
  const float pc = e * inv;
  long av;
  if (a.action_in) av = a.action_in[b];
  else {
    const Philox4 r = philox4x32_10(a.seed, offset, (uint32_t)b);
    const float u = (float)(r.x >> 8) * (1.0f / 16777216.0f);
    float cum = pc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float t = __shfl_up(cum, o, 64);
      if (lane >= o) cum += t;
    }
    const unsigned long long hit = __ballot(in && u < cum), live = __ballot(in && pc > 0.f);
    av = hit ? (long)(__ffsll((long long)hit) - 1) : (live ? (long)(63 - __clzll((long long)live)) : 0);
  }
  const float lc = __logf(fminf(fmaxf(pc, eps), 1.f - eps));
  const float H = -wave_sum(in ? pc * lc : 0.f);
  const float la = (av >= 0 && av < C) ? __shfl(lc, (int)av, 64) : 0.f;
  if (in) a.probs[(long)b * C + lane] = pc;
  if (lane == 0) {
    if (a.action_out) a.action_out[b] = av;
    if (a.action_host) __hip_atomic_store(a.action_host + b, (long long)av, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    a.logp[b] = la;
    a.ent[b] = H;
  }
}
int cand_logits_sample(hipStream_t st, const void* cand, int ctype, SlabVec q, float* logits, const uint8_t* mask, const int64_t* action_in,
                       int64_t* action_out, int64_t* action_host, float* probs, float* logp, float* ent, uint64_t seed, uint64_t offset,
                       const uint64_t* offset_base_dev, int B, int C, int D) {
  if (!cand || !q.p || !logits || !probs || !logp || !ent || B <= 0 || C <= 0 || C > 64 || D <= 0 || (!action_in && !action_out)) {
    set_error("cand_logits_sample: bad args (at most 64 candidates)");
    return VLN_ERR_ARG;
  }
  const int V = (ctype == W_BF16) ? 8 : 4;
  CandSample a{cand, q, logits, mask, (const long long*)action_in, (long long*)action_out, (long long*)action_host, probs, logp, ent, seed, offset,
               reinterpret_cast<const unsigned long long*>(offset_base_dev), B, C, D,
               (aligned16(cand) && aligned16(q.p) && (D % V == 0) && (q.ld % 4 == 0) && (q.stride % 4 == 0)) ? 1 : 0};
  const double bytes = (double)B * C * D * (ctype == W_BF16 ? 2 : 4) + 4.0 * B * D + 12.0 * B * C;
  if (ctype == W_BF16) launch_timed(K_ATTN_DOT, bytes, cand_sample_kernel<bf16_raw>, dim3(B), dim3(kCandWaves * 64), 0, st, a);
  else launch_timed(K_ATTN_DOT, bytes, cand_sample_kernel<float>, dim3(B), dim3(kCandWaves * 64), 0, st, a);
  VLN_CHECK_LAUNCH("cand_logits_sample");
  return VLN_OK;
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
template <typename TC>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const TC* ctx, const float* attn, const float* dalpha,
                                                       const float* dattn_ext, const float* dwc, long lddwc,
                                                       const float* vec, long ldvec, float* dvec, long lddvec,
                                                       float* dctx, float* dl_out, int S, int D, int vec_ok) {
  constexpr int V = Elt<TC>::kVec;
  constexpr int DC = 64 * V;
  __shared__ float sa[kMaxS];   // attn
  __shared__ float sl[kMaxS];   // dlogits
  __shared__ float red[4][DC];
  const int b = blockIdx.x, chunk = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave == 0) {
    float dot = 0.f;
    for (int s = lane; s < S; s += 64) {
      const float a = attn[(long)b * S + s];
      float g = dalpha ? dalpha[(long)b * S + s] : 0.f;
      if (dattn_ext) g += dattn_ext[(long)b * S + s];
      sa[s] = a;
      sl[s] = g;
      dot += a * g;
    }
    dot = wave_sum(dot);
    for (int s = lane; s < S; s += 64) {
      const float v = sa[s] * (sl[s] - dot);
      sl[s] = v;
      if (dl_out && chunk == 0) dl_out[(long)b * S + s] = v;
    }
  }
  __syncthreads();
  const int d0 = chunk * DC + lane * V;
  float acc[V], gw[V], qv[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    acc[j] = 0.f;
    gw[j] = (dctx && d0 + j < D) ? dwc[(long)b * lddwc + d0 + j] : 0.f;
    qv[j] = (dctx && d0 + j < D) ? vec[(long)b * ldvec + d0 + j] : 0.f;
  }
  const TC* base = ctx + (long)b * S * D;
  float* dbase = dctx ? dctx + (long)b * S * D : nullptr;
  if (vec_ok) {
    if (d0 < D) {
      for (int s = wave; s < S; s += 4) {
        const float dl = sl[s], a = sa[s];
        const TC* p = base + (long)s * D + d0;
        float x[V];
        Elt<TC>::ld16(p, x);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += dl * x[j];
        if (dbase) {
          float* q = dbase + (long)s * D + d0;
#pragma unroll
          for (int j = 0; j < V; j += 4) {
            float4 t = *reinterpret_cast<float4*>(q + j);
            t.x += a * gw[j] + dl * qv[j];
            t.y += a * gw[j + 1] + dl * qv[j + 1];
            t.z += a * gw[j + 2] + dl * qv[j + 2];
            t.w += a * gw[j + 3] + dl * qv[j + 3];
            *reinterpret_cast<float4*>(q + j) = t;
          }
        }
      }
    }
  } else {
    for (int s = wave; s < S; s += 4) {
      const float dl = sl[s], a = sa[s];
#pragma unroll
      for (int j = 0; j < V; ++j) {
        if (d0 + j < D) {
          acc[j] += dl * Elt<TC>::ld(base + (long)s * D + d0 + j);
          if (dbase) dbase[(long)s * D + d0 + j] += a * gw[j] + dl * qv[j];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) red[wave][lane * V + j] = acc[j];
  __syncthreads();
  if (dvec)
    for (int i = threadIdx.x; i < DC; i += 256) {
      const int d = chunk * DC + i;
      if (d < D) dvec[(long)b * lddvec + d] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    }
}

int attn_bwd(hipStream_t st, const void* ctx, int ctype, const float* attn, const float* dalpha,
             const float* dattn_ext, const float* dwc, long lddwc, const float* vec, long ldvec, float* dvec,
             long lddvec, float* dctx, float* dl_out, int B, int S, int D) {
  if (B <= 0 || S <= 0 || D <= 0 || S > kMaxS) { set_error("attn_bwd: bad dims"); return VLN_ERR_ARG; }
  if (dctx && (!dwc || !vec)) { set_error("attn_bwd: dctx needs dwc and vec"); return VLN_ERR_ARG; }
  const int V = (ctype == W_BF16) ? 8 : 4;
  const int DC = 64 * V;
  int vec_ok = aligned16(ctx) && (D % V == 0) && (!dctx || aligned16(dctx));
  dim3 grid(B, (D + DC - 1) / DC), block(256);
  const double bytes = (double)B * S * D * (ctype == W_BF16 ? 2 : 4) + (dctx ? 8.0 * B * S * D : 0.0) +
                       4.0 * B * D * (dctx ? 3 : 1) + 12.0 * B * S;
  if (ctype == W_BF16)
    launch_timed(K_ATTN_BWD, bytes, attn_bwd_kernel<bf16_raw>, grid, block, 0, st, (const bf16_raw*)ctx, attn, dalpha, dattn_ext, dwc, lddwc, vec, ldvec, dvec, lddvec, dctx, dl_out, S, D, vec_ok);
  else
    launch_timed(K_ATTN_BWD, bytes, attn_bwd_kernel<float>, grid, block, 0, st, (const float*)ctx, attn, dalpha, dattn_ext, dwc, lddwc, vec, ldvec, dvec, lddvec, dctx, dl_out, S, D, vec_ok);
  VLN_CHECK_LAUNCH("attn_bwd");
  return VLN_OK;
}


#include "attention_fused.h"
#include "attention_split.h"
#include "attention_textk.h"

// dots + softmax + weighted sum in one launch when a register-resident configuration fits, else the two-kernel path
// (`dots_scratch` [B,S] is only touched by the fallback).
int attn_fwd_rows(hipStream_t st, const void* ctx, int ctype, const float* vec, long ldv, const uint8_t* mask, float* attn,
                  float* out, long ldo, float* dots_scratch, int B, int S, int D) {
  return attn_fwd_rows_sv(st, ctx, ctype, plain_vec(vec, ldv), nullptr, 0, mask, attn, out, ldo, dots_scratch, B, S, D);
}
// `vec` may still lie in split-K slabs; vec_out (nullable) receives the summed vector
int attn_fwd_rows_sv(hipStream_t st, const void* ctx, int ctype, SlabVec vec, float* vec_out, long ldvo, const uint8_t* mask,
                     float* attn, float* out, long ldo, float* dots_scratch, int B, int S, int D, void* sync, long sync_bytes) {
  if (B <= 0 || S <= 0 || D <= 0 || S > kMaxS) { set_error("attn_fwd_rows: bad dims B=%d S=%d D=%d", B, S, D); return VLN_ERR_ARG; }
  AttnFusedArgs a{ctx, vec, vec_out, ldvo, mask, attn, nullptr, nullptr, out, ldo, S, D};
  if (attn_split_try(st, ctype, a, B, false, sync, sync_bytes, device_cus())) { VLN_CHECK_LAUNCH("attn_split_fwd"); return VLN_OK; }
  if (attn_fused_try(st, ctype, a, B, false)) { VLN_CHECK_LAUNCH("attn_fused_fwd"); return VLN_OK; }
  if (!dots_scratch) { set_error("attn_fwd_rows: shape needs the two-kernel path and no scratch was given"); return VLN_ERR_ARG; }
  if (vec_out) {                        // the caller wants the finished vector: sum the slabs into it first
    int r0 = reduce_epilogue(st, vec.p, vec.n, vec.stride, vec.ld, vec_out, ldvo, B, D, vec.bias, ACT_NONE, nullptr, 0, DropSpec{0, 0, 0.f});
    if (r0 != VLN_OK) return r0;
    vec = plain_vec(vec_out, ldvo);
  }
  int r = attn_dot_sv(st, ctx, ctype, vec, dots_scratch, B, S, D);
  if (r != VLN_OK) return r;
  return attn_softmax_wsum(st, ctx, ctype, dots_scratch, mask, attn, out, ldo, B, S, D);
}

// d alpha = ctx . dwc, softmax backward, d query = sum_s dl ctx; dl_out (nullable) feeds attn_dctx_deferred
int attn_bwd_rows(hipStream_t st, const void* ctx, int ctype, const float* attn, const float* dwc, long lddwc,
                  const float* dattn_ext, float* dvec, long lddvec, float* dl_out, float* dots_scratch, int B, int S, int D) {
  return attn_bwd_rows_sv(st, ctx, ctype, attn, plain_vec(dwc, lddwc), nullptr, 0, dattn_ext, dvec, lddvec, dl_out, dots_scratch, B, S, D);
}
int attn_bwd_rows_sv(hipStream_t st, const void* ctx, int ctype, const float* attn, SlabVec dwc, float* dwc_out, long lddo,
                     const float* dattn_ext, float* dvec, long lddvec, float* dl_out, float* dots_scratch, int B, int S, int D,
                     void* sync, long sync_bytes) {
  if (B <= 0 || S <= 0 || D <= 0 || S > kMaxS) { set_error("attn_bwd_rows: bad dims"); return VLN_ERR_ARG; }
  AttnFusedArgs a{ctx, dwc, dwc_out, lddo, nullptr, const_cast<float*>(attn), dattn_ext, dl_out, dvec, lddvec, S, D};
  if (attn_split_try(st, ctype, a, B, true, sync, sync_bytes, device_cus())) { VLN_CHECK_LAUNCH("attn_split_bwd"); return VLN_OK; }
  if (attn_fused_try(st, ctype, a, B, true)) { VLN_CHECK_LAUNCH("attn_fused_bwd"); return VLN_OK; }
  if (!dots_scratch) { set_error("attn_bwd_rows: shape needs the two-kernel path and no scratch was given"); return VLN_ERR_ARG; }
  if (dwc_out) {
    int r0 = reduce_epilogue(st, dwc.p, dwc.n, dwc.stride, dwc.ld, dwc_out, lddo, B, D, dwc.bias, ACT_NONE, nullptr, 0, DropSpec{0, 0, 0.f});
    if (r0 != VLN_OK) return r0;
    dwc = plain_vec(dwc_out, lddo);
  }
  int r = attn_dot_sv(st, ctx, ctype, dwc, dots_scratch, B, S, D);
  if (r != VLN_OK) return r;
  return attn_bwd(st, ctx, ctype, attn, dots_scratch, dattn_ext, nullptr, 0, nullptr, 0, dvec, lddvec, nullptr, dl_out, B, S, D);
}

int attn_dctx_deferred(hipStream_t st, const float* const* alpha, const float* const* dl, const float* const* g, long ldg,
                       const float* const* q, long ldq, int T, float* dctx, int B, int S, int D, int accumulate,
                       const uint64_t* drop_seed, const uint64_t* drop_off, const float* drop_p, float* dk) {
  if (T <= 0 || B <= 0 || S <= 0 || D <= 0 || (D & 3) || !aligned16(dctx) || (dk && (!aligned16(dk) || drop_p))) {
    set_error("attn_dctx_deferred: bad args (T=%d B=%d S=%d D=%d)", T, B, S, D);
    return VLN_ERR_ARG;
  }
  int tmax = (int)((60 * 1024) / (2 * (D + 16) * sizeof(float)));
  if (tmax > kDctxMaxSteps) tmax = kDctxMaxSteps;
  if (tmax < 1) { set_error("attn_dctx_deferred: D=%d too wide", D); return VLN_ERR_ARG; }
  for (int t0 = 0; t0 < T; t0 += tmax) {
    DctxArgs a{};
    a.T = (T - t0 < tmax) ? T - t0 : tmax;
    a.vec_ok = ((ldg & 3) == 0 && (ldq & 3) == 0) ? 1 : 0;
    for (int t = 0; t < a.T; ++t) {
      a.alpha[t] = alpha[t0 + t]; a.dl[t] = dl[t0 + t]; a.g[t] = g[t0 + t]; a.q[t] = q[t0 + t];
      // a step carries an (alpha, g) pair, a (dl, q) pair or both
      if ((!a.alpha[t]) != (!a.g[t]) || (!a.dl[t]) != (!a.q[t]) || (!a.alpha[t] && !a.dl[t])) { set_error("attn_dctx_deferred: null step pointer"); return VLN_ERR_ARG; }
      if ((a.g[t] && !aligned16(a.g[t])) || (a.q[t] && !aligned16(a.q[t]))) a.vec_ok = 0;
      a.drop_seed[t] = drop_seed ? drop_seed[t0 + t] : 0; a.drop_off[t] = drop_off ? drop_off[t0 + t] : 0;
      a.drop_p[t] = drop_p ? drop_p[t0 + t] : 0.f;
    }
    a.drop_base = drop_base_tls();
    a.ldg = ldg; a.ldq = ldq; a.dctx = dctx; a.dk = dk; a.S = S; a.D = D; a.accumulate = (accumulate || t0 > 0) ? 1 : 0; a.dk_accumulate = t0 > 0 ? 1 : 0;
    const unsigned lds = (unsigned)(2 * a.T * (D + 16) * sizeof(float));
    VLN_LAUNCH(attn_dctx_deferred_kernel, dim3(B, (S + 15) / 16), dim3(256), lds, st, a);
    VLN_CHECK_LAUNCH("attn_dctx_deferred");
  }
  return VLN_OK;
}

}  // namespace vln

extern "C" int64_t vln_attn_sync_bytes(int B) { return B > 0 ? vln::attn_split_sync_bytes(B) : -1; }
extern "C" int vln_attn_textk_ok(int ctype, int B, int S, int D, const void* sync, int64_t sync_bytes) {
  return vln::attn_textk_ok(ctype, B, S, D, sync, (long)sync_bytes) ? 1 : 0;
}
