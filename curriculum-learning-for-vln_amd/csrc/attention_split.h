// One attention per launch on FOUR workgroups per batch row (included by attention.hip inside namespace vln).
//
// attn_fused_kernel gives a batch row's whole [S, D] block to ONE workgroup: at B = 64 that is 64 of 256 CUs, each pulling
// 80-157 KB through one CU's memory path (~30 GB/s for bytes another kernel just wrote): 5 of the launch's 8-9 us.  Here the
// block's COLUMNS are split over kSplitNS = 4 workgroups (256 workgroups at B = 64, 20-39 KB each).  The weighted sum needs
// no exchange (every workgroup produces its own columns of the output); the row dots do: each workgroup computes the dots of
// its column slice and the four partials of a row are exchanged ONCE, as data-tagged granules (MI355X_MICROARCH.md
// "handoff-1to1": one 16-byte write-through store = two {fp32 value, tag} granules; the reader polls the bytes themselves,
// no flag, no fence), and summed by every workgroup in the same fixed order (part 0 + 1 + 2 + 3: deterministic, and all four
// workgroups derive bit-identical softmax weights).
//
// Tags: seq[b] + 1, where seq[b] is a per-batch-row launch count in the caller's sync buffer (zero-initialised once, never
// cleared: 32-bit tags only grow).  Part 0 bumps seq[b] after its sweep has seen all four parts' granules -- by then every
// part has read seq[b] -- and the next launch on the stream reads the new value (kernel boundary).  No host-side sequence:
// the launch arguments repeat, so the launch can be captured in a whole-iteration graph.
// The four workgroups of a row must be co-resident: the host takes this path only when B * 4 <= the device's CU count;
// spins are bounded and a timeout raises the sticky word of vln_persistent_check (encoder.hip).
#pragma once

constexpr int kSplitNS = 4;
constexpr int kSplitSMax = 128;                       // rows per block the exchange has room for (kMaxS)
constexpr unsigned kSplitSpinLimit = 1u << 20;

struct AttnSplitSync {
  unsigned* seq;           // [B]
  unsigned char* gran;     // [B][kSplitNS][kSplitSMax / 2] x 16 bytes
  unsigned* sticky;        // host-mapped timeout counter
};
__host__ __device__ inline long attn_split_sync_bytes(int B) { return (long)B * 64 + (long)B * kSplitNS * (kSplitSMax / 2) * 16; }

// The exchange of a row's partial dots between its kSplitNS workgroups (512 threads each; S <= kSplitSMax).  In: sdots[s] = this
// part's partial dot of row s (written, visible after the caller's __syncthreads).  Out: sdots[s] = part 0 + 1 + 2 + 3 in that order
// (identical bits in all four workgroups), visible to every thread on return.
__device__ __forceinline__ void attn_split_exchange(const AttnSplitSync& sy, int b, int part, int B, int S, unsigned tag, float* sdots,
                                                    float (*spart)[kSplitSMax], int* s_abort) {
  const int lane = threadIdx.x & 63;
  // publish this part's partials: thread i < ceil(S / 2) stores {dot[2i], tag, dot[2i + 1], tag}, one 16-byte write-through store
  const int nh = (S + 1) >> 1;
  __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(sy.gran, 0, (unsigned)((long)B * kSplitNS * (kSplitSMax / 2) * 16), 0x00020000);
  const unsigned rbase = (unsigned)(b * kSplitNS) * (unsigned)(kSplitSMax / 2) * 16u;
  if ((int)threadIdx.x < nh) {
    const int s0 = 2 * threadIdx.x;
    const float v0 = sdots[s0], v1 = (s0 + 1 < S) ? sdots[s0 + 1] : 0.f;
    const u32x4_t o = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
    __builtin_amdgcn_raw_buffer_store_b128(o, xres, rbase + (unsigned)(part * (kSplitSMax / 2) + (int)threadIdx.x) * 16u, 0, 16);   // sc1
  }
  // sweep all four parts' partials (thread -> part q = tid / 64, pair i = tid % 64; S <= 128) until the tags match
  float p0 = 0.f, p1 = 0.f;
  {
    const int q = threadIdx.x >> 6, i = threadIdx.x & 63;
    const bool mine = q < kSplitNS && i < nh;
    unsigned spins = 0;
    for (;;) {
      bool ok = true;
      if (mine) {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(xres, rbase + (unsigned)(q * (kSplitSMax / 2) + i) * 16u, 0, 16);
        ok = (v.y == tag) && (v.w == tag);
        p0 = __uint_as_float(v.x); p1 = __uint_as_float(v.z);
      }
      if (__all(ok) || *(volatile int*)s_abort) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kSplitSpinLimit) {               // a sibling workgroup is not resident / died: report, then drain
        if (lane == 0) { __hip_atomic_fetch_add(sy.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); *s_abort = 1; }
        break;
      }
    }
  }
  __syncthreads();                                   // sdots has been read by the publishers: reuse it for the exchange
  {
    const int q = threadIdx.x >> 6, i = threadIdx.x & 63;
    if (q < kSplitNS && i < nh) { spart[q][2 * i] = p0; spart[q][2 * i + 1] = p1; }
  }
  __syncthreads();
  if (part == 0 && threadIdx.x == 0) VLN_AGENT_STORE(sy.seq + b, tag);     // every part has read seq[b] (its granules carry the tag)
  for (int s = threadIdx.x; s < S; s += 8 * 64) sdots[s] = ((spart[0][s] + spart[1][s]) + spart[2][s]) + spart[3][s];
  __syncthreads();
}

// Geometry: 8 waves; one wave instruction covers RPI = 64 / LPR rows x (LPR lanes x V elements) columns; lane (rsub = lane /
// LPR, cl = lane % LPR) owns segments cl, cl + LPR, ... (SLP of them) of rows (wave + i * 8) * RPI + rsub, i < RWI.
template <typename TC, int LPR, int SLP, int RWI, bool kBwd>
__global__ __launch_bounds__(512) void attn_split_kernel(AttnFusedArgs a, AttnSplitSync sy, int B) {
  constexpr int V = Elt<TC>::kVec;
  constexpr int NW = 8, RPI = 64 / LPR;
  constexpr int DPP = LPR * SLP * V;                 // columns a part covers (padded)
  __shared__ __attribute__((aligned(16))) float sq[DPP];
  __shared__ __attribute__((aligned(16))) float red[NW / 2][DPP];
  __shared__ float sdots[kSplitSMax];
  __shared__ int s_abort;
  __shared__ float spart[kSplitNS][kSplitSMax];
  // workgroup -> (batch row, part): the four parts of a row get block ids that are equal mod 8 = one XCD under round-robin
  // placement (a speed matter only)
  const int blk = blockIdx.x;
  const int b = (blk / (8 * kSplitNS)) * 8 + (blk & 7), part = (blk >> 3) & (kSplitNS - 1);
  if (b >= B) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rsub = lane / LPR, cl = lane % LPR;
  const int S = a.S, D = a.D;
  const int nseg_p = D / V / kSplitNS;               // segments per part (host: D / V divisible by 4)
  const int c0 = part * nseg_p * V;                  // first column of this part
  const TC* base = reinterpret_cast<const TC*>(a.ctx) + (long)b * S * D + c0;
  if (threadIdx.x == 0) s_abort = 0;

  // (1) every load of the slice in flight at once
  uint4 data[RWI][SLP];
#pragma unroll
  for (int i = 0; i < RWI; ++i) {
    const int s = (wave + i * NW) * RPI + rsub;
#pragma unroll
    for (int j = 0; j < SLP; ++j) {
      const int seg = cl + j * LPR;
      if (s < S && seg < nseg_p) data[i][j] = *reinterpret_cast<const uint4*>(base + (long)s * D + (long)seg * V);
      else data[i][j] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  const unsigned tag = VLN_AGENT_LOAD(sy.seq + b) + 1u;
  // the query / gradient vector slice (summed from split-K slabs while staged)
  for (int i = threadIdx.x * 4; i < DPP; i += NW * 64 * 4) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < nseg_p * V) {
      t = a.vec.at4(b, c0 + i);
      if (a.vec_out) *reinterpret_cast<float4*>(a.vec_out + (long)b * a.ldvo + c0 + i) = t;
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

  // (2) partial row dots over this part's columns
  float dot[RWI];
#pragma unroll
  for (int i = 0; i < RWI; ++i) dot[i] = 0.f;
#pragma unroll
  for (int j = 0; j < SLP; ++j) {
    float qv[V];
#pragma unroll
    for (int e = 0; e < V; e += 4) {
      const float4 t = *reinterpret_cast<const float4*>(&sq[(cl + j * LPR) * V + e]);
      qv[e] = t.x; qv[e + 1] = t.y; qv[e + 2] = t.z; qv[e + 3] = t.w;
    }
#pragma unroll
    for (int i = 0; i < RWI; ++i) {
      float x[V];
      unpack(data[i][j], x);
#pragma unroll
      for (int e = 0; e < V; ++e) dot[i] += x[e] * qv[e];
    }
  }
#pragma unroll
  for (int i = 0; i < RWI; ++i) {
    float t = dot[i];
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    const int s = (wave + i * NW) * RPI + rsub;
    if (cl == 0 && s < kSplitSMax) sdots[s] = t;
  }
  __syncthreads();

  // (3) + (4) exchange the partial row dots of the four parts; sdots[s] = their sum in part order
  attn_split_exchange(sy, b, part, B, S, tag, sdots, spart, &s_abort);

  // (5) row weights: every wave derives all of them (S <= 128: two per lane)
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
      if (part == 0 && wave == 0 && a.alpha) {
        if (s0 < S) a.alpha[ro + s0] = w0;
        if (s1 < S) a.alpha[ro + s1] = w1;
      }
    } else {
      float a0 = 0.f, a1 = 0.f, g0 = 0.f, g1 = 0.f;
      if (s0 < S) { a0 = a.alpha[ro + s0]; g0 = sdots[s0] + (a.ext ? a.ext[ro + s0] : 0.f); }
      if (s1 < S) { a1 = a.alpha[ro + s1]; g1 = sdots[s1] + (a.ext ? a.ext[ro + s1] : 0.f); }
      const float tot = wave_sum(a0 * g0 + a1 * g1);
      w0 = a0 * (g0 - tot); w1 = a1 * (g1 - tot);
      if (part == 0 && wave == 0 && a.wout) {
        if (s0 < S) a.wout[ro + s0] = w0;
        if (s1 < S) a.wout[ro + s1] = w1;
      }
    }
  }

  // (6) weighted sums of this lane's rows over its columns, reduced over the RPI row groups of the wave, then over the waves
  float acc[SLP][V];
#pragma unroll
  for (int j = 0; j < SLP; ++j)
#pragma unroll
    for (int e = 0; e < V; ++e) acc[j][e] = 0.f;
#pragma unroll
  for (int i = 0; i < RWI; ++i) {
    const int s = (wave + i * NW) * RPI + rsub;
    const float wlo = __shfl(w0, s & 63, 64), whi = __shfl(w1, s & 63, 64);
    const float w = (s < S) ? ((s < 64) ? wlo : whi) : 0.f;
#pragma unroll
    for (int j = 0; j < SLP; ++j) {
      float x[V];
      unpack(data[i][j], x);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[j][e] += w * x[e];
    }
  }
  if constexpr (RPI > 1) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
      for (int j = 0; j < SLP; ++j)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[j][e] += __shfl_xor(acc[j][e], o, 64);
  }
#pragma unroll
  for (int half = NW / 2; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half && rsub == 0) {
#pragma unroll
      for (int j = 0; j < SLP; ++j)
#pragma unroll
        for (int e = 0; e < V; e += 4)
          *reinterpret_cast<float4*>(&red[wave - half][(cl + j * LPR) * V + e]) = make_float4(acc[j][e], acc[j][e + 1], acc[j][e + 2], acc[j][e + 3]);
    }
    __syncthreads();
    if (wave < half && rsub == 0) {
#pragma unroll
      for (int j = 0; j < SLP; ++j)
#pragma unroll
        for (int e = 0; e < V; e += 4) {
          const float4 t = *reinterpret_cast<const float4*>(&red[wave][(cl + j * LPR) * V + e]);
          acc[j][e] += t.x; acc[j][e + 1] += t.y; acc[j][e + 2] += t.z; acc[j][e + 3] += t.w;
        }
    }
    __syncthreads();
  }
  if (wave == 0 && rsub == 0) {
    float* o = a.out + (long)b * a.ldo + c0;
#pragma unroll
    for (int j = 0; j < SLP; ++j) {
      const int seg = cl + j * LPR;
      if (seg < nseg_p) {
#pragma unroll
        for (int e = 0; e < V; e += 4)
          *reinterpret_cast<float4*>(o + seg * V + e) = make_float4(acc[j][e], acc[j][e + 1], acc[j][e + 2], acc[j][e + 3]);
      }
    }
  }
}

// Returns true when a split configuration covers (ctype, S, D), the four workgroups of every row are co-resident and the
// pointers are aligned; the launch is then issued.
static bool attn_split_try(hipStream_t st, int ctype, const AttnFusedArgs& a, int B, bool bwd, void* sync, long sync_bytes, int cus) {
  if (!sync || g_tunable[4] == 2 || !g_split_attn_enabled) return false;     // tunable[4] = 2: one workgroup per row (A/B); the flag: off after a timeout
  const int V = (ctype == W_BF16) ? 8 : 4;
  const int S = a.S, D = a.D;
  if (S > kSplitSMax || D % (V * kSplitNS) != 0 || B * kSplitNS > cus || sync_bytes < attn_split_sync_bytes(B) || !aligned16(sync) ||
      !aligned16(a.ctx) || !aligned16(a.vec.p) || !aligned16(a.out) || (a.vec.ld & 3) || (a.vec.stride & 3) || !aligned16(a.vec.bias) || (a.ldo & 3) ||
      (((long)D / kSplitNS) & 3) || (a.vec_out && (!aligned16(a.vec_out) || (a.ldvo & 3)))) return false;
  unsigned* sticky = sticky_dev_word();
  if (!sticky) return false;
  AttnSplitSync sy{reinterpret_cast<unsigned*>(sync), static_cast<unsigned char*>(sync) + (long)B * 64, sticky + 2};   // word 2: its own report
  const int nseg_p = D / V / kSplitNS;
  const dim3 grid(((B + 7) / 8) * 8 * kSplitNS), block(512);
  const double bytes = (double)B * S * D * (ctype == W_BF16 ? 2 : 4) + 8.0 * B * D + 8.0 * B * S;
#define VLN_SPLIT_CASE(TC, LPRv, SLPv, RWIv) \
  if (nseg_p <= LPRv * SLPv && S <= 8 * RWIv * (64 / LPRv)) { \
    if (bwd) launch_timed(K_ATTN_BWD, bytes, attn_split_kernel<TC, LPRv, SLPv, RWIv, true>, grid, block, 0, st, a, sy, B); \
    else launch_timed(K_ATTN_WSUM, bytes, attn_split_kernel<TC, LPRv, SLPv, RWIv, false>, grid, block, 0, st, a, sy, B); \
    return true; }
  if (ctype == W_BF16) {
    VLN_SPLIT_CASE(bf16_raw, 16, 1, 3)        // instruction context: S <= 96, D <= 512
    VLN_SPLIT_CASE(bf16_raw, 64, 2, 5)        // panorama: S <= 40, D <= 4096
  } else {
    VLN_SPLIT_CASE(float, 32, 1, 5)           // S <= 80, D <= 512
    VLN_SPLIT_CASE(float, 64, 3, 5)           // S <= 40, D <= 3072
  }
#undef VLN_SPLIT_CASE
  return false;
}
