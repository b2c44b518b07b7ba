// The EnvDrop step's TEXT attention with the query projection folded into the context and the LSTM cell's pointwise stage
// folded into the same launch (included by attention.hip inside namespace vln, after attention_split.h).
//
// SoftDotAttention (units.py:106-117) scores logits[b,l] = ctx[b,l,:] . (W_in h[b,:]).  The context of a rollout is fixed, so
//   logits[b,l] = (ctx W_in)[b,l,:] . h[b,:] = K[b,l,:] . h[b,:]        K = ctx W_in, ONE product per rollout (M = B * L rows)
// and the M = B product W_in h -- a dependent launch of every decoder step, and its transpose in every step's backward -- leaves
// the step.  With the four-workgroups-per-row split of attention_split.h the query is consumed COLUMN-wise: part p of row b needs
// h[b, p * H/4 .. (p + 1) * H/4) only, exactly the units whose LSTM pointwise stage (policy.py:237-240: gates -> c_1, h_1,
// dropout) it can run itself from the gate product's split-K slabs; in the backward the same part produces d drop(h_1) for its
// units (sum_l dl[l] K[l, units]) and runs the cell's pointwise backward on them.  Per step and direction: gate GEMM -> THIS
// launch -> linear_out GEMM, instead of gate GEMM -> lstm_pw -> query GEMM -> attention -> linear_out GEMM.
//
//   forward : (i,f,g,o) = sum of slabs + biases -> c1, h1, hd = drop(h1)      [units of this part]
//             dots = K[:, units] . hd -> exchange -> alpha = softmax(mask(dots)) -> out[cols] = sum_l alpha[l] ctx[l, cols]
//   backward: dots = ctx[:, cols] . d(weighted ctx)[cols] -> exchange -> dl = alpha * (dots - sum alpha dots)
//             dq[cols] = sum_l dl[l] ctx[l, cols]       (the dY rows of d W_in = dq^T hd, contracted once per rollout)
//             dhd[units] = sum_l dl[l] K[l, units]  -> + linear_out's share -> dropout -> cell backward -> dgates, dc0
// The context gradient of the rollout then is  sum_t alpha_t g_t  +  (sum_t dl_t hd_t) W_in^T  (attn_dctx_deferred + one GEMM).
// K is fp32 whatever the streamed type of ctx: a bf16 K would put a 2^-9 rounding in front of the softmax (what W_in is
// streamed in fp32 for).
#pragma once

struct TextKArgs {
  const void* ctx;            // [B,S,D] streamed (fp32 or bf16)
  const float* kctx;          // [B,S,D] fp32: ctx W_in
  const uint8_t* mask;        // fwd: [B,S] 1 = masked, nullable
  float* alpha;               // fwd: out [B,S]; bwd: in
  float* out; long ldo;       // fwd: weighted context [B,D]; bwd: dq [B,D]
  int S, D;
  LstmPwFwd pw;               // fwd: the cell's pointwise stage (pw.H == D; h1_drop = the query)
  SlabVec dwc;                // bwd: [B,2D], cols [0,D) d(weighted ctx), cols [D,2D) linear_out's share of d drop(h1); maybe in slabs
  float* dwc_out; long lddo;  // bwd, nullable: the summed d(weighted ctx) written back (g operand of the deferred dctx)
  float* wout;                // bwd: d logits [B,S], nullable
  LstmPwBwd pb;               // bwd: the cell's pointwise backward (dh1_b / dh1_b2 unused: produced here)
};

// A part's [S, D/4] slice of one row block, resident in registers.  Geometry as attn_split_kernel: one wave instruction covers
// RPI = 64 / LPR rows x (LPR lanes x V elements); lane (rsub, cl) owns segment cl of rows (wave + i * 8) * RPI + rsub, i < RWI.
template <typename TC, int LPR, int RWI>
struct TextKSlice {
  static constexpr int V = Elt<TC>::kVec, NW = 8, RPI = 64 / LPR, DPP = LPR * V;
  uint4 data[RWI];

  __device__ __forceinline__ void load(const TC* base, int S, int D, int nseg_p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rsub = lane / LPR, cl = lane % LPR;
#pragma unroll
    for (int i = 0; i < RWI; ++i) {
      const int s = (wave + i * NW) * RPI + rsub;
      if (s < S && cl < nseg_p) data[i] = *reinterpret_cast<const uint4*>(base + (long)s * D + (long)cl * V);
      else data[i] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  static __device__ __forceinline__ void unpack(const uint4& u, float (&x)[V]) {
    if constexpr (V == 8) {
      x[0] = __uint_as_float(u.x << 16); x[1] = __uint_as_float(u.x & 0xffff0000u);
      x[2] = __uint_as_float(u.y << 16); x[3] = __uint_as_float(u.y & 0xffff0000u);
      x[4] = __uint_as_float(u.z << 16); x[5] = __uint_as_float(u.z & 0xffff0000u);
      x[6] = __uint_as_float(u.w << 16); x[7] = __uint_as_float(u.w & 0xffff0000u);
    } else {
      x[0] = __uint_as_float(u.x); x[1] = __uint_as_float(u.y); x[2] = __uint_as_float(u.z); x[3] = __uint_as_float(u.w);
    }
  }
  // sdots[s] = sum over this part's columns of slice[s, :] * sq[:]     (sq: DPP floats in LDS, zero past the part's columns)
  __device__ __forceinline__ void dots(const float* sq, float* sdots) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rsub = lane / LPR, cl = lane % LPR;
    float qv[V];
#pragma unroll
    for (int e = 0; e < V; e += 4) {
      const float4 t = *reinterpret_cast<const float4*>(&sq[cl * V + e]);
      qv[e] = t.x; qv[e + 1] = t.y; qv[e + 2] = t.z; qv[e + 3] = t.w;
    }
#pragma unroll
    for (int i = 0; i < RWI; ++i) {
      float x[V];
      unpack(data[i], x);
      float t = 0.f;
#pragma unroll
      for (int e = 0; e < V; ++e) t += x[e] * qv[e];
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      const int s = (wave + i * NW) * RPI + rsub;
      if (cl == 0 && s < kSplitSMax) sdots[s] = t;
    }
  }
  // partial weighted sums of this lane's rows, reduced over the RPI row groups of its wave: acc[e] valid in lanes rsub == 0
  __device__ __forceinline__ void wsum_wave(float w0, float w1, int S, float (&acc)[V]) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rsub = lane / LPR;
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < RWI; ++i) {
      const int s = (wave + i * NW) * RPI + rsub;
      const float wlo = __shfl(w0, s & 63, 64), whi = __shfl(w1, s & 63, 64);
      const float w = (s < S) ? ((s < 64) ? wlo : whi) : 0.f;
      float x[V];
      unpack(data[i], x);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += w * x[e];
    }
    if constexpr (RPI > 1) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
    }
  }
};

// Cross-wave tree (fixed order 8 -> 4 -> 2 -> 1) of per-wave partial sums `acc` (valid in lanes rsub == 0, column cl * V + e);
// the result lands in wave 0's lanes rsub == 0.  `red`: [4][DPP] floats of LDS.
template <int V, int LPR>
__device__ __forceinline__ void textk_tree(float (&acc)[V], float* red) {
  constexpr int DPP = LPR * V;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rsub = lane / LPR, cl = lane % LPR;
#pragma unroll
  for (int half = 4; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half && rsub == 0) {
#pragma unroll
      for (int e = 0; e < V; e += 4)
        *reinterpret_cast<float4*>(&red[(wave - half) * DPP + cl * V + e]) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
    }
    __syncthreads();
    if (wave < half && rsub == 0) {
#pragma unroll
      for (int e = 0; e < V; e += 4) {
        const float4 t = *reinterpret_cast<const float4*>(&red[wave * DPP + cl * V + e]);
        acc[e] += t.x; acc[e + 1] += t.y; acc[e + 2] += t.z; acc[e + 3] += t.w;
      }
    }
    __syncthreads();
  }
}

constexpr int kTextKCols = 128;          // columns (= LSTM units) of a part: D / 4 <= 128
constexpr int kTextKRWI_F32 = 6;         // fp32 slices: 32 lanes per row, 2 rows per instruction, 8 waves x 6 x 2 = 96 rows
constexpr int kTextKRWI_BF16 = 3;        // bf16 slices: 16 lanes per row, 4 rows per instruction, 8 waves x 3 x 4 = 96 rows
constexpr int kTextKSMax = 96;

template <typename TC> struct TextKGeom;
template <> struct TextKGeom<float> { static constexpr int LPR = 32, RWI = kTextKRWI_F32; };
template <> struct TextKGeom<bf16_raw> { static constexpr int LPR = 16, RWI = kTextKRWI_BF16; };

template <typename TC>
__global__ __launch_bounds__(512) void attn_textk_fwd_kernel(TextKArgs a, AttnSplitSync sy, int B) {
  typedef TextKSlice<TC, TextKGeom<TC>::LPR, TextKGeom<TC>::RWI> CtxS;
  typedef TextKSlice<float, 32, kTextKRWI_F32> KS;
  constexpr int VC = CtxS::V;
  __shared__ __attribute__((aligned(16))) float sq[kTextKCols];
  __shared__ __attribute__((aligned(16))) float red[4 * kTextKCols];
  __shared__ __attribute__((aligned(16))) float sg[4][4][kTextKCols];      // [slab residue][gate][unit]
  __shared__ float sdots[kSplitSMax];
  __shared__ float spart[kSplitNS][kSplitSMax];
  __shared__ int s_abort;
  const int blk = blockIdx.x;
  const int b = (blk / (8 * kSplitNS)) * 8 + (blk & 7), part = (blk >> 3) & (kSplitNS - 1);
  if (b >= B) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = a.S, D = a.D, H = D;
  const int DP = D / kSplitNS;                        // columns / units of this part (host: D % 32 == 0, DP <= 128)
  const int c0 = part * DP;
  if (threadIdx.x == 0) s_abort = 0;

  // (1) every load of the launch in flight at once, in the order the results are needed (the memory counter retires in issue
  // order): the gate product's slabs (the previous launch's output: the longest latency) and the cell's other operands first,
  // then K (the dots), then the context slice (the weighted sum), the mask last.
  // The LSTM cell's pointwise stage of units [c0, c0 + DP) of row b (policy.py:237-240; gate order i,f,g,o): thread -> (unit
  // group g of 4, gate q, slab residue r) sums slabs r, r + 4, r + 8, ... of its gate, 16-byte loads.
  const LstmPwFwd& p = a.pw;
  const int g = threadIdx.x & 31, q = (threadIdx.x >> 5) & 3, r = threadIdx.x >> 7;
  const bool gon = 4 * g < DP;
  const float* gp = p.gates + (long)b * 4 * H + (long)q * H + c0 + 4 * g;
  float4 t[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int sl = r + 4 * k;
    t[k] = (gon && sl < p.nsplit) ? *reinterpret_cast<const float4*>(gp + (long)sl * p.slab_stride) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int j = threadIdx.x;
  float cprev = 0.f, bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (j < DP) {
    cprev = p.c0[(long)b * p.ldc0 + c0 + j];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (p.bias_a) bias[k] += p.bias_a[k * H + c0 + j];
      if (p.bias_b) bias[k] += p.bias_b[k * H + c0 + j];
    }
  }
  KS ks; CtxS cs;
  ks.load(a.kctx + (long)b * S * D + c0, S, D, DP / 4);
  cs.load(reinterpret_cast<const TC*>(a.ctx) + (long)b * S * D + c0, S, D, DP / VC);
  const unsigned tag = VLN_AGENT_LOAD(sy.seq + b) + 1u;
  bool m0 = false, m1 = false;
  if (a.mask) {
    if (lane < S) m0 = a.mask[(long)b * S + lane] != 0;
    if (lane + 64 < S) m1 = a.mask[(long)b * S + lane + 64] != 0;
  }

  // (2) the cell
  {
    float4 v;
    v.x = (t[0].x + t[1].x) + (t[2].x + t[3].x); v.y = (t[0].y + t[1].y) + (t[2].y + t[3].y);
    v.z = (t[0].z + t[1].z) + (t[2].z + t[3].z); v.w = (t[0].w + t[1].w) + (t[2].w + t[3].w);
    if (gon) {
      for (int s0 = r + 16; s0 < p.nsplit; s0 += 16) {       // more than 16 slabs (small shapes only): further rounds of four
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int sl = s0 + 4 * k;
          t[k] = (sl < p.nsplit) ? *reinterpret_cast<const float4*>(gp + (long)sl * p.slab_stride) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        v.x += (t[0].x + t[1].x) + (t[2].x + t[3].x); v.y += (t[0].y + t[1].y) + (t[2].y + t[3].y);
        v.z += (t[0].z + t[1].z) + (t[2].z + t[3].z); v.w += (t[0].w + t[1].w) + (t[2].w + t[3].w);
      }
    }
    *reinterpret_cast<float4*>(&sg[r][q][4 * g]) = v;
    __syncthreads();
    if (j < kTextKCols) {
      float hd = 0.f;
      if (j < DP) {
        const int u = c0 + j;
        float pre[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) pre[k] = ((sg[0][k][j] + sg[1][k][j]) + (sg[2][k][j] + sg[3][k][j])) + bias[k];
        const LstmCellPw c = lstm_cell_pw(pre[0], pre[1], pre[2], pre[3], cprev);
        const long e = (long)b * H + u;
        p.h1[(long)b * p.ldh1 + u] = c.hn;
        p.c1[(long)b * p.ldc1 + u] = c.cn;
        if (p.act) {
          float* ap = p.act + (long)b * 4 * H + u;
          ap[0] = c.si; ap[H] = c.sf; ap[2 * H] = c.tg; ap[3 * H] = c.so;
        }
        if (p.tanh_c1) p.tanh_c1[e] = c.tc;
        hd = c.hn * dropout_scale1(p.drop.seed, p.drop.off(), (uint32_t)e, p.drop.p);
        p.h1_drop[(long)b * p.ldh1d + u] = hd;
      }
      sq[j] = hd;
    }
  }
  __syncthreads();

  // (3) partial row dots K[:, units] . drop(h1)[units], exchanged between the row's four workgroups
  ks.dots(sq, sdots);
  __syncthreads();
  attn_split_exchange(sy, b, part, B, S, tag, sdots, spart, &s_abort);

  // (4) softmax over the S rows: every wave derives all weights (S <= 96: two per lane)
  float w0, w1;
  {
    const int s0 = lane, s1 = lane + 64;
    const long ro = (long)b * S;
    float v0 = -INFINITY, v1 = -INFINITY;
    if (s0 < S && !m0) v0 = sdots[s0];
    if (s1 < S && !m1) v1 = sdots[s1];
    const float mx = wave_max(fmaxf(v0, v1));
    const float e0 = (s0 < S) ? __expf(v0 - mx) : 0.f, e1 = (s1 < S) ? __expf(v1 - mx) : 0.f;
    const float inv = 1.0f / wave_sum(e0 + e1);
    w0 = e0 * inv; w1 = e1 * inv;
    if (part == 0 && wave == 0 && a.alpha) {
      if (s0 < S) a.alpha[ro + s0] = w0;
      if (s1 < S) a.alpha[ro + s1] = w1;
    }
  }

  // (5) weighted context of this part's columns
  float acc[VC];
  cs.wsum_wave(w0, w1, S, acc);
  textk_tree<VC, CtxS::DPP / VC>(acc, red);
  if (wave == 0 && lane < DP / VC) {
    float* o = a.out + (long)b * a.ldo + c0 + lane * VC;
#pragma unroll
    for (int e = 0; e < VC; e += 4) *reinterpret_cast<float4*>(o + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
  }
}

template <typename TC>
__global__ __launch_bounds__(512) void attn_textk_bwd_kernel(TextKArgs a, AttnSplitSync sy, int B) {
  typedef TextKSlice<TC, TextKGeom<TC>::LPR, TextKGeom<TC>::RWI> CtxS;
  typedef TextKSlice<float, 32, kTextKRWI_F32> KS;
  constexpr int VC = CtxS::V;
  __shared__ __attribute__((aligned(16))) float sq[kTextKCols];
  __shared__ __attribute__((aligned(16))) float sdh[kTextKCols];        // linear_out's share of d drop(h1), then + K's share
  __shared__ __attribute__((aligned(16))) float red[4 * kTextKCols];
  __shared__ float sdots[kSplitSMax];
  __shared__ float spart[kSplitNS][kSplitSMax];
  __shared__ int s_abort;
  const int blk = blockIdx.x;
  const int b = (blk / (8 * kSplitNS)) * 8 + (blk & 7), part = (blk >> 3) & (kSplitNS - 1);
  if (b >= B) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = a.S, D = a.D, H = D;
  const int DP = D / kSplitNS;
  const int c0 = part * DP;
  if (threadIdx.x == 0) s_abort = 0;

  // (1) every load in flight at once, in the order the results are needed: d(weighted ctx) and linear_out's share of d drop(h1)
  // for this part's columns (the previous launch's split-K slabs, summed while staged) first, then the context slice (the dots
  // and dq), then K (d drop(h1)), then alpha and the operands of the cell's pointwise backward
  if (threadIdx.x < 2 * (kTextKCols / 4)) {
    const int half = threadIdx.x / (kTextKCols / 4), i = (threadIdx.x % (kTextKCols / 4)) * 4;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < DP) {
      t = a.dwc.at4(b, (long)half * D + c0 + i);
      if (half == 0 && a.dwc_out) *reinterpret_cast<float4*>(a.dwc_out + (long)b * a.lddo + c0 + i) = t;
    }
    *reinterpret_cast<float4*>(half ? &sdh[i] : &sq[i]) = t;
  }
  CtxS cs; KS ks;
  cs.load(reinterpret_cast<const TC*>(a.ctx) + (long)b * S * D + c0, S, D, DP / VC);
  ks.load(a.kctx + (long)b * S * D + c0, S, D, DP / 4);
  const unsigned tag = VLN_AGENT_LOAD(sy.seq + b) + 1u;
  float a0 = 0.f, a1 = 0.f;
  if (lane < S) a0 = a.alpha[(long)b * S + lane];
  if (lane + 64 < S) a1 = a.alpha[(long)b * S + lane + 64];
  const LstmPwBwd& p = a.pb;
  const int j = threadIdx.x;
  float e_dh = 0.f, e_si = 0.f, e_sf = 0.f, e_tg = 0.f, e_so = 0.f, e_tc = 0.f, e_c0 = 0.f, e_dc = 0.f;
  if (j < DP) {
    const int u = c0 + j;
    if (p.dh1_a) e_dh = p.dh1_a[(long)b * p.ld_a + u];
    const float* act = p.act + (long)b * 4 * H + u;
    e_si = act[0]; e_sf = act[H]; e_tg = act[2 * H]; e_so = act[3 * H];
    e_tc = p.tanh_c1[(long)b * H + u];
    e_c0 = p.c0[(long)b * p.ldc0 + u];
    if (p.dc1) e_dc = p.dc1[(long)b * p.lddc1 + u];
  }
  __syncthreads();

  // (2) d alpha (partial over this part's columns), exchanged
  cs.dots(sq, sdots);
  __syncthreads();
  attn_split_exchange(sy, b, part, B, S, tag, sdots, spart, &s_abort);

  // (3) softmax backward
  float w0, w1;
  {
    const int s0 = lane, s1 = lane + 64;
    const long ro = (long)b * S;
    float g0 = 0.f, g1 = 0.f;
    if (s0 < S) g0 = sdots[s0];
    if (s1 < S) g1 = sdots[s1];
    const float tot = wave_sum(a0 * g0 + a1 * g1);
    w0 = a0 * (g0 - tot); w1 = a1 * (g1 - tot);
    if (part == 0 && wave == 0 && a.wout) {
      if (s0 < S) a.wout[ro + s0] = w0;
      if (s1 < S) a.wout[ro + s1] = w1;
    }
  }

  // (4) dq[cols] = sum_l dl[l] ctx[l, cols]  and  d drop(h1)[units] += sum_l dl[l] K[l, units]: both trees share the barriers' cost
  // by running back to back on disjoint halves of `red`?  No: the tree uses all four rows of `red`; run them in sequence.
  {
    float acc[VC];
    cs.wsum_wave(w0, w1, S, acc);
    textk_tree<VC, CtxS::DPP / VC>(acc, red);
    if (wave == 0 && lane < DP / VC) {
      float* o = a.out + (long)b * a.ldo + c0 + lane * VC;
#pragma unroll
      for (int e = 0; e < VC; e += 4) *reinterpret_cast<float4*>(o + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
    }
  }
  {
    float acc[4];
    ks.wsum_wave(w0, w1, S, acc);
    textk_tree<4, 32>(acc, red);
    if (wave == 0 && lane < DP / 4) {
      float4 t = *reinterpret_cast<const float4*>(&sdh[lane * 4]);
      t.x += acc[0]; t.y += acc[1]; t.z += acc[2]; t.w += acc[3];
      *reinterpret_cast<float4*>(&sdh[lane * 4]) = t;
    }
  }
  __syncthreads();

  // (5) the cell's pointwise backward on units [c0, c0 + DP) (lstm_pw_bwd_body's formulas)
  if (j < DP) {
    const int u = c0 + j;
    const long e = (long)b * H + u;
    const float dh = e_dh + sdh[j] * dropout_scale1(p.drop.seed, p.drop.off(), (uint32_t)e, p.drop.p);
    const float dc = e_dc + dh * e_so * (1.f - e_tc * e_tc);
    float* dg = p.dgates + (long)b * p.lddg + u;
    dg[0] = dc * e_tg * e_si * (1.f - e_si);
    dg[H] = dc * e_c0 * e_sf * (1.f - e_sf);
    dg[2 * H] = dc * e_si * (1.f - e_tg * e_tg);
    dg[3 * H] = dh * e_tc * e_so * (1.f - e_so);
    p.dc0[(long)b * p.lddc0 + u] = dc * e_sf;
  }
}

// Whether the folded text attention covers this shape on this device (the four workgroups of every row co-resident, a part's
// columns in one slice, the exchange buffer present and the four-workgroup path not switched off by a timeout or a tunable).
// CAPABLE: what the kernels need.  OK: capable AND chosen -- the switches (tunable[4], the four-workgroup path turned off by a timeout
// report) decide which form a NEW rollout takes; a rollout that already runs on the projected context has no per-step query to fall
// back to, so its launches only ask `capable` and finish the rollout on the kernels it started with (their waits stay bounded)
// instead of failing its remaining forward steps and its whole backward.
static bool attn_textk_capable(int ctype, int B, int S, int D, const void* sync, long sync_bytes) {
  if (!sync) return false;
  if (ctype != W_BF16 && ctype != W_F32) return false;
  if (S <= 0 || S > kTextKSMax || D <= 0 || (D % 32) != 0 || D / kSplitNS > kTextKCols) return false;
  if (B * kSplitNS > device_cus() || sync_bytes < attn_split_sync_bytes(B) || !aligned16(sync)) return false;
  return sticky_dev_word() != nullptr;
}
bool attn_textk_ok(int ctype, int B, int S, int D, const void* sync, long sync_bytes) {
  if (g_tunable[4] != 0 || !g_split_attn_enabled) return false;
  return attn_textk_capable(ctype, B, S, D, sync, sync_bytes);
}

static int textk_launch(hipStream_t st, int ctype, const TextKArgs& a, int B, bool bwd, void* sync, long sync_bytes) {
  if (!attn_textk_capable(ctype, B, a.S, a.D, sync, sync_bytes)) {
    set_error("attn_textk: shape / device / exchange buffer not supported (B=%d S=%d D=%d)", B, a.S, a.D);
    return VLN_ERR_ARG;
  }
  if (!a.ctx || !a.kctx || !a.alpha || !a.out || !aligned16(a.ctx) || !aligned16(a.kctx) || !aligned16(a.out) || (a.ldo & 3)) {
    set_error("attn_textk: null or misaligned operand");
    return VLN_ERR_ARG;
  }
  AttnSplitSync sy{reinterpret_cast<unsigned*>(sync), static_cast<unsigned char*>(sync) + (long)B * 64, sticky_dev_word() + 2};
  const dim3 grid(((B + 7) / 8) * 8 * kSplitNS), block(512);
  const double bytes = (double)B * a.S * a.D * ((ctype == W_BF16 ? 2 : 4) + 4) + 8.0 * B * a.D + 8.0 * B * a.S + 4.0 * B * a.D * 12;
  if (ctype == W_BF16) {
    if (bwd) launch_timed(K_ATTN_BWD, bytes, attn_textk_bwd_kernel<bf16_raw>, grid, block, 0, st, a, sy, B);
    else launch_timed(K_ATTN_WSUM, bytes, attn_textk_fwd_kernel<bf16_raw>, grid, block, 0, st, a, sy, B);
  } else {
    if (bwd) launch_timed(K_ATTN_BWD, bytes, attn_textk_bwd_kernel<float>, grid, block, 0, st, a, sy, B);
    else launch_timed(K_ATTN_WSUM, bytes, attn_textk_fwd_kernel<float>, grid, block, 0, st, a, sy, B);
  }
  VLN_CHECK_LAUNCH("attn_textk");
  return VLN_OK;
}

int attn_textk_fwd(hipStream_t st, const void* ctx, int ctype, const float* kctx, const uint8_t* mask, float* alpha, float* out,
                   long ldo, const LstmPwFwd& pw, int B, int S, int D, void* sync, long sync_bytes) {
  if (pw.H != D || pw.B != B || !pw.gates || pw.nsplit < 1 || !pw.c0 || !pw.h1 || !pw.c1 || !pw.h1_drop ||
      !aligned16(pw.gates) || (pw.slab_stride & 3)) {
    set_error("attn_textk_fwd: bad cell arguments (H=%d D=%d nsplit=%d)", pw.H, D, pw.nsplit);
    return VLN_ERR_ARG;
  }
  TextKArgs a{};
  a.ctx = ctx; a.kctx = kctx; a.mask = mask; a.alpha = alpha; a.out = out; a.ldo = ldo; a.S = S; a.D = D; a.pw = pw;
  return textk_launch(st, ctype, a, B, false, sync, sync_bytes);
}

int attn_textk_bwd(hipStream_t st, const void* ctx, int ctype, const float* kctx, const float* alpha, SlabVec dwc, float* dwc_out,
                   long lddo, float* dq, long lddq, float* dl_out, const LstmPwBwd& pb, int B, int S, int D, void* sync,
                   long sync_bytes) {
  if (pb.H != D || pb.B != B || !pb.act || !pb.tanh_c1 || !pb.c0 || !pb.dgates || !pb.dc0 || !dwc.p || !aligned16(dwc.p) ||
      (dwc.ld & 3) || (dwc.stride & 3) || (dwc_out && (!aligned16(dwc_out) || (lddo & 3)))) {
    set_error("attn_textk_bwd: bad cell / gradient arguments");
    return VLN_ERR_ARG;
  }
  TextKArgs a{};
  a.ctx = ctx; a.kctx = kctx; a.alpha = const_cast<float*>(alpha); a.out = dq; a.ldo = lddq; a.S = S; a.D = D;
  a.dwc = dwc; a.dwc_out = dwc_out; a.lddo = lddo; a.wout = dl_out; a.pb = pb;
  return textk_launch(st, ctype, a, B, true, sync, sync_bytes);
}
