// Packed (bi)LSTM instruction encoder (reference units.py:48-74: embedding -> dropout ->
// pack_padded_sequence -> nn.LSTM -> pad_packed_sequence -> dropout).
//
// The input projection of ALL time steps is one GEMM (gemm_nt, M = L*B, time-major rows); only the
// h_{t-1} * W_hh^T recurrence is sequential.  One launch per time step handles BOTH directions:
//   workgroup = 4 waves = the 4 gates (i,f,g,o) of 16 hidden units for 16 batch rows,
//   grid = (Hd/16, dirs, B/16)  ->  128 workgroups at B=64, Hd=256, so the fp32 MFMA work of a step is spread
//   over half the chip; W_hh fragments stream global->VGPR (L2-resident across steps), the gate
//   pre-activations meet in LDS and each of the 256 threads finishes one (row, unit) of the fused cell.
// Packed-sequence semantics: row b only advances while t < len[b]; outputs at padded steps are 0; the
// reverse direction starts at each row's own last token because its state stays 0 until t < len[b].
// Backward-through-time mirrors it: wave w contracts gate block w of the previous step's dgates with the
// transposed W_hh shadow, LDS-reduce, then the pointwise cell backward.
#include <mutex>
#include <unordered_map>

#include "vln_internal.h"
#include "graph_cache.h"
#include "gather_ride.h"
#include "prologue_bodies.h"
#include "shadow_bodies.h"
#include "layout_bodies.h"
#include "../../include/vln_hip.h"

namespace vln {

template <typename TW> struct RecCfg;
template <> struct RecCfg<float> { static constexpr int BK = 32, VK = 8; };
template <> struct RecCfg<bf16_raw> { static constexpr int BK = 64, VK = 16; };

// Fast path: a K range of exactly NS K-steps, aligned, all rows valid: every load of the whole range is issued
// before the first MFMA (one memory latency per launch instead of one per K-step).
template <typename TW, int NS>
__device__ __forceinline__ void wave_tile_16x16_fast(const float* arow, const TW* wrow, int kbeg, int fq, f32x4& acc) {
  constexpr int BK = RecCfg<TW>::BK, VK = RecCfg<TW>::VK;
  float4 av[NS][VK / 4];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int v = 0; v < VK / 4; ++v) av[s][v] = *reinterpret_cast<const float4*>(arow + kbeg + s * BK + fq * VK + v * 4);
  if constexpr (sizeof(TW) == 4) {
    float4 wv[NS][2];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      wv[s][0] = *reinterpret_cast<const float4*>(wrow + kbeg + s * BK + fq * VK);
      wv[s][1] = *reinterpret_cast<const float4*>(wrow + kbeg + s * BK + fq * VK + 4);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][0].x, wv[s][0].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][0].y, wv[s][0].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][0].z, wv[s][0].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][0].w, wv[s][0].w, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][1].x, wv[s][1].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][1].y, wv[s][1].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][1].z, wv[s][1].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][1].w, wv[s][1].w, acc, 0, 0, 0);
    }
  } else {
    bf16x8 wv[NS][2];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      wv[s][0] = *reinterpret_cast<const bf16x8*>(wrow + kbeg + s * BK + fq * VK);
      wv[s][1] = *reinterpret_cast<const bf16x8*>(wrow + kbeg + s * BK + fq * VK + 8);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float x[16] = {av[s][0].x, av[s][0].y, av[s][0].z, av[s][0].w, av[s][1].x, av[s][1].y, av[s][1].z, av[s][1].w,
                           av[s][2].x, av[s][2].y, av[s][2].z, av[s][2].w, av[s][3].x, av[s][3].y, av[s][3].z, av[s][3].w};
      bf16x8 a0, a1, l0, l1;   // activations split hi + lo: only the weight stream is quantised
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a0[j] = (__bf16)x[j];
        a1[j] = (__bf16)x[8 + j];
        l0[j] = (__bf16)(x[j] - (float)a0[j]);
        l1[j] = (__bf16)(x[8 + j] - (float)a1[j]);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0, wv[s][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, wv[s][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wv[s][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wv[s][1], acc, 0, 0, 0);
    }
  }
}

// acc += A[16 rows, kbeg:kend] * Wrow[16 cols, kbeg:kend]^T ; A fp32 row-major (lda), W rows K-contiguous.
// Lane (fi = lane&15, fq = lane>>4) owns A row fi / W row fi and k-slots fq*VK..+VK of every K-step.
template <typename TW>
__device__ __forceinline__ void wave_tile_16x16(const float* arow, bool a_ok, const TW* wrow, bool w_ok, int kbeg,
                                                int kend, int fq, bool vec, f32x4& acc) {
  if (vec && (kend - kbeg) == 256 && __all(a_ok && w_ok)) {   // the reference's 2x256 encoder (wave-uniform test)
    wave_tile_16x16_fast<TW, 256 / RecCfg<TW>::BK>(arow, wrow, kbeg, fq, acc);
    return;
  }
  constexpr int BK = RecCfg<TW>::BK, VK = RecCfg<TW>::VK;
#pragma unroll 4
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const int k = k0 + fq * VK;
    float av[VK];
    if (a_ok && vec && k + VK <= kend) {
#pragma unroll
      for (int v = 0; v < VK / 4; ++v) {
        float4 t = *reinterpret_cast<const float4*>(arow + k + v * 4);
        av[v * 4] = t.x; av[v * 4 + 1] = t.y; av[v * 4 + 2] = t.z; av[v * 4 + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < VK; ++j) av[j] = (a_ok && (k + j) < kend) ? arow[k + j] : 0.f;
    }
    if constexpr (sizeof(TW) == 4) {
      float wv[8];
      if (w_ok && vec && k + 8 <= kend) {
        float4 t0 = *reinterpret_cast<const float4*>(wrow + k);
        float4 t1 = *reinterpret_cast<const float4*>(wrow + k + 4);
        wv[0] = t0.x; wv[1] = t0.y; wv[2] = t0.z; wv[3] = t0.w; wv[4] = t1.x; wv[5] = t1.y; wv[6] = t1.z; wv[7] = t1.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = (w_ok && (k + j) < kend) ? wrow[k + j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv[j], acc, 0, 0, 0);
    } else {
      bf16x8 a0, a1, l0, l1, w0, w1;   // activations split hi + lo: only the weight stream is quantised
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a0[j] = (__bf16)av[j];
        a1[j] = (__bf16)av[8 + j];
        l0[j] = (__bf16)(av[j] - (float)a0[j]);
        l1[j] = (__bf16)(av[8 + j] - (float)a1[j]);
      }
      if (w_ok && vec && k + 16 <= kend) {
        w0 = *reinterpret_cast<const bf16x8*>(wrow + k);
        w1 = *reinterpret_cast<const bf16x8*>(wrow + k + 8);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          bf16_raw r0 = (w_ok && (k + j) < kend) ? wrow[k + j] : (bf16_raw)0;
          bf16_raw r1 = (w_ok && (k + 8 + j) < kend) ? wrow[k + 8 + j] : (bf16_raw)0;
          w0[j] = __builtin_bit_cast(__bf16, r0);
          w1[j] = __builtin_bit_cast(__bf16, r1);
        }
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0, w0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, w1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, w1, acc, 0, 0, 0);
    }
  }
}

struct RecFwdArgs {
  const float* xproj;     // [L*B, dirs*4Hd] (biases already added)
  const void* w_hh;       // [dirs][4Hd, Hd]
  const int* lengths;     // [B]
  float* hprev;           // [dirs][L][B][Hd]  state fed into the cell at time t
  float* cprev;           // [dirs][L][B][Hd]
  float* y;               // [L*B, dirs*Hd]
  float* act;             // [L*B, dirs*4Hd]
  float* tanh_c;          // [L*B, dirs*Hd]
  float* hcat;            // [B, dirs*Hd] final h
  float* ccat;            // [B, dirs*Hd] final c
  int B, L, Hd, dirs, step, vec;
  int init;               // 1: the first time slot of hprev/cprev holds a caller-given initial state (else zeros)
  int nbb_per;            // persistent granule kernel: row blocks per PASS (0 = all of them in one pass), see persist_passes()
  // in-kernel input projection (persistent granule kernel, round 6; x == nullptr: the gate inputs come from xproj):
  const float* x;         // [L*B, kInprojE] time-major embedded inputs
  const void* w_ih;       // [dirs*4Hd, kInprojE] in the recurrence's weight type
  const float* bsum;      // [dirs*4Hd] b_ih + b_hh
};

template <typename TW>
__global__ __launch_bounds__(256) void lstm_rec_fwd_kernel(RecFwdArgs a) {
  __shared__ float sg[4][16][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 15, fq = lane >> 4;
  const int j0 = blockIdx.x * 16, d = blockIdx.y, b0 = blockIdx.z * 16;
  const int Hd = a.Hd, B = a.B, L = a.L;
  const int t = (d == 0) ? a.step : (L - 1 - a.step);
  const long sbase = ((long)d * L + t) * B;          // row offset into hprev/cprev for (d, t)
  // pointwise operands of this thread's (row, unit): issued first so their latency hides under the MFMAs
  const int bl = threadIdx.x >> 4, jl = threadIdx.x & 15;
  const int b = b0 + bl, j = j0 + jl;
  const bool live = (b < B) && (j < Hd);
  const int G = a.dirs * 4 * Hd, Y = a.dirs * Hd;
  const long row = (long)t * B + (live ? b : 0);
  float xi = 0.f, xf = 0.f, xg = 0.f, xo = 0.f, hp = 0.f, cp = 0.f;
  int len = 0;
  if (live) {
    const float* xp = a.xproj + row * G + (long)d * 4 * Hd + j;
    xi = xp[0]; xf = xp[Hd]; xg = xp[2 * Hd]; xo = xp[3 * Hd];
    hp = a.hprev[(sbase + b) * Hd + j];
    cp = a.cprev[(sbase + b) * Hd + j];
    len = a.lengths[b];
  }
  // gate `wave` of units j0..j0+15 for rows b0..b0+15
  {
    const bool a_ok = (b0 + fi) < B, w_ok = (j0 + fi) < Hd;
    const float* arow = a.hprev + (sbase + (a_ok ? b0 + fi : 0)) * Hd;
    const TW* wrow = reinterpret_cast<const TW*>(a.w_hh) + ((long)d * 4 * Hd + (long)wave * Hd + (w_ok ? j0 + fi : 0)) * Hd;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    wave_tile_16x16<TW>(arow, a_ok, wrow, w_ok, 0, Hd, fq, a.vec, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) sg[wave][fq * 4 + r][fi] = acc[r];
  }
  __syncthreads();
  if (!live) return;
  const float pi = sg[0][bl][jl] + xi, pf = sg[1][bl][jl] + xf;
  const float pg = sg[2][bl][jl] + xg, po = sg[3][bl][jl] + xo;
  const bool valid = t < len;
  const LstmCellPw cw = lstm_cell_pw(pi, pf, pg, po, cp);
  const float si = cw.si, sf = cw.sf, tg = cw.tg, so = cw.so, cn = cw.cn, tc = cw.tc, hn = cw.hn;
  float* ac = a.act + row * G + (long)d * 4 * Hd + j;
  ac[0] = si; ac[Hd] = sf; ac[2 * Hd] = tg; ac[3 * Hd] = so;
  a.tanh_c[row * Y + d * Hd + j] = tc;
  a.y[row * Y + d * Hd + j] = valid ? hn : 0.f;
  const float hs = valid ? hn : hp, cs = valid ? cn : cp;
  const int tn = (d == 0) ? t + 1 : t - 1;
  if (tn >= 0 && tn < L) {
    const long nb = ((long)d * L + tn) * B + b;
    a.hprev[nb * Hd + j] = hs;
    a.cprev[nb * Hd + j] = cs;
  } else {
    a.hcat[(long)b * Y + d * Hd + j] = hs;
    a.ccat[(long)b * Y + d * Hd + j] = cs;
  }
}

struct RecBwdArgs {
  const float* dy;        // [L*B, dirs*Hd] grad of the layer output (nullable)
  const void* w_hh_t;     // [dirs][Hd, 4Hd] transposed shadow
  const int* lengths;
  const float* act; const float* tanh_c; const float* cprev;
  float* dgates;          // [L*B, dirs*4Hd]
  float* dh_pass;         // [dirs][B][Hd]  in: grad wrt the state after this step; out: before it (frozen rows)
  float* dc_carry;        // [dirs][B][Hd]
  int B, L, Hd, dirs, step, first, vec;
  // optional: the gradients of the FINAL states as the caller has them, [B, dirs*Hd] (hcat / ccat layout).  The persistent
  // kernels read their initial dh / dc from these instead of dh_pass / dc_carry (which are then pure scratch): the caller's
  // two transposing copies (two launches of ~5 us in front of the recurrence) disappear.
  const float* dh_bm; const float* dc_bm;
  // optional (persistent counter-protocol kernel): [dirs][ceil(B / 16)][4 * Hd] partial column sums of dgates -- the LSTM's bias
  // gradients summed over each workgroup's 16 rows and all L steps by the threads that form the values (vln_lstm_seq_bwd)
  float* bias_part;
  int nbb_per;            // persistent counter-protocol kernel: row blocks per PASS (0 = all in one pass), see persist_passes()
  // in-launch weight gradients (persistent counter-protocol kernel, round 6; wg_part == nullptr: none):
  const float* x;         // [L*B, kWgE] the layer's inputs (time-major)
  const float* hprev;     // [dirs][L][B][Hd] the state fed into the cell at time t
  float* wg_part;         // [dirs][row blocks per pass][4Hd][Hd + kWgE] the workgroups' partial sums
};
__global__ __launch_bounds__(256) void state_bm_to_db_kernel(const float* dh_bm, const float* dc_bm, float* dh, float* dc, int B, int dirs, int Hd) {
  const long n = (long)B * dirs * Hd;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % Hd), b = (int)((i / Hd) % B), d = (int)(i / ((long)Hd * B));
    const long src = ((long)b * dirs + d) * Hd + j;
    dh[i] = dh_bm[src];
    dc[i] = dc_bm[src];
  }
}

template <typename TW>
__global__ __launch_bounds__(256) void lstm_rec_bwd_kernel(RecBwdArgs a) {
  __shared__ float sp[4][16][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 15, fq = lane >> 4;
  const int j0 = blockIdx.x * 16, d = blockIdx.y, b0 = blockIdx.z * 16;
  const int Hd = a.Hd, B = a.B, L = a.L;
  const int t = (d == 0) ? a.step : (L - 1 - a.step);     // a.step counts DOWN on the host side
  const int G = a.dirs * 4 * Hd, Y = a.dirs * Hd;
  // pointwise operands first (their loads overlap the recurrent product)
  const int bl = threadIdx.x >> 4, jl = threadIdx.x & 15;
  const int b = b0 + bl, j = j0 + jl;
  const bool live = (b < B) && (j < Hd);
  const long ci = ((long)d * B + (live ? b : 0)) * Hd + (live ? j : 0);
  const long row = (long)t * B + (live ? b : 0);
  float dh = 0.f, dyv = 0.f, si = 0.f, sf = 0.f, tg = 0.f, so = 0.f, tc = 0.f, cp = 0.f, dcc = 0.f;
  int len = 0;
  if (live) {
    dh = a.dh_pass[ci];
    len = a.lengths[b];
    if (t < len) {
      if (a.dy) dyv = a.dy[row * Y + d * Hd + j];
      const float* ac = a.act + row * G + (long)d * 4 * Hd + j;
      si = ac[0]; sf = ac[Hd]; tg = ac[2 * Hd]; so = ac[3 * Hd];
      tc = a.tanh_c[row * Y + d * Hd + j];
      cp = a.cprev[(((long)d * L + t) * B + b) * Hd + j];
      dcc = a.dc_carry[ci];
    }
  }
  if (!a.first) {
    // dh_rec[b, j] = sum_k dgates[t_later][b, k] * W_hh[k, j]; wave w contracts gate block w
    const int tl = (d == 0) ? t + 1 : t - 1;
    const bool a_ok = (b0 + fi) < B, w_ok = (j0 + fi) < Hd;
    const float* arow = a.dgates + ((long)tl * B + (a_ok ? b0 + fi : 0)) * G + (long)d * 4 * Hd;
    const TW* wrow = reinterpret_cast<const TW*>(a.w_hh_t) + ((long)d * Hd + (w_ok ? j0 + fi : 0)) * 4 * Hd;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    wave_tile_16x16<TW>(arow, a_ok, wrow, w_ok, wave * Hd, (wave + 1) * Hd, fq, a.vec, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) sp[wave][fq * 4 + r][fi] = acc[r];
  }
  __syncthreads();
  if (!live) return;
  if (!a.first) dh += sp[0][bl][jl] + sp[1][bl][jl] + sp[2][bl][jl] + sp[3][bl][jl];
  float* dg = a.dgates + row * G + (long)d * 4 * Hd + j;
  if (t < len) {
    dh += dyv;
    const float dc = dcc + dh * so * (1.f - tc * tc);
    dg[0] = dc * tg * si * (1.f - si);
    dg[Hd] = dc * cp * sf * (1.f - sf);
    dg[2 * Hd] = dc * si * (1.f - tg * tg);
    dg[3 * Hd] = dh * tc * so * (1.f - so);
    a.dc_carry[ci] = dc * sf;
    a.dh_pass[ci] = 0.f;
  } else {
    dg[0] = 0.f; dg[Hd] = 0.f; dg[2 * Hd] = 0.f; dg[3 * Hd] = 0.f;
    a.dh_pass[ci] = dh;
  }
}

#include "wgrad_ride.h"
#include "encoder_persist.h"
#include "encoder_persist_g.h"

// ---- embedding gather (+dropout) to time-major rows, and its scatter-add backward ----------------------
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* tokens, const float* E, float* out, int B,
                                                        int L, int D, DropSpec dr, int vec) {
  if (vec == 2) {                                        // D % 8 == 0, aligned: 8 columns per thread = ONE Philox call
    const int D8 = D >> 3;
    const long total8 = (long)L * B * D8;
    for (long e8 = (long)blockIdx.x * blockDim.x + threadIdx.x; e8 < total8; e8 += (long)gridDim.x * blockDim.x) {
      const int c8 = (int)(e8 % D8);
      const long rb = e8 / D8;
      const int b = (int)(rb % B), t = (int)(rb / B);
      const long tok = tokens[(long)b * L + t];
      const float4 x0 = *reinterpret_cast<const float4*>(E + tok * D + c8 * 8), x1 = *reinterpret_cast<const float4*>(E + tok * D + c8 * 8 + 4);
      float m[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
      if (dr.p > 0.f) dropout_scale8(dr.seed, dr.off(), (uint32_t)(((long)b * L + t) * D8 + c8), dr.p, m);
      *reinterpret_cast<float4*>(out + e8 * 8) = make_float4(x0.x * m[0], x0.y * m[1], x0.z * m[2], x0.w * m[3]);
      *reinterpret_cast<float4*>(out + e8 * 8 + 4) = make_float4(x1.x * m[4], x1.y * m[5], x1.z * m[6], x1.w * m[7]);
    }
    return;
  }
  if (vec) {                                             // D % 4 == 0, aligned: float4 rows, half a Philox call per four
    const int D4 = D >> 2;
    const long total4 = (long)L * B * D4;
    for (long e4 = (long)blockIdx.x * blockDim.x + threadIdx.x; e4 < total4; e4 += (long)gridDim.x * blockDim.x) {
      const int c4 = (int)(e4 % D4);
      const long rb = e4 / D4;
      const int b = (int)(rb % B), t = (int)(rb / B);
      const long tok = tokens[(long)b * L + t];
      const float4 x = *reinterpret_cast<const float4*>(E + tok * D + c4 * 4);
      float m[4] = {1.f, 1.f, 1.f, 1.f};
      if (dr.p > 0.f) dropout_scale4(dr.seed, dr.off(), (uint32_t)(((long)b * L + t) * D4 + c4), dr.p, m);
      *reinterpret_cast<float4*>(out + e4 * 4) = make_float4(x.x * m[0], x.y * m[1], x.z * m[2], x.w * m[3]);
    }
    return;
  }
  const long total = (long)L * B * D;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % D);
    const long rb = e / D;
    const int b = (int)(rb % B), t = (int)(rb / B);
    const long tok = tokens[(long)b * L + t];
    const long idx = ((long)b * L + t) * D + c;           // mask index follows the [B,L,D] layout
    out[e] = E[tok * D + c] * dropout_scale1(dr.seed, dr.off(), (uint32_t)idx, dr.p);
  }
}
__global__ __launch_bounds__(256) void embed_bwd_kernel(const long long* tokens, const int* lengths, const float* dx,
                                                        float* dE, int B, int L, int D, long padding_idx,
                                                        DropSpec dr) {
  const long total = (long)L * B * D;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % D);
    const long rb = e / D;
    const int b = (int)(rb % B), t = (int)(rb / B);
    if (t >= lengths[b]) continue;                        // padded steps carry no gradient
    const long tok = tokens[(long)b * L + t];
    if (tok == padding_idx) continue;
    const long idx = ((long)b * L + t) * D + c;
    const float g = dx[e] * dropout_scale1(dr.seed, dr.off(), (uint32_t)idx, dr.p);
    atomicAdd(dE + tok * D + c, g);
  }
}

// Deterministic form (no float atomics; opt-in, ~5x the time of the atomic kernel): one workgroup per vocabulary row
// scans the B*L tokens through a ballot compaction and adds the matching positions' gradient rows in (t, b) order --
// with it a training iteration is reproducible bit for bit from run to run (every other reduction on the path already
// has a fixed order).
__global__ __launch_bounds__(256) void embed_bwd_det_kernel(const long long* tokens, const int* lengths, const float* dx,
                                                            float* dE, int B, int L, int D, long padding_idx, DropSpec dr) {
  __shared__ int s_hit[256];
  __shared__ int s_wcnt[4];
  const long v = blockIdx.x;
  if (v == padding_idx) return;
  const int P = L * B;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};               // columns threadIdx.x + k*256 (D <= 1024)
  for (int p0 = 0; p0 < P; p0 += 256) {
    const int p = p0 + threadIdx.x;                  // time-major position p = t*B + b
    bool hit = false;
    if (p < P) {
      const int t = p / B, b = p % B;
      hit = t < lengths[b] && tokens[(long)b * L + t] == v;
    }
    const unsigned long long bal = __ballot(hit);
    __syncthreads();                                 // the previous chunk's hits are consumed
    if (lane == 0) s_wcnt[wave] = __popcll(bal);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_wcnt[w];
    if (hit) s_hit[base + __popcll(bal & ((1ull << lane) - 1ull))] = p;
    const int n = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    __syncthreads();
    for (int i = 0; i < n; ++i) {
      const int q = s_hit[i], t = q / B, b = q % B;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = threadIdx.x + k * 256;
        if (c < D) {
          const long idx = ((long)b * L + t) * D + c;
          acc[k] += dx[(long)q * D + c] * dropout_scale1(dr.seed, dr.off(), (uint32_t)idx, dr.p);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = threadIdx.x + k * 256;
    if (c < D) dE[v * D + c] += acc[k];
  }
}

// ---- layout changes with fused dropout: time-major [L,B,W] <-> batch-major [B,L,W] -------------------------
// [L,B,W] <-> [B,L,W] with the context dropout (units.py:71-72): four consecutive columns per thread (one Philox call,
// 16-byte accesses) when W % 4 == 0 and the pointers are aligned (`vec`), else one element per thread.
__global__ __launch_bounds__(256) void tm_to_bm_kernel(const float* tm, float* bm, bf16_raw* bm_lp, int B, int L,
                                                       int W, DropSpec dr, int vec) {
  tm_to_bm_body(tm, bm, bm_lp, B, L, W, dr, vec, (long)blockIdx.x, (long)gridDim.x);       // layout_bodies.h
}
__global__ __launch_bounds__(256) void bm_to_tm_kernel(const float* bm, float* tm, int B, int L, int W, DropSpec dr, int vec) {
  bm_to_tm_body(bm, tm, B, L, W, dr, vec, (long)blockIdx.x, (long)gridDim.x);
}

static inline int nblk(long n, int cap = 4096) {
  long b = (n + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace vln

using namespace vln;

extern "C" int vln_embed_fwd(const int64_t* tokens, const float* E, float* out_tm, int B, int L, int D,
                             uint64_t seed, uint64_t offset, float p, const uint64_t* offset_base_dev, vln_stream_t s) {
  if (!tokens || !E || !out_tm || B <= 0 || L <= 0 || D <= 0) { set_error("vln_embed_fwd: bad args"); return VLN_ERR_ARG; }
  int vec = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(E) | reinterpret_cast<uintptr_t>(out_tm)) & 15) == 0;
  if (vec && D % 8 == 0) vec = 2;
  VLN_LAUNCH(embed_fwd_kernel, dim3(nblk(vec == 2 ? (long)B * L * D / 8 : vec ? (long)B * L * D / 4 : (long)B * L * D)), dim3(256), 0, (hipStream_t)s,
                     (const long long*)tokens, E, out_tm, B, L, D, drop_spec(seed, offset, p, offset_base_dev), vec);
  VLN_CHECK_LAUNCH("embed_fwd");
  return VLN_OK;
}
extern "C" int vln_embed_bwd(const int64_t* tokens, const int32_t* lengths, const float* dx_tm, float* dE, int B,
                             int L, int D, int64_t padding_idx, uint64_t seed, uint64_t offset, float p,
                             const uint64_t* offset_base_dev, vln_stream_t s) {
  if (!tokens || !lengths || !dx_tm || !dE) { set_error("vln_embed_bwd: null pointer"); return VLN_ERR_ARG; }
  VLN_LAUNCH(embed_bwd_kernel, dim3(nblk((long)B * L * D)), dim3(256), 0, (hipStream_t)s,
                     (const long long*)tokens, lengths, dx_tm, dE, B, L, D, (long)padding_idx, drop_spec(seed, offset, p, offset_base_dev));
  VLN_CHECK_LAUNCH("embed_bwd");
  return VLN_OK;
}
extern "C" int vln_embed_bwd_det(const int64_t* tokens, const int32_t* lengths, const float* dx_tm, float* dE, int B, int L,
                                 int D, int V, int64_t padding_idx, uint64_t seed, uint64_t offset, float p,
                                 const uint64_t* offset_base_dev, vln_stream_t s) {
  if (!tokens || !lengths || !dx_tm || !dE || V <= 0 || D <= 0 || D > 1024) { set_error("vln_embed_bwd_det: bad args (D <= 1024)"); return VLN_ERR_ARG; }
  VLN_LAUNCH(embed_bwd_det_kernel, dim3(V), dim3(256), 0, (hipStream_t)s, (const long long*)tokens, lengths, dx_tm, dE, B, L,
                     D, (long)padding_idx, drop_spec(seed, offset, p, offset_base_dev));
  VLN_CHECK_LAUNCH("embed_bwd_det");
  return VLN_OK;
}
extern "C" int vln_tm_to_bm(const float* tm, float* bm, void* bm_bf16, int B, int L, int W, uint64_t seed,
                            uint64_t offset, float p, const uint64_t* offset_base_dev, vln_stream_t s) {
  if (!tm || !bm) { set_error("vln_tm_to_bm: null pointer"); return VLN_ERR_ARG; }
  int vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(tm) | reinterpret_cast<uintptr_t>(bm) | reinterpret_cast<uintptr_t>(bm_bf16)) & 15) == 0;
  if (vec && W % 8 == 0) vec = 2;
  VLN_LAUNCH(tm_to_bm_kernel, dim3(nblk(vec == 2 ? (long)B * L * W / 8 : vec ? (long)B * L * W / 4 : (long)B * L * W)), dim3(256), 0, (hipStream_t)s, tm, bm,
                     (bf16_raw*)bm_bf16, B, L, W, drop_spec(seed, offset, p, offset_base_dev), vec);
  VLN_CHECK_LAUNCH("tm_to_bm");
  return VLN_OK;
}
extern "C" int vln_bm_to_tm(const float* bm, float* tm, int B, int L, int W, uint64_t seed, uint64_t offset, float p,
                            const uint64_t* offset_base_dev, vln_stream_t s) {
  if (!tm || !bm) { set_error("vln_bm_to_tm: null pointer"); return VLN_ERR_ARG; }
  int vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(tm) | reinterpret_cast<uintptr_t>(bm)) & 15) == 0;
  if (vec && W % 8 == 0) vec = 2;
  VLN_LAUNCH(bm_to_tm_kernel, dim3(nblk(vec == 2 ? (long)B * L * W / 8 : vec ? (long)B * L * W / 4 : (long)B * L * W)), dim3(256), 0, (hipStream_t)s, bm, tm, B,
                     L, W, drop_spec(seed, offset, p, offset_base_dev), vec);
  VLN_CHECK_LAUNCH("bm_to_tm");
  return VLN_OK;
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// initial states: time 0 of the forward direction, time L-1 of the reverse direction; zeros unless h0/c0 [dirs][B][Hd]
__global__ void copy_f32_kernel(const float* src, float* dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
static int seed_initial_state(hipStream_t st, const float* h0, const float* c0, float* hprev, float* cprev, int B, int L,
                              int Hd, int dirs) {
  const long blk = (long)B * Hd;
  for (int d = 0; d < dirs; ++d) {
    const long off = ((long)d * L + (d == 0 ? 0 : L - 1)) * blk;
    const float* src[2] = {h0 ? h0 + d * blk : nullptr, c0 ? c0 + d * blk : nullptr};
    float* dst[2] = {hprev + off, cprev + off};
    for (int k = 0; k < 2; ++k) {
      if (src[k]) {
        int nb = (int)((blk + 255) / 256); if (nb > 1024) nb = 1024;
        VLN_LAUNCH(copy_f32_kernel, dim3(nb), dim3(256), 0, st, src[k], dst[k], blk);
      } else {
        int r = fill_f32(st, dst[k], blk, 0.f); if (r) return r;
      }
    }
  }
  VLN_CHECK_LAUNCH("lstm initial state");
  return VLN_OK;
}

static int lstm_seq_fwd_issue(hipStream_t st, const float* xproj, const void* w_hh, int wtype, const int32_t* lengths,
                              float* hprev, float* cprev, float* y_tm, float* act, float* tanh_c, float* hcat,
                              float* ccat, int B, int L, int Hd, int dirs, const float* h0, const float* c0) {
  int r0 = seed_initial_state(st, h0, c0, hprev, cprev, B, L, Hd, dirs);
  if (r0) return r0;
  RecFwdArgs a{xproj, w_hh, lengths, hprev, cprev, y_tm, act, tanh_c, hcat, ccat, B, L, Hd, dirs, 0, 0, 0};
  a.vec = al16(w_hh) && al16(hprev) && (Hd % (wtype == VLN_BF16 ? 8 : 4) == 0) && (Hd % 4 == 0);
  dim3 grid((Hd + 15) / 16, dirs, (B + 15) / 16), block(256);
  // per-launch algorithmic bytes: W_hh once, h/c state in+out, xproj in, y/act/tanh_c out
  const double step_bytes = (double)dirs * (4.0 * Hd * Hd * (wtype == VLN_BF16 ? 2 : 4) + 4.0 * B * Hd * (4 + 4 + 1 + 4 + 1));
  for (int step = 0; step < L; ++step) {
    a.step = step;
    ProfScope prof(st, K_LSTM_REC_FWD, step_bytes);
    if (wtype == VLN_BF16) VLN_LAUNCH(lstm_rec_fwd_kernel<bf16_raw>, grid, block, 0, st, a);
    else VLN_LAUNCH(lstm_rec_fwd_kernel<float>, grid, block, 0, st, a);
  }
  VLN_CHECK_LAUNCH("lstm_rec_fwd");
  return VLN_OK;
}

// ---- persistent single-launch path (encoder_persist.h) ------------------------------------------------------
// 0 = per-step launch chain; 1 (default) = persistent kernels, forward with data-tagged granule hand-offs
// (encoder_persist_g.h), backward with the counter + payload hand-off (encoder_persist.h); 2 = both on the counter protocol
// (round 1); 3 = both on granules.  Measured on MI355X (scripts/lstm_probe, B=64, Hd=256, L=80, bf16, interleaved on one box,
// profiles/round2_lstm_probe.txt): forward 259 us (counter) -> 193 us (granules); backward 196 us (counter) vs 291 us
// (granules) -- its hand-off is a PARTIAL-SUM exchange of 32 KB per workgroup per step, and doubling those bytes with tags
// costs more than the removed drain + barrier + atomic.  All variants produce identical bits.
int g_persist_enabled = 1;
int vln::g_split_attn_enabled = 1;       // cleared when a four-workgroup attention exchange timed out (attention_split.h reads it)
// Which sync workspaces hold a header that the counter-protocol backward left CLEAN (group counters zero: it resets them
// itself).  Anything else -- a buffer this process has not launched on yet (the caller may not have zeroed it), a header the
// counter-protocol FORWARD or a mode switch touched -- gets the fill launch in front of the next counter-protocol backward.
static std::mutex g_hdr_mu;
static std::unordered_map<const void*, bool> g_hdr_clean;
static bool header_clean(const void* ws) {
  std::lock_guard<std::mutex> lock(g_hdr_mu);
  auto it = g_hdr_clean.find(ws);
  return it != g_hdr_clean.end() && it->second;
}
static void header_mark(const void* ws, bool clean) {
  std::lock_guard<std::mutex> lock(g_hdr_mu);
  g_hdr_clean[ws] = clean;
}
extern "C" int vln_lstm_sync_ws_forget(const void* sync_ws) {
  std::lock_guard<std::mutex> lock(g_hdr_mu);
  g_hdr_clean.erase(sync_ws);       // (the allocator may hand an old buffer's address to a new one: what is known about it is void)
  return VLN_OK;
}
extern "C" int vln_set_persistent(int on) {
  g_persist_enabled = (on >= 0 && on <= 3) ? on : 1;
  std::lock_guard<std::mutex> lock(g_hdr_mu);
  g_hdr_clean.clear();              // another protocol may run on the same buffers next: every header is dirty
  return VLN_OK;
}
static inline bool fwd_granules() { return g_persist_enabled == 1 || g_persist_enabled == 3; }
static inline bool bwd_granules() { return g_persist_enabled == 3; }

// Co-residency: every workgroup of the persistent grid spins on its neighbours, so the WHOLE grid must be resident at once.
// The kernels keep their W_hh slice in registers (about one workgroup per CU), so the capacity is the device's CU count --
// queried, not assumed: a partitioned (CPX) or smaller device takes the per-step chain instead of spinning into a timeout.
int vln::device_cus() {
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return 0; }
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    cus[dev] = n > 0 ? n : -1;
  }
  return cus[dev] > 0 ? cus[dev] : 0;
}
template <typename K>
static bool kernel_fits_one_per_cu(K kernel, int threads = 256) {      // the occupancy API's answer for this instantiation, cached by the caller
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
  return n >= 1;
}

// A bounded spin that timed out leaves wrong numbers behind.  The per-launch status word is zeroed by the next launch, so
// the kernels also count timeouts in a STICKY word that lives in host-mapped pinned memory (one per device; a system-scope
// atomic from the timeout path, i.e. no traffic at all in a healthy run); the next entry into the library (sequence
// forward / backward, optimizer step: vln_persistent_check) reports it ONCE as an error and switches this process to the
// per-step chain.
static unsigned* g_sticky_host = nullptr;       // pinned + mapped, one word per device (16 words apart)
static unsigned* g_sticky_dev = nullptr;        // the same memory as the device sees it
unsigned* vln::sticky_dev_word() {
  if (!g_sticky_host) {
    if (hipHostMalloc(reinterpret_cast<void**>(&g_sticky_host), 16 * 16 * sizeof(unsigned), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&g_sticky_dev), g_sticky_host, 0) != hipSuccess) {
      (void)hipGetLastError();
      g_sticky_host = nullptr; g_sticky_dev = nullptr;
      return nullptr;
    }
    for (int i = 0; i < 16 * 16; ++i) g_sticky_host[i] = 0u;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return nullptr; }
  return g_sticky_dev + dev * 16;
}
extern "C" int vln_debug_raise_sticky(int word) {      // test hook: what a timed-out recurrence wait (0) / a bad gather index (1) / a timed-out attention exchange (2) leaves behind
  if (word < 0 || word > 2 || !sticky_dev_word()) { set_error("vln_debug_raise_sticky: bad word / no host-mapped status line"); return VLN_ERR_ARG; }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return VLN_ERR_HIP; }
  __atomic_fetch_add(&g_sticky_host[dev * 16 + word], 1u, __ATOMIC_RELAXED);
  return VLN_OK;
}
extern "C" int vln_persistent_check(void) {
  // (the status line is created here if no launch has needed it yet: every graph capture starts with this call, and the pinned
  //  allocation must not happen INSIDE a capture -- a first vln_host_wait in a fresh process would invalidate it)
  if (!g_sticky_host) (void)sticky_dev_word();
  unsigned* h = g_sticky_host;
  if (!h) return VLN_OK;
  for (int d = 0; d < 16; ++d) {
    if (__atomic_load_n(&h[d * 16], __ATOMIC_RELAXED)) {
      const unsigned n = __atomic_exchange_n(&h[d * 16], 0u, __ATOMIC_RELAXED);
      g_persist_enabled = 0;
      set_error("%u bounded in-kernel wait(s) timed out on device %d in an EARLIER launch (persistent LSTM recurrence: its "
                "workgroups were not co-resident); that iteration's numbers are invalid.  The persistent recurrence is now off "
                "for this process (per-step launches)", n, d);
      return VLN_ERR_HIP;
    }
    if (__atomic_load_n(&h[d * 16 + 1], __ATOMIC_RELAXED)) {
      const unsigned n = __atomic_exchange_n(&h[d * 16 + 1], 0u, __ATOMIC_RELAXED);
      set_error("%u feature-gather index(es) were out of range of the registered table on device %d in an EARLIER launch "
                "(viewpoint row, view index or candidate view; or a row index of vln_select_rows_multi): those rows were gathered "
                "as zeros; that iteration's numbers are invalid", n, d);
      return VLN_ERR_ARG;
    }
    if (__atomic_load_n(&h[d * 16 + 2], __ATOMIC_RELAXED)) {
      const unsigned n = __atomic_exchange_n(&h[d * 16 + 2], 0u, __ATOMIC_RELAXED);
      g_split_attn_enabled = 0;
      set_error("%u bounded in-kernel wait(s) timed out on device %d in an EARLIER launch (four-workgroup attention: the "
                "partial-dot exchange of a batch row never completed); that iteration's numbers are invalid.  The split "
                "attention is now off for this process (one workgroup per batch row)", n, d);
      return VLN_ERR_HIP;
    }
    if (__atomic_load_n(&h[d * 16 + 3], __ATOMIC_RELAXED)) {
      const unsigned n = __atomic_exchange_n(&h[d * 16 + 3], 0u, __ATOMIC_RELAXED);
      set_error("%u wait(s) for the host timed out on device %d in an EARLIER launch (vln_host_wait: the host never wrote the flag "
                "of its turn); the steps behind it ran on whatever their inputs held: that iteration's numbers are invalid", n, d);
      return VLN_ERR_HIP;
    }
    if (__atomic_load_n(&h[d * 16 + 4], __ATOMIC_RELAXED)) {
      const unsigned n = __atomic_exchange_n(&h[d * 16 + 4], 0u, __ATOMIC_RELAXED);
      set_error("%u weight(s) of an input BatchNorm were exactly 0, or below 1/%d of their bias in magnitude, on device %d in an EARLIER "
                "launch (vln_bn0_grads_from_wgrad divides the first layer's weight gradient by them: a zero weight's d gamma was left 0, a "
                "tiny one's carries the weight gradient's rounding amplified by |beta| / |gamma|); form the gradients by the direct path "
                "instead (functional.set_bn0_grads_from_wgrad(False))", n, VLN_BN0_MAX_AMPLIFICATION, d);
      return VLN_ERR_ARG;
    }
  }
  return VLN_OK;
}
// Test hook: the cumulative tallies of the backward recurrence's hand-off decisions kept in a sync workspace's header (synchronous
// device-to-host copy): dependency groups whose launches verified one XCD (stores kept in its L2) / found several (write-through).
extern "C" int vln_lstm_handoff_stats(const void* sync_ws, uint32_t* xcd_local, uint32_t* spanning) {
  if (!sync_ws || !xcd_local || !spanning) { set_error("vln_lstm_handoff_stats: null pointer"); return VLN_ERR_ARG; }
  uint32_t w[2] = {0, 0};
  if (hipMemcpy(w, static_cast<const char*>(sync_ws) + (32 + 8) * 4, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    set_error("vln_lstm_handoff_stats: copy failed");
    return VLN_ERR_HIP;
  }
  *xcd_local = w[0]; *spanning = w[1];
  return VLN_OK;
}
extern "C" int vln_lstm_fwd_handoff_stats(const void* sync_ws, uint32_t* xcd_local, uint32_t* spanning) {
  if (!sync_ws || !xcd_local || !spanning) { set_error("vln_lstm_fwd_handoff_stats: null pointer"); return VLN_ERR_ARG; }
  uint32_t w[2] = {0, 0};
  if (hipMemcpy(w, static_cast<const char*>(sync_ws) + (32 + 10) * 4, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    set_error("vln_lstm_fwd_handoff_stats: copy failed");
    return VLN_ERR_HIP;
  }
  *xcd_local = w[0]; *spanning = w[1];
  return VLN_OK;
}
extern "C" int vln_set_split_attention(int on) { g_split_attn_enabled = on ? 1 : 0; return VLN_OK; }
extern "C" int vln_get_split_attention(void) { return g_split_attn_enabled; }

// A batch whose one-workgroup-per-(slice, direction, 16 rows) grid exceeds the device (B > 128 with two directions of 256 units on 256
// CUs) runs the SAME persistent launch in PASSES (round 6): the grid holds `nbb_per` row blocks per direction, and a workgroup that has
// finished the L steps of row block bb starts over on row block bb + nbb_per -- the rows of a batch never interact, every dependency
// group (direction, row block) keeps its own flag line and exchange region, the resident W_hh fragments are loaded once.  Returns the
// row blocks per pass (= all of them when the batch fits), 0 when even one row block per direction does not fit.  Only the default
// protocol pair takes passes (granules forward, counters backward): `passes_ok`.
static int persist_passes(int B, int Hd, int dirs, bool passes_ok) {
  const int nbb = (B + 15) / 16, cus = device_cus();
  if (cus <= 0) return 0;
  const int fit = cus / ((Hd / 16) * dirs);          // row blocks per direction that are co-resident
  if (fit <= 0) return 0;
  if (nbb <= fit) return nbb;
  if (!passes_ok || g_tunable[13] == 7) return 0;     // (tunable[13] = 7: never in passes -- the per-step chain instead, A/B)
  const int passes = (nbb + fit - 1) / fit;
  return (nbb + passes - 1) / passes;                 // balanced: B = 192 -> 2 x 6 row blocks, not 8 + 4
}
static bool persist_ok(int B, int L, int Hd, int dirs, const void* sync_ws, bool passes_ok = false) {
  if (!g_persist_enabled || !sync_ws) return false;
  if (Hd != 128 && Hd != 256 && Hd != 512) return false;
  if (persist_passes(B, Hd, dirs, passes_ok) <= 0 || dirs * ((B + 15) / 16) > 32) return false;   // every workgroup of a pass co-resident; 32 flag lines
  if ((long)L * B * dirs * 4 * Hd * 4 >= (1L << 32)) return false;       // 32-bit buffer offsets
  if ((fwd_granules() || bwd_granules()) && L > 255) return false;      // granule tags hold the step in 8 bits
  return true;
}

template <typename TW>
static int launch_persist_fwd(hipStream_t st, const RecFwdArgs& a, unsigned* counters, unsigned* status, dim3 grid) {
  unsigned* sticky = sticky_dev_word();
  if (!sticky) { set_error("persistent lstm: no host-mapped status word"); return VLN_ERR_HIP; }
  const dim3 g1(grid.x * grid.y * grid.z);          // one-dimensional: the kernel decodes (slice, direction, row block)
  const int xm = g_tunable[7] != 1;                 // tunable[7] = 1: dispatch-order mapping (A/B)
  constexpr int BK = RecCfg<TW>::BK;
#define VLN_PERSIST_FWD(NS_)                                                                                              \
  {                                                                                                                       \
    static const bool fits = kernel_fits_one_per_cu(lstm_persist_fwd_kernel<TW, NS_>);                                    \
    if (!fits) { set_error("persistent lstm fwd: the kernel does not fit one workgroup per CU on this device"); return VLN_ERR_HIP; } \
    VLN_LAUNCH((lstm_persist_fwd_kernel<TW, NS_>), g1, dim3(256), 0, st, a, counters, status, sticky, xm);                 \
  }                                                                                                                       \
  break
  switch (a.Hd / BK) {
    case 2: VLN_PERSIST_FWD(2);
    case 4: VLN_PERSIST_FWD(4);
    case 8: VLN_PERSIST_FWD(8);
    case 16: VLN_PERSIST_FWD(16);
    default: set_error("persistent lstm fwd: unsupported Hd"); return VLN_ERR_ARG;
  }
#undef VLN_PERSIST_FWD
  VLN_CHECK_LAUNCH("lstm_persist_fwd");
  return VLN_OK;
}

static int ride_passengers(int nrec);
// Recurrence workgroups on XCDs 0-3, passengers on XCDs 4-7 (persist_role, encoder_persist.h): when the recurrence's workgroups come
// in fours and fit half of the device's compute units -- ride_passengers() then allows as many passengers as the other half holds.
static bool ride_partitioned(int nrec) {
  return g_tunable[15] != 1 && nrec > 0 && (nrec & 3) == 0 && 2 * nrec <= device_cus();
}
// the shadow jobs a gather ride carries, as their own launch (no passengers to carry them)
static int ride_shadows_launch(hipStream_t st, const ::vln_gather_ride& r) {
  if (r.n_shadow_jobs < 0 || (r.n_shadow_jobs > 0 && !r.shadow_jobs)) { set_error("gather ride: bad shadow jobs"); return VLN_ERR_ARG; }
  return r.n_shadow_jobs > 0 ? shadow_refresh(st, r.shadow_jobs, r.n_shadow_jobs) : VLN_OK;
}
static constexpr unsigned kRideLdsClaim = 96u * 1024u;
template <typename TW>
static int launch_persist_bwd(hipStream_t st, const RecBwdArgs& a, unsigned* counters, unsigned* status, float* exch, dim3 grid,
                              const WgradRideArgs* ride) {
  unsigned* sticky = sticky_dev_word();
  if (!sticky) { set_error("persistent lstm: no host-mapped status word"); return VLN_ERR_HIP; }
  dim3 g1(grid.x * grid.y * grid.z);
  const int nrec = (int)g1.x;
  // bit 0: one XCD per dependency group; bit 1: the hand-off stores may stay in that XCD's L2 once the group has verified that it
  // does run on one XCD (encoder_persist.h; tunable[14] = 1: always write-through, A/B)
  int xm = (g_tunable[7] != 1 ? 1 : 0) | (g_tunable[14] != 1 ? 2 : 0);
  static const WgradRideArgs no_ride{};
  unsigned lds_claim = 0;
  int np = 0;
  if (ride) {
    // passengers on the CUs the recurrence leaves idle; the dynamic-LDS claim keeps the launch at ONE workgroup per CU (the
    // passengers' barrier needs all of them resident, and they must not share a CU with the latency-bound recurrence)
    // HALF as many passengers as recurrence workgroups by default: their traffic slows the recurrence's hand-offs, and the ride only
    // has to finish inside the launch (B = 64: 128 passengers 1.680 ms, 96 1.667, 48-80 1.665, 32 1.705 -- the ride outlasts the
    // BPTT --, own launches 1.698; profiles/round4_notes.md).  tunable[11] >= 8 sets the cap (A/B).  (Round 5, partitioned form below:
    // 64, 96 or 128 passengers cost the BPTT the same 184 us -- the cap no longer matters there.)
    np = ride_passengers(nrec) & ~7;
    const int cap = g_tunable[11] >= 8 ? (g_tunable[11] & ~7) : (nrec / 2 > 8 ? (nrec / 2) & ~7 : 8);
    if (np > cap) np = cap;
    if (np <= 0 || (nrec & 7)) { set_error("persistent lstm bwd: a gradient ride was handed to a launch with no room for passengers"); return VLN_ERR_ARG; }
    // the two kinds of workgroup on disjoint XCDs when the recurrence fits four of them (persist_role; tunable[15] = 1: interleaved, A/B)
    if (ride_partitioned(nrec)) { xm |= 4; g1.x = 2u * (unsigned)nrec; }
    else g1.x += (unsigned)np;
    lds_claim = kRideLdsClaim;
  }
  const WgradRideArgs& rd = ride ? *ride : no_ride;
#define VLN_PERSIST_BWD(NT_, WG_)                                                                                         \
  {                                                                                                                       \
    static const bool fits = kernel_fits_one_per_cu(lstm_persist_bwd_kernel<TW, NT_, WG_>, WG_ ? 512 : 256);              \
    if (!fits) { set_error("persistent lstm bwd: the kernel does not fit one workgroup per CU on this device"); return VLN_ERR_HIP; } \
    if (lds_claim) {                                                                                                      \
      static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_persist_bwd_kernel<TW, NT_, WG_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRideLdsClaim) == hipSuccess; \
      if (!ok) { (void)hipGetLastError(); set_error("persistent lstm bwd: the dynamic-LDS claim of the passenger launch was refused"); return VLN_ERR_HIP; } \
    }                                                                                                                     \
    VLN_LAUNCH((lstm_persist_bwd_kernel<TW, NT_, WG_>), g1, dim3(WG_ ? 512 : 256), lds_claim, st, a, counters, status, sticky, exch, xm, nrec, np, rd); \
  }                                                                                                                       \
  break
  if (a.wg_part) {      // the layer's own weight gradients inside the launch (wgrad_inlaunch_ok: Hd 256, bf16 weights)
    if constexpr (sizeof(TW) == 2) {
      if (a.Hd != 256) { set_error("persistent lstm bwd: the in-launch weight gradients take Hd = 256"); return VLN_ERR_ARG; }
      switch (0) { default: VLN_PERSIST_BWD(4, true); }
    } else {
      set_error("persistent lstm bwd: the in-launch weight gradients are the bf16 mode's"); return VLN_ERR_ARG;
    }
  } else {
    switch (a.Hd / 64) {
      case 2: VLN_PERSIST_BWD(2, false);
      case 4: VLN_PERSIST_BWD(4, false);
      case 8: VLN_PERSIST_BWD(8, false);
      default: set_error("persistent lstm bwd: unsupported Hd"); return VLN_ERR_ARG;
    }
  }
#undef VLN_PERSIST_BWD
  VLN_CHECK_LAUNCH("lstm_persist_bwd");
  return VLN_OK;
}

// sync_ws layout: kSyncHeaderBytes (status word, arrival-flag lines of the counter protocol; zeroed before every
// counter-protocol launch) | counter protocol: the backward's partial-dh exchange floats | granule protocol: forward
// exchange | granule protocol: backward exchange.  Granule regions are never cleared between launches: their tags carry a
// per-buffer launch sequence (persist_tag_base).
static long sync_off_gfwd(int B, int Hd, int dirs) { return kSyncHeaderBytes + persist_bwd_exchange_floats(B, Hd, dirs) * 4; }
static long sync_off_gbwd(int B, int Hd, int dirs) { return sync_off_gfwd(B, Hd, dirs) + persist_g_fwd_bytes(B, Hd, dirs); }
// ... | one 128-byte line: word 0 = the device-resident launch sequence of the granule protocol (device_seq form)
static long sync_off_seq(int B, int Hd, int dirs) { return sync_off_gbwd(B, Hd, dirs) + persist_g_bwd_bytes(B, Hd, dirs); }
extern "C" int64_t vln_lstm_sync_ws_bytes(int B, int Hd, int dirs) {
  if (B <= 0 || Hd <= 0 || dirs < 1) return kSyncHeaderBytes + 128;
  return sync_off_seq(B, Hd, dirs) + 128;
}
extern "C" int64_t vln_lstm_sync_seq_offset(int B, int Hd, int dirs) {
  if (B <= 0 || Hd <= 0 || dirs < 1) return -1;
  return sync_off_seq(B, Hd, dirs);
}
extern "C" int vln_lstm_sync_granule_range(int B, int Hd, int dirs, int64_t* offset, int64_t* bytes) {
  if (B <= 0 || Hd <= 0 || dirs < 1 || !offset || !bytes) { set_error("vln_lstm_sync_granule_range: bad args"); return VLN_ERR_ARG; }
  *offset = sync_off_gfwd(B, Hd, dirs);
  *bytes = persist_g_fwd_bytes(B, Hd, dirs) + persist_g_bwd_bytes(B, Hd, dirs);
  return VLN_OK;
}

// Tag base of the next granule-protocol launch on this buffer: (sequence << 8), sequence = 1, 2, ... per buffer address.
// A buffer starts zeroed (tags 0 never match a base >= 256); when the 24-bit sequence wraps the granule regions are cleared
// once, so no stale granule can carry a live tag.
static int persist_tag_base(hipStream_t st, void* sync_ws, long gran_off, long gran_bytes, unsigned* base_out) {
  static std::mutex mu;
  static std::unordered_map<const void*, unsigned> seq;
  std::lock_guard<std::mutex> lock(mu);
  unsigned& q = seq[sync_ws];
  if (++q >= (1u << 24)) {
    if (hipMemsetAsync(static_cast<char*>(sync_ws) + gran_off, 0, (size_t)gran_bytes, st) != hipSuccess) {
      (void)hipGetLastError();
      set_error("persistent lstm: clearing the granule exchange on sequence wrap failed");
      return VLN_ERR_HIP;
    }
    q = 1;
  }
  *base_out = q << 8;
  return VLN_OK;
}

// Passenger workgroups a recurrence launch of `nrec` workgroups can carry: the CUs it leaves idle (at most as many as it has
// workgroups itself), and only on a device whose workgroups may claim kRideLdsClaim bytes of dynamic LDS (the claim keeps the
// launch at one workgroup per CU).  0 = the gather must be its own launch: the CALLER decides before it commits to passengers.
static int ride_passengers(int nrec) {
  static const int max_lds = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return v;
  }();
  if (max_lds < (int)(kRideLdsClaim + 24u * 1024u)) return 0;
  int np = device_cus() - nrec;
  if (np > nrec) np = nrec;
  return np >= 8 ? np : 0;
}

template <typename TW>
static int launch_persist_g_fwd(hipStream_t st, const RecFwdArgs& a, unsigned* status, unsigned char* exch, unsigned tag_base, dim3 grid,
                                const unsigned* seq_dev, unsigned seq_rel, const GatherRolloutArgs* ride, const FetchPart& fetch, const RideShadows& shadows) {
  unsigned* sticky = sticky_dev_word();
  if (!sticky) { set_error("persistent lstm: no host-mapped status word"); return VLN_ERR_HIP; }
  static_assert(sizeof(RecFwdArgs) + sizeof(GatherRolloutArgs) + sizeof(FetchPart) + sizeof(RideShadows) + 96 <= 4096,
                "the forward recurrence launch's arguments must fit the 4 KB a launch may pass");
  dim3 g1(grid.x * grid.y * grid.z);
  const int nrec = (int)g1.x;
  // bit 0: one XCD per dependency group; bit 1: the granule stores may stay in that XCD's L2 once the group has verified that it runs
  // on one XCD (encoder_persist_g.h; tunable[14] = 1: always write-through, A/B)
  int xm = (g_tunable[7] != 1 ? 1 : 0) | (g_tunable[14] != 1 ? 2 : 0);
  constexpr int BK = RecCfg<TW>::BK;
  static const GatherRolloutArgs no_ride{};
  unsigned lds_claim = 0;
  int np = 0;
  if (ride) {
    // passengers on the CUs the recurrence leaves idle (at most as many as it has workgroups); 96 KB of dynamic LDS on top of
    // the kernel's own ~20 KB: ONE workgroup per compute unit, so the two kinds never share a CU
    np = ride_passengers(nrec);
    if (np <= 0) { set_error("persistent lstm fwd: a gather ride was handed to a launch with no room for passengers (caller must check ride_passengers)"); return VLN_ERR_ARG; }
    // the two kinds of workgroup on disjoint XCDs when the recurrence fits four of them (persist_role; tunable[15] = 1: interleaved, A/B)
    if (ride_partitioned(nrec)) { xm |= 4; g1.x = 2u * (unsigned)nrec; }
    else g1.x += (unsigned)np;
    lds_claim = kRideLdsClaim;
  }
  const GatherRolloutArgs& rd = ride ? *ride : no_ride;
#define VLN_PERSIST_GF(NS_, XP_)                                                                                          \
  {                                                                                                                       \
    static const bool fits = kernel_fits_one_per_cu(lstm_persist_g_fwd_kernel<TW, NS_, XP_>, XP_ ? 512 : 256);            \
    if (!fits) { set_error("persistent lstm fwd: the kernel does not fit one workgroup per CU on this device"); return VLN_ERR_HIP; } \
    if (lds_claim) {                                                                                                      \
      static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_persist_g_fwd_kernel<TW, NS_, XP_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRideLdsClaim) == hipSuccess; \
      if (!ok) { (void)hipGetLastError(); set_error("persistent lstm fwd: the dynamic-LDS claim of the passenger launch was refused"); return VLN_ERR_HIP; } \
    }                                                                                                                     \
    VLN_LAUNCH((lstm_persist_g_fwd_kernel<TW, NS_, XP_>), g1, dim3(XP_ ? 512 : 256), lds_claim, st, a, status, sticky, exch, tag_base, xm, seq_dev, seq_rel, nrec, np, rd, \
               (ride ? fetch : FetchPart{}), (ride ? shadows : RideShadows{}));                                                   \
  }                                                                                                                       \
  break
  if (a.x) {       // the input projection inside the recurrence (inproj_ok: Hd 256 / 512, kInprojE input features)
    // (Hd = 256 only: at 512 the two wave groups' 256-register budget spills -- 29 VGPRs in bf16, 124 in fp32)
    if (a.Hd != 256) { set_error("persistent lstm fwd: the in-kernel input projection takes Hd = 256"); return VLN_ERR_ARG; }
    switch (0) { default: VLN_PERSIST_GF(256 / BK, true); }
  } else {
    switch (a.Hd / BK) {
      case 2: VLN_PERSIST_GF(2, false);
      case 4: VLN_PERSIST_GF(4, false);
      case 8: VLN_PERSIST_GF(8, false);
      case 16: VLN_PERSIST_GF(16, false);
      default: set_error("persistent lstm fwd: unsupported Hd"); return VLN_ERR_ARG;
    }
  }
#undef VLN_PERSIST_GF
  VLN_CHECK_LAUNCH("lstm_persist_g_fwd");
  return VLN_OK;
}

template <typename TW>
static int launch_persist_g_bwd(hipStream_t st, const RecBwdArgs& a, unsigned* status, unsigned char* exch, unsigned tag_base, dim3 grid,
                                const unsigned* seq_dev, unsigned seq_rel) {
  unsigned* sticky = sticky_dev_word();
  if (!sticky) { set_error("persistent lstm: no host-mapped status word"); return VLN_ERR_HIP; }
  const dim3 g1(grid.x * grid.y * grid.z);
  const int xm = g_tunable[7] != 1;
#define VLN_PERSIST_GB(NT_)                                                                                               \
  {                                                                                                                       \
    static const bool fits = kernel_fits_one_per_cu(lstm_persist_g_bwd_kernel<TW, NT_>);                                  \
    if (!fits) { set_error("persistent lstm bwd: the kernel does not fit one workgroup per CU on this device"); return VLN_ERR_HIP; } \
    VLN_LAUNCH((lstm_persist_g_bwd_kernel<TW, NT_>), g1, dim3(256), 0, st, a, status, sticky, exch, tag_base, xm, seq_dev, seq_rel); \
  }                                                                                                                       \
  break
  switch (a.Hd / 64) {
    case 2: VLN_PERSIST_GB(2);
    case 4: VLN_PERSIST_GB(4);
    case 8: VLN_PERSIST_GB(8);
    default: set_error("persistent lstm bwd: unsupported Hd"); return VLN_ERR_ARG;
  }
#undef VLN_PERSIST_GB
  VLN_CHECK_LAUNCH("lstm_persist_g_bwd");
  return VLN_OK;
}

// Whether vln_lstm_seq_fwd_x forms the input projection INSIDE the persistent recurrence for this shape on this device (round 6):
// the granule-protocol forward launch, Hd 256, E = kInprojE input features.  tunable[13] = 5: never (A/B: the GEMM launch).
static bool inproj_ok(int B, int L, int Hd, int dirs, int E, const void* sync_ws) {
  return g_tunable[13] != 5 && fwd_granules() && E == kInprojE && Hd == 256 && persist_ok(B, L, Hd, dirs, sync_ws, true);
}
extern "C" int vln_lstm_inproj_ok(int B, int L, int Hd, int dirs, int E, const void* sync_ws, int64_t sync_ws_bytes) {
  return (inproj_ok(B, L, Hd, dirs, E, sync_ws) && sync_ws_bytes >= vln_lstm_sync_ws_bytes(B, Hd, dirs) && al16(sync_ws)) ? 1 : 0;
}
static int lstm_seq_fwd_impl(const float* xproj, const float* x, const void* w_ih, const float* bsum, const void* w_hh, int wtype,
                             const int32_t* lengths, float* hprev, float* cprev, float* y_tm, float* act, float* tanh_c, float* hcat, float* ccat,
                             int B, int L, int Hd, int dirs, const float* h0, const float* c0, void* sync_ws, int64_t sync_ws_bytes,
                             int64_t device_seq, const vln_gather_ride* ride, vln_stream_t s);
extern "C" int vln_lstm_seq_fwd(const float* xproj, const void* w_hh, int wtype, const int32_t* lengths, float* hprev,
                                float* cprev, float* y_tm, float* act, float* tanh_c, float* hcat, float* ccat, int B,
                                int L, int Hd, int dirs, const float* h0, const float* c0, void* sync_ws,
                                int64_t sync_ws_bytes, int64_t device_seq, const vln_gather_ride* ride, vln_stream_t s) {
  return lstm_seq_fwd_impl(xproj, nullptr, nullptr, nullptr, w_hh, wtype, lengths, hprev, cprev, y_tm, act, tanh_c, hcat, ccat, B, L, Hd, dirs,
                           h0, c0, sync_ws, sync_ws_bytes, device_seq, ride, s);
}
extern "C" int vln_lstm_seq_fwd_x(const float* x, int E, const void* w_ih, const float* bsum, const void* w_hh, int wtype,
                                  const int32_t* lengths, float* hprev, float* cprev, float* y_tm, float* act, float* tanh_c, float* hcat,
                                  float* ccat, int B, int L, int Hd, int dirs, const float* h0, const float* c0, void* sync_ws,
                                  int64_t sync_ws_bytes, int64_t device_seq, const vln_gather_ride* ride, vln_stream_t s) {
  if (!x || !w_ih || !bsum || !al16(x) || !al16(w_ih)) { set_error("vln_lstm_seq_fwd_x: null or misaligned x / w_ih / bsum"); return VLN_ERR_ARG; }
  if (!vln_lstm_inproj_ok(B, L, Hd, dirs, E, sync_ws, sync_ws_bytes)) {
    set_error("vln_lstm_seq_fwd_x: this shape does not take the in-kernel input projection (ask vln_lstm_inproj_ok; B=%d L=%d Hd=%d E=%d)", B, L, Hd, E);
    return VLN_ERR_ARG;
  }
  return lstm_seq_fwd_impl(nullptr, x, w_ih, bsum, w_hh, wtype, lengths, hprev, cprev, y_tm, act, tanh_c, hcat, ccat, B, L, Hd, dirs, h0, c0,
                           sync_ws, sync_ws_bytes, device_seq, ride, s);
}
static int lstm_seq_fwd_impl(const float* xproj, const float* x, const void* w_ih, const float* bsum, const void* w_hh, int wtype,
                             const int32_t* lengths, float* hprev, float* cprev, float* y_tm, float* act, float* tanh_c, float* hcat, float* ccat,
                             int B, int L, int Hd, int dirs, const float* h0, const float* c0, void* sync_ws, int64_t sync_ws_bytes,
                             int64_t device_seq, const vln_gather_ride* ride, vln_stream_t s) {
  if ((!xproj && !x) || !w_hh || !lengths || !hprev || !cprev || !y_tm || !act || !tanh_c || !hcat || !ccat || B <= 0 ||
      L <= 0 || Hd <= 0 || dirs < 1 || dirs > 2) { set_error("vln_lstm_seq_fwd: bad args"); return VLN_ERR_ARG; }
  if (persist_ok(B, L, Hd, dirs, sync_ws, fwd_granules()) && sync_ws_bytes >= vln_lstm_sync_ws_bytes(B, Hd, dirs) && al16(w_hh) && al16(hprev) &&
      al16(sync_ws)) {
    hipStream_t st = (hipStream_t)s;
    int r = vln_persistent_check(); if (r) return r;
    if (!fwd_granules()) {    // counter protocol: status word [32] + flag lines from word 64 start at zero (the granule kernel
      r = fill_f32(st, (float*)sync_ws, kSyncHeaderBytes / 4, 0.f);        // clears its own status word and has no counters)
      if (r) return r;
    }
    const int init = (h0 || c0) ? 1 : 0;
    if (init) { r = seed_initial_state(st, h0, c0, hprev, cprev, B, L, Hd, dirs); if (r) return r; }
    const int nbbp = persist_passes(B, Hd, dirs, fwd_granules());       // row blocks per pass (persist_ok: > 0)
    RecFwdArgs a{xproj, w_hh, lengths, hprev, cprev, y_tm, act, tanh_c, hcat, ccat, B, L, Hd, dirs, 0, 1, init, nbbp < (B + 15) / 16 ? nbbp : 0,
                 x, w_ih, bsum};
    dim3 grid(Hd / 16, dirs, nbbp);
    unsigned* cw = (unsigned*)sync_ws;
    // algorithmic bytes of the whole sequence: W_hh ONCE (register-resident), per step state/xproj/outputs
    unsigned tag_base = 0;
    unsigned char* gex = static_cast<unsigned char*>(sync_ws) + sync_off_gfwd(B, Hd, dirs);
    // the ride: as passengers of the granule-protocol launch when it fits one argument block, else its own launch(es) first
    GatherRolloutArgs ride_args{};
    const GatherRolloutArgs* riders = nullptr;
    FetchPart fetch{};
    RideShadows shadows{};
    if (ride) { r = fetch_part_args(*ride, &fetch); if (r) return r; }
    if (ride) {
      // passengers need idle CUs (B = 128 with two directions of 256 units fills all 256) and the 96 KB dynamic-LDS claim
      if (fwd_granules() && ride->T <= kGatherMaxSteps && g_tunable[6] != 3 &&       // tunable[6] = 3: never as passengers (A/B)
          ride_passengers((int)(grid.x * grid.y * grid.z)) > 0) {
        r = gather_ride_args(*ride, 0, &ride_args); if (r) return r;
        riders = &ride_args;
        // the shadows the ride carries: the passengers' when one argument block holds the jobs, else their own launch first
        if (ride->n_shadow_jobs > 0 && ride->n_shadow_jobs <= kRideShadowJobs) {
          r = shadow_jobs(ride->shadow_jobs, ride->n_shadow_jobs, &shadows.jobs, &shadows.tiles); if (r) return r;
        } else {
          r = ride_shadows_launch(st, *ride); if (r) return r;
        }
      } else {
        r = gather_ride_launch(st, *ride); if (r) return r;
        r = launch_fetch_part(st, fetch); if (r) return r;          // the batch tail as its own one-block launch
        r = ride_shadows_launch(st, *ride); if (r) return r;        // and the shadows as theirs
      }
    }
    const unsigned* seq_dev = device_seq >= 0 ? reinterpret_cast<const unsigned*>(static_cast<char*>(sync_ws) + sync_off_seq(B, Hd, dirs)) : nullptr;
    if (fwd_granules() && !seq_dev) {
      r = persist_tag_base(st, sync_ws, sync_off_gfwd(B, Hd, dirs), persist_g_fwd_bytes(B, Hd, dirs) + persist_g_bwd_bytes(B, Hd, dirs), &tag_base);
      if (r) return r;
    }
    {
      // algorithmic bytes: the resident weights once, per (step, row, unit) the gate inputs (xproj, 4 floats -- or, with the projection
      // inside the launch, the row's E inputs once per direction and W_ih once) + activations / states / outputs written (10 floats)
      const double wb = wtype == VLN_BF16 ? 2 : 4;
      ProfScope prof(st, K_LSTM_REC_FWD, x ? (double)dirs * (4.0 * Hd * (Hd + kInprojE) * wb + (double)L * 4.0 * B * (kInprojE + Hd * 10.0))
                                           : (double)dirs * (4.0 * Hd * Hd * wb + (double)L * 4.0 * B * Hd * (4 + 4 + 1 + 4 + 1)));
      if (fwd_granules())
        r = (wtype == VLN_BF16) ? launch_persist_g_fwd<bf16_raw>(st, a, cw + 32, gex, tag_base, grid, seq_dev, (unsigned)device_seq, riders, fetch, shadows)
                                : launch_persist_g_fwd<float>(st, a, cw + 32, gex, tag_base, grid, seq_dev, (unsigned)device_seq, riders, fetch, shadows);
      else
        r = (wtype == VLN_BF16) ? launch_persist_fwd<bf16_raw>(st, a, cw + 64, cw + 32, grid)
                                : launch_persist_fwd<float>(st, a, cw + 64, cw + 32, grid);
    }
    if (!fwd_granules()) header_mark(sync_ws, false);       // the counter-protocol forward leaves its counters behind
    return r;
  }
  if (!xproj) { set_error("vln_lstm_seq_fwd_x: the persistent path was refused after vln_lstm_inproj_ok said yes (mode switched?)"); return VLN_ERR_ARG; }
  if (ride) {
    int rr = gather_ride_launch((hipStream_t)s, *ride); if (rr) return rr;
    FetchPart fetch{};
    rr = fetch_part_args(*ride, &fetch); if (rr) return rr;
    rr = launch_fetch_part((hipStream_t)s, fetch); if (rr) return rr;
    rr = ride_shadows_launch((hipStream_t)s, *ride); if (rr) return rr;
  }
  // the L-launch chain is a pure function of this argument block -> memoised as a hipGraph (graph_cache.h)
  struct { const void* p[12]; int v[5]; } key = {{xproj, w_hh, lengths, hprev, cprev, y_tm, act, tanh_c, hcat, ccat, h0, c0},
                                                 {wtype, B, L, Hd, dirs}};
  static GraphCache cache;
  return cache.run((hipStream_t)s, &key, sizeof(key), [&](hipStream_t st) {
    return lstm_seq_fwd_issue(st, xproj, w_hh, wtype, lengths, hprev, cprev, y_tm, act, tanh_c, hcat, ccat, B, L, Hd, dirs, h0, c0);
  });
}

// ---- gradient rides (wgrad_ride.h) ------------------------------------------------------------------------------------
// vln_wgrad_ride_post leaves a module's grouped weight / bias gradient jobs PENDING on a stream; the next vln_lstm_seq_bwd on
// that stream carries them as passengers of its persistent launch when it can (counter-protocol kernel, idle CUs, jobs the
// packed kernels take without a reduce launch) and otherwise issues them as their own launches first; vln_wgrad_ride_flush
// issues whatever is still pending (no backward recurrence followed).  Either way the gradients are final when the stream
// reaches the end of that call.
struct PendingRide {
  vln_wgrad_job w[kRideWgradJobs]; vln_colsum_job c[kRideColsumJobs];
  int nw = 0, nc = 0, rows = 0, precision = 0; float* ws = nullptr; long ws_floats = 0;
};
static std::mutex g_ride_mu;
static std::unordered_map<hipStream_t, PendingRide> g_rides;
static int64_t g_ride_stats[2] = {0, 0};       // rides carried as passengers, rides issued as their own launches
static bool ride_take(hipStream_t st, PendingRide* out) {
  std::lock_guard<std::mutex> lock(g_ride_mu);
  auto it = g_rides.find(st);
  if (it == g_rides.end()) return false;
  *out = it->second;
  g_rides.erase(it);
  return true;
}
static int ride_issue_alone(hipStream_t st, const PendingRide& r) {
  int rc = wgrad_grouped(st, r.w, r.nw, r.rows, r.precision, r.ws, r.ws_floats);
  if (rc) return rc;
  if (r.nc) rc = colsum_grouped(st, r.c, r.nc, r.rows, r.ws, r.ws_floats);
  { std::lock_guard<std::mutex> lock(g_ride_mu); g_ride_stats[1]++; }
  return rc;
}
extern "C" int vln_wgrad_ride_flush(vln_stream_t s) {
  PendingRide r;
  if (!ride_take((hipStream_t)s, &r)) return VLN_OK;
  return ride_issue_alone((hipStream_t)s, r);
}
extern "C" int vln_wgrad_ride_add(const vln_wgrad_job* job, int rows, int precision, vln_stream_t s) {
  if (!job || rows <= 0 || !job->dy || !job->x || !job->dw) return 0;
  std::lock_guard<std::mutex> lock(g_ride_mu);
  auto it = g_rides.find((hipStream_t)s);
  if (it == g_rides.end()) return 0;
  PendingRide& r = it->second;
  if (r.precision != precision || r.nw >= kRideWgradJobs || rows > r.rows) return 0;
  vln_wgrad_job q = *job;
  q.rows = rows < r.rows ? rows : 0;
  r.w[r.nw] = q;
  if (wgrad_grouped_ws_floats(r.w, r.nw + 1, r.rows) > r.ws_floats) return 0;      // the pack area would not hold its operands
  r.nw++;
  return 1;
}
// Forget a pending ride WITHOUT issuing it (its iteration was abandoned: a backward pass that raised never reached the carrying
// launch or the flush, and the jobs' operands belong to a dead rollout).  Returns 1 if one was pending.
extern "C" int vln_wgrad_ride_drop(vln_stream_t s) {
  PendingRide r;
  return ride_take((hipStream_t)s, &r) ? 1 : 0;
}
extern "C" int vln_wgrad_ride_post(const vln_wgrad_job* jobs, int n_jobs, const vln_colsum_job* cjobs, int n_cjobs, int rows, int precision,
                                   float* ws, int64_t ws_floats, vln_stream_t s) {
  if (!jobs || n_jobs <= 0 || n_jobs > kRideWgradJobs || n_cjobs < 0 || n_cjobs > kRideColsumJobs || (n_cjobs && !cjobs) || rows <= 0 ||
      precision < 0 || precision > 2) {
    set_error("vln_wgrad_ride_post: bad args (at most %d products and %d column sums)", kRideWgradJobs, kRideColsumJobs); return VLN_ERR_ARG;
  }
  int rc = vln_wgrad_ride_flush(s);          // one pending ride per stream: an older one goes out now
  if (rc) return rc;
  PendingRide r;
  for (int i = 0; i < n_jobs; ++i) r.w[i] = jobs[i];
  for (int i = 0; i < n_cjobs; ++i) r.c[i] = cjobs[i];
  r.nw = n_jobs; r.nc = n_cjobs; r.rows = rows; r.precision = precision; r.ws = ws; r.ws_floats = (long)ws_floats;
  std::lock_guard<std::mutex> lock(g_ride_mu);
  g_rides[(hipStream_t)s] = r;
  return VLN_OK;
}
extern "C" int vln_wgrad_ride_stats(int64_t out[2]) {
  if (!out) { set_error("vln_wgrad_ride_stats: null pointer"); return VLN_ERR_ARG; }
  std::lock_guard<std::mutex> lock(g_ride_mu);
  out[0] = g_ride_stats[0]; out[1] = g_ride_stats[1];
  return VLN_OK;
}

// Fallback producer of vln_lstm_seq_bwd's `bias_partials` for the recurrence forms that do not accumulate them themselves (granule
// protocol, per-step launches): the same [dirs][ceil(B / 16)][4 * Hd] layout from the finished dgates.
__global__ __launch_bounds__(256) void bias_part_from_dgates_kernel(const float* dgates, float* part, int B, int L, int Hd, int dirs) {
  const int nbb = (B + 15) / 16, G4 = 4 * Hd;
  const int d = (int)blockIdx.x / nbb, bb = (int)blockIdx.x % nbb;
  const int c = (int)blockIdx.y * 256 + (int)threadIdx.x;
  if (c >= G4) return;
  float t = 0.f;
  for (int s = 0; s < L; ++s)
    for (int r = 0; r < 16; ++r) {
      const int b = bb * 16 + r;
      if (b < B) t += dgates[((long)s * B + b) * (dirs * G4) + (long)d * G4 + c];
    }
  part[((long)d * nbb + bb) * G4 + c] = t;
}
static int bias_part_fallback(hipStream_t st, const float* dgates, float* part, int B, int L, int Hd, int dirs) {
  if (!part) return VLN_OK;
  VLN_LAUNCH(bias_part_from_dgates_kernel, dim3(dirs * ((B + 15) / 16), (4 * Hd + 255) / 256), dim3(256), 0, st, dgates, part, B, L, Hd, dirs);
  VLN_CHECK_LAUNCH("bias_part_from_dgates");
  return VLN_OK;
}

static int lstm_seq_bwd_issue(hipStream_t st, const float* dy_tm, const void* w_hh_t, int wtype, const int32_t* lengths,
                              const float* act, const float* tanh_c, const float* cprev, float* dgates, float* dh_pass,
                              float* dc_carry, int B, int L, int Hd, int dirs, const float* dh_bm, const float* dc_bm) {
  RecBwdArgs a{dy_tm, w_hh_t, lengths, act, tanh_c, cprev, dgates, dh_pass, dc_carry, B, L, Hd, dirs, 0, 1, 0, nullptr, nullptr, nullptr};
  if (dh_bm) {          // per-step launches keep their running state in dh_pass / dc_carry: bring the initial values into that layout
    long nb = ((long)B * dirs * Hd + 255) / 256;
    VLN_LAUNCH(state_bm_to_db_kernel, dim3((unsigned)(nb > 1024 ? 1024 : nb)), dim3(256), 0, st, dh_bm, dc_bm, dh_pass, dc_carry, B, dirs, Hd);
  }
  a.vec = al16(w_hh_t) && al16(dgates) && (Hd % (wtype == VLN_BF16 ? 8 : 4) == 0) && (Hd % 4 == 0);
  dim3 grid((Hd + 15) / 16, dirs, (B + 15) / 16), block(256);
  for (int step = L - 1; step >= 0; --step) {
    a.step = step;
    a.first = (step == L - 1);
    ProfScope prof(st, K_LSTM_REC_BWD, (double)dirs * (4.0 * Hd * Hd * (wtype == VLN_BF16 ? 2 : 4) + 4.0 * B * Hd * (4 + 4 + 4 + 1 + 1 + 1 + 4)));
    if (wtype == VLN_BF16) VLN_LAUNCH(lstm_rec_bwd_kernel<bf16_raw>, grid, block, 0, st, a);
    else VLN_LAUNCH(lstm_rec_bwd_kernel<float>, grid, block, 0, st, a);
  }
  VLN_CHECK_LAUNCH("lstm_rec_bwd");
  return VLN_OK;
}

// Whether vln_lstm_seq_bwd_w accumulates the layer's own weight gradients INSIDE the persistent BPTT launch for this shape (round 6):
// the counter-protocol launch, bf16-streamed weights, Hd = 256, E = kWgE inputs, the plain-bf16 weight-gradient precision (the
// process default: ops.set_wgrad_precision("bf16")).  tunable[13] = 6: never (A/B: the pack + contraction launches).
extern "C" int vln_lstm_wgrad_inlaunch_ok(int B, int L, int Hd, int dirs, int E, int wtype, int precision, const void* sync_ws, int64_t sync_ws_bytes) {
  return (g_tunable[13] != 6 && !bwd_granules() && wtype == VLN_BF16 && precision == 2 && Hd == 256 && E == kWgE &&
          persist_ok(B, L, Hd, dirs, sync_ws, true) && sync_ws_bytes >= vln_lstm_sync_ws_bytes(B, Hd, dirs) && al16(sync_ws)) ? 1 : 0;
}
extern "C" int64_t vln_lstm_wgrad_part_floats(int B, int Hd, int dirs, int E) {
  const int nbbp = persist_passes(B, Hd, dirs, true);
  return nbbp <= 0 ? 0 : (int64_t)dirs * nbbp * 4 * Hd * (Hd + E);
}
namespace vln {
// partial sums of the in-launch weight gradients -> the gradients: out_hh[d] [4Hd, Hd] (+)= sum_pb part[d][pb][:, :Hd], out_ih[d]
// [4Hd, E] (+)= sum_pb part[d][pb][:, Hd:], the row blocks' partials added in order
struct WgReduce { const float* part; float* out_hh[2]; float* out_ih[2]; int acc_hh[2], acc_ih[2]; int nparts, Hd, E, dirs; };
__global__ __launch_bounds__(256) void lstm_wgrad_reduce_kernel(WgReduce a) {
  const int W = a.Hd + a.E, W4 = W / 4;
  const long n4 = (long)a.dirs * 4 * a.Hd * W4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % W4);
    const long rr = i / W4;
    const int row = (int)(rr % (4 * a.Hd)), d = (int)(rr / (4 * a.Hd));
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < a.nparts; ++p) {
      const float4 v = *reinterpret_cast<const float4*>(a.part + (((long)d * a.nparts + p) * 4 * a.Hd + row) * W + c4 * 4);
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    const int c = c4 * 4;
    float* o; int acc;
    if (c < a.Hd) { o = a.out_hh[d] ? a.out_hh[d] + (long)row * a.Hd + c : nullptr; acc = a.acc_hh[d]; }
    else { o = a.out_ih[d] ? a.out_ih[d] + (long)row * a.E + (c - a.Hd) : nullptr; acc = a.acc_ih[d]; }
    if (!o) continue;
    if (acc) { const float4 g = *reinterpret_cast<const float4*>(o); sum.x += g.x; sum.y += g.y; sum.z += g.z; sum.w += g.w; }
    *reinterpret_cast<float4*>(o) = sum;
  }
}
}  // namespace vln
extern "C" int vln_lstm_wgrad_reduce(const float* part, int B, int Hd, int dirs, int E, float* const* out_hh, float* const* out_ih,
                                     const int* acc_hh, const int* acc_ih, vln_stream_t s) {
  if (!part || !out_hh || !out_ih || !acc_hh || !acc_ih || dirs < 1 || dirs > 2 || (Hd & 3) || (E & 3)) { set_error("vln_lstm_wgrad_reduce: bad args"); return VLN_ERR_ARG; }
  WgReduce a{};
  a.part = part; a.nparts = persist_passes(B, Hd, dirs, true); a.Hd = Hd; a.E = E; a.dirs = dirs;
  if (a.nparts <= 0) { set_error("vln_lstm_wgrad_reduce: this batch takes no persistent launch"); return VLN_ERR_ARG; }
  for (int d = 0; d < dirs; ++d) {
    a.out_hh[d] = out_hh[d]; a.out_ih[d] = out_ih[d]; a.acc_hh[d] = acc_hh[d]; a.acc_ih[d] = acc_ih[d];
    if ((out_hh[d] && !al16(out_hh[d])) || (out_ih[d] && !al16(out_ih[d]))) { set_error("vln_lstm_wgrad_reduce: gradients must be 16-byte aligned"); return VLN_ERR_ARG; }
  }
  VLN_LAUNCH(lstm_wgrad_reduce_kernel, dim3(512), dim3(256), 0, (hipStream_t)s, a);
  VLN_CHECK_LAUNCH("lstm_wgrad_reduce");
  return VLN_OK;
}
static int lstm_seq_bwd_impl(const float* dy_tm, const void* w_hh_t, int wtype, const int32_t* lengths,
                             const float* act, const float* tanh_c, const float* cprev, float* dgates,
                             float* dh_pass, float* dc_carry, const float* dh_init_bm, const float* dc_init_bm, int B, int L,
                             int Hd, int dirs, void* sync_ws, int64_t sync_ws_bytes, int64_t device_seq, float* bias_partials,
                             const float* wg_x, const float* wg_hprev, float* wg_part, vln_stream_t s);
extern "C" int vln_lstm_seq_bwd(const float* dy_tm, const void* w_hh_t, int wtype, const int32_t* lengths,
                                const float* act, const float* tanh_c, const float* cprev, float* dgates,
                                float* dh_pass, float* dc_carry, const float* dh_init_bm, const float* dc_init_bm, int B, int L,
                                int Hd, int dirs, void* sync_ws, int64_t sync_ws_bytes, int64_t device_seq, float* bias_partials, vln_stream_t s) {
  return lstm_seq_bwd_impl(dy_tm, w_hh_t, wtype, lengths, act, tanh_c, cprev, dgates, dh_pass, dc_carry, dh_init_bm, dc_init_bm, B, L, Hd, dirs,
                           sync_ws, sync_ws_bytes, device_seq, bias_partials, nullptr, nullptr, nullptr, s);
}
extern "C" int vln_lstm_seq_bwd_w(const float* dy_tm, const void* w_hh_t, int wtype, const int32_t* lengths,
                                  const float* act, const float* tanh_c, const float* cprev, float* dgates,
                                  float* dh_pass, float* dc_carry, const float* dh_init_bm, const float* dc_init_bm, int B, int L,
                                  int Hd, int dirs, void* sync_ws, int64_t sync_ws_bytes, int64_t device_seq, float* bias_partials,
                                  const float* x, int E, const float* hprev, float* wg_part, int64_t wg_part_floats, vln_stream_t s) {
  if (!x || !hprev || !wg_part || !al16(x) || !al16(hprev) || !al16(wg_part)) { set_error("vln_lstm_seq_bwd_w: null or misaligned x / hprev / wg_part"); return VLN_ERR_ARG; }
  if (!vln_lstm_wgrad_inlaunch_ok(B, L, Hd, dirs, E, wtype, 2, sync_ws, sync_ws_bytes) || wg_part_floats < vln_lstm_wgrad_part_floats(B, Hd, dirs, E)) {
    set_error("vln_lstm_seq_bwd_w: this shape does not take the in-launch weight gradients (ask vln_lstm_wgrad_inlaunch_ok), or wg_part is too small");
    return VLN_ERR_ARG;
  }
  return lstm_seq_bwd_impl(dy_tm, w_hh_t, wtype, lengths, act, tanh_c, cprev, dgates, dh_pass, dc_carry, dh_init_bm, dc_init_bm, B, L, Hd, dirs,
                           sync_ws, sync_ws_bytes, device_seq, bias_partials, x, hprev, wg_part, s);
}
static int lstm_seq_bwd_impl(const float* dy_tm, const void* w_hh_t, int wtype, const int32_t* lengths,
                             const float* act, const float* tanh_c, const float* cprev, float* dgates,
                             float* dh_pass, float* dc_carry, const float* dh_init_bm, const float* dc_init_bm, int B, int L,
                             int Hd, int dirs, void* sync_ws, int64_t sync_ws_bytes, int64_t device_seq, float* bias_partials,
                             const float* wg_x, const float* wg_hprev, float* wg_part, vln_stream_t s) {
  if (!w_hh_t || !lengths || !act || !tanh_c || !cprev || !dgates || !dh_pass || !dc_carry || B <= 0 || L <= 0 ||
      Hd <= 0 || dirs < 1 || dirs > 2 || ((dh_init_bm == nullptr) != (dc_init_bm == nullptr))) { set_error("vln_lstm_seq_bwd: bad args"); return VLN_ERR_ARG; }
  const float* dh_bm = dh_init_bm; const float* dc_bm = dc_init_bm;
  PendingRide pend;
  bool have_ride = ride_take((hipStream_t)s, &pend);
  if (persist_ok(B, L, Hd, dirs, sync_ws, !bwd_granules()) && sync_ws_bytes >= vln_lstm_sync_ws_bytes(B, Hd, dirs) && al16(w_hh_t) &&
      al16(sync_ws)) {
    hipStream_t st = (hipStream_t)s;
    int r = vln_persistent_check(); if (r) return r;
    // a pending gradient ride: passengers of the counter-protocol launch when it has idle CUs and the jobs need no reduce
    // launch, else its own launches in front of the recurrence
    WgradRideArgs ride_args{};
    const WgradRideArgs* riders = nullptr;
    if (have_ride) {
      const int nrec = (Hd / 16) * dirs * persist_passes(B, Hd, dirs, !bwd_granules());
      if (!bwd_granules() && g_tunable[10] != 1 && (nrec & 7) == 0 && (ride_passengers(nrec) & ~7) > 0 &&       // tunable[10] = 1: never as passengers (A/B)
          wgrad_ride_prepare(pend.w, pend.nw, pend.rows, pend.precision, pend.c, pend.nc, pend.ws, pend.ws_floats, &ride_args)) {
        ride_args.bar = reinterpret_cast<unsigned*>(sync_ws) + kRideBarWord;
        riders = &ride_args;
        std::lock_guard<std::mutex> lock(g_ride_mu);
        g_ride_stats[0]++;
      } else {
        r = ride_issue_alone(st, pend); if (r) return r;
      }
      have_ride = false;
    }
    // The counter-protocol backward kernel resets its group counters itself and clears its status word; the header only
    // needs a fill when another kernel left counters behind (the counter-protocol FORWARD of mode 2) or in the flag variant.
    // The self-reset only holds for a header this protocol left behind itself: a first launch on a buffer, or one after a mode
    // switch / a counter-protocol forward, gets the fill (header_clean).
    const bool self_cleaning = !VLN_SYNC_FLAGS && fwd_granules() && !bwd_granules();
    if (!self_cleaning || !header_clean(sync_ws)) {
      r = fill_f32(st, (float*)sync_ws, kSyncHeaderBytes / 4, 0.f);
      if (r) return r;
    }
    const int nbbp = persist_passes(B, Hd, dirs, !bwd_granules());
    RecBwdArgs a{dy_tm, w_hh_t, lengths, act, tanh_c, cprev, dgates, dh_pass, dc_carry, B, L, Hd, dirs, 0, 1, 1, dh_bm, dc_bm,
                 bwd_granules() ? nullptr : bias_partials, nbbp < (B + 15) / 16 ? nbbp : 0, wg_x, wg_hprev, wg_part};
    dim3 grid(Hd / 16, dirs, nbbp);
    unsigned* cw = (unsigned*)sync_ws;
    float* exch = reinterpret_cast<float*>(static_cast<char*>(sync_ws) + kSyncHeaderBytes);
    unsigned tag_base = 0;
    unsigned char* gex = static_cast<unsigned char*>(sync_ws) + sync_off_gbwd(B, Hd, dirs);
    const unsigned* seq_dev = device_seq >= 0 ? reinterpret_cast<const unsigned*>(static_cast<char*>(sync_ws) + sync_off_seq(B, Hd, dirs)) : nullptr;
    if (bwd_granules() && !seq_dev) {
      r = persist_tag_base(st, sync_ws, sync_off_gfwd(B, Hd, dirs), persist_g_fwd_bytes(B, Hd, dirs) + persist_g_bwd_bytes(B, Hd, dirs), &tag_base);
      if (r) return r;
    }
    {
      ProfScope prof(st, K_LSTM_REC_BWD, (double)dirs * (4.0 * Hd * Hd * (wtype == VLN_BF16 ? 2 : 4) + (double)L * 4.0 * B * Hd * (4 + 4 + 4 + 1 + 1 + 1 + 4)));
      if (bwd_granules())
        r = (wtype == VLN_BF16) ? launch_persist_g_bwd<bf16_raw>(st, a, cw + 32, gex, tag_base, grid, seq_dev, (unsigned)device_seq)
                                : launch_persist_g_bwd<float>(st, a, cw + 32, gex, tag_base, grid, seq_dev, (unsigned)device_seq);
      else
        r = (wtype == VLN_BF16) ? launch_persist_bwd<bf16_raw>(st, a, cw + 64, cw + 32, exch, grid, riders)
                                : launch_persist_bwd<float>(st, a, cw + 64, cw + 32, exch, grid, riders);
    }
    header_mark(sync_ws, r == VLN_OK && self_cleaning);
    if (r == VLN_OK && bwd_granules()) r = bias_part_fallback(st, dgates, bias_partials, B, L, Hd, dirs);
    return r;
  }
  if (have_ride) { int rr = ride_issue_alone((hipStream_t)s, pend); if (rr) return rr; }
  if (wg_part) { set_error("vln_lstm_seq_bwd_w: the persistent path was refused after vln_lstm_wgrad_inlaunch_ok said yes (mode switched?)"); return VLN_ERR_ARG; }
  struct { const void* p[11]; int v[5]; } key = {{dy_tm, w_hh_t, lengths, act, tanh_c, cprev, dgates, dh_pass, dc_carry, dh_bm, dc_bm},
                                                 {wtype, B, L, Hd, dirs}};
  static GraphCache cache;
  int rc = cache.run((hipStream_t)s, &key, sizeof(key), [&](hipStream_t st) {
    return lstm_seq_bwd_issue(st, dy_tm, w_hh_t, wtype, lengths, act, tanh_c, cprev, dgates, dh_pass, dc_carry, B, L, Hd, dirs, dh_bm, dc_bm);
  });
  if (rc == VLN_OK) rc = bias_part_fallback((hipStream_t)s, dgates, bias_partials, B, L, Hd, dirs);
  return rc;
}
