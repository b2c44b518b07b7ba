// gemm_nt: the workgroup body of Y[M,N] = X[M,K] W[N,K]^T (included by gemm.hip inside namespace vln).
#pragma once

// kF32: exact fp32 MFMA on fp32 weights.  kWS: fp32 weights in memory, split into hi + lo bf16 planes in registers (W_F32S).
template <typename TW> struct GemmCfg;
template <> struct GemmCfg<float> { static constexpr int BK = 32, VK = 8, kPlanes = 1; static constexpr bool kF32 = true, kWS = false, kX6 = false; };
template <> struct GemmCfg<bf16_raw> { static constexpr int BK = 64, VK = 16, kPlanes = 2; static constexpr bool kF32 = false, kWS = false, kX6 = false; };
template <> struct GemmCfg<f32s_raw> { static constexpr int BK = 64, VK = 16, kPlanes = 2; static constexpr bool kF32 = false, kWS = true, kX6 = false; };
// W_F32X: fp32 operands as THREE bf16 pieces each (x = hi + mid + lo, 24 bits), six products (every pair whose weight is above
// 2^-24 of the full product): fp32-grade results -- the truncation is below the fp32 accumulation's own rounding -- at 6/16 of
// the exact fp32 MFMA's time.  For products in FRONT of a ReLU (the BN-MLP's forward Linear): the three-product W_F32S form's 2^-16
// flips ~10 of 1.2 M units per call against the exact product, this form none beyond what exact fp32 itself flips against fp64.
template <> struct GemmCfg<f32x_raw> { static constexpr int BK = 64, VK = 16, kPlanes = 3; static constexpr bool kF32 = false, kWS = true, kX6 = true; };

constexpr int kLdsRow = 144;  // 128 B of data + 16 B pad per staged X row

struct GemmNTArgs {
  const float* X; long ldx;
  const void* W; long ldw;
  float* Y; long ldy; long slab_stride;
  const float* bias; int act;
  int M, N, K, kchunk;
  int xvec, wvec;
  int xcd;               // 1: the launch's workgroups take their tiles in XCD-aware order (gemm_nt_kernel; many row tiles)
};

// PD = register prefetch depth: the loads of K-steps s+1 .. s+PD are in flight while step s is multiplied.  Every
// first-touch load in these launches crosses the fabric (~1.3 us: the operands were written by another XCD or come
// from the MALL / HBM), and with PD = 1 a workgroup's K-steps are a chain of such round trips (13.4 us for the 8 steps
// of an un-split H->F projection, 10.9 us for 6 steps of the LSTM gate product); PD = 4 overlaps them.
// kFast: every load is unconditional (rows clamped to the last valid one, K a multiple of BK per chunk, 16-byte
// aligned operands -- checked on the host).  The bounds-checked form branches per thread between a vector and a
// scalar load, and the compiler closes every such divergent region with `s_waitcnt vmcnt(0)`: the prefetch never
// overlaps anything.  Without branches the waits become vmcnt(N) and PD loads really are in flight.
// NT = 16-column tiles per wave: the workgroup's tile is 64 rows x 64*NT columns.  NT = 2 halves the re-reads of X (every
// column tile streams the whole activation slice: 2x the weight bytes at NT = 1) for the wide products (LSTM gates, d xcat).
// Virtual block of a launch (`nbar` = the K-step count the workgroup's barriers walk through).
struct VBlock { int bx, by, bz; int tid; unsigned char* smem; };

constexpr int gemm_nt_smem_bytes(int planes) { return 2 * planes * 64 * kLdsRow; }

__device__ __forceinline__ int gemm_nt_nsteps(const GemmNTArgs& a, int by, int BK) {
  const int kbeg = by * a.kchunk;
  const int kend = min(a.K, kbeg + a.kchunk);
  return (kend - kbeg + BK - 1) / BK;
}

// `pre()` is called once, after the first weight fragments have been requested and before anything that depends on the
// producer of X is touched: the chained kernel waits for the previous stage there (its weight loads are in flight meanwhile).
template <typename TW, int PD, bool kFast, int NT, typename Pre>
__device__ __forceinline__ void gemm_nt_body(const GemmNTArgs& a, const VBlock& vb, bool active, int nbar, Pre pre) {
  constexpr int BK = GemmCfg<TW>::BK, VK = GemmCfg<TW>::VK;
  constexpr bool kF32 = GemmCfg<TW>::kF32, kWS = GemmCfg<TW>::kWS, kX6 = GemmCfg<TW>::kX6;
  typedef typename std::conditional<kWS, float, TW>::type TM;       // the weights' element type IN MEMORY
  // bf16 path: the fp32 activations are split x = hi + lo (two bf16 planes) so only the STREAMED operand is
  // quantised; the second MFMA pair is free in these weight-bandwidth-bound shapes.  (kX6: three planes, hi / mid / lo.)
  constexpr int kPlanes = GemmCfg<TW>::kPlanes;
  typedef unsigned char (*SmemT)[kPlanes][64 * kLdsRow];
  SmemT smem = reinterpret_cast<SmemT>(vb.smem);

  const int tid = vb.tid, lane = tid & 63, wave = tid >> 6;
  const int n0 = vb.bx * 64 * NT, m0 = vb.bz * 64;
  const int kbeg = vb.by * a.kchunk;
  const int kend = min(a.K, kbeg + a.kchunk);
  const int nsteps = active ? (kend - kbeg + BK - 1) / BK : 0;

  // staging role: thread -> (row, 32-byte segment) of the X tile
  const int srow = tid >> 2, sseg = tid & 3;
  const bool srow_ok = (m0 + srow) < a.M;
  const float* xrow = a.X + (long)(srow_ok ? (m0 + srow) : (kFast ? a.M - 1 : 0)) * a.ldx;
  // fragment role
  const int fi = lane & 15, fq = lane >> 4;
  int wn[NT]; bool wn_ok[NT]; const TM* wrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    wn[t] = n0 + (wave * NT + t) * 16 + fi;
    wn_ok[t] = wn[t] < a.N;
    wrow[t] = reinterpret_cast<const TM*>(a.W) + (long)(wn_ok[t] ? wn[t] : (kFast ? a.N - 1 : 0)) * a.ldw;
  }
  const int mrows = min(64, a.M - m0);
  const int nrb = (mrows + 15) >> 4;

  float xs[PD][VK];
  float wf32[PD][NT][kF32 ? 8 : (kWS ? 16 : 1)];
  bf16x8 wb16[PD][NT][(kF32 || kWS) ? 1 : 2];

  // kFast staging role: each 16-lane group reads 256 contiguous bytes of ONE row (two cache lines) per instruction.
  // (The bounds-checked role above gives every lane its own 64-byte run: a wave instruction then touches 32 lines for
  // 1 KiB, and the CU's address path, not the fabric, sets the pace -- scripts/stream_probe.hip, "fragment shape".)
  constexpr int NXI = VK / 4;                        // float4 loads per thread per K-step
  constexpr int XLPR = BK / 4;                       // lanes per row: 16 (BK = 64) or 8 (BK = 32)
  const int xpiece = lane % XLPR;
  const float* xrow_i[NXI];
  int xr_i[NXI];
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    xr_i[i] = wave * 16 + i * (64 / XLPR) + lane / XLPR;
    const int gr = min(m0 + xr_i[i], a.M - 1);
    xrow_i[i] = a.X + (long)gr * a.ldx + xpiece * 4;
  }
  auto load_x = [&](float (&x)[VK], int kb) {
    if constexpr (kFast) {
#pragma unroll
      for (int i = 0; i < NXI; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(xrow_i[i] + kb);
        x[i * 4 + 0] = t.x; x[i * 4 + 1] = t.y; x[i * 4 + 2] = t.z; x[i * 4 + 3] = t.w;
      }
      return;
    }
    const int k = kb + sseg * VK;
    if (srow_ok && a.xvec && k + VK <= kend) {
#pragma unroll
      for (int v = 0; v < VK / 4; ++v) {
        float4 t = *reinterpret_cast<const float4*>(xrow + k + v * 4);
        x[v * 4 + 0] = t.x; x[v * 4 + 1] = t.y; x[v * 4 + 2] = t.z; x[v * 4 + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < VK; ++j) x[j] = (srow_ok && (k + j) < kend) ? xrow[k + j] : 0.0f;
    }
  };
  auto store_x = [&](const float (&x)[VK], int buf) {
    if constexpr (kFast) {
#pragma unroll
      for (int i = 0; i < NXI; ++i) {
        if constexpr (kF32) {
          *reinterpret_cast<float4*>(&smem[buf][0][xr_i[i] * kLdsRow + xpiece * 16]) =
              make_float4(x[i * 4], x[i * 4 + 1], x[i * 4 + 2], x[i * 4 + 3]);
        } else {
          typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
          bf16x4 h, l, m3;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            h[j] = (__bf16)x[i * 4 + j];
            const float r1 = x[i * 4 + j] - (float)h[j];
            l[j] = (__bf16)r1;
            if constexpr (kX6) m3[j] = (__bf16)(r1 - (float)l[j]);
          }
          *reinterpret_cast<bf16x4*>(&smem[buf][0][xr_i[i] * kLdsRow + xpiece * 8]) = h;
          *reinterpret_cast<bf16x4*>(&smem[buf][1][xr_i[i] * kLdsRow + xpiece * 8]) = l;         // (two planes: kPlanes - 1 == 1)
          if constexpr (kX6) *reinterpret_cast<bf16x4*>(&smem[buf][2][xr_i[i] * kLdsRow + xpiece * 8]) = m3;
        }
      }
      return;
    }
    unsigned char* dst = &smem[buf][0][srow * kLdsRow + sseg * 32];
    if constexpr (kF32) {
      *reinterpret_cast<float4*>(dst) = make_float4(x[0], x[1], x[2], x[3]);
      *reinterpret_cast<float4*>(dst + 16) = make_float4(x[4], x[5], x[6], x[7]);
    } else {
      bf16x8 h0, h1, l0, l1, t0, t1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        h0[j] = (__bf16)x[j];
        h1[j] = (__bf16)x[8 + j];
        const float r0 = x[j] - (float)h0[j], r1 = x[8 + j] - (float)h1[j];
        l0[j] = (__bf16)r0;
        l1[j] = (__bf16)r1;
        if constexpr (kX6) { t0[j] = (__bf16)(r0 - (float)l0[j]); t1[j] = (__bf16)(r1 - (float)l1[j]); }
      }
      *reinterpret_cast<bf16x8*>(dst) = h0;
      *reinterpret_cast<bf16x8*>(dst + 16) = h1;
      unsigned char* dlo = &smem[buf][1][srow * kLdsRow + sseg * 32];
      *reinterpret_cast<bf16x8*>(dlo) = l0;
      *reinterpret_cast<bf16x8*>(dlo + 16) = l1;
      if constexpr (kX6) {
        unsigned char* d3 = &smem[buf][2][srow * kLdsRow + sseg * 32];
        *reinterpret_cast<bf16x8*>(d3) = t0;
        *reinterpret_cast<bf16x8*>(d3 + 16) = t1;
      }
    }
  };
  auto load_w = [&](float (&w32)[kF32 ? 8 : (kWS ? 16 : 1)], bf16x8 (&w16)[(kF32 || kWS) ? 1 : 2], int kb, int t) {
    const TM* wrow_t = wrow[t]; const bool wn_ok_t = wn_ok[t];
    const int k = kb + fq * VK;
    if constexpr (kF32 || kWS) {
      constexpr int NW = kF32 ? 8 : 16;
      if (kFast || (wn_ok_t && a.wvec && k + VK <= kend)) {
#pragma unroll
        for (int v = 0; v < NW / 4; ++v) {
          const float4 t0 = *reinterpret_cast<const float4*>(wrow_t + k + v * 4);
          w32[v * 4 + 0] = t0.x; w32[v * 4 + 1] = t0.y; w32[v * 4 + 2] = t0.z; w32[v * 4 + 3] = t0.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < NW; ++j) w32[j] = (wn_ok_t && (k + j) < kend) ? wrow_t[k + j] : 0.0f;
      }
    } else {
      if (kFast || (wn_ok_t && a.wvec && k + VK <= kend)) {
        w16[0] = *reinterpret_cast<const bf16x8*>(wrow_t + k);
        w16[1] = *reinterpret_cast<const bf16x8*>(wrow_t + k + 8);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          bf16_raw r0 = (wn_ok_t && (k + j) < kend) ? wrow_t[k + j] : (bf16_raw)0;
          bf16_raw r1 = (wn_ok_t && (k + 8 + j) < kend) ? wrow_t[k + 8 + j] : (bf16_raw)0;
          w16[0][j] = __builtin_bit_cast(__bf16, r0);
          w16[1][j] = __builtin_bit_cast(__bf16, r1);
        }
      }
    }
  };

  f32x4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int p = 0; p < PD; ++p) {
    if (p < nsteps) {
#pragma unroll
      for (int t = 0; t < NT; ++t) load_w(wf32[p][t], wb16[p][t], kbeg + p * BK, t);
    }
  }
  pre();
#pragma unroll
  for (int p = 0; p < PD; ++p) {
    if (p < nsteps) load_x(xs[p], kbeg + p * BK);
  }
  for (int s0 = 0; s0 < nbar; s0 += PD) {
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const int s = s0 + p;
      if (s >= nbar) continue;
      const bool on = s < nsteps;          // the other half of a chained workgroup may have more K-steps: barriers only
      const int buf = s & 1;
      // current W fragment -> private copy before the prefetch overwrites the registers
      float wc32[NT][kF32 ? 8 : 1];
      bf16x8 wc16[NT][kF32 ? 1 : 2];
      bf16x8 wcl[NT][kWS ? 2 : 1];         // W_F32S: the weights' lo plane (w - bf16(w))
      bf16x8 wc3[NT][kX6 ? 2 : 1];         // W_F32X: the third piece (w - hi - lo)
      if (on) {
        store_x(xs[p], buf);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if constexpr (kF32) {
#pragma unroll
            for (int j = 0; j < 8; ++j) wc32[t][j] = wf32[p][t][j];
          } else if constexpr (kWS) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const float w = wf32[p][t][h * 8 + j];
                const __bf16 hi = (__bf16)w;
                wc16[t][h][j] = hi;
                const float r1 = w - (float)hi;
                wcl[t][h][j] = (__bf16)r1;
                if constexpr (kX6) wc3[t][h][j] = (__bf16)(r1 - (float)wcl[t][h][j]);
              }
          } else {
            wc16[t][0] = wb16[p][t][0]; wc16[t][1] = wb16[p][t][1];
          }
        }
      }
      __syncthreads();
      if (on) {
        if (s + PD < nsteps) {
          load_x(xs[p], kbeg + (s + PD) * BK);
#pragma unroll
          for (int t = 0; t < NT; ++t) load_w(wf32[p][t], wb16[p][t], kbeg + (s + PD) * BK, t);
        }
#ifdef VLN_PROBE_NO_MFMA       // scripts/gemm_probe.hip: what the K-step costs without the LDS reads and MFMAs
        if constexpr (!kF32) { acc[0][0][0] += (float)wc16[0][0][0] + (float)wc16[NT - 1][1][7]; acc[0][1][0] += smem[buf][0][tid]; }
        else { acc[0][0][0] += wc32[0][0] + wc32[NT - 1][7]; acc[0][1][0] += smem[buf][0][tid]; }
        continue;
#endif
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          if (rb < nrb) {
            const unsigned char* src = &smem[buf][0][(rb * 16 + fi) * kLdsRow + fq * 32];
            if constexpr (kF32) {
              float4 a0 = *reinterpret_cast<const float4*>(src);
              float4 a1 = *reinterpret_cast<const float4*>(src + 16);
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, wc32[t][0], acc[t][rb], 0, 0, 0);
                acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, wc32[t][1], acc[t][rb], 0, 0, 0);
                acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, wc32[t][2], acc[t][rb], 0, 0, 0);
                acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, wc32[t][3], acc[t][rb], 0, 0, 0);
                acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, wc32[t][4], acc[t][rb], 0, 0, 0);
                acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, wc32[t][5], acc[t][rb], 0, 0, 0);
                acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, wc32[t][6], acc[t][rb], 0, 0, 0);
                acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, wc32[t][7], acc[t][rb], 0, 0, 0);
              }
            } else {
              bf16x8 a0 = *reinterpret_cast<const bf16x8*>(src);
              bf16x8 a1 = *reinterpret_cast<const bf16x8*>(src + 16);
              const unsigned char* slo = &smem[buf][1][(rb * 16 + fi) * kLdsRow + fq * 32];
              bf16x8 b0 = *reinterpret_cast<const bf16x8*>(slo);
              bf16x8 b1 = *reinterpret_cast<const bf16x8*>(slo + 16);
              if constexpr (kX6) {         // the smallest terms first: x3 w1, x1 w3, x2 w2; then the three of the W_F32S form
                const unsigned char* s3 = &smem[buf][2][(rb * 16 + fi) * kLdsRow + fq * 32];
                const bf16x8 c0 = *reinterpret_cast<const bf16x8*>(s3);
                const bf16x8 c1 = *reinterpret_cast<const bf16x8*>(s3 + 16);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                  acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, wc16[t][0], acc[t][rb], 0, 0, 0);
                  acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c1, wc16[t][1], acc[t][rb], 0, 0, 0);
                  acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wc3[t][0], acc[t][rb], 0, 0, 0);
                  acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wc3[t][1], acc[t][rb], 0, 0, 0);
                  acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, wcl[t][0], acc[t][rb], 0, 0, 0);
                  acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, wcl[t][1], acc[t][rb], 0, 0, 0);
                }
              }
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, wc16[t][0], acc[t][rb], 0, 0, 0);
                acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, wc16[t][1], acc[t][rb], 0, 0, 0);
                if constexpr (kWS) {
                  acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wcl[t][0], acc[t][rb], 0, 0, 0);
                  acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wcl[t][1], acc[t][rb], 0, 0, 0);
                }
                acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wc16[t][0], acc[t][rb], 0, 0, 0);
                acc[t][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wc16[t][1], acc[t][rb], 0, 0, 0);
              }
            }
          }
        }
      }
    }
  }

  if (!active) return;
  // C/D layout of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
  float* y = a.Y + (long)vb.by * a.slab_stride;
  const bool fused = (a.slab_stride == 0);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (!wn_ok[t]) continue;
    const float bv = (fused && a.bias) ? a.bias[wn[t]] : 0.0f;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + rb * 16 + fq * 4 + r;
        if (row < a.M) {
          float v = acc[t][rb][r] + bv;
          if (fused) {
            if ((a.act & 3) == ACT_TANH) v = tanhf(v);
            else if ((a.act & 3) == ACT_RELU) v = fmaxf(v, 0.0f);
            if (a.act & ACT_ACCUM) v += y[(long)row * a.ldy + wn[t]];
          }
          y[(long)row * a.ldy + wn[t]] = v;
        }
      }
    }
  }
}
