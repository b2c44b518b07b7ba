// gemm_wres: Y[M,N] = act(X[M,K] W[N,K]^T + bias) for TALL activations over a SHALLOW K (K <= 256) with bf16-streamed weights --
// the instruction encoder's input projection, M = B * L = 5120 rows of 256 embedding features onto 2048 gate columns.
// Included by gemm.hip inside namespace vln, after gemm_nt_body.h.
//
// gemm_nt gives that product 80 x 32 tiles of 64 x 64, four K-steps each: a workgroup's life is one round trip for its first
// operands, four short steps and a store, and every one of the 32 column tiles streams the whole activation again (31.5 us for
// 5.4 GFLOP and 42 MB of output: neither MFMA- nor HBM-bound, a chain of first-touch latencies).  Here the WEIGHTS of a workgroup's
// 256 columns stay in registers for the whole launch -- wave w holds columns 64 w .. 64 w + 63 over all of K: 128 VGPRs -- and
// the workgroup walks down its share of the rows in stages of 32: the stage's activations go global -> registers (prefetched one
// stage ahead) -> hi / lo bf16 planes in LDS (double-buffered, one barrier per stage), every wave multiplies the stage by its own
// columns and stores.  One workgroup per CU (N / 256 column tiles x row groups = the device's CUs), XCD-aware: the column tiles
// that share a row group run behind one L2.
// The SAME MFMA sequence per output element as gemm_nt's bf16 form (per 64-wide K-step: lo x w0, lo x w1, hi x w0, hi x w1), the
// same hi / lo split of the activations and the same epilogue: the results are bit-identical to gemm_nt's
// (tests/test_hip_ops.py::test_shallow_tall_products_with_resident_weights_are_bit_identical).
// REJECTED (round 5, scripts/wres_probe.py): bit-identical to gemm_nt, but 31.8 vs 34.1 us on the [5120, 256] x [2048, 256]^T
// projection and SLOWER on its siblings (21.0 vs 20.0, 20.8 vs 18.7, 24.5 vs 22.8 us); headline unchanged (1.432 / 1.433 vs 1.430 /
// 1.436 ms).  One workgroup per CU with one stage of prefetch is again a chain of load latencies (5 stages x ~6 us), and the
// 4-byte-per-lane result stores are the same as gemm_nt's.  Kept for the record; not built.
#pragma once

constexpr int kWresStage = 32;                 // rows per stage
constexpr int kWresMaxK = 256;
constexpr int kWresRow = kWresMaxK * 2 + 16;   // bytes of one staged row of one plane: K bf16 + 16 B pad
constexpr int gemm_wres_smem_bytes() { return 2 /*buffers*/ * 2 /*planes*/ * kWresStage * kWresRow; }

struct WresPlan { int col_tiles, row_groups, rows_per, xcd; };

// KS = K / 64
template <int KS>
__global__ __launch_bounds__(256) void gemm_wres_kernel(GemmNTArgs a, WresPlan pl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef unsigned char (*SmemT)[2][kWresStage * kWresRow];     // [buffer][plane][...]
  SmemT smem = reinterpret_cast<SmemT>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, fq = lane >> 4;
  int t = (int)blockIdx.x, bx, rg;
  if (pl.xcd) { const int x = t & 7, q = t >> 3; rg = x + 8 * (q / pl.col_tiles); bx = q % pl.col_tiles; }
  else { bx = t % pl.col_tiles; rg = t / pl.col_tiles; }
  const int n0 = bx * 256 + wave * 64;
  const int m_beg = rg * pl.rows_per, m_end = min(m_beg + pl.rows_per, a.M);
  if (m_beg >= m_end) return;                                   // (uniform per workgroup)
  const int nstages = (m_end - m_beg + kWresStage - 1) / kWresStage;

  // the wave's weights: 4 column tiles x KS K-steps x {w0, w1}; lane (fi, fq) of tile c holds column n0 + 16 c + fi, k = 64 s + 16 fq ..
  bf16x8 w[4][KS][2];
  const bf16_raw* W = reinterpret_cast<const bf16_raw*>(a.W);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bf16_raw* wrow = W + (long)(n0 + c * 16 + fi) * a.ldw + fq * 16;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      w[c][s][0] = *reinterpret_cast<const bf16x8*>(wrow + s * 64);
      w[c][s][1] = *reinterpret_cast<const bf16x8*>(wrow + s * 64 + 8);
    }
  }

  // staging role: chunk = 16 consecutive floats of a row; the stage has 32 rows x (K / 16) chunks, NCH per thread
  constexpr int CPR = KS * 4;                          // chunks per row
  constexpr int NCH = kWresStage * CPR / 256;          // 2 (K = 256), 1 (K = 128); K = 64, 192: see the host's conditions
  float xs[NCH][16];
  auto load_x = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = tid + 256 * j, r = c / CPR, piece = c % CPR;
      const int row = min(m_beg + stage * kWresStage + r, m_end - 1);
      const float* p = a.X + (long)row * a.ldx + piece * 16;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float4 q = *reinterpret_cast<const float4*>(p + v * 4);
        xs[j][v * 4 + 0] = q.x; xs[j][v * 4 + 1] = q.y; xs[j][v * 4 + 2] = q.z; xs[j][v * 4 + 3] = q.w;
      }
    }
  };
  auto store_x = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = tid + 256 * j, r = c / CPR, piece = c % CPR;
      bf16x8 h0, h1, l0, l1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        h0[e] = (__bf16)xs[j][e];
        h1[e] = (__bf16)xs[j][8 + e];
        l0[e] = (__bf16)(xs[j][e] - (float)h0[e]);
        l1[e] = (__bf16)(xs[j][8 + e] - (float)h1[e]);
      }
      unsigned char* dh = &smem[buf][0][r * kWresRow + piece * 32];
      unsigned char* dl = &smem[buf][1][r * kWresRow + piece * 32];
      *reinterpret_cast<bf16x8*>(dh) = h0; *reinterpret_cast<bf16x8*>(dh + 16) = h1;
      *reinterpret_cast<bf16x8*>(dl) = l0; *reinterpret_cast<bf16x8*>(dl + 16) = l1;
    }
  };

  const bool fused = true;                             // (never a slab launch: the host keeps split-K products on gemm_nt)
  float bv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) bv[c] = (fused && a.bias) ? a.bias[n0 + c * 16 + fi] : 0.0f;

  load_x(0);
  for (int st = 0; st < nstages; ++st) {
    const int buf = st & 1;
    store_x(buf);
    __syncthreads();                                   // the stage is in LDS; the other buffer's readers of stage st - 1 are done
    if (st + 1 < nstages) load_x(st + 1);
    f32x4 acc[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c) { acc[c][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[c][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const unsigned char* sh = &smem[buf][0][(rb * 16 + fi) * kWresRow + s * 128 + fq * 32];
        const unsigned char* sl = &smem[buf][1][(rb * 16 + fi) * kWresRow + s * 128 + fq * 32];
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(sh), a1 = *reinterpret_cast<const bf16x8*>(sh + 16);
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(sl), b1 = *reinterpret_cast<const bf16x8*>(sl + 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[c][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, w[c][s][0], acc[c][rb], 0, 0, 0);
          acc[c][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, w[c][s][1], acc[c][rb], 0, 0, 0);
          acc[c][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w[c][s][0], acc[c][rb], 0, 0, 0);
          acc[c][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, w[c][s][1], acc[c][rb], 0, 0, 0);
        }
      }
    }
    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    const int mrow0 = m_beg + st * kWresStage;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = n0 + c * 16 + fi;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = mrow0 + rb * 16 + fq * 4 + r;
          if (row < m_end) {
            float v = acc[c][rb][r] + bv[c];
            if ((a.act & 3) == ACT_TANH) v = tanhf(v);
            else if ((a.act & 3) == ACT_RELU) v = fmaxf(v, 0.0f);
            if (a.act & ACT_ACCUM) v += a.Y[(long)row * a.ldy + col];
            a.Y[(long)row * a.ldy + col] = v;
          }
        }
      }
    }
  }
}

// 1 and the plan when the product takes this kernel on a device of `cus` compute units
static inline int gemm_wres_plan(int M, int N, int K, int cus, WresPlan* pl) {
  if ((K != 256 && K != 128) || (N & 255) || M < 1024 || cus < 8) return 0;
  const int ct = N / 256;
  int rgs = cus / ct;
  if (rgs < 1) return 0;
  int rows_per = ((M + rgs - 1) / rgs + kWresStage - 1) / kWresStage * kWresStage;
  rgs = (M + rows_per - 1) / rows_per;                 // (no empty row groups)
  pl->col_tiles = ct; pl->row_groups = rgs; pl->rows_per = rows_per;
  pl->xcd = (rgs % 8 == 0) ? 1 : 0;
  return 1;
}
