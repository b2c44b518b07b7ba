// gemm_rows: Y[M,N] = act(X[M,K] W[N,K]^T + bias) for TALL activations (the BN-MLP's M = B * (C + 1) = 1152 rows: the one
// MFMA-bound shape of the library) -- included by gemm.hip inside namespace vln, after gemm_nt_body.h.
//
// gemm_nt's fixed 64 x 64 tiles give such a product a workgroup count that does not divide over the 256 CUs: 18 x 16 = 288
// tiles = one round of 256 and a second of 32, i.e. the launch takes the time of TWO tiles per CU for 1.125 tiles per CU of work
// (measured: 91 us for 5.1 GFLOP of exact-fp32 MFMA = 36 % of the 157 TF/s peak; 68 us in the six-product form).  Here the row
// tiles are 16-row blocks dealt so that (row tiles x column tiles) fills whole rounds: tiles of rb_big blocks first, rb_big - 1
// after (gemm_rows_plan).  Same operand paths as gemm_nt's fast form -- W global -> VGPR per wave (16 columns), X through LDS
// (144-B padded rows), depth-2 register prefetch, one barrier per K-step -- and the SAME MFMA sequence per output element, so
// the results are bit-identical to gemm_nt's; a W fragment now feeds up to NRB row blocks instead of 4.
#pragma once

struct RowTiling { int n_big, rb_big; };   // row tiles [0, n_big) hold rb_big 16-row blocks, the others rb_big - 1

// LDS rows of a tile: the staged rows of whole passes of the 256 threads (the stores then need no predicate)
template <typename TW, int NRB>
constexpr int gemm_rows_lds_rows() { return (NRB * 16 + 256 / (GemmCfg<TW>::BK / 4) - 1) / (256 / (GemmCfg<TW>::BK / 4)) * (256 / (GemmCfg<TW>::BK / 4)); }
template <typename TW, int NRB>
constexpr int gemm_rows_smem_bytes() { return 2 * GemmCfg<TW>::kPlanes * gemm_rows_lds_rows<TW, NRB>() * kLdsRow; }

template <typename TW, int NRB>
__global__ __launch_bounds__(256) void gemm_rows_kernel(GemmNTArgs a, RowTiling rt) {
  constexpr int BK = GemmCfg<TW>::BK, VK = GemmCfg<TW>::VK, kPlanes = GemmCfg<TW>::kPlanes;
  constexpr bool kF32 = GemmCfg<TW>::kF32, kWS = GemmCfg<TW>::kWS, kX6 = GemmCfg<TW>::kX6;
  typedef typename std::conditional<kWS, float, TW>::type TM;       // the weights' element type IN MEMORY
  constexpr int ROWS = NRB * 16;
  constexpr int XLPR = BK / 4;                       // lanes per staged row: 16 (BK = 64) or 8 (BK = 32)
  constexpr int RPP = 256 / XLPR;                    // rows staged per pass of the 256 threads: 16 or 32
  constexpr int NLD = (ROWS + RPP - 1) / RPP;        // float4 loads per thread per K-step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef unsigned char (*SmemT)[kPlanes][gemm_rows_lds_rows<TW, NRB>() * kLdsRow];
  SmemT smem = reinterpret_cast<SmemT>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = (a.N + 63) >> 6;
  // XCD-aware tile order (the host sets a.xcd only when the tile count is a multiple of 8): XCD x takes a contiguous range of
  // tiles, column tiles of one row tile adjacent -- the sharers of an X block run behind one L2 (as gemm_nt_kernel)
  int t = (int)blockIdx.x;
  if (a.xcd) t = (t & 7) * ((int)gridDim.x >> 3) + (t >> 3);
  const int bx = t % nb, bz = t / nb;
  const int nrb = bz < rt.n_big ? rt.rb_big : rt.rb_big - 1;
  const int m0 = 16 * (bz < rt.n_big ? bz * rt.rb_big : rt.n_big * rt.rb_big + (bz - rt.n_big) * (rt.rb_big - 1));
  const int n0 = bx * 64;
  const int nsteps = a.K / BK;                       // (K % BK == 0: checked on the host)

  // fragment role: lane (fi, fq) of wave w owns column n0 + 16 w + fi and k = k0 + fq * VK .. + VK
  const int fi = lane & 15, fq = lane >> 4;
  const int wn = n0 + wave * 16 + fi;
  const bool wn_ok = wn < a.N;
  const TM* wrow = reinterpret_cast<const TM*>(a.W) + (long)(wn_ok ? wn : a.N - 1) * a.ldw;
  // staging role: thread -> (row j * RPP + tid / XLPR, 16-byte piece tid % XLPR); rows past the tile repeat its last row
  const int xpiece = tid % XLPR, xr0 = tid / XLPR;
  const int last_row = min(m0 + nrb * 16, a.M) - 1;
  const float* xrow[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) xrow[j] = a.X + (long)min(m0 + j * RPP + xr0, last_row) * a.ldx + xpiece * 4;

  float xs[2][NLD][4];
  float wf32[2][kF32 ? 8 : (kWS ? 16 : 1)];
  bf16x8 wb16[2][(kF32 || kWS) ? 1 : 2];

  auto load_x = [&](float (&x)[NLD][4], int kb) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(xrow[j] + kb);
      x[j][0] = v.x; x[j][1] = v.y; x[j][2] = v.z; x[j][3] = v.w;
    }
  };
  auto store_x = [&](const float (&x)[NLD][4], int buf) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int r = j * RPP + xr0;                 // (the LDS image holds whole passes: no predicate)
      if constexpr (kF32) {
        *reinterpret_cast<float4*>(&smem[buf][0][r * kLdsRow + xpiece * 16]) = make_float4(x[j][0], x[j][1], x[j][2], x[j][3]);
      } else {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 h, l, m3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          h[q] = (__bf16)x[j][q];
          const float r1 = x[j][q] - (float)h[q];
          l[q] = (__bf16)r1;
          if constexpr (kX6) m3[q] = (__bf16)(r1 - (float)l[q]);
        }
        *reinterpret_cast<bf16x4*>(&smem[buf][0][r * kLdsRow + xpiece * 8]) = h;
        *reinterpret_cast<bf16x4*>(&smem[buf][1][r * kLdsRow + xpiece * 8]) = l;
        if constexpr (kX6) *reinterpret_cast<bf16x4*>(&smem[buf][2][r * kLdsRow + xpiece * 8]) = m3;
      }
    }
  };
  auto load_w = [&](float (&w32)[kF32 ? 8 : (kWS ? 16 : 1)], bf16x8 (&w16)[(kF32 || kWS) ? 1 : 2], int kb) {
    const int k = kb + fq * VK;
    if constexpr (kF32 || kWS) {
      constexpr int NW = kF32 ? 8 : 16;
#pragma unroll
      for (int v = 0; v < NW / 4; ++v) {
        const float4 t0 = *reinterpret_cast<const float4*>(wrow + k + v * 4);
        w32[v * 4 + 0] = t0.x; w32[v * 4 + 1] = t0.y; w32[v * 4 + 2] = t0.z; w32[v * 4 + 3] = t0.w;
      }
    } else {
      w16[0] = *reinterpret_cast<const bf16x8*>(wrow + k);
      w16[1] = *reinterpret_cast<const bf16x8*>(wrow + k + 8);
    }
  };

  f32x4 acc[NRB];
#pragma unroll
  for (int r = 0; r < NRB; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (p < nsteps) { load_w(wf32[p], wb16[p], p * BK); load_x(xs[p], p * BK); }
  }
  for (int s0 = 0; s0 < nsteps; s0 += 2) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int s = s0 + p;
      if (s >= nsteps) continue;
      const int buf = s & 1;
      // current W fragment -> private copy (split into its bf16 pieces) before the prefetch overwrites the registers
      float wc32[kF32 ? 8 : 1];
      bf16x8 wc16[kF32 ? 1 : 2];
      bf16x8 wcl[kWS ? 2 : 1];
      bf16x8 wc3[kX6 ? 2 : 1];
      store_x(xs[p], buf);
      if constexpr (kF32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) wc32[j] = wf32[p][j];
      } else if constexpr (kWS) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float w = wf32[p][h * 8 + j];
            const __bf16 hi = (__bf16)w;
            wc16[h][j] = hi;
            const float r1 = w - (float)hi;
            wcl[h][j] = (__bf16)r1;
            if constexpr (kX6) wc3[h][j] = (__bf16)(r1 - (float)wcl[h][j]);
          }
      } else {
        wc16[0] = wb16[p][0]; wc16[1] = wb16[p][1];
      }
      __syncthreads();
      if (s + 2 < nsteps) { load_x(xs[p], (s + 2) * BK); load_w(wf32[p], wb16[p], (s + 2) * BK); }
      // The row blocks' X fragments come from LDS through a ring of DEPTH + 1 register sets: block rb + DEPTH is requested before
      // block rb is multiplied, so one LDS round trip per K-step is exposed instead of one per row block (the compiler keeps the
      // program order of the reads and waits with counted lgkmcnt).  No branch on the tile's height in here: a tile of rb_big - 1
      // blocks multiplies one block of repeated rows whose results are never stored -- it finishes with the tall tiles anyway.
      constexpr int DEPTH = kF32 ? 1 : (kX6 ? 1 : (kWS ? 2 : 3));
      constexpr int NF = kF32 ? 2 : 2 * kPlanes;           // 16-byte fragments per row block
      float4 fa[DEPTH + 1][kF32 ? 2 : 1];
      bf16x8 fb[DEPTH + 1][kF32 ? 1 : NF];
      auto request = [&](int rb, int slot) {
        const int off = (rb * 16 + fi) * kLdsRow + fq * 32;
        if constexpr (kF32) {
          fa[slot][0] = *reinterpret_cast<const float4*>(&smem[buf][0][off]);
          fa[slot][1] = *reinterpret_cast<const float4*>(&smem[buf][0][off + 16]);
        } else {
#pragma unroll
          for (int pl = 0; pl < kPlanes; ++pl) {
            fb[slot][2 * pl] = *reinterpret_cast<const bf16x8*>(&smem[buf][pl][off]);
            fb[slot][2 * pl + 1] = *reinterpret_cast<const bf16x8*>(&smem[buf][pl][off + 16]);
          }
        }
      };
#pragma unroll
      for (int rb = 0; rb < DEPTH && rb < NRB; ++rb) request(rb, rb);
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        if (rb + DEPTH < NRB) request(rb + DEPTH, (rb + DEPTH) % (DEPTH + 1));
        __builtin_amdgcn_sched_barrier(0);          // (the scheduler otherwise sinks the reads to just before their use)
        const int sl = rb % (DEPTH + 1);
        if constexpr (kF32) {
          const float4 a0 = fa[sl][0], a1 = fa[sl][1];
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, wc32[0], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, wc32[1], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, wc32[2], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, wc32[3], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, wc32[4], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, wc32[5], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, wc32[6], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, wc32[7], acc[rb], 0, 0, 0);
        } else {
          const bf16x8 a0 = fb[sl][0], a1 = fb[sl][1], b0 = fb[sl][2], b1 = fb[sl][3];     // planes: 0 = hi, 1 = lo, 2 = third piece
          if constexpr (kX6) {           // the smallest terms first (the order of gemm_nt_body.h)
            const bf16x8 c0 = fb[sl][4], c1 = fb[sl][5];
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, wc16[0], acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c1, wc16[1], acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wc3[0], acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wc3[1], acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, wcl[0], acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, wcl[1], acc[rb], 0, 0, 0);
          }
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, wc16[0], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, wc16[1], acc[rb], 0, 0, 0);
          if constexpr (kWS) {
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wcl[0], acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wcl[1], acc[rb], 0, 0, 0);
          }
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wc16[0], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wc16[1], acc[rb], 0, 0, 0);
        }
      }
    }
  }

  // C/D layout of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
  if (!wn_ok) return;
  const float bv = a.bias ? a.bias[wn] : 0.0f;
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) {
    if (rb >= nrb) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + rb * 16 + fq * 4 + r;
      if (row < a.M) {
        float v = acc[rb][r] + bv;
        if ((a.act & 3) == ACT_TANH) v = tanhf(v);
        else if ((a.act & 3) == ACT_RELU) v = fmaxf(v, 0.0f);
        if (a.act & ACT_ACCUM) v += a.Y[(long)row * a.ldy + wn];
        a.Y[(long)row * a.ldy + wn] = v;
      }
    }
  }
}

// The row tiling of a tall product: R row tiles (of rb_big and rb_big - 1 blocks of 16 rows) x nb column tiles, chosen to minimise
// (rounds of workgroups over the CUs) x (blocks of the tallest tile); ties go to the taller tile (a W fragment feeds more rows).
// Returns false when nothing beats gemm_nt's 64-row tiles (or no tiling with <= 8 blocks per tile exists).
static bool gemm_rows_plan(int M, int N, int cus, RowTiling* rt, int* tiles, int* rb_max) {
  if (cus <= 0) return false;
  const int mbk = (M + 15) / 16, nb = (N + 63) / 64;
  const long old_cost = (((long)((M + 63) / 64) * nb + cus - 1) / cus) * 4;
  long best = -1; int best_r = 0, best_rb = 0;
  for (int R = 1; R <= mbk; ++R) {
    const int rb = (mbk + R - 1) / R;
    if (rb > 8 || rb < 3) continue;
    const long cost = (((long)R * nb + cus - 1) / cus) * rb;
    // ties go to the tallest tile (ascending R: the first seen), and among tilings of that height to the one with the MOST tiles
    // (fewer of them are tall; every CU of the round has one)
    if (best < 0 || cost < best || (cost == best && rb == best_rb)) { best = cost; best_r = R; best_rb = rb; }
  }
  if (best < 0 || best >= old_cost) return false;
  // best_r tiles: n_big of best_rb blocks, the rest best_rb - 1, covering exactly mbk blocks
  const int n_big = mbk - best_r * (best_rb - 1);
  rt->n_big = n_big; rt->rb_big = best_rb;
  *tiles = best_r * nb; *rb_max = best_rb;
  return true;
}
