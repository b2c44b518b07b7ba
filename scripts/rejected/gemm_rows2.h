// REJECTED (round 5, profiles/round5_notes.md section 8): gemm_rows_kernel with step s + 1's operand preparation (fp32 -> bf16 planes of
// X into the other LDS buffer, W's pieces into a second register set) placed between step s's row blocks, so that the conversions run
// under the MFMAs.  Bit-identical, NOT faster: BN-MLP forward exact fp32 91.7 vs 71.7 us, f32s 55.4 vs 48.0, f32x 60.6 vs 59.2
// (scripts/rows_probe.py; hip events incl. ~12 us launch overhead) -- left to the compiler's scheduler the stores and conversions
// do not interleave with the matrix instructions (it would take sched_group_barrier patterns per form), and the second register
// set costs occupancy.  Kept for the record; not part of the build (it was included by gemm.hip after gemm_rows_kernel).
// ---- gemm_rows_kernel with the NEXT K-step's operand preparation under the current K-step's MFMAs -------------------------------------
// In the bf16 forms every K-step first converts its operands -- X: fp32 -> hi / lo (/ third) bf16 planes into LDS, W: the same split
// in registers -- and only then multiplies: with one wave per SIMD the VALU phase and the MFMA phase alternate (a K-step of the
// six-product form: ~900 VALU cycles, then 960 MFMA cycles).  Here step s + 1's operands are prepared in per-row-block chunks
// BETWEEN step s's row blocks (LDS buffer (s + 1) & 1 is free from the barrier that opened step s; W's pieces go to a second register
// set), so the matrix pipe runs under the conversions.  Same MFMA sequence per output element: bit-identical to gemm_rows_kernel.
template <typename TW, int NRB>
__global__ __launch_bounds__(256) void gemm_rows2_kernel(GemmNTArgs a, RowTiling rt) {
  constexpr int BK = GemmCfg<TW>::BK, VK = GemmCfg<TW>::VK, kPlanes = GemmCfg<TW>::kPlanes;
  constexpr bool kF32 = GemmCfg<TW>::kF32, kWS = GemmCfg<TW>::kWS, kX6 = GemmCfg<TW>::kX6;
  typedef typename std::conditional<kWS, float, TW>::type TM;
  constexpr int XLPR = BK / 4, RPP = 256 / XLPR;
  constexpr int LROWS = gemm_rows_lds_rows<TW, NRB>();
  constexpr int NLD = LROWS / RPP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef unsigned char (*SmemT)[kPlanes][LROWS * kLdsRow];
  SmemT smem = reinterpret_cast<SmemT>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = (a.N + 63) >> 6;
  int t = (int)blockIdx.x;
  if (a.xcd) t = (t & 7) * ((int)gridDim.x >> 3) + (t >> 3);
  const int bx = t % nb, bz = t / nb;
  const int nrb = bz < rt.n_big ? rt.rb_big : rt.rb_big - 1;
  const int m0 = 16 * (bz < rt.n_big ? bz * rt.rb_big : rt.n_big * rt.rb_big + (bz - rt.n_big) * (rt.rb_big - 1));
  const int n0 = bx * 64;
  const int nsteps = a.K / BK;
  const int fi = lane & 15, fq = lane >> 4;
  const int wn = n0 + wave * 16 + fi;
  const bool wn_ok = wn < a.N;
  const TM* wrow = reinterpret_cast<const TM*>(a.W) + (long)(wn_ok ? wn : a.N - 1) * a.ldw;
  const int xpiece = tid % XLPR, xr0 = tid / XLPR;
  const int last_row = min(m0 + nrb * 16, a.M) - 1;
  const float* xrow[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) xrow[j] = a.X + (long)min(m0 + j * RPP + xr0, last_row) * a.ldx + xpiece * 4;

  struct WPieces { float f[kF32 ? 8 : 1]; bf16x8 hi[kF32 ? 1 : 2]; bf16x8 lo[kWS ? 2 : 1]; bf16x8 t3[kX6 ? 2 : 1]; };
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
  // one pass (RPP rows) of a K-step's X tile: convert and store into LDS buffer `buf`
  auto store_pass = [&](const float (&x)[4], int j, int buf) {
    const int r = j * RPP + xr0;
    if constexpr (kF32) {
      *reinterpret_cast<float4*>(&smem[buf][0][r * kLdsRow + xpiece * 16]) = make_float4(x[0], x[1], x[2], x[3]);
    } else {
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 h, l, m3;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h[q] = (__bf16)x[q];
        const float r1 = x[q] - (float)h[q];
        l[q] = (__bf16)r1;
        if constexpr (kX6) m3[q] = (__bf16)(r1 - (float)l[q]);
      }
      *reinterpret_cast<bf16x4*>(&smem[buf][0][r * kLdsRow + xpiece * 8]) = h;
      *reinterpret_cast<bf16x4*>(&smem[buf][1][r * kLdsRow + xpiece * 8]) = l;
      if constexpr (kX6) *reinterpret_cast<bf16x4*>(&smem[buf][2][r * kLdsRow + xpiece * 8]) = m3;
    }
  };
  // half h (8 values) of a K-step's W fragment -> its pieces
  auto split_w_half = [&](const float (&w32)[kF32 ? 8 : (kWS ? 16 : 1)], const bf16x8 (&w16)[(kF32 || kWS) ? 1 : 2], WPieces& o, int h) {
    if constexpr (kF32) {
      if (h == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o.f[j] = w32[j];
      }
    } else if constexpr (kWS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float w = w32[h * 8 + j];
        const __bf16 hi = (__bf16)w;
        o.hi[h][j] = hi;
        const float r1 = w - (float)hi;
        o.lo[h][j] = (__bf16)r1;
        if constexpr (kX6) o.t3[h][j] = (__bf16)(r1 - (float)o.lo[h][j]);
      }
    } else {
      o.hi[h] = w16[h];
    }
  };

  f32x4 acc[NRB];
#pragma unroll
  for (int r = 0; r < NRB; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int DEPTH = kF32 ? 1 : (kX6 ? 1 : (kWS ? 2 : 3));
  constexpr int NF = kF32 ? 2 : 2 * kPlanes;
  // the MFMAs of K-step (LDS buffer `buf`, W pieces `wc`); kPrep: step s + 1's operands (raw in xs[Q] / wf32[Q]) are prepared between the
  // row blocks into LDS buffer buf ^ 1 and `wn`
  auto k_step = [&](int buf, const WPieces& wc, auto prep) {
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
      __builtin_amdgcn_sched_barrier(0);
      prep(rb);
      const int sl = rb % (DEPTH + 1);
      if constexpr (kF32) {
        const float4 a0 = fa[sl][0], a1 = fa[sl][1];
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, wc.f[0], acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, wc.f[1], acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, wc.f[2], acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, wc.f[3], acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, wc.f[4], acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, wc.f[5], acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, wc.f[6], acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, wc.f[7], acc[rb], 0, 0, 0);
      } else {
        const bf16x8 a0 = fb[sl][0], a1 = fb[sl][1], b0 = fb[sl][2], b1 = fb[sl][3];
        if constexpr (kX6) {
          const bf16x8 c0 = fb[sl][4], c1 = fb[sl][5];
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, wc.hi[0], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c1, wc.hi[1], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wc.t3[0], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wc.t3[1], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, wc.lo[0], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, wc.lo[1], acc[rb], 0, 0, 0);
        }
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, wc.hi[0], acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, wc.hi[1], acc[rb], 0, 0, 0);
        if constexpr (kWS) {
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wc.lo[0], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wc.lo[1], acc[rb], 0, 0, 0);
        }
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wc.hi[0], acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wc.hi[1], acc[rb], 0, 0, 0);
      }
    }
  };

  // prologue: steps 0 and 1 requested; step 0 prepared; step 2 requested into step 0's registers
  WPieces wA, wB;
  load_w(wf32[0], wb16[0], 0); load_x(xs[0], 0);
  if (nsteps > 1) { load_w(wf32[1], wb16[1], BK); load_x(xs[1], BK); }
#pragma unroll
  for (int j = 0; j < NLD; ++j) store_pass(xs[0][j], j, 0);
  split_w_half(wf32[0], wb16[0], wA, 0); split_w_half(wf32[0], wb16[0], wA, 1);
  __syncthreads();
  if (nsteps > 2) { load_x(xs[0], 2 * BK); load_w(wf32[0], wb16[0], 2 * BK); }

  // chunk rb of the preparation of the step whose raw operands sit in register set Q, into LDS buffer `nbuf` and pieces `wn`:
  // X pass rb (and the passes beyond NRB with the last block), W half 0 with block 0, half 1 with block 1 (or 0 when NRB == 1)
  auto make_prep = [&](auto Qc, int nbuf, WPieces& wn) {
    return [&, nbuf](int rb) {
      constexpr int Q = decltype(Qc)::value;
      if (rb < NLD) store_pass(xs[Q][rb], rb, nbuf);
      if (rb == NRB - 1) {
#pragma unroll
        for (int j = NRB; j < NLD; ++j) store_pass(xs[Q][j], j, nbuf);
      }
      if (rb == 0) split_w_half(wf32[Q], wb16[Q], wn, 0);
      if (rb == (NRB > 1 ? 1 : 0)) split_w_half(wf32[Q], wb16[Q], wn, 1);
    };
  };
  auto no_prep = [](int) {};
  int s = 0;
  for (; s + 2 < nsteps; s += 2) {
    // step s (buffer 0, pieces wA) prepares step s + 1 (raw set 1) into buffer 1 / wB; then step s + 3 is requested into set 1
    k_step(0, wA, make_prep(std::integral_constant<int, 1>{}, 1, wB));
    __syncthreads();
    if (s + 3 < nsteps) { load_x(xs[1], (s + 3) * BK); load_w(wf32[1], wb16[1], (s + 3) * BK); }
    // step s + 1 (buffer 1, pieces wB) prepares step s + 2 (raw set 0) into buffer 0 / wA; then step s + 4 into set 0
    k_step(1, wB, make_prep(std::integral_constant<int, 0>{}, 0, wA));
    __syncthreads();
    if (s + 4 < nsteps) { load_x(xs[0], (s + 4) * BK); load_w(wf32[0], wb16[0], (s + 4) * BK); }
  }
  // tail: one step (buffer 0, nothing to prepare) or two (step s prepares step s + 1)
  if (s + 1 < nsteps) {
    k_step(0, wA, make_prep(std::integral_constant<int, 1>{}, 1, wB));
    __syncthreads();
    k_step(1, wB, no_prep);
  } else {
    k_step(0, wA, no_prep);
  }

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

