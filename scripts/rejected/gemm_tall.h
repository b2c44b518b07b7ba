// gemm_tall: Y[M,N] = X[M,K] W[N,K]^T for TALL products -- M = B * L rows (the instruction encoder's input projection, the
// projected context K = ctx W_in and its gradient), K <= 512 (included by gemm.hip inside namespace vln).
//
// gemm_nt gives every 64 x 64 output tile its own workgroup and walks K in 64-wide steps through a double-buffered LDS stage
// of X with a barrier per step: at M = 5120 that is 640 workgroups of 8 dependent (global load -> LDS -> barrier -> MFMA)
// round trips each -- 24-38 us for 8 GFLOP of bf16-pipe work and 22 MB of traffic.  Here the roles are swapped:
//   * the workgroup's 64-column slice of W -- ALL of K -- is staged ONCE into LDS as bf16 planes (hi, and lo for fp32 weights:
//     <= 128 KB of the CU's 160 KB), conflict-free 16-byte rows;
//   * X never touches LDS: every wave streams its own 32 rows straight from global memory into MFMA A-fragments (lane (row i,
//     k-group q) reads 32 contiguous bytes of row i), split hi + lo in registers, four K-steps of loads in flight;
//   * no barrier after the staging one: 8 waves x 32 rows = 256 rows per workgroup run independently;
//   * W is the MFMA's A operand and X its B operand, and LDS slot t * 16 + i holds column (i / 4) * 16 + t * 4 + i % 4 of the
//     slice: a lane then ends up with 16 CONSECUTIVE output columns of one row (four 16-byte stores to 64 contiguous bytes, the
//     four lane groups of a row 256 contiguous bytes) instead of 4-byte stores scattered over four rows -- the scattered form's
//     stores cost more than the whole product (8 of 28 us at N = 512, 21 of 52 us at N = 2048).
// Arithmetic as gemm_nt's bf16 forms: W_BF16 = x_hi w + x_lo w, W_F32S = x_hi w_hi + x_lo w_hi + x_hi w_lo, fp32 accumulate
// (summation order differs from gemm_nt: same bound, different last bits).
#pragma once

struct GemmTallArgs {
  const float* X; long ldx;
  const void* W; long ldw;
  float* Y; long ldy;
  const float* bias; int act;
  int M, N, K;
};

constexpr int kTallCols = 64;            // columns per workgroup
constexpr int kTallRows = 256;           // rows per workgroup: 8 waves x 2 blocks of 16
constexpr int kTallPF = 4;               // K-steps of X loads in flight per wave

__host__ __device__ constexpr int gemm_tall_smem_bytes(int planes, int K) { return planes * kTallCols * (K * 2 + 16); }

template <typename TW, int KS>           // KS = K / 32 (8: K = 256, 16: K = 512)
__global__ __launch_bounds__(512) void gemm_tall_kernel(GemmTallArgs a) {
  constexpr bool kWS = std::is_same<TW, f32s_raw>::value;
  constexpr int K = KS * 32;
  constexpr int kRow = K * 2 + 16;                         // bytes per staged column (16-byte pad: conflict-free ds_read_b128)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Workgroup -> (column tile, row range), XCD-aware: the column tiles of one row range all stream the same 256 rows of X.
  // Workgroups are dealt to the 8 XCDs round-robin in launch order, so id & 7 names the XCD: row range = 8 * (group) + XCD and the
  // column tile walks within the XCD -- the sharers of an X block run behind ONE L2 (in grid order they sat on 8 different
  // XCDs and every L2 fetched every row for itself: 8x the fabric traffic, what bounded the first form of this kernel).
  const int ncol = (a.N + kTallCols - 1) / kTallCols, nrow = (a.M + kTallRows - 1) / kTallRows;
  const int in_xcd = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
  const int rowr = (in_xcd / ncol) * 8 + xcd;
  if (rowr >= nrow) return;
  const int n0 = (in_xcd % ncol) * kTallCols, m0 = rowr * kTallRows;

  // ---- stage W[n0 .. n0 + 64, 0 .. K) -> bf16 planes ---------------------------------------------------------------------------------
  if constexpr (kWS) {
    const float* W = reinterpret_cast<const float*>(a.W);
    constexpr int PER = kTallCols * K / 4 / 512;           // float4 loads per thread
#pragma unroll 4
    for (int i = 0; i < PER; ++i) {
      const int idx = tid + i * 512;
      const int slot = idx / (K / 4), k4 = idx % (K / 4);
      const int col = slot, wcol = ((slot & 15) >> 2) * 16 + (slot >> 4) * 4 + (slot & 3);      // slot -> the column it holds
      const float4 t = *reinterpret_cast<const float4*>(W + (long)min(n0 + wcol, a.N - 1) * a.ldw + k4 * 4);
      const float x[4] = {t.x, t.y, t.z, t.w};
      bf16x4 h, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) { h[j] = (__bf16)x[j]; l[j] = (__bf16)(x[j] - (float)h[j]); }
      *reinterpret_cast<bf16x4*>(smem + col * kRow + k4 * 8) = h;
      *reinterpret_cast<bf16x4*>(smem + kTallCols * kRow + col * kRow + k4 * 8) = l;
    }
  } else {
    const bf16_raw* W = reinterpret_cast<const bf16_raw*>(a.W);
    constexpr int PER = kTallCols * K / 8 / 512;           // 16-byte loads per thread
#pragma unroll 4
    for (int i = 0; i < PER; ++i) {
      const int idx = tid + i * 512;
      const int slot = idx / (K / 8), k8 = idx % (K / 8);
      const int col = slot, wcol = ((slot & 15) >> 2) * 16 + (slot >> 4) * 4 + (slot & 3);
      const uint4 t = *reinterpret_cast<const uint4*>(W + (long)min(n0 + wcol, a.N - 1) * a.ldw + k8 * 8);
      *reinterpret_cast<uint4*>(smem + col * kRow + k8 * 16) = t;
    }
  }

  // ---- this wave's 32 rows: two 16-row blocks share every W fragment read ------------------------------------------------------------
  const int fi = lane & 15, fq = lane >> 4;
  const int r0 = m0 + wave * 32;
  const float* xr[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) xr[blk] = a.X + (long)min(r0 + blk * 16 + fi, a.M - 1) * a.ldx + fq * 8;
  float4 xa[kTallPF][2][2];
  auto load_x = [&](int slot, int ks) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      xa[slot][blk][0] = *reinterpret_cast<const float4*>(xr[blk] + ks * 32);
      xa[slot][blk][1] = *reinterpret_cast<const float4*>(xr[blk] + ks * 32 + 4);
    }
  };
#pragma unroll
  for (int p = 0; p < kTallPF; ++p) load_x(p, p);
  f32x4 acc[2][4];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[blk][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  if (r0 >= a.M) return;                                   // (after the only barrier) a wave past the last row

#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int slot = ks % kTallPF;
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const float x[8] = {xa[slot][blk][0].x, xa[slot][blk][0].y, xa[slot][blk][0].z, xa[slot][blk][0].w,
                          xa[slot][blk][1].x, xa[slot][blk][1].y, xa[slot][blk][1].z, xa[slot][blk][1].w};
#pragma unroll
      for (int j = 0; j < 8; ++j) { ah[blk][j] = (__bf16)x[j]; al[blk][j] = (__bf16)(x[j] - (float)ah[blk][j]); }
    }
    if (ks + kTallPF < KS) load_x(slot, ks + kTallPF);
    // all W fragments of the K-step first, then the products TERM-major: between two MFMAs into the same accumulator stand seven
    // independent ones (back-to-back dependent MFMAs wait out the whole pipeline: 3 x the issue time in the tile-major order)
    bf16x8 bh[4], bl[kWS ? 4 : 1];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned char* wp = smem + (t * 16 + fi) * kRow + ks * 64 + fq * 16;
      bh[t] = *reinterpret_cast<const bf16x8*>(wp);
      if constexpr (kWS) bl[t] = *reinterpret_cast<const bf16x8*>(wp + kTallCols * kRow);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) acc[blk][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[t], al[blk], acc[blk][t], 0, 0, 0);
    if constexpr (kWS) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) acc[blk][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[t], ah[blk], acc[blk][t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) acc[blk][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[t], ah[blk], acc[blk][t], 0, 0, 0);
  }

  // ---- epilogue.  D = W-fragment x X-fragment: lane (fi, fq) holds D[slot row fq * 4 + r][X row fi] of tile t, i.e. output
  // row r0 + blk * 16 + fi, columns n0 + fq * 16 + t * 4 + r: 16 consecutive columns per lane and block ------------------------------
  const int actk = a.act & 3;
  const int cb = n0 + fq * 16;
  __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, (unsigned)(((long)(a.M - 1) * a.ldy + a.N) * 4), 0x00020000);
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int row = r0 + blk * 16 + fi;
    if (row >= a.M) continue;
    float* y = a.Y + (long)row * a.ldy + cb;
    if (cb + 16 <= a.N && (a.ldy & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.Y) & 15) == 0)) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[blk][t][r] + (a.bias ? a.bias[cb + t * 4 + r] : 0.f);
          if (actk == ACT_TANH) v[r] = tanhf(v[r]);
          else if (actk == ACT_RELU) v[r] = fmaxf(v[r], 0.f);
        }
        float4* yp = reinterpret_cast<float4*>(y + t * 4);
        if (a.act & ACT_ACCUM) { const float4 o = *yp; v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
        // write-through (sc1): the tile's bytes leave the XCD's L2 while the launch still computes, instead of as one write-back
        // of every dirty line at the kernel's end (MI355X_MICROARCH.md "publish-large")
        const u32x4_t o4 = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
        __builtin_amdgcn_raw_buffer_store_b128(o4, yres, (unsigned)(((long)row * a.ldy + cb + t * 4) * 4), 0, 16);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = cb + t * 4 + r;
          if (col < a.N) {
            float v = acc[blk][t][r] + (a.bias ? a.bias[col] : 0.f);
            if (actk == ACT_TANH) v = tanhf(v);
            else if (actk == ACT_RELU) v = fmaxf(v, 0.f);
            if (a.act & ACT_ACCUM) v += y[t * 4 + r];
            y[t * 4 + r] = v;
          }
        }
    }
  }
}

// The tall form takes the product when it pays (many row tiles, the whole K resident in LDS) and the operands allow its
// unconditional 16-byte loads; tunable[9] = 1 switches it off (A/B).
static bool gemm_tall_applies(const float* X, long ldx, const void* W, int wtype, long ldw, int M, int N, int K) {
  if (g_tunable[9] == 1) return false;
  if (wtype != W_BF16 && wtype != W_F32S) return false;
  if (M < 1024 || (K != 256 && K != 512) || N < kTallCols) return false;
  if (!aligned16(X) || (ldx & 3) || !aligned16(W) || (ldw % (wtype == W_BF16 ? 8 : 4)) != 0) return false;
  return true;
}

static int gemm_tall(hipStream_t st, const float* X, long ldx, const void* W, int wtype, long ldw, float* Y, long ldy, int M, int N,
                     int K, const float* bias, int act) {
  GemmTallArgs a{X, ldx, W, ldw, Y, ldy, bias, act, M, N, K};
  const int ncol = (N + kTallCols - 1) / kTallCols, nrow = (M + kTallRows - 1) / kTallRows;
  const dim3 grid(ncol * 8 * ((nrow + 7) / 8)), block(512);          // (row ranges padded to a multiple of 8: one per XCD and group)
  const double bytes = (double)N * K * (wtype == W_BF16 ? 2 : 4) + 4.0 * M * K + 4.0 * M * N;
  const unsigned lds = (unsigned)gemm_tall_smem_bytes(wtype == W_F32S ? 2 : 1, K);
#define VLN_TALL(TW, KSv) do { \
    static bool attr_set = false; \
    if (!attr_set) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tall_kernel<TW, KSv>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess) { (void)hipGetLastError(); } attr_set = true; } \
    launch_timed(K_GEMM_NT, bytes, gemm_tall_kernel<TW, KSv>, grid, block, lds, st, a); } while (0)
  if (wtype == W_F32S) { if (K == 512) VLN_TALL(f32s_raw, 16); else VLN_TALL(f32s_raw, 8); }
  else { if (K == 512) VLN_TALL(bf16_raw, 16); else VLN_TALL(bf16_raw, 8); }
#undef VLN_TALL
  VLN_CHECK_LAUNCH("gemm_tall");
  return VLN_OK;
}
