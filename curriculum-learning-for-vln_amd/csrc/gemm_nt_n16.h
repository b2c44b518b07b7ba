// gemm_nt for NARROW outputs (N <= 1024: the H-sized products of the decoder step, forward and dX): a workgroup owns
// 64 rows x 16 columns and its 4 waves split K four ways; the partial tiles meet in LDS, so there are no split-K
// slabs in global memory and no reduce launch -- bias / tanh / relu / the dropped second output are applied here.
// W fragments and X fragments both go global -> VGPR (each lane reads 32 contiguous bytes of a weight row, and 32 or
// 64 contiguous bytes of an activation row); X (<= 64 x K fp32, L2-resident) is re-read once per 16 columns.
// Included by gemm.hip after GemmCfg / GemmNTArgs.
#pragma once

struct GemmN16Args {
  const float* X; long ldx;
  const void* W; long ldw;
  float* Y; long ldy;
  const float* bias; int act;
  float* Y2; long ldy2; DropSpec drop;     // optional second output Y2 = Y * dropout mask (index r*N + c)
  int M, N, K, kq;                          // kq = K range per wave (multiple of BK)
  int xvec, wvec;
};

// the tile of column block bx, row block bz; red: 4 x 64 x 17 floats of LDS
template <typename TW>
__device__ __forceinline__ void gemm_nt_n16_body(const GemmN16Args& a, int bx, int bz, float (*red)[64][17]) {
  constexpr int BK = GemmCfg<TW>::BK, VK = GemmCfg<TW>::VK;
  constexpr bool kF32 = (sizeof(TW) == 4);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 15, fq = lane >> 4;
  const int n0 = bx * 16, m0 = bz * 64;
  const int wn = n0 + fi;
  const bool wn_ok = wn < a.N;
  const TW* wrow = reinterpret_cast<const TW*>(a.W) + (long)(wn_ok ? wn : 0) * a.ldw;
  const int kbeg = wave * a.kq, kend = min(a.K, kbeg + a.kq);
  const int mrows = min(64, a.M - m0);
  const int nrb = (mrows + 15) >> 4;

  f32x4 acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const int k = k0 + fq * VK;
    // ---- W fragment ----
    float wf[kF32 ? 8 : 1];
    bf16x8 wb[kF32 ? 1 : 2];
    if constexpr (kF32) {
      if (wn_ok && a.wvec && k + VK <= kend) {
        const float4 t0 = *reinterpret_cast<const float4*>(wrow + k), t1 = *reinterpret_cast<const float4*>(wrow + k + 4);
        wf[0] = t0.x; wf[1] = t0.y; wf[2] = t0.z; wf[3] = t0.w; wf[4] = t1.x; wf[5] = t1.y; wf[6] = t1.z; wf[7] = t1.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[j] = (wn_ok && (k + j) < kend) ? wrow[k + j] : 0.f;
      }
    } else {
      if (wn_ok && a.wvec && k + VK <= kend) {
        wb[0] = *reinterpret_cast<const bf16x8*>(wrow + k);
        wb[1] = *reinterpret_cast<const bf16x8*>(wrow + k + 8);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bf16_raw r0 = (wn_ok && (k + j) < kend) ? wrow[k + j] : (bf16_raw)0;
          const bf16_raw r1 = (wn_ok && (k + 8 + j) < kend) ? wrow[k + 8 + j] : (bf16_raw)0;
          wb[0][j] = __builtin_bit_cast(__bf16, r0);
          wb[1][j] = __builtin_bit_cast(__bf16, r1);
        }
      }
    }
    // ---- X fragments of the 4 row blocks + MFMA ----
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      if (rb < nrb) {
        const int row = m0 + rb * 16 + fi;
        const bool r_ok = row < a.M;
        const float* xr = a.X + (long)(r_ok ? row : 0) * a.ldx;
        float xv[VK];
        if (r_ok && a.xvec && k + VK <= kend) {
#pragma unroll
          for (int v = 0; v < VK / 4; ++v) {
            const float4 t = *reinterpret_cast<const float4*>(xr + k + v * 4);
            xv[v * 4] = t.x; xv[v * 4 + 1] = t.y; xv[v * 4 + 2] = t.z; xv[v * 4 + 3] = t.w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < VK; ++j) xv[j] = (r_ok && (k + j) < kend) ? xr[k + j] : 0.f;
        }
        if constexpr (kF32) {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[j], wf[j], acc[rb], 0, 0, 0);
        } else {
          bf16x8 a0, a1, l0, l1;   // activations split hi + lo: only the weight stream is quantised
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            a0[j] = (__bf16)xv[j];
            a1[j] = (__bf16)xv[8 + j];
            l0[j] = (__bf16)(xv[j] - (float)a0[j]);
            l1[j] = (__bf16)(xv[8 + j] - (float)a1[j]);
          }
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0, wb[0], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, wb[1], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wb[0], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wb[1], acc[rb], 0, 0, 0);
        }
      }
    }
  }
  // ---- cross-wave K reduction through LDS, then the epilogue ----
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][rb * 16 + fq * 4 + r][fi] = acc[rb][r];
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 16; e += 256) {
    const int r = e >> 4, c = e & 15;
    const int row = m0 + r, col = n0 + c;
    if (row < a.M && col < a.N) {
      float v = (red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c]);
      if (a.bias) v += a.bias[col];
      if (a.act == ACT_TANH) v = tanhf(v);
      else if (a.act == ACT_RELU) v = fmaxf(v, 0.f);
      a.Y[(long)row * a.ldy + col] = v;
      if (a.Y2) a.Y2[(long)row * a.ldy2 + col] = v * dropout_scale1(a.drop.seed, a.drop.off(), (uint32_t)((long)row * a.N + col), a.drop.p);
    }
  }
}

template <typename TW>
__global__ __launch_bounds__(256) void gemm_nt_n16_kernel(GemmN16Args a) {
  __shared__ float red[4][64][17];
  gemm_nt_n16_body<TW>(a, (int)blockIdx.x, (int)blockIdx.z, red);
}

// A POSTED layout change (vln_layout_post: the encoder's [L,B,W] <-> [B,L,W] copies with their dropout, layout_bodies.h) as extra
// workgroups of a narrow product's launch: the product's few tiles (32 workgroups for an H x H one) are a chain of latencies, the copy
// is HBM-bound, and in the encoder's forward and backward the two do not depend on each other.
struct LayoutArgs { int kind; const float* src; float* dst; bf16_raw* dst_lp; int B, L, W; DropSpec dr; int vec; int blocks; };   // kind 0: tm -> bm, 1: bm -> tm
template <typename TW>
__global__ __launch_bounds__(256) void gemm_nt_n16_layout_kernel(GemmN16Args a, int gx, int gz, LayoutArgs lay) {
  __shared__ float red[4][64][17];
  const int ng = gx * gz;
  if ((int)blockIdx.x < ng) { gemm_nt_n16_body<TW>(a, (int)blockIdx.x % gx, (int)blockIdx.x / gx, red); return; }
  const long vb = (long)blockIdx.x - ng;
  if (lay.kind == 0) tm_to_bm_body(lay.src, lay.dst, lay.dst_lp, lay.B, lay.L, lay.W, lay.dr, lay.vec, vb, (long)lay.blocks);
  else bm_to_tm_body(lay.src, lay.dst, lay.B, lay.L, lay.W, lay.dr, lay.vec, vb, (long)lay.blocks);
}
