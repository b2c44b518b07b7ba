// Device bodies of the grouped weight-gradient launches (pack -> packed contraction) and of the grouped bias-gradient column
// sums, written against a VIRTUAL block index so that two callers share them: the stand-alone kernels of gemm.hip (block index =
// blockIdx) and the PASSENGER workgroups of the backward recurrence launch (encoder_persist.h / wgrad_ride.h: a passenger walks
// the blocks p, p + NP, ...).  The job tables are templates over their capacity: the stand-alone launches take the library-wide
// maxima, a ride travels by value inside another kernel's argument block and uses small ones.
#pragma once

struct PackJob { const float* src; long ld; int C; long dst; long seg_stride; int rows; };   // dst: byte offset of the hi plane in the pack area; rows: the operand's own row count (<= Mt; rows past it pack as zeros)
// SEGMENTED rows (rollout-level weight gradients of per-step C calls): row m of an operand lives at
// src + (m / seg_rows) * seg_stride + (m % seg_rows) * ld -- step t's [seg_rows, C] block sits seg_stride floats after step
// t-1's (the steps' saved-activation / scratch blocks are slots of one arena).  seg_rows == 0: plain rows, m * ld.
template <int NJ>
struct PackJobsT {
  PackJob j[2 * NJ];
  int blk0[2 * NJ + 1];
  unsigned char* area; int n, Mt, MS;
  int lo;                 // 1: hi + lo planes (split-bf16 contraction); 0: hi plane only (plain bf16 operands)
  int seg_rows;
};
__device__ __forceinline__ long pack_plane_bytes(int C, int MS) { return (long)((C + 15) / 16) * MS * 1024; }
template <class PJ>
__device__ __forceinline__ void wgrad_pack_block(const PJ& a, int vbx) {
  int ji = 0;
  while (ji + 1 < a.n && vbx >= a.blk0[ji + 1]) ++ji;
  const PackJob q = a.j[ji];
  const int blk = vbx - a.blk0[ji];
  const int ncb = (q.C + 127) / 128;
  const int cb = blk % ncb, rb = blk / ncb;                       // 128 columns x 64 rows per workgroup
  const int half = threadIdx.x >> 7, t = threadIdx.x & 127, cg = t & 31, mq = t >> 5;
  const int ms = rb * 2 + half;
  if (ms >= a.MS) return;
  const int col = cb * 128 + cg * 4;
  if (col >= ((q.C + 15) & ~15)) return;                          // beyond the last (zero-padded) column tile
  const bool col_ok = col < q.C;                                  // C % 4 == 0
  const float* pc = q.src + (col_ok ? col : 0);
  float4 r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = ms * 32 + mq * 8 + i;
    const int mc = min(m, q.rows - 1);
    const long roff = a.seg_rows ? (long)(mc / a.seg_rows) * q.seg_stride + (long)(mc % a.seg_rows) * q.ld : (long)mc * q.ld;
    const float4 v = *reinterpret_cast<const float4*>(pc + roff);
    const bool ok = col_ok && m < q.rows;
    r[i] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
  }
  unsigned char* hi = a.area + q.dst;
  unsigned char* lo = hi + pack_plane_bytes(q.C, a.MS);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    bf16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = (c == 0) ? r[i].x : (c == 1) ? r[i].y : (c == 2) ? r[i].z : r[i].w;
      h[i] = (__bf16)v;
      l[i] = (__bf16)(v - (float)h[i]);
    }
    const int column = col + c;
    const long off = (((long)(column >> 4) * a.MS + ms) * 64 + (mq * 16 + (column & 15))) * 16;
    *reinterpret_cast<bf16x8*>(hi + off) = h;
    if (a.lo) *reinterpret_cast<bf16x8*>(lo + off) = l;
  }
}


#ifndef VLN_WGRAD_PACKED_BUFS
#define VLN_WGRAD_PACKED_BUFS 2
#endif
template <int NJ>
struct PackedJobsT {
  vln_wgrad_job j[NJ];
  long pa[NJ], px[NJ];   // byte offsets of the packed dy / x operands (hi plane)
  int tile0[NJ + 1];
  long slab0[NJ];
  unsigned char* area; float* ws;
  int n, Mt, MS, msplit, schunk, ntiles, per_xcd;          // schunk: row steps per split
};
// TERMS = 3: D = Ah Xh + Ah Xl + Al Xh (split bf16: fp32-grade products, error 2^-16); TERMS = 1: D = Ah Xh (plain bf16 operands,
// fp32 accumulation: what mixed-precision training computes; a third of the MFMAs, half the fragment loads)
template <int TERMS, class PJ>
__device__ __forceinline__ void wgrad_packed_tile(const PJ& a, int vbx, int vby) {
  const int lt = (vbx & 7) * a.per_xcd + (vbx >> 3);       // XCD-aware tile order (vbx % 8 = the XCD the caller runs on)
  if ((vbx >> 3) >= a.per_xcd || lt >= a.ntiles) return;
  int ji = 0;
  while (ji + 1 < a.n && lt >= a.tile0[ji + 1]) ++ji;
  const vln_wgrad_job& q = a.j[ji];
  const int tile = lt - a.tile0[ji];
  const int nbk = (q.K + 127) / 128;
  const int n0 = (tile / nbk) * 128, k0 = (tile % nbk) * 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fi = lane & 15, fq = lane >> 4;
  const int wn = n0 + (wave >> 1) * 64, wk = k0 + (wave & 1) * 64;
  const int s_beg = vby * a.schunk, s_end = min(a.MS, s_beg + a.schunk);
  const int nct = (q.N + 15) / 16, kct = (q.K + 15) / 16;
  const unsigned char* Ah = a.area + a.pa[ji];
  const unsigned char* Al = Ah + pack_plane_bytes(q.N, a.MS);
  const unsigned char* Xh = a.area + a.px[ji];
  const unsigned char* Xl = Xh + pack_plane_bytes(q.K, a.MS);
  long oa[4], ox[4];                       // per-fragment base offsets (column tiles past the edge are clamped: their
#pragma unroll                             // products land in rows / columns the epilogue does not write)
  for (int i = 0; i < 4; ++i) {
    oa[i] = ((long)min(wn / 16 + i, nct - 1) * a.MS * 64 + lane) * 16;
    ox[i] = ((long)min(wk / 16 + i, kct - 1) * a.MS * 64 + lane) * 16;
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NB = VLN_WGRAD_PACKED_BUFS;      // 2: next step's fragments load behind this step's MFMAs (256 VGPRs, one
                                                 // workgroup per CU); 1: half the registers, co-resident workgroups overlap instead
  bf16x8 ah[NB][4], al[NB][4], xh[NB][4], xl[NB][4];
  auto load = [&](int buf, int ms) {
    const long so = (long)ms * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[buf][i] = *reinterpret_cast<const bf16x8*>(Ah + oa[i] + so);
      xh[buf][i] = *reinterpret_cast<const bf16x8*>(Xh + ox[i] + so);
      if constexpr (TERMS == 3) {
        al[buf][i] = *reinterpret_cast<const bf16x8*>(Al + oa[i] + so);
        xl[buf][i] = *reinterpret_cast<const bf16x8*>(Xl + ox[i] + so);
      }
    }
  };
  auto mma = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (TERMS == 3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[buf][i], xh[buf][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[buf][i], xl[buf][j], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[buf][i], xh[buf][j], acc[i][j], 0, 0, 0);
      }
  };
  if constexpr (NB == 2) {
    if (s_beg < s_end) load(0, s_beg);
    for (int ms = s_beg; ms < s_end; ms += 2) {
      if (ms + 1 < s_end) load(NB - 1, ms + 1);
      mma(0);
      if (ms + 1 < s_end) {
        if (ms + 2 < s_end) load(0, ms + 2);
        mma(NB - 1);
      }
    }
  } else {
    for (int ms = s_beg; ms < s_end; ++ms) { load(0, ms); mma(0); }
  }
  float* D = q.dw; long ldd = q.ld_dw; int accumulate = q.accumulate;
  if (a.msplit > 1) { D = a.ws + a.slab0[ji] + (long)vby * q.N * q.K; ldd = q.K; accumulate = 0; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float old[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kcol = min(wk + j * 16 + fi, q.K - 1), nrow = min(wn + i * 16 + fq * 4 + r, q.N - 1);
        old[j][r] = accumulate ? D[(long)nrow * ldd + kcol] : 0.f;
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kcol = wk + j * 16 + fi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nrow = wn + i * 16 + fq * 4 + r;
        if (nrow < q.N && kcol < q.K) D[(long)nrow * ldd + kcol] = acc[i][j][r] + old[j][r];
      }
    }
  }
}


template <int NJ>
struct ColsumJobsT {
  vln_colsum_job j[NJ];
  int blk0[NJ + 1];
  int col0[NJ + 1];      // first column of the job in the partial buffer
  float* ws; int n, rows, rsplit, rchunk, total_cols;
  int seg_rows; long seg_stride[NJ];      // segmented rows as in PackJobs (seg_rows == 0: plain)
};
// part: 64 x 4 float4 of LDS (4 KB), free at entry; every thread of the workgroup calls this (two barriers inside)
template <class CJ>
__device__ __forceinline__ void colsum_grouped_block(const CJ& a, int vbx, int vby, float4 (*part)[4]) {
  __syncthreads();                                    // a previous block's readers are done with `part`
  int ji = 0;
  while (ji + 1 < a.n && vbx >= a.blk0[ji + 1]) ++ji;
  const vln_colsum_job& q = a.j[ji];
  const int cb = vbx - a.blk0[ji];
  const int cg = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = cb * 16 + cg * 4;
  const int rbeg = vby * a.rchunk, rend = min(a.rows, rbeg + a.rchunk);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (c < q.cols) {                                   // cols % 4 == 0
    const float* p = q.A + c;
    const long sst = a.seg_stride[ji];
    auto roff = [&](int r) { return a.seg_rows ? (long)(r / a.seg_rows) * sst + (long)(r % a.seg_rows) * q.lda : (long)r * q.lda; };
    int r = rbeg + rl;
    for (; r + 64 < rend; r += 128) {
      const float4 x = *reinterpret_cast<const float4*>(p + roff(r));
      const float4 y = *reinterpret_cast<const float4*>(p + roff(r + 64));
      s0.x += x.x; s0.y += x.y; s0.z += x.z; s0.w += x.w;
      s1.x += y.x; s1.y += y.y; s1.z += y.z; s1.w += y.w;
    }
    if (r < rend) {
      const float4 x = *reinterpret_cast<const float4*>(p + roff(r));
      s0.x += x.x; s0.y += x.y; s0.z += x.z; s0.w += x.w;
    }
  }
  part[rl][cg] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
  __syncthreads();
  if (threadIdx.x < 16) {
    const int g = threadIdx.x >> 2, e = threadIdx.x & 3, cc = cb * 16 + threadIdx.x;
    if (cc < q.cols) {
      float t = 0.f;
#pragma unroll 8
      for (int k = 0; k < 64; ++k) t += reinterpret_cast<const float*>(&part[k][g])[e];
      if (a.rsplit > 1) a.ws[(long)vby * a.total_cols + a.col0[ji] + cc] = t;
      else {
        if (q.out1) q.out1[cc] = q.accumulate ? q.out1[cc] + t : t;
        if (q.out2) q.out2[cc] = q.accumulate ? q.out2[cc] + t : t;
      }
    }
  }
}
