// Layout changes with fused dropout, time-major [L,B,W] <-> batch-major [B,L,W] (units.py:71-72 context dropout), as device bodies over
// VIRTUAL blocks of 256 threads: their own launches (encoder.hip) and extra workgroups of a narrow product's launch that carries a
// posted one (gemm.hip: vln_layout_post).
#pragma once
#include "common.h"

namespace vln {

__device__ __forceinline__ void tm_to_bm_body(const float* tm, float* bm, bf16_raw* bm_lp, int B, int L, int W, DropSpec dr, int vec,
                                              long vblk, long nvblk) {
  if (vec == 2) {                                        // W % 8 == 0: 8 columns per thread = ONE Philox call
    const int W8 = W >> 3;
    const long total8 = (long)B * L * W8;
    for (long e8 = vblk * 256 + threadIdx.x; e8 < total8; e8 += nvblk * 256) {
      const int c8 = (int)(e8 % W8);
      const long rb = e8 / W8;
      const int t = (int)(rb % L), b = (int)(rb / L);
      const float* src = tm + ((long)t * B + b) * W + c8 * 8;
      const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
      float m[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
      if (dr.p > 0.f) dropout_scale8(dr.seed, dr.off(), (uint32_t)e8, dr.p, m);
      const float v0[4] = {x0.x * m[0], x0.y * m[1], x0.z * m[2], x0.w * m[3]}, v1[4] = {x1.x * m[4], x1.y * m[5], x1.z * m[6], x1.w * m[7]};
      *reinterpret_cast<float4*>(bm + e8 * 8) = make_float4(v0[0], v0[1], v0[2], v0[3]);
      *reinterpret_cast<float4*>(bm + e8 * 8 + 4) = make_float4(v1[0], v1[1], v1[2], v1[3]);
      if (bm_lp) { Elt<bf16_raw>::st4(bm_lp + e8 * 8, v0); Elt<bf16_raw>::st4(bm_lp + e8 * 8 + 4, v1); }
    }
    return;
  }
  if (vec) {
    const int W4 = W >> 2;
    const long total4 = (long)B * L * W4;
    for (long e4 = vblk * 256 + threadIdx.x; e4 < total4; e4 += nvblk * 256) {
      const int c4 = (int)(e4 % W4);
      const long rb = e4 / W4;
      const int t = (int)(rb % L), b = (int)(rb / L);
      const float4 x = *reinterpret_cast<const float4*>(tm + ((long)t * B + b) * W + c4 * 4);
      float m[4] = {1.f, 1.f, 1.f, 1.f};
      if (dr.p > 0.f) dropout_scale4(dr.seed, dr.off(), (uint32_t)e4, dr.p, m);
      const float v[4] = {x.x * m[0], x.y * m[1], x.z * m[2], x.w * m[3]};
      *reinterpret_cast<float4*>(bm + e4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
      if (bm_lp) Elt<bf16_raw>::st4(bm_lp + e4 * 4, v);
    }
    return;
  }
  const long total = (long)B * L * W;
  for (long e = vblk * 256 + threadIdx.x; e < total; e += nvblk * 256) {
    const int c = (int)(e % W);
    const long rb = e / W;
    const int t = (int)(rb % L), b = (int)(rb / L);
    const float v = tm[((long)t * B + b) * W + c] * dropout_scale1(dr.seed, dr.off(), (uint32_t)e, dr.p);
    bm[e] = v;
    if (bm_lp) bm_lp[e] = f32_to_bf16_bits(v);
  }
}
__device__ __forceinline__ void bm_to_tm_body(const float* bm, float* tm, int B, int L, int W, DropSpec dr, int vec, long vblk, long nvblk) {
  if (vec == 2) {
    const int W8 = W >> 3;
    const long total8 = (long)B * L * W8;
    for (long e8 = vblk * 256 + threadIdx.x; e8 < total8; e8 += nvblk * 256) {
      const int c8 = (int)(e8 % W8);
      const long rb = e8 / W8;
      const int t = (int)(rb % L), b = (int)(rb / L);
      const float4 x0 = *reinterpret_cast<const float4*>(bm + e8 * 8), x1 = *reinterpret_cast<const float4*>(bm + e8 * 8 + 4);
      float m[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
      if (dr.p > 0.f) dropout_scale8(dr.seed, dr.off(), (uint32_t)e8, dr.p, m);
      float* dst = tm + ((long)t * B + b) * W + c8 * 8;
      *reinterpret_cast<float4*>(dst) = make_float4(x0.x * m[0], x0.y * m[1], x0.z * m[2], x0.w * m[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(x1.x * m[4], x1.y * m[5], x1.z * m[6], x1.w * m[7]);
    }
    return;
  }
  if (vec) {
    const int W4 = W >> 2;
    const long total4 = (long)B * L * W4;
    for (long e4 = vblk * 256 + threadIdx.x; e4 < total4; e4 += nvblk * 256) {
      const int c4 = (int)(e4 % W4);
      const long rb = e4 / W4;
      const int t = (int)(rb % L), b = (int)(rb / L);
      const float4 x = *reinterpret_cast<const float4*>(bm + e4 * 4);
      float m[4] = {1.f, 1.f, 1.f, 1.f};
      if (dr.p > 0.f) dropout_scale4(dr.seed, dr.off(), (uint32_t)e4, dr.p, m);
      *reinterpret_cast<float4*>(tm + ((long)t * B + b) * W + c4 * 4) = make_float4(x.x * m[0], x.y * m[1], x.z * m[2], x.w * m[3]);
    }
    return;
  }
  const long total = (long)B * L * W;
  for (long e = vblk * 256 + threadIdx.x; e < total; e += nvblk * 256) {
    const int c = (int)(e % W);
    const long rb = e / W;
    const int t = (int)(rb % L), b = (int)(rb / L);
    tm[((long)t * B + b) * W + c] = bm[e] * dropout_scale1(dr.seed, dr.off(), (uint32_t)e, dr.p);
  }
}


}  // namespace vln
