// Feature gather of one decoder step, one table row per 256-thread (virtual) block (included inside namespace vln by
// features.hip and chain.hip).
#pragma once
#include "envdrop_prep.h"

namespace vln {

// ---- one launch per decoder step: panorama rows + candidate rows ------------------------------------------------
// Same outputs, same Philox indexing as the two kernels above (so vln_dropout_mask exports the same masks); a thread
// handles 8 consecutive elements (one 16-byte load from a bf16 table), a row of 2176 is one pass of 272 threads.
template <typename TT>
__device__ __forceinline__ void gather_step_row(const GatherStepArgs& a, int r, int tid) {
  const int F = a.IMG + a.ANG, IMG = a.IMG;
  const TT* table = reinterpret_cast<const TT*>(a.table);
  const bool pano = r < a.B * a.V;
  const TT* src; float* dst; bf16_raw* dlp; DropSpec dr;
  const float* ang = nullptr;
  float sh = 0.f, ch = 0.f, se = 0.f, ce = 0.f;
  bool empty = false;
  if (pano) {
    const int b = r / a.V, v = r % a.V;
    src = table + ((long)a.rows[b] * a.V + v) * IMG;
    ang = a.angle_table + ((long)a.view_index[b] * a.V + v) * a.ANG;
    dst = a.out ? a.out + (long)r * F : nullptr;
    dlp = a.out_lp ? a.out_lp + (long)r * F : nullptr;
    dr = a.dr_pano;
  } else {
    r -= a.B * a.V;
    const long row = a.crows[r];
    empty = row < 0;                      // STOP slot / padding: all-zero feature (base.py:152-153)
    if (!empty) { sh = sinf(a.heading[r]); ch = cosf(a.heading[r]); se = sinf(a.elevation[r]); ce = cosf(a.elevation[r]); }
    src = empty ? table : table + (row * a.V + a.cviews[r]) * IMG;
    dst = a.cout ? a.cout + (long)r * F : nullptr;
    dlp = a.cout_lp ? a.cout_lp + (long)r * F : nullptr;
    dr = a.dr_cand;
  }
  const int q = a.ANG >> 2;
  for (int c = tid * 8; c < F; c += 256 * 8) {       // IMG % 8 == 0, ANG % 8 == 0: a group never straddles
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!empty) {
      if (c < IMG) {
        if constexpr (sizeof(TT) == 2) Elt<bf16_raw>::ld16(reinterpret_cast<const bf16_raw*>(src) + c, x);
        else {
          const float4 t0 = *reinterpret_cast<const float4*>(src + c), t1 = *reinterpret_cast<const float4*>(src + c + 4);
          x[0] = t0.x; x[1] = t0.y; x[2] = t0.z; x[3] = t0.w; x[4] = t1.x; x[5] = t1.y; x[6] = t1.z; x[7] = t1.w;
        }
        if (dr.p > 0.f) {
          float m[4];
          const uint32_t i4 = (uint32_t)(((long)r * IMG + c) >> 2);
          dropout_scale4(dr.seed, dr.off(), i4, dr.p, m);
          x[0] *= m[0]; x[1] *= m[1]; x[2] *= m[2]; x[3] *= m[3];
          dropout_scale4(dr.seed, dr.off(), i4 + 1, dr.p, m);
          x[4] *= m[0]; x[5] *= m[1]; x[6] *= m[2]; x[7] *= m[3];
        }
      } else if (pano) {
        const float4 t0 = *reinterpret_cast<const float4*>(ang + (c - IMG)), t1 = *reinterpret_cast<const float4*>(ang + (c - IMG) + 4);
        x[0] = t0.x; x[1] = t0.y; x[2] = t0.z; x[3] = t0.w; x[4] = t1.x; x[5] = t1.y; x[6] = t1.z; x[7] = t1.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int g = (c - IMG + j) / q;
          x[j] = g == 0 ? sh : (g == 1 ? ch : (g == 2 ? se : ce));
        }
      }
    }
    if (dst) {
      *reinterpret_cast<float4*>(dst + c) = make_float4(x[0], x[1], x[2], x[3]);
      *reinterpret_cast<float4*>(dst + c + 4) = make_float4(x[4], x[5], x[6], x[7]);
    }
    if (dlp) {
      uint4 v;
      v.x = (uint32_t)f32_to_bf16_bits(x[0]) | ((uint32_t)f32_to_bf16_bits(x[1]) << 16);
      v.y = (uint32_t)f32_to_bf16_bits(x[2]) | ((uint32_t)f32_to_bf16_bits(x[3]) << 16);
      v.z = (uint32_t)f32_to_bf16_bits(x[4]) | ((uint32_t)f32_to_bf16_bits(x[5]) << 16);
      v.w = (uint32_t)f32_to_bf16_bits(x[6]) | ((uint32_t)f32_to_bf16_bits(x[7]) << 16);
      *reinterpret_cast<uint4*>(dlp + c) = v;
    }
  }
}
}  // namespace vln
