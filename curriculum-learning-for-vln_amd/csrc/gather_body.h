// Feature gather of one decoder step, one table row per 256-thread (virtual) block (included inside namespace vln by
// features.hip and chain.hip).
#pragma once
#include "envdrop_prep.h"

namespace vln {

// ---- one launch per decoder step: panorama rows + candidate rows ------------------------------------------------
// Same outputs, same Philox indexing as the two kernels above (so vln_dropout_mask exports the same masks).  A 256-thread
// block takes RPB consecutive output rows; a thread handles 8 consecutive image elements of each (one 16-byte load from a
// bf16 table: 2048 elements = exactly one pass of the block) with the loads of all RPB rows in flight together -- the
// rows are scattered over a 1.5 GB table, every one is a cold HBM access behind an index lookup -- and the first ANG / 8
// threads also write the row's angle columns (a copy for panorama rows, sin / cos of the heading for candidates).
template <typename TT, int RPB>
__device__ __forceinline__ void gather_step_rows(const GatherStepArgs& a, int r0, int nrows, int tid) {
  const int F = a.IMG + a.ANG, IMG = a.IMG;
  const TT* table = reinterpret_cast<const TT*>(a.table);
  const int npano = a.B * a.V;
  const TT* src[RPB]; float* dst[RPB]; bf16_raw* dlp[RPB]; bool pano[RPB], empty[RPB], live[RPB]; int rr[RPB];
#pragma unroll
  for (int k = 0; k < RPB; ++k) {
    int r = r0 + k;
    live[k] = r < nrows;
    if (!live[k]) r = nrows - 1;
    pano[k] = r < npano;
    if (pano[k]) {
      const int b = r / a.V, v = r % a.V;
      const long row = a.rows[b];
      const bool bad = a.chk.n_rows && (row < 0 || row >= a.chk.n_rows || a.view_index[b] < 0 || a.view_index[b] >= a.chk.n_aviews);
      if (bad && live[k] && tid == 0) __hip_atomic_fetch_add(a.chk.bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      src[k] = bad ? table : table + (row * a.V + v) * IMG;
      dst[k] = a.out ? a.out + (long)r * F : nullptr;
      dlp[k] = a.out_lp ? a.out_lp + (long)r * F : nullptr;
      empty[k] = bad;                          // an out-of-range index reads nothing: zeros, counted (GatherCheck)
      rr[k] = r;
    } else {
      const int rc = r - npano;
      const long row = a.crows[rc];
      const int cv = a.cviews[rc];
      const bool bad = a.chk.n_rows && row >= 0 && (row >= a.chk.n_rows || cv < 0 || cv >= a.V);
      if (bad && live[k] && tid == 0) __hip_atomic_fetch_add(a.chk.bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      empty[k] = row < 0 || bad;               // STOP slot / padding: all-zero feature (base.py:152-153)
      src[k] = empty[k] ? table : table + (row * a.V + cv) * IMG;
      dst[k] = a.cout ? a.cout + (long)rc * F : nullptr;
      dlp[k] = a.cout_lp ? a.cout_lp + (long)rc * F : nullptr;
      rr[k] = rc;
    }
  }
  auto store8 = [&](int k, int c, const float (&x)[8]) {
    if (dst[k]) {
      *reinterpret_cast<float4*>(dst[k] + c) = make_float4(x[0], x[1], x[2], x[3]);
      *reinterpret_cast<float4*>(dst[k] + c + 4) = make_float4(x[4], x[5], x[6], x[7]);
    }
    if (dlp[k]) {
      uint4 v;
      v.x = (uint32_t)f32_to_bf16_bits(x[0]) | ((uint32_t)f32_to_bf16_bits(x[1]) << 16);
      v.y = (uint32_t)f32_to_bf16_bits(x[2]) | ((uint32_t)f32_to_bf16_bits(x[3]) << 16);
      v.z = (uint32_t)f32_to_bf16_bits(x[4]) | ((uint32_t)f32_to_bf16_bits(x[5]) << 16);
      v.w = (uint32_t)f32_to_bf16_bits(x[6]) | ((uint32_t)f32_to_bf16_bits(x[7]) << 16);
      *reinterpret_cast<uint4*>(dlp[k] + c) = v;
    }
  };
  // image columns
  for (int c = tid * 8; c < IMG; c += 256 * 8) {                 // IMG % 8 == 0
    float x[RPB][8];
#pragma unroll
    for (int k = 0; k < RPB; ++k) {       // every row's load goes out before any is used: unconditional (an empty slot reads table row 0)
      if constexpr (sizeof(TT) == 2) {
        Elt<bf16_raw>::ld16(reinterpret_cast<const bf16_raw*>(src[k]) + c, x[k]);
      } else {
        const float4 t0 = *reinterpret_cast<const float4*>(src[k] + c), t1 = *reinterpret_cast<const float4*>(src[k] + c + 4);
        x[k][0] = t0.x; x[k][1] = t0.y; x[k][2] = t0.z; x[k][3] = t0.w; x[k][4] = t1.x; x[k][5] = t1.y; x[k][6] = t1.z; x[k][7] = t1.w;
      }
    }
#pragma unroll
    for (int k = 0; k < RPB; ++k)
      if (empty[k]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[k][j] = 0.f;
      }
#pragma unroll
    for (int k = 0; k < RPB; ++k) {
      if (!live[k]) continue;
      const DropSpec& dr = pano[k] ? a.dr_pano : a.dr_cand;
      if (!empty[k] && dr.p > 0.f) {
        float m[8];                    // IMG % 8 == 0 and c % 8 == 0: the thread's 8 elements are exactly one Philox call
        dropout_scale8(dr.seed, dr.off(), (uint32_t)(((long)rr[k] * IMG + c) >> 3), dr.p, m);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[k][j] *= m[j];
      }
      store8(k, c, x[k]);
    }
  }
  // angle columns (ANG % 8 == 0): a handful of threads per row
  const int q = a.ANG >> 2;
  for (int ca = tid * 8; ca < a.ANG; ca += 256 * 8) {
#pragma unroll
    for (int k = 0; k < RPB; ++k) {
      if (!live[k]) continue;
      float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (pano[k] && !empty[k]) {
        const int b = rr[k] / a.V, v = rr[k] % a.V;
        const float* ang = a.angle_table + ((long)a.view_index[b] * a.V + v) * a.ANG;
        const float4 t0 = *reinterpret_cast<const float4*>(ang + ca), t1 = *reinterpret_cast<const float4*>(ang + ca + 4);
        x[0] = t0.x; x[1] = t0.y; x[2] = t0.z; x[3] = t0.w; x[4] = t1.x; x[5] = t1.y; x[6] = t1.z; x[7] = t1.w;
      } else if (!pano[k] && !empty[k]) {
        const float sh = sinf(a.heading[rr[k]]), ch = cosf(a.heading[rr[k]]), se = sinf(a.elevation[rr[k]]), ce = cosf(a.elevation[rr[k]]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int g = (ca + j) / q;
          x[j] = g == 0 ? sh : (g == 1 ? ch : (g == 2 ? se : ce));
        }
      }
      store8(k, IMG + ca, x);
    }
  }
}
template <typename TT>
__device__ __forceinline__ void gather_step_row(const GatherStepArgs& a, int r, int tid) {
  gather_step_rows<TT, 1>(a, r, a.B * a.V + a.B * a.C, tid);
}
}  // namespace vln
