// The rollout-wide feature gather as PASSENGER workgroups of the persistent recurrence launch (encoder_persist_g.h).
//
// A passenger compute unit holds ONE 256-thread workgroup (the launch claims a CU's LDS per workgroup so that passengers never
// share a CU with a recurrence workgroup), i.e. one wave per SIMD: nothing hides a wave's memory latency or its Philox
// arithmetic but the wave itself.  Measured with the plain per-row loop: 287 us for the 7 x 2944 rows of the headline
// rollout, longer than the 179 us recurrence it rides in (profiles/round3_notes.md); this loop: 150 us alone.  It is a three-stage
// software pipeline over groups of 8 output rows, every memory operation unconditional (clamped addresses, idempotent
// duplicate stores for the rows past the end of a step) so that the wait counts the compiler places stay exact:
//   A(i+2): the rows' table indices (wave-uniform scalar loads) and the angle item's index / heading;
//   B(i+1): the 8 rows' 16-byte loads per thread + the angle item's load;
//   C(i)  : Philox scale, pack, store.
// Same outputs, same Philox indexing as gather_step_rows (tests/test_hip_staging.py compares them bit for bit).
// Requires IMG == 2048 (one 8-element chunk per thread per row), ANG == 128 (one float4 per lane of a half-wave per row) and
// exactly one output precision (GatherRolloutArgs::pipe: 1 = bf16 outputs, 2 = fp32 outputs); else the plain loop runs.
#pragma once
#include "../../include/vln_hip.h"
#include "gather_body.h"

namespace vln {

constexpr int kRideRows = 8;

template <typename TT> struct RideRaw;
template <> struct RideRaw<bf16_raw> {
  uint4 v;
  __device__ __forceinline__ void load(const bf16_raw* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void unpack(float (&o)[8]) const {
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
  }
};
template <> struct RideRaw<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
  __device__ __forceinline__ void unpack(float (&o)[8]) const {
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  }
};

struct RideIdx {            // stage A of one group: wave-uniform except the angle item
  int t, r0;
  long src[kRideRows];      // table row (x V + view) of each output row, -1 = empty candidate slot
  // the thread's angle item: output row r0 + (tid >> 5), columns IMG + (tid & 31) * 4 ..
  long ang_src;             // panorama row: element offset into the angle table
  bool ang_empty;           // empty candidate slot, or an out-of-range index: zeros
  int nbad;                 // out-of-range indices among the group's rows (GatherCheck)
  float theta;              // candidate row: the heading (lanes 0-15 of the half-wave) or the elevation (16-31)
};

template <typename TT, bool LP>
__device__ __forceinline__ void gather_ride_pipelined(const GatherRolloutArgs& ride, int first, int np, int tid) {
  constexpr int IMG = 2048, ANG = 128, F = IMG + ANG;
  const int nrows = ride.nrows, groups = (nrows + kRideRows - 1) / kRideRows, total = ride.T * groups;
  if (first >= total) return;
  const int ka = tid >> 5, ja = tid & 31;

  auto row_of = [&](int r0, int k) { const int r = r0 + k; return r < nrows ? r : nrows - 1; };     // rows past the end repeat the last one
  auto stage_a = [&](int i, RideIdx& g) {
    if (i >= total) i = total - 1;                     // past the end: a valid group again (its loads are dropped)
    g.t = i / groups; g.r0 = (i % groups) * kRideRows;
    const GatherStepArgs& a = ride.step[g.t];
    const int npano = a.B * a.V;
    const long nr = a.chk.n_rows;
    g.nbad = 0;
#pragma unroll
    for (int k = 0; k < kRideRows; ++k) {
      const int r = row_of(g.r0, k);
      const int rp = r < npano ? r : npano - 1, rc = r < npano ? 0 : r - npano;
      const long prow = a.rows[rp / a.V];
      const int pvi = a.view_index[rp / a.V];
      const bool pbad = nr && (prow < 0 || prow >= nr || pvi < 0 || pvi >= a.chk.n_aviews);
      const long crow = a.crows[rc];
      const int cv = a.cviews[rc];
      const bool cbad = nr && crow >= 0 && (crow >= nr || cv < 0 || cv >= a.V);
      const long cidx = (crow < 0 || cbad) ? -1 : crow * a.V + cv;
      g.src[k] = r < npano ? (pbad ? -1 : prow * a.V + rp % a.V) : cidx;
      g.nbad += (g.r0 + k < nrows && (r < npano ? pbad : cbad)) ? 1 : 0;
    }
    const int r = row_of(g.r0, ka);
    const int rp = r < npano ? r : npano - 1, rc = r < npano ? 0 : r - npano;
    const long prow = a.rows[rp / a.V];
    const int pvi = a.view_index[rp / a.V];
    const bool pbad = nr && (prow < 0 || prow >= nr || pvi < 0 || pvi >= a.chk.n_aviews);
    g.ang_src = pbad ? 0 : ((long)pvi * a.V + rp % a.V) * ANG + ja * 4;
    const float h = a.heading[rc], e = a.elevation[rc];
    g.theta = ja < 16 ? h : e;
    const long crow = a.crows[rc];
    const int cv = a.cviews[rc];
    g.ang_empty = r < npano ? pbad : (crow < 0 || (nr && (crow >= nr || cv < 0 || cv >= a.V)));
  };
  auto stage_b = [&](const RideIdx& g, RideRaw<TT> (&raw)[kRideRows], float4& ang) {
    const GatherStepArgs& a = ride.step[g.t];
    const TT* table = reinterpret_cast<const TT*>(a.table);
#pragma unroll
    for (int k = 0; k < kRideRows; ++k) raw[k].load(table + (g.src[k] < 0 ? 0 : g.src[k]) * IMG + tid * 8);
    ang = *reinterpret_cast<const float4*>(a.angle_table + g.ang_src);
  };
  auto stage_c = [&](const RideIdx& g, const RideRaw<TT> (&raw)[kRideRows], const float4& ang) {
    const GatherStepArgs& a = ride.step[g.t];
    const int npano = a.B * a.V;
    const uint64_t off_p = a.dr_pano.off(), off_c = a.dr_cand.off();
#pragma unroll
    for (int k = 0; k < kRideRows; ++k) {
      const int r = row_of(g.r0, k);
      const bool pano = r < npano, empty = g.src[k] < 0;
      const int rr = pano ? r : r - npano;
      float x[8];
      raw[k].unpack(x);
      const DropSpec& dr = pano ? a.dr_pano : a.dr_cand;
      if (dr.p > 0.f) {                                          // uniform, no memory operation inside
        float m[8];
        dropout_scale8(dr.seed, pano ? off_p : off_c, (uint32_t)(((long)rr * IMG + tid * 8) >> 3), dr.p, m);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] *= m[j];
      }
      if (empty) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = 0.f;
      }
      if constexpr (LP) {
        bf16_raw* d = (pano ? a.out_lp : a.cout_lp) + (long)rr * F + tid * 8;
        uint4 v;
        v.x = (uint32_t)f32_to_bf16_bits(x[0]) | ((uint32_t)f32_to_bf16_bits(x[1]) << 16);
        v.y = (uint32_t)f32_to_bf16_bits(x[2]) | ((uint32_t)f32_to_bf16_bits(x[3]) << 16);
        v.z = (uint32_t)f32_to_bf16_bits(x[4]) | ((uint32_t)f32_to_bf16_bits(x[5]) << 16);
        v.w = (uint32_t)f32_to_bf16_bits(x[6]) | ((uint32_t)f32_to_bf16_bits(x[7]) << 16);
        *reinterpret_cast<uint4*>(d) = v;
      } else {
        float* d = (pano ? a.out : a.cout) + (long)rr * F + tid * 8;
        *reinterpret_cast<float4*>(d) = make_float4(x[0], x[1], x[2], x[3]);
        *reinterpret_cast<float4*>(d + 4) = make_float4(x[4], x[5], x[6], x[7]);
      }
    }
    // the angle columns of row r0 + ka: a copy of the view's angle feature, or sin / cos of the candidate's heading / elevation
    {
      const int r = row_of(g.r0, ka);
      const bool pano = r < npano;
      const int rr = pano ? r : r - npano;
      const bool empty = g.ang_empty;
      const float s = sinf(g.theta), c = cosf(g.theta);
      const float cv = empty ? 0.f : (((ja >> 3) & 1) ? c : s);       // ANG / 4 = 32 columns each of sin h, cos h, sin e, cos e
      const float4 x = pano ? (empty ? make_float4(0.f, 0.f, 0.f, 0.f) : ang) : make_float4(cv, cv, cv, cv);
      if constexpr (LP) {
        bf16_raw* d = (pano ? a.out_lp : a.cout_lp) + (long)rr * F + IMG + ja * 4;
        uint2 v;
        v.x = (uint32_t)f32_to_bf16_bits(x.x) | ((uint32_t)f32_to_bf16_bits(x.y) << 16);
        v.y = (uint32_t)f32_to_bf16_bits(x.z) | ((uint32_t)f32_to_bf16_bits(x.w) << 16);
        *reinterpret_cast<uint2*>(d) = v;
      } else {
        float* d = (pano ? a.out : a.cout) + (long)rr * F + IMG + ja * 4;
        *reinterpret_cast<float4*>(d) = x;
      }
    }
  };

  RideIdx g0, g1, g2;
  RideRaw<TT> raw0[kRideRows], raw1[kRideRows];
  float4 ang0, ang1;
  stage_a(first, g0);
  stage_b(g0, raw0, ang0);
  stage_a(first + np, g1);
  int nbad = 0;
  for (int i = first; i < total; i += np) {
    stage_b(g1, raw1, ang1);
    stage_a(i + 2 * np, g2);
    stage_c(g0, raw0, ang0);
    nbad += g0.nbad;
    g0 = g1; g1 = g2;
#pragma unroll
    for (int k = 0; k < kRideRows; ++k) raw0[k] = raw1[k];
    ang0 = ang1;
  }
  if (nbad && tid == 0) __hip_atomic_fetch_add(ride.step[0].chk.bad, (unsigned)nbad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the passenger workgroup `first` of `np`
__device__ __forceinline__ void gather_ride_passenger(const GatherRolloutArgs& ride, int first, int np, int tid) {
  if (ride.pipe == 1 && ride.ttype == VLN_BF16) return gather_ride_pipelined<bf16_raw, true>(ride, first, np, tid);
  if (ride.pipe == 2 && ride.ttype != VLN_BF16) return gather_ride_pipelined<float, false>(ride, first, np, tid);
  const int groups = (ride.nrows + kRideRows - 1) / kRideRows, total = ride.T * groups;
  for (int i = first; i < total; i += np) {
    const int t = i / groups, r = (i % groups) * kRideRows;
    if (ride.ttype == VLN_BF16) gather_step_rows<bf16_raw, kRideRows>(ride.step[t], r, ride.nrows, tid);
    else gather_step_rows<float, kRideRows>(ride.step[t], r, ride.nrows, tid);
  }
}
}  // namespace vln
