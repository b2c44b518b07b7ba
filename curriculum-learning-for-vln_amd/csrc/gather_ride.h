// The rollout-wide feature gather as PASSENGER workgroups of the persistent recurrence launch (encoder_persist_g.h).
//
// A passenger compute unit holds ONE 256-thread workgroup (the launch claims a CU's LDS per workgroup so that passengers never
// share a CU with a recurrence workgroup), i.e. one wave per SIMD: nothing hides a wave's memory latency or its Philox
// arithmetic but the wave itself.  Measured with the plain per-row loop: 287 us for the 7 x 2944 rows of the headline
// rollout, longer than the 179 us recurrence it rides in (profiles/round3_notes.md); round 3's pipeline 175 us alone; this loop
// 95 us (round 5, profiles/round5_notes.md section 11).  It is a software pipeline over groups of 8 output rows, every memory
// operation unconditional (clamped addresses, idempotent duplicate stores for the rows past the end of a step) so that the
// wait counts the compiler places stay exact -- there is no vmcnt(0) in the loop:
//   A1(i+2): lanes 0-7 load the 8 rows' table indices, one vector load per index array, + the dropout offset words;
//   A2(i+1): the same arithmetic for the 8 rows at once, v_readlane into the wave-uniform RideIdx (+ the angle item's);
//   B(i+1) : the 8 rows' 16-byte non-temporal loads per thread + the angle item's load;
//   C(i)   : Philox scale, pack, non-temporal store.
// Same outputs, same Philox indexing as gather_step_rows (tests/test_hip_staging.py compares them bit for bit).
// Requires IMG == 2048 (one 8-element chunk per thread per row), ANG == 128 (one float4 per lane of a half-wave per row) and
// exactly one output precision (GatherRolloutArgs::pipe: 1 = bf16 outputs, 2 = fp32 outputs); else the plain loop runs.
#pragma once
#include "../../include/vln_hip.h"
#include "gather_body.h"

namespace vln {

constexpr int kRideRows = 8;
constexpr int kRideDepth = 1;      // groups whose feature loads are in flight ahead of the one being stored (2, 3: no faster)

// The passengers' feature rows are touched once: read and written NON-TEMPORALLY, they do not push the recurrence's hand-off
// lines out of the L2s the two kinds of workgroup share.  (Round 5, B = 64 bf16: the encoder forward is 261 us without
// passengers, 278 us with plain accesses, 262 us with these; scripts/ride_probe.py.)
__device__ __forceinline__ void ride_store16(void* d, uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {x, y, z, w};
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(d));
}

template <typename TT> struct RideRaw;
template <> struct RideRaw<bf16_raw> {
  uint4 v;
  __device__ __forceinline__ void load(const bf16_raw* p) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);      // (one global_load_dwordx4 ... nt)
    v.x = __builtin_nontemporal_load(q); v.y = __builtin_nontemporal_load(q + 1); v.z = __builtin_nontemporal_load(q + 2); v.w = __builtin_nontemporal_load(q + 3);
  }
  __device__ __forceinline__ void unpack(float (&o)[8]) const {
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
  }
};
template <> struct RideRaw<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a.x = __builtin_nontemporal_load(p); a.y = __builtin_nontemporal_load(p + 1); a.z = __builtin_nontemporal_load(p + 2); a.w = __builtin_nontemporal_load(p + 3);
    b.x = __builtin_nontemporal_load(p + 4); b.y = __builtin_nontemporal_load(p + 5); b.z = __builtin_nontemporal_load(p + 6); b.w = __builtin_nontemporal_load(p + 7);
  }
  __device__ __forceinline__ void unpack(float (&o)[8]) const {
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  }
};

struct RideLoad {           // stage A1 of one group: lane k (< 8) of every wave holds the raw indices of output row r0 + k
  int t, r0;
  long prow, crow;
  int pvi, cv;
  float hd, el;
  unsigned long long base_p, base_c;   // the device words of the two dropout sites' Philox offsets (DropSpec::step), every lane the same
};
struct RideIdx {            // stage A2 of one group: wave-uniform except the angle item
  int t, r0;
  long src[kRideRows];      // table row (x V + view) of each output row, -1 = empty candidate slot
  // the thread's angle item: output row r0 + (tid >> 5), columns IMG + (tid & 31) * 4 ..
  long ang_src;             // panorama row: element offset into the angle table
  bool ang_empty;           // empty candidate slot, or an out-of-range index: zeros
  int nbad;                 // out-of-range indices among the group's rows (GatherCheck)
  float theta;              // candidate row: the heading (lanes 0-15 of the half-wave) or the elevation (16-31)
  uint64_t off_p, off_c;    // Philox offsets of the step's panorama / candidate dropout
};

template <typename TT, bool LP, int D>
__device__ __forceinline__ void gather_ride_pipelined(const GatherRolloutArgs& ride, int first, int np, int tid) {
  constexpr int IMG = 2048, ANG = 128, F = IMG + ANG;
  const int nrows = ride.nrows, groups = (nrows + kRideRows - 1) / kRideRows, total = ride.T * groups;
  if (first >= total) return;
  const int ka = tid >> 5, ja = tid & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);                 // the wave: its angle items are rows 2 wv, 2 wv + 1
  const bool odd = ka & 1;
  const int kl = (tid & 63) < kRideRows ? (tid & 63) : kRideRows - 1;      // the row of the group this LANE indexes

  auto row_of = [&](int r0, int k) __attribute__((always_inline)) { const int r = r0 + k; return r < nrows ? r : nrows - 1; };     // rows past the end repeat the last one
  auto lane64 = [&](long v, int lane) __attribute__((always_inline)) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long)v, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long)v >> 32), lane);
    return (long)(((unsigned long)hi << 32) | lo);
  };
  // Stage A.  The 8 rows' indices are read by lanes 0-7 of each wave, ONE vector load per index array (A1), a whole iteration
  // before they are used (A2: the same arithmetic for the 8 rows at once, then v_readlane into the wave-uniform RideIdx).  The
  // loads are older than every feature load in flight, so the in-order vector counter lets A2 take them without waiting for
  // stage B.  (Round 5: as 8 x 6 loads of a uniform address inside the row loop, each was waited for with vmcnt(0) --
  // together with the feature loads just issued; 8 rows x 3 dependent waits per group were the passenger's critical path.)
  auto stage_a1 = [&](int i, RideLoad& L) __attribute__((always_inline)) {
    if (i >= total) i = total - 1;                     // past the end: a valid group again (its loads are dropped)
    L.t = i / groups; L.r0 = (i % groups) * kRideRows;
    const GatherStepArgs& a = ride.step[L.t];
    const int npano = a.B * a.V;
    const int r = row_of(L.r0, kl);
    const int rp = r < npano ? r : npano - 1, rc = r < npano ? 0 : r - npano;
    L.prow = a.rows[rp / a.V];
    L.pvi = a.view_index[rp / a.V];
    L.crow = a.crows[rc];
    L.cv = a.cviews[rc];
    L.hd = a.heading[rc];
    L.el = a.elevation[rc];
    // DropSpec::off()'s device word, fetched here and not in stage C, where it was a load to wait for with vmcnt(0) at the
    // head of every group (no site word: any readable word, dropped in A2)
    const unsigned long long* any = reinterpret_cast<const unsigned long long*>(a.rows);
    L.base_p = *(a.dr_pano.step ? a.dr_pano.step : any);
    L.base_c = *(a.dr_cand.step ? a.dr_cand.step : any);
  };
  auto stage_a2 = [&](const RideLoad& L, RideIdx& g) __attribute__((always_inline)) {
    g.t = L.t; g.r0 = L.r0;
    const GatherStepArgs& a = ride.step[g.t];
    const int npano = a.B * a.V;
    const long nr = a.chk.n_rows;
    const bool chk = nr != 0;
    const int r = row_of(g.r0, kl);
    const bool pano = r < npano;
    const int rp = pano ? r : npano - 1;
    const bool pbad = chk & ((L.prow < 0) | (L.prow >= nr) | (L.pvi < 0) | (L.pvi >= a.chk.n_aviews));
    const bool cbad = chk & (L.crow >= 0) & ((L.crow >= nr) | (L.cv < 0) | (L.cv >= a.V));
    const bool cempty = (L.crow < 0) | cbad;
    const long src = pano ? (pbad ? -1 : L.prow * a.V + rp % a.V) : (cempty ? -1 : L.crow * a.V + L.cv);
    const bool bad = ((tid & 63) < kRideRows) & (g.r0 + kl < nrows) & (pano ? pbad : cbad);
    g.nbad = __popcll(__ballot(bad));
#pragma unroll
    for (int k = 0; k < kRideRows; ++k) g.src[k] = lane64(src, k);
    // the angle item of row r0 + ka = lane ka's
    const long asrc = pbad ? 0 : ((long)L.pvi * a.V + rp % a.V) * ANG;
    const int aempty = (pano ? pbad : cempty) ? 1 : 0;
    const long asrc0 = lane64(asrc, 2 * wv), asrc1 = lane64(asrc, 2 * wv + 1);
    const int em0 = __builtin_amdgcn_readlane(aempty, 2 * wv), em1 = __builtin_amdgcn_readlane(aempty, 2 * wv + 1);
    const float h0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.hd), 2 * wv)), h1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.hd), 2 * wv + 1));
    const float e0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.el), 2 * wv)), e1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.el), 2 * wv + 1));
    g.ang_src = (odd ? asrc1 : asrc0) + ja * 4;
    g.theta = ja < 16 ? (odd ? h1 : h0) : (odd ? e1 : e0);
    g.ang_empty = (odd ? em1 : em0) != 0;
    g.off_p = a.dr_pano.step ? (uint64_t)lane64((long)L.base_p, 0) * 8ull + a.dr_pano.offset : a.dr_pano.offset;
    g.off_c = a.dr_cand.step ? (uint64_t)lane64((long)L.base_c, 0) * 8ull + a.dr_cand.offset : a.dr_cand.offset;
  };
  auto stage_b = [&](const RideIdx& g, RideRaw<TT> (&raw)[kRideRows], float4& ang) __attribute__((always_inline)) {
    const GatherStepArgs& a = ride.step[g.t];
    const TT* table = reinterpret_cast<const TT*>(a.table);
#pragma unroll
    for (int k = 0; k < kRideRows; ++k) raw[k].load(table + (g.src[k] < 0 ? 0 : g.src[k]) * IMG + tid * 8);
    ang = *reinterpret_cast<const float4*>(a.angle_table + g.ang_src);
  };
  auto stage_c = [&](const RideIdx& g, const RideRaw<TT> (&raw)[kRideRows], const float4& ang) __attribute__((always_inline)) {
    const GatherStepArgs& a = ride.step[g.t];
    const int npano = a.B * a.V;
    // the step's constants once per group, not per row (as `pano ? a.dr_pano : a.dr_cand` they were three argument loads,
    // each with its wait, in every row)
    const float p_p = a.dr_pano.p, p_c = a.dr_cand.p;
    const uint64_t seed_p = a.dr_pano.seed, seed_c = a.dr_cand.seed;
    bf16_raw* const lp_p = a.out_lp; bf16_raw* const lp_c = a.cout_lp;
    float* const f_p = a.out; float* const f_c = a.cout;
#pragma unroll
    for (int k = 0; k < kRideRows; ++k) {
      const int r = row_of(g.r0, k);
      const bool pano = r < npano, empty = g.src[k] < 0;
      const int rr = pano ? r : r - npano;
      float x[8];
      raw[k].unpack(x);
      const float p = pano ? p_p : p_c;
      if (p > 0.f) {                                             // uniform, no memory operation inside
        float m[8];
        dropout_scale8(pano ? seed_p : seed_c, pano ? g.off_p : g.off_c, (uint32_t)(((long)rr * IMG + tid * 8) >> 3), p, m);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] *= m[j];
      }
      if (empty) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = 0.f;
      }
      if constexpr (LP) {
        bf16_raw* d = (pano ? lp_p : lp_c) + (long)rr * F + tid * 8;
        uint4 v;
        v.x = (uint32_t)f32_to_bf16_bits(x[0]) | ((uint32_t)f32_to_bf16_bits(x[1]) << 16);
        v.y = (uint32_t)f32_to_bf16_bits(x[2]) | ((uint32_t)f32_to_bf16_bits(x[3]) << 16);
        v.z = (uint32_t)f32_to_bf16_bits(x[4]) | ((uint32_t)f32_to_bf16_bits(x[5]) << 16);
        v.w = (uint32_t)f32_to_bf16_bits(x[6]) | ((uint32_t)f32_to_bf16_bits(x[7]) << 16);
        ride_store16(d, v.x, v.y, v.z, v.w);
      } else {
        float* d = (pano ? f_p : f_c) + (long)rr * F + tid * 8;
        ride_store16(d, __float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3]));
        ride_store16(d + 4, __float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7]));
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
        bf16_raw* d = (pano ? lp_p : lp_c) + (long)rr * F + IMG + ja * 4;
        uint2 v;
        v.x = (uint32_t)f32_to_bf16_bits(x.x) | ((uint32_t)f32_to_bf16_bits(x.y) << 16);
        v.y = (uint32_t)f32_to_bf16_bits(x.z) | ((uint32_t)f32_to_bf16_bits(x.w) << 16);
        *reinterpret_cast<uint2*>(d) = v;
      } else {
        float* d = (pano ? f_p : f_c) + (long)rr * F + IMG + ja * 4;
        *reinterpret_cast<float4*>(d) = x;
      }
    }
  };

  // D groups' feature loads are in flight ahead of the group being stored: g[0 .. D-1] are loaded, g[D] is indexed, L holds
  // the index loads of the group after it
  RideLoad L;
  RideIdx g[D + 2];
  RideRaw<TT> raw[D + 1][kRideRows];
  float4 ang[D + 1];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    stage_a1(first + d * np, L);
    stage_a2(L, g[d]);
    stage_b(g[d], raw[d], ang[d]);
  }
  stage_a1(first + D * np, L);
  stage_a2(L, g[D]);
  stage_a1(first + (D + 1) * np, L);
  int nbad = 0;
  for (int i = first; i < total; i += np) {
    stage_b(g[D], raw[D], ang[D]);
    stage_a2(L, g[D + 1]);
    stage_a1(i + (D + 2) * np, L);
    stage_c(g[0], raw[0], ang[0]);
    nbad += g[0].nbad;
#pragma unroll
    for (int d = 0; d <= D; ++d) g[d] = g[d + 1];
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
      for (int k = 0; k < kRideRows; ++k) raw[d][k] = raw[d + 1][k];
      ang[d] = ang[d + 1];
    }
  }
  if (nbad && tid == 0) __hip_atomic_fetch_add(ride.step[0].chk.bad, (unsigned)nbad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the passenger workgroup `first` of `np`
__device__ __forceinline__ void gather_ride_passenger(const GatherRolloutArgs& ride, int first, int np, int tid) {
  if (ride.pipe == 1 && ride.ttype == VLN_BF16) {
    return gather_ride_pipelined<bf16_raw, true, kRideDepth>(ride, first, np, tid);
  }
  if (ride.pipe == 2 && ride.ttype != VLN_BF16) return gather_ride_pipelined<float, false, kRideDepth>(ride, first, np, tid);
  const int groups = (ride.nrows + kRideRows - 1) / kRideRows, total = ride.T * groups;
  for (int i = first; i < total; i += np) {
    const int t = i / groups, r = (i % groups) * kRideRows;
    if (ride.ttype == VLN_BF16) gather_step_rows<bf16_raw, kRideRows>(ride.step[t], r, ride.nrows, tid);
    else gather_step_rows<float, kRideRows>(ride.step[t], r, ride.nrows, tid);
  }
}
}  // namespace vln
