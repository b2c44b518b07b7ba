// The first launch of an EnvDrop decoder step, shared by envdrop.hip (the step's own prep launch) and features.hip (the same work
// as extra workgroups of the feature-gather launch when the step reads its features from the resident table).
#pragma once
#include "common.h"

namespace vln {

struct PrepArgs {
  const float* a; const float* act_w; const float* act_b; const float* htp;
  float* e; float* xcat; long ldx; float* hq;
  int B, ANG, AE, F, H;
  DropSpec d_act, d_h;
  float* a_stash;   // nullable: copy of `a` kept for the deferred act_embed weight gradient
  // Chained steps (round 4, vln_envdrop_step.chain): the PREVIOUS step left the epilogue of its linear_out product pending --
  // h_tilde = tanh(sum of `pend_n` split-K slabs), drop(h_tilde) (its site 3).  This launch finishes it on the way: the value is
  // written to pend_ht (= `htp`'s memory, the tensor the previous step returned) and pend_htd (its stash row) and used here.
  const float* pend_slabs; int pend_n; long pend_stride; float* pend_ht; float* pend_htd; DropSpec pend_drop;
};
// e = tanh(a W_a^T + b); xcat[:, :AE] = drop(e); xcat[:, AE+F:] = h_tilde_prev; hq = drop(h_tilde_prev)
// Work items: B*AE dot products of length ANG, 8 lanes each (a lane group reads 128 contiguous bytes of the weight row
// per step: whole cache lines, where one thread per output walked 64 rows x 16 B per wave instruction), then B*H/4
// float4 copies of h_tilde_prev and B*ANG/4 of a_prev.
__device__ __forceinline__ void envdrop_prep_body(const PrepArgs& p, long first, long stride) {
  const long ne = (long)p.B * p.AE, nh4 = (long)p.B * p.H / 4;
  const long na4 = p.a_stash ? (long)p.B * p.ANG / 4 : 0;
  const long nitems = ne * 8 + nh4 + na4;
  for (long i = first; i < nitems; i += stride) {
    if (i < ne * 8) {                                  // ne*8 is a multiple of 64: a wave never straddles this branch
      const long o = i >> 3;
      const int sub = (int)(i & 7);
      const int b = (int)(o / p.AE), j = (int)(o % p.AE);
      const float* a = p.a + (long)b * p.ANG;
      const float* w = p.act_w + (long)j * p.ANG;
      float acc = 0.f;
      for (int k = sub * 4; k < p.ANG; k += 32) {
        const float4 x = *reinterpret_cast<const float4*>(a + k);
        const float4 y = *reinterpret_cast<const float4*>(w + k);
        acc += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
      }
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      acc += __shfl_xor(acc, 4, 64);
      if (sub == 0) {
        const float e = tanhf(acc + p.act_b[j]);
        p.e[o] = e;
        p.xcat[(long)b * p.ldx + j] = e * dropout_scale1(p.d_act.seed, p.d_act.off(), (uint32_t)o, p.d_act.p);
      }
    } else if (i < ne * 8 + nh4) {
      const long k4 = i - ne * 8;
      const long k = k4 * 4;
      const int b = (int)(k / p.H), j = (int)(k % p.H);
      float4 v;
      if (p.pend_slabs) {        // the previous step's reduce + tanh + dropout epilogue (step_bodies.h::reduce_epilogue_body: the
        // same per-element sums -- four partial sums over the slabs s mod 4, then (a1 + a2) + a3 -- on 16-byte loads)
        const float* sp = p.pend_slabs + k;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        auto ld = [&](int s_) { return *reinterpret_cast<const float4*>(sp + (long)s_ * p.pend_stride); };
        auto add = [](float4& x, const float4& y) { x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w; };
        int s = 0;
        for (; s + 3 < p.pend_n; s += 4) {
          const float4 t0 = ld(s), t1 = ld(s + 1), t2 = ld(s + 2), t3 = ld(s + 3);
          add(a0, t0); add(a1, t1); add(a2, t2); add(a3, t3);
        }
        for (; s < p.pend_n; ++s) add(a0, ld(s));
        float o[4] = {a0.x + ((a1.x + a2.x) + a3.x), a0.y + ((a1.y + a2.y) + a3.y), a0.z + ((a1.z + a2.z) + a3.z), a0.w + ((a1.w + a2.w) + a3.w)};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          o[c] = tanhf(o[c]);
          p.pend_htd[k + c] = o[c] * dropout_scale1(p.pend_drop.seed, p.pend_drop.off(), (uint32_t)(k + c), p.pend_drop.p);
        }
        v = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(p.pend_ht + k) = v;
      } else {
        v = *reinterpret_cast<const float4*>(p.htp + k);
      }
      *reinterpret_cast<float4*>(p.xcat + (long)b * p.ldx + p.AE + p.F + j) = v;
      float m[4] = {1.f, 1.f, 1.f, 1.f};
      if (p.d_h.p > 0.f) dropout_scale4(p.d_h.seed, p.d_h.off(), (uint32_t)k4, p.d_h.p, m);
      *reinterpret_cast<float4*>(p.hq + k) = make_float4(v.x * m[0], v.y * m[1], v.z * m[2], v.w * m[3]);
    } else {
      const long k = (i - ne * 8 - nh4) * 4;
      *reinterpret_cast<float4*>(p.a_stash + k) = *reinterpret_cast<const float4*>(p.a + k);
    }
  }
}

static inline long envdrop_prep_items(const PrepArgs& p) {
  return (long)p.B * p.AE * 8 + (long)p.B * (p.H + (p.a_stash ? p.ANG : 0)) / 4;
}

// one launch per decoder step: panorama rows + candidate rows from the resident table (features.hip)
// Range check of the gather's indices: a viewpoint row outside [0, n_rows) (candidates: >= n_rows; < 0 is the empty slot), a
// panorama view index outside [0, n_aviews) or a candidate view outside [0, V) reads NOTHING -- the output row is all zeros --
// and is counted in a host-mapped sticky word that the next vln_persistent_check() reports (features.hip).
struct GatherCheck { long n_rows; int n_aviews; unsigned* bad; };
struct GatherStepArgs {
  const void* table; const float* angle_table;
  const long long* rows; const int* view_index;                                     // panorama: [B], [B]
  const long long* crows; const int* cviews; const float* heading; const float* elevation;   // candidates: [B*C]
  float* out; bf16_raw* out_lp; float* cout; bf16_raw* cout_lp;
  int B, V, C, IMG, ANG;
  DropSpec dr_pano, dr_cand;
  GatherCheck chk;          // the table's registered extent (vln_feature_table_extent); n_rows == 0: indices are not checked
};
// every step of a teacher-forced rollout (features.hip: one launch; encoder_persist_g.h: as PASSENGER workgroups of the
// instruction encoder's persistent recurrence launch, on the compute units that launch leaves idle)
constexpr int kGatherMaxSteps = 12;
struct GatherRolloutArgs { GatherStepArgs step[kGatherMaxSteps]; int T, nrows, ttype, pipe; };   // pipe: gather_ride.h
// gather of a step + the step's prep work as extra workgroups of the same launch (features.hip)
int gather_step_prep(hipStream_t st, const GatherStepArgs& a, int ttype, const PrepArgs& p);

}  // namespace vln
