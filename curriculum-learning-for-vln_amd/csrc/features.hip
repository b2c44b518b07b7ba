// Per-step feature marshalling ON the device (reference: agent/base.py:141-157, environ/common_env.py:307-308,
// utils/misc.py:285-317).  The reference gathers 36x2176 floats per episode on the host every step and ships
// 20 MB (+ candidates) over PCIe; here the whole ResNet feature table (2.9 GB fp32 / 1.5 GB bf16 -- 1 % of the
// MI355X's 288 GB) stays resident in HBM and a step ships only indices (B viewpoint rows + view indices + the
// candidates' (row, view, heading, elevation)).  One pass gathers the rows, appends the angle features, applies
// the environmental feature dropout (policy.py:226-231; same Philox indexing as feat_dropout_inplace) and
// optionally emits the bf16 copy the attention kernels stream.  16-byte accesses, one workgroup per output row.
#include <mutex>
#include <unordered_map>
#include "vln_internal.h"
#include "envdrop_prep.h"
#include "gather_body.h"
#include "../../include/vln_hip.h"

namespace vln {

// ---- registered table extents: every gather entry point range-checks its indices against them -----------------------
namespace {
struct Extent { long rows; int aviews; };
std::mutex g_ext_mu;
std::unordered_map<const void*, Extent> g_ext;
}  // namespace
GatherCheck gather_check(const void* table) {
  std::lock_guard<std::mutex> lock(g_ext_mu);
  auto it = g_ext.find(table);
  if (it == g_ext.end()) return GatherCheck{0, 0, nullptr};
  unsigned* w = sticky_dev_word();
  if (!w) return GatherCheck{0, 0, nullptr};
  return GatherCheck{it->second.rows, it->second.aviews, w + 1};       // sticky word 1 of the device: bad gather indices
}

template <typename TT>
__global__ __launch_bounds__(256) void gather_pano_kernel(const TT* table, const long long* rows, const int* view_index,
                                                          const float* angle_table, float* out, bf16_raw* out_lp, int V,
                                                          int IMG, int ANG, DropSpec dr, GatherCheck chk) {
  const int r = blockIdx.x;             // output row = b*V + v
  const int b = r / V, v = r % V;
  const int F = IMG + ANG;
  const bool bad = chk.n_rows && (rows[b] < 0 || rows[b] >= chk.n_rows || view_index[b] < 0 || view_index[b] >= chk.n_aviews);
  if (bad && threadIdx.x == 0) __hip_atomic_fetch_add(chk.bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const TT* src = table + (bad ? 0 : ((long)rows[b] * V + v) * IMG);
  const float* ang = angle_table + (bad ? 0 : ((long)view_index[b] * V + v) * ANG);
  float* dst = out ? out + (long)r * F : nullptr;
  bf16_raw* dlp = out_lp ? out_lp + (long)r * F : nullptr;
  for (int c = threadIdx.x * 4; c < F; c += 256 * 4) {
    float x[4];
    if (c < IMG) {
      Elt<TT>::ld4(src + c, x);
      if (dr.p > 0.f) {
        float m[4];
        dropout_scale4(dr.seed, dr.off(), (uint32_t)(((long)r * IMG + c) >> 2), dr.p, m);
        x[0] *= m[0]; x[1] *= m[1]; x[2] *= m[2]; x[3] *= m[3];
      }
    } else {
      Elt<float>::ld4(ang + (c - IMG), x);
    }
    if (bad) { x[0] = 0.f; x[1] = 0.f; x[2] = 0.f; x[3] = 0.f; }      // an out-of-range index reads nothing (GatherCheck)
    if (dst) Elt<float>::st4(dst + c, x);
    if (dlp) Elt<bf16_raw>::st4(dlp + c, x);
  }
}

template <typename TT>
__global__ __launch_bounds__(256) void gather_cands_kernel(const TT* table, const long long* rows, const int* views,
                                                           const float* heading, const float* elevation, float* out,
                                                           bf16_raw* out_lp, int V, int IMG, int ANG, DropSpec dr, GatherCheck chk) {
  const int r = blockIdx.x;             // output row = b*C + c
  const int F = IMG + ANG;
  const long row = rows[r];
  float* dst = out ? out + (long)r * F : nullptr;
  bf16_raw* dlp = out_lp ? out_lp + (long)r * F : nullptr;
  const bool bad = chk.n_rows && row >= 0 && (row >= chk.n_rows || views[r] < 0 || views[r] >= V);
  if (bad && threadIdx.x == 0) __hip_atomic_fetch_add(chk.bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const bool empty = row < 0 || bad;    // STOP slot / padding: all-zero feature (base.py:152-153); out-of-range: the same, counted
  float sh = 0.f, ch = 0.f, se = 0.f, ce = 0.f;
  if (!empty) { sh = sinf(heading[r]); ch = cosf(heading[r]); se = sinf(elevation[r]); ce = cosf(elevation[r]); }
  const TT* src = empty ? table : table + (row * V + views[r]) * IMG;
  const int q = ANG >> 2;               // [sin h]*q [cos h]*q [sin e]*q [cos e]*q   (misc.py:285-293)
  for (int c = threadIdx.x * 4; c < F; c += 256 * 4) {
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    if (!empty) {
      if (c < IMG) {
        Elt<TT>::ld4(src + c, x);
        if (dr.p > 0.f) {
          float m[4];
          dropout_scale4(dr.seed, dr.off(), (uint32_t)(((long)r * IMG + c) >> 2), dr.p, m);
          x[0] *= m[0]; x[1] *= m[1]; x[2] *= m[2]; x[3] *= m[3];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int g = (c - IMG + j) / q;
          x[j] = g == 0 ? sh : (g == 1 ? ch : (g == 2 ? se : ce));
        }
      }
    }
    if (dst) Elt<float>::st4(dst + c, x);
    if (dlp) Elt<bf16_raw>::st4(dlp + c, x);
  }
}

template <typename TT>
__global__ __launch_bounds__(256) void gather_step_kernel(GatherStepArgs a) { gather_step_row<TT>(a, (int)blockIdx.x, (int)threadIdx.x); }

// The same gather + the decoder step's prep work (act embedding, h_tilde_prev copy / dropout: envdrop_prep.h) as the LAST
// `nprep` workgroups of the launch: both only depend on what the previous step left behind, and as two launches the second
// one was ~6.5 us of pure dependent-launch latency per decoder step.
template <typename TT, int kGatherRows>
__global__ __launch_bounds__(256) void gather_step_prep_kernel(GatherStepArgs a, PrepArgs p, int nrows, int nprep) {
  const int nrb = (nrows + kGatherRows - 1) / kGatherRows;
  if ((int)blockIdx.x < nrb) gather_step_rows<TT, kGatherRows>(a, (int)blockIdx.x * kGatherRows, nrows, (int)threadIdx.x);
  else envdrop_prep_body(p, (long)((int)blockIdx.x - nrb) * 256 + threadIdx.x, (long)nprep * 256);
}

// ---- every step of a teacher-forced rollout in one launch ---------------------------------------------------------
// With teacher forcing the path -- hence every step's viewpoint and candidate rows -- is known when the rollout starts
// (the reference steps its simulator along the ground-truth actions, base.py:141-157 + follower.py:140): T gather launches of
// ~15 us on the steps' dependent chains become one launch of T * (B*V + B*C) row blocks ahead of the first step.
template <typename TT>
__global__ __launch_bounds__(256) void gather_rollout_kernel(GatherRolloutArgs a) {
  const int t = (int)blockIdx.x / a.nrows, r = (int)blockIdx.x % a.nrows;
  gather_step_rows<TT, 1>(a.step[t], r, a.nrows, (int)threadIdx.x);
}

int gather_step_prep(hipStream_t st, const GatherStepArgs& a, int ttype, const PrepArgs& p) {
  if (!a.table || !a.angle_table || !a.rows || !a.view_index || !a.crows || !a.cviews || !a.heading || !a.elevation ||
      (!a.out && !a.out_lp) || (!a.cout && !a.cout_lp) || a.B <= 0 || a.V <= 0 || a.C <= 0 || (a.IMG & 7) || (a.ANG & 7)) {
    set_error("gather_step_prep: bad args (IMG and ANG must be multiples of 8)");
    return VLN_ERR_ARG;
  }
  const int nrows = a.B * a.V + a.B * a.C;
  long nb = (envdrop_prep_items(p) + 255) / 256;
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  const int nprep = (int)nb;
  // one table row per block: 2 / 4 rows per block with their loads in flight together measured SLOWER (1.815 / 1.855 vs 1.796 ms
  // per iteration, profiles/round2_notes.md) -- thousands of small workgroups hide the cold HBM accesses better than fat ones
  if (ttype == VLN_BF16) VLN_LAUNCH((gather_step_prep_kernel<bf16_raw, 1>), dim3(nrows + nprep), dim3(256), 0, st, a, p, nrows, nprep);
  else VLN_LAUNCH((gather_step_prep_kernel<float, 1>), dim3(nrows + nprep), dim3(256), 0, st, a, p, nrows, nprep);
  VLN_CHECK_LAUNCH("gather_step_prep");
  return VLN_OK;
}

}  // namespace vln

using namespace vln;

extern "C" int vln_gather_step(const void* table, int ttype, const float* angle_table, const int64_t* rows,
                               const int32_t* view_index, const int64_t* crows, const int32_t* cviews, const float* heading,
                               const float* elevation, float* out, void* out_bf16, float* cout, void* cout_bf16, int B, int V,
                               int C, int IMG, int ANG, uint64_t seed, uint64_t offset_pano, uint64_t offset_cand,
                               float p_feat, vln_stream_t s) {
  if (!table || !angle_table || !rows || !view_index || !crows || !cviews || !heading || !elevation || (!out && !out_bf16) ||
      (!cout && !cout_bf16) || B <= 0 || V <= 0 || C <= 0 || IMG <= 0 || ANG <= 0 || (IMG & 7) || (ANG & 7)) {
    set_error("vln_gather_step: bad args (IMG and ANG must be multiples of 8)");
    return VLN_ERR_ARG;
  }
  GatherStepArgs a{table, angle_table, (const long long*)rows, view_index, (const long long*)crows, cviews, heading, elevation,
                   out, (bf16_raw*)out_bf16, cout, (bf16_raw*)cout_bf16, B, V, C, IMG, ANG,
                   DropSpec{seed, offset_pano, p_feat}, DropSpec{seed, offset_cand, p_feat}, gather_check(table)};
  dim3 grid(B * V + B * C), block(256);
  if (ttype == VLN_BF16) VLN_LAUNCH(gather_step_kernel<bf16_raw>, grid, block, 0, (hipStream_t)s, a);
  else VLN_LAUNCH(gather_step_kernel<float>, grid, block, 0, (hipStream_t)s, a);
  VLN_CHECK_LAUNCH("gather_step");
  return VLN_OK;
}

namespace vln {
// steps [t0, t0 + a->T) of a ride as the kernels' argument block
int gather_ride_args(const ::vln_gather_ride& r, int t0, GatherRolloutArgs* a) {
  if (!r.table || !r.angle_table || !r.steps || r.T <= 0 || r.B <= 0 || r.V <= 0 || r.C <= 0 || r.IMG <= 0 || r.ANG <= 0 || (r.IMG & 7) || (r.ANG & 7)) {
    set_error("gather ride: bad args (IMG and ANG must be multiples of 8)");
    return VLN_ERR_ARG;
  }
  a->T = (r.T - t0 < kGatherMaxSteps) ? r.T - t0 : kGatherMaxSteps;
  a->nrows = r.B * r.V + r.B * r.C;
  a->ttype = r.ttype;
  // the pipelined passenger loop (gather_ride.h): one 8-element chunk per thread per row, one float4 of angle columns per
  // lane of a half-wave, ONE output precision, the same in every step
  int pipe = (r.IMG == 2048 && r.ANG == 128) ? 3 : 0;
  const GatherCheck chk = gather_check(r.table);
  for (int t = 0; t < a->T; ++t) {
    const vln_gather_rollout_step& q = r.steps[t0 + t];
    if (!q.rows || !q.view_index || !q.crows || !q.cviews || !q.heading || !q.elevation || (!q.out && !q.out_bf16) || (!q.cout && !q.cout_bf16)) {
      set_error("gather ride: null pointer in step %d", t0 + t);
      return VLN_ERR_ARG;
    }
    a->step[t] = GatherStepArgs{r.table, r.angle_table, (const long long*)q.rows, q.view_index, (const long long*)q.crows, q.cviews,
                                q.heading, q.elevation, q.out, (bf16_raw*)q.out_bf16, q.cout, (bf16_raw*)q.cout_bf16, r.B, r.V, r.C, r.IMG, r.ANG,
                                drop_spec(r.seed, q.offset_pano, r.p_feat, r.offset_base_dev), drop_spec(r.seed, q.offset_cand, r.p_feat, r.offset_base_dev), chk};
    const int lp = (q.out_bf16 && q.cout_bf16 && !q.out && !q.cout) ? 1 : 0, f32 = (q.out && q.cout && !q.out_bf16 && !q.cout_bf16) ? 2 : 0;
    pipe &= (lp | f32);
  }
  a->pipe = (pipe == 1 && r.ttype == VLN_BF16) || (pipe == 2 && r.ttype != VLN_BF16) ? pipe : 0;
  return VLN_OK;
}
int gather_ride_launch(hipStream_t st, const ::vln_gather_ride& r) {
  for (int t0 = 0; t0 < r.T; t0 += kGatherMaxSteps) {
    GatherRolloutArgs a{};
    int rc = gather_ride_args(r, t0, &a);
    if (rc != VLN_OK) return rc;
    dim3 grid((unsigned)(a.T * a.nrows)), block(256);
    if (r.ttype == VLN_BF16) VLN_LAUNCH(gather_rollout_kernel<bf16_raw>, grid, block, 0, st, a);
    else VLN_LAUNCH(gather_rollout_kernel<float>, grid, block, 0, st, a);
    VLN_CHECK_LAUNCH("gather_rollout");
  }
  return VLN_OK;
}
}  // namespace vln

extern "C" int vln_feature_table_extent(const void* table, int64_t n_rows, int n_angle_views) {
  if (!table || n_rows < 0 || (n_rows > 0 && n_angle_views <= 0)) { set_error("vln_feature_table_extent: bad args"); return VLN_ERR_ARG; }
  std::lock_guard<std::mutex> lock(g_ext_mu);
  if (n_rows == 0) g_ext.erase(table);
  else g_ext[table] = Extent{(long)n_rows, n_angle_views};
  return VLN_OK;
}

extern "C" int vln_gather_rollout(const void* table, int ttype, const float* angle_table, const vln_gather_rollout_step* steps, int T,
                                  int B, int V, int C, int IMG, int ANG, uint64_t seed, float p_feat, const uint64_t* offset_base_dev,
                                  vln_stream_t s) {
  const vln_gather_ride r{table, angle_table, steps, ttype, T, B, V, C, IMG, ANG, 0, seed, p_feat, 0.f, offset_base_dev};
  return gather_ride_launch((hipStream_t)s, r);
}

extern "C" int vln_gather_pano(const void* table, int ttype, const int64_t* rows, const int32_t* view_index,
                               const float* angle_table, float* out, void* out_bf16, int B, int V, int IMG, int ANG,
                               uint64_t seed, uint64_t offset, float p_feat, vln_stream_t s) {
  if (!table || !rows || !view_index || !angle_table || (!out && !out_bf16) || B <= 0 || V <= 0 || IMG <= 0 || ANG <= 0 || (IMG & 7) || (ANG & 7)) {
    set_error("vln_gather_pano: bad args (IMG and ANG must be multiples of 8)");
    return VLN_ERR_ARG;
  }
  dim3 grid(B * V), block(256);
  DropSpec dr{seed, offset, p_feat};
  const GatherCheck chk = gather_check(table);
  if (ttype == VLN_BF16)
    VLN_LAUNCH(gather_pano_kernel<bf16_raw>, grid, block, 0, (hipStream_t)s, (const bf16_raw*)table, (const long long*)rows, view_index, angle_table, out, (bf16_raw*)out_bf16, V, IMG, ANG, dr, chk);
  else
    VLN_LAUNCH(gather_pano_kernel<float>, grid, block, 0, (hipStream_t)s, (const float*)table, (const long long*)rows, view_index, angle_table, out, (bf16_raw*)out_bf16, V, IMG, ANG, dr, chk);
  VLN_CHECK_LAUNCH("gather_pano");
  return VLN_OK;
}

extern "C" int vln_gather_cands(const void* table, int ttype, const int64_t* rows, const int32_t* views,
                                const float* heading, const float* elevation, float* out, void* out_bf16, int BC,
                                int V, int IMG, int ANG, uint64_t seed, uint64_t offset, float p_feat, vln_stream_t s) {
  if (!table || !rows || !views || !heading || !elevation || (!out && !out_bf16) || BC <= 0 || V <= 0 || (IMG & 7) || (ANG & 7) || ANG <= 0 || IMG <= 0) {
    set_error("vln_gather_cands: bad args (IMG and ANG must be multiples of 8)");
    return VLN_ERR_ARG;
  }
  dim3 grid(BC), block(256);
  DropSpec dr{seed, offset, p_feat};
  const GatherCheck chk = gather_check(table);
  if (ttype == VLN_BF16)
    VLN_LAUNCH(gather_cands_kernel<bf16_raw>, grid, block, 0, (hipStream_t)s, (const bf16_raw*)table, (const long long*)rows, views, heading, elevation, out, (bf16_raw*)out_bf16, V, IMG, ANG, dr, chk);
  else
    VLN_LAUNCH(gather_cands_kernel<float>, grid, block, 0, (hipStream_t)s, (const float*)table, (const long long*)rows, views, heading, elevation, out, (bf16_raw*)out_bf16, V, IMG, ANG, dr, chk);
  VLN_CHECK_LAUNCH("gather_cands");
  return VLN_OK;
}
