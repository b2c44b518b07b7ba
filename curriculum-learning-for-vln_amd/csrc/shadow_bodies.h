// Weight shadows: the compute-dtype copy and / or the transposed copy of an fp32 parameter matrix, refreshed once per optimizer
// step (vln_shadow_refresh, the prologue launch, or -- round 5 -- passenger workgroups of the encoder's forward recurrence launch
// for the modules that launch does not read: vln_gather_ride::shadow_jobs).  Device bodies shared by those carriers.
#pragma once
#include "../../include/vln_hip.h"
#include "vln_internal.h"
#include "common.h"

namespace vln {

// ---- all shadows of a module in ONE launch ------------------------------------------------------------------------
// job = one fp32 matrix [N,K] (optionally the sum of two: b_ih + b_hh) -> its compute-dtype copy and/or its transposed
// copy.  64x64 tiles; a workgroup finds its job from the prefix sums of the jobs' tile counts.
template <int NJ>
struct ShadowJobsT {
  vln_shadow_job j[NJ];
  int tile0[NJ + 1];
  int n;
};
// the 4 fp32 of a source row chunk; NT: non-temporal (a passenger of the recurrence launch reads every source once and must not
// push the recurrence's hand-off lines out of the L2, gather_ride.h)
template <bool NT>
__device__ __forceinline__ void shadow_ld4(const float* p, float (&v)[4]) {
  if constexpr (NT) {
    v[0] = __builtin_nontemporal_load(p); v[1] = __builtin_nontemporal_load(p + 1); v[2] = __builtin_nontemporal_load(p + 2); v[3] = __builtin_nontemporal_load(p + 3);
  } else {
    Elt<float>::ld4(p, v);
  }
}
template <typename TO>
__device__ __forceinline__ void shadow_tile(const vln_shadow_job& q, int tile, float (*lds)[65]) {
  const int tk = (q.K + 63) / 64;
  const int n0 = (tile / tk) * 64, k0 = (tile % tk) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int n = n0 + r, k = k0 + tx;
    float v = 0.f;
    if (n < q.N && k < q.K) {
      v = q.src[(long)n * q.ld_src + k];
      if (q.src2) v += q.src2[(long)n * q.ld_src + k];
      if (q.dst) Elt<TO>::st(reinterpret_cast<TO*>(q.dst) + (long)n * q.ld_dst + k, v);
    }
    lds[r][tx] = v;
  }
  if (!q.dst_t) return;
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int k = k0 + r, n = n0 + tx;
    if (k < q.K && n < q.N) Elt<TO>::st(reinterpret_cast<TO*>(q.dst_t) + (long)k * q.ld_dst_t + n, lds[tx][r]);
  }
}
// Same tile with 16-byte loads and 8/16-byte stores (N, K and every leading dimension multiples of 4, 16-byte aligned
// bases): a thread owns 4 consecutive k of a row on the way in and 4 consecutive n of a transposed row on the way out.  The
// scalar form above moved 4 bytes in and 2 bytes out per lane and instruction (43 us for the 40 MB of EnvDrop weights).
template <typename TO, bool NT>
__device__ __forceinline__ void shadow_tile_v4(const vln_shadow_job& q, int tile, float (*lds)[65]) {
  const int tk = (q.K + 63) / 64;
  const int n0 = (tile / tk) * 64, k0 = (tile % tk) * 64;
  const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + 16 * i, n = n0 + r, k = k0 + c4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < q.N && k < q.K) {
      shadow_ld4<NT>(q.src + (long)n * q.ld_src + k, v);
      if (q.src2) {
        float w[4];
        shadow_ld4<NT>(q.src2 + (long)n * q.ld_src + k, w);
        v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
      }
      if (q.dst) Elt<TO>::st4(reinterpret_cast<TO*>(q.dst) + (long)n * q.ld_dst + k, v);
    }
    lds[r][c4] = v[0]; lds[r][c4 + 1] = v[1]; lds[r][c4 + 2] = v[2]; lds[r][c4 + 3] = v[3];
  }
  if (!q.dst_t) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + 16 * i, k = k0 + r, n = n0 + c4;
    if (k < q.K && n < q.N) {
      const float v[4] = {lds[c4][r], lds[c4 + 1][r], lds[c4 + 2][r], lds[c4 + 3][r]};
      Elt<TO>::st4(reinterpret_cast<TO*>(q.dst_t) + (long)k * q.ld_dst_t + n, v);
    }
  }
}
__device__ __forceinline__ bool shadow_vec_ok(const vln_shadow_job& q) {
  const uintptr_t al = (uintptr_t)q.src | (uintptr_t)q.src2 | (uintptr_t)q.dst | (uintptr_t)q.dst_t;
  return !((q.N | q.K | (int)q.ld_src | (int)q.ld_dst | (int)q.ld_dst_t) & 3) && !(al & 15);
}
template <bool NT, int NJ>
__device__ __forceinline__ void shadow_block(const ShadowJobsT<NJ>& a, int block, float (*lds)[65]) {
  int ji = 0;
  while (ji + 1 < a.n && block >= a.tile0[ji + 1]) ++ji;
  const vln_shadow_job& q = a.j[ji];
  const int tile = block - a.tile0[ji];
  if (shadow_vec_ok(q)) {
    if (q.out_type == W_BF16) shadow_tile_v4<bf16_raw, NT>(q, tile, lds);
    else shadow_tile_v4<float, NT>(q, tile, lds);
  } else if (q.out_type == W_BF16) shadow_tile<bf16_raw>(q, tile, lds);
  else shadow_tile<float>(q, tile, lds);
}

// host: the argument block of up to NJ jobs; *tiles = its 64 x 64 tiles
template <int NJ>
static inline int shadow_jobs(const vln_shadow_job* jobs, int n, ShadowJobsT<NJ>* a, int* tiles) {
  if (n < 0 || n > NJ) { set_error("shadow_refresh: %d jobs do not fit an argument block of %d", n, NJ); return VLN_ERR_ARG; }
  a->n = n;
  int t = 0;
  for (int i = 0; i < n; ++i) {
    const vln_shadow_job& q = jobs[i];
    if (!q.src || q.N <= 0 || q.K <= 0 || (!q.dst && !q.dst_t)) { set_error("shadow_refresh: bad job %d", i); return VLN_ERR_ARG; }
    a->j[i] = q;
    a->tile0[i] = t;
    t += ((q.N + 63) / 64) * ((q.K + 63) / 64);
  }
  a->tile0[n] = t;
  *tiles = t;
  return VLN_OK;
}

// the shadows a gather ride carries (vln_gather_ride::shadow_jobs): a smaller block, the recurrence launch's arguments are near
// the 4 KB a launch may pass
constexpr int kRideShadowJobs = 8;
struct RideShadows { ShadowJobsT<kRideShadowJobs> jobs; int tiles; };

}  // namespace vln
