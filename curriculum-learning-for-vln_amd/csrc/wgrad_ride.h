// A module's grouped weight / bias gradients as PASSENGER workgroups of the backward recurrence launch (encoder_persist.h).
//
// The encoder's BPTT is 128 workgroups of latency-bound hand-offs for ~220 us; the other half of the chip idles.  The DECODER's
// parameter gradients (one pack of the rollout's operands into bf16 fragment order, one packed contraction over steps x batch
// rows, the bias column sums: 57-69 us as three launches in front of the BPTT, profiles/round4_timeline_chained.txt) depend on
// nothing the BPTT produces and nothing before the optimizer reads them: they ride here instead.
//
//   phase 1   passenger p takes pack blocks p, p + NP, ... and then the column-sum blocks (independent of each other)
//   barrier   passengers only: release (the pack area is plain stores) -> one agent-scope counter -> acquire
//   phase 2   contraction tiles p, p + NP, ...   (NP % 8 == 0 and the recurrence's workgroup count % 8 == 0: tile `lt` of the
//             XCD-aware order still runs on XCD lt / per_xcd, as in the stand-alone launch)
//
// Same arithmetic per block / tile as the stand-alone launches (wgrad_bodies.h): results are bit-identical to them.
// The passenger barrier needs all NP passengers resident: the launch claims a CU's LDS per workgroup (one workgroup per CU) and
// nrec + NP <= the device's CUs (ride_passengers, encoder.hip); its spin is bounded and raises the recurrence's sticky word.
#pragma once
#include "wgrad_bodies.h"

constexpr int kRideWgradJobs = 8;       // weight-gradient products of one ride
constexpr int kRideColsumJobs = 4;      // bias column sums of one ride

struct WgradRideArgs {
  PackJobsT<kRideWgradJobs> pk;
  PackedJobsT<kRideWgradJobs> g;
  ColsumJobsT<kRideColsumJobs> cs;
  unsigned* bar;            // two zero-initialised words of the sync workspace: arrivals, passes (self-resetting)
  int pack_blocks, cs_blocks, tiles, terms;     // tiles = g.per_xcd * 8 virtual blocks; terms: 1 plain bf16 operands, 3 split
  int on;
};

// every thread of every passenger calls this; `part` = 4 KB of LDS for the column sums
__device__ __forceinline__ void wgrad_ride_passenger(const WgradRideArgs& r, int p, int np, unsigned* status, unsigned* sticky,
                                                     float4 (*part)[4]) {
  for (int b = p; b < r.pack_blocks; b += np) wgrad_pack_block(r.pk, b);
  for (int b = p; b < r.cs_blocks; b += np) colsum_grouped_block(r.cs, b, 0, part);
  // ---- passengers-only barrier
  __shared__ int s_ok;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // this workgroup's pack blocks reach memory before its arrival does
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(r.bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int ok = 1;
    while (__hip_atomic_load(r.bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)np) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 22)) {                           // ~1 s: a passenger is not resident / died
        VLN_AGENT_STORE(status, 1u);
        __hip_atomic_fetch_add(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = 0;
        break;
      }
    }
    s_ok = ok;
    // the last passenger through leaves both words as the launch found them (zero)
    const unsigned through = __hip_atomic_fetch_add(r.bar + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (through + 1u == (unsigned)np) {
      __hip_atomic_store(r.bar, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(r.bar + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // the other passengers' pack blocks, not this XCD's stale lines
  if (!s_ok) return;                                        // timed out: the error is raised, the gradients are not formed
  for (int t = p; t < r.tiles; t += np) {
    if (r.terms == 3) wgrad_packed_tile<3>(r.g, t, 0);
    else wgrad_packed_tile<1>(r.g, t, 0);
  }
}
