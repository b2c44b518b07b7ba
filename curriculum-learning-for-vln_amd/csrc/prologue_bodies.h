// Bodies of the iteration's PROLOGUE (round 4): the pull of the batch out of pinned host memory (vln_host_fetch), the device clock's
// tick (vln_tick) and the weight-shadow refresh (vln_shadow_refresh) -- three launches at the top of every iteration that do not
// depend on each other -- also run as block ranges of ONE launch (vln_prologue, gemm.hip).
#pragma once
#include "vln_internal.h"
#include "../../include/vln_hip.h"

namespace vln {

struct TickArgs { unsigned long long* w64[VLN_TICK_MAX]; unsigned long long inc64[VLN_TICK_MAX]; unsigned* w32[VLN_TICK_MAX];
                  unsigned inc32[VLN_TICK_MAX]; int n; };
__device__ __forceinline__ void tick_body(const TickArgs& a, int i) {
  if (i < a.n) {
    if (a.w64[i]) *a.w64[i] += a.inc64[i];
    if (a.w32[i]) *a.w32[i] += a.inc32[i];
  }
}
int tick_args(const vln_tick_item* items, int n, TickArgs* a);        // api.hip: validates and fills

// The host runs AHEAD of the device (it enqueues many replays), so one slot would be overwritten before the launch that should read
// it has run: the slots form a RING of `ring` words and the kernel picks slot (*seq % ring), where `seq` is a device word counting
// the fetches that have run; the last workgroup to finish bumps it (every workgroup has read it by then).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct FetchArgs { const unsigned long long* slots; int ring; unsigned long long* seq; unsigned* done; u32x4* dst; long n16; };
__device__ __forceinline__ void host_fetch_body(const FetchArgs& f, int block, int nblocks, int tid) {
  const unsigned long long n = *f.seq;
  const u32x4* src = reinterpret_cast<const u32x4*>(__hip_atomic_load(f.slots + (n % (unsigned long long)f.ring), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
  if (src)                                       // (a slot the host never selected holds 0: nothing to pull, the sequence still advances)
    for (long i = (long)block * 256 + tid; i < f.n16; i += (long)nblocks * 256)
      f.dst[i] = __builtin_nontemporal_load(src + i);
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(f.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (unsigned)nblocks - 1u) {           // every workgroup has read *seq: the next launch sees the next slot
      __hip_atomic_store(f.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(f.seq, n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// The TAIL of the current batch's blob, pulled later in the same iteration by one workgroup (a passenger of the forward recurrence
// launch, or its own one-block launch): the slot is the one the iteration's head fetch used, i.e. (*seq - 1) % ring.
struct FetchPart { const unsigned long long* slots; const unsigned long long* seq; u32x4* dst; long off16, n16; int ring; int on; };
__device__ __forceinline__ void host_fetch_part_body(const FetchPart& f, int tid) {
  const unsigned long long done = *f.seq;
  if (done == 0ull) return;                    // no head fetch has run on this ring yet: there is no current batch to take the tail of
  const unsigned long long n = done - 1ull;
  const u32x4* base = reinterpret_cast<const u32x4*>(__hip_atomic_load(f.slots + (n % (unsigned long long)f.ring), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
  if (!base) return;                           // an empty slot (never selected)
  const u32x4* src = base + f.off16;
  long i = tid;
  for (; i + 7 * 256 < f.n16; i += 8 * 256) {           // eight PCIe reads in flight per thread
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(src + i + k * 256);
#pragma unroll
    for (int k = 0; k < 8; ++k) f.dst[i + k * 256] = v[k];
  }
  for (; i < f.n16; i += 256) f.dst[i] = __builtin_nontemporal_load(src + i);
}
int fetch_part_args(const ::vln_gather_ride& r, FetchPart* f);        // api.hip: validates and fills (on = 0 when the ride carries none)
int launch_fetch_part(hipStream_t st, const FetchPart& f);            // api.hip: the same body as a one-block launch

int fetch_args(const uint64_t* slots_dev, int ring, uint64_t* seq, uint32_t* done, void* dst, int64_t nbytes, FetchArgs* f, int* blocks);

}  // namespace vln
