// The chained step kernel and its host-side recorder (chain.h has the design).
#include <string.h>

#include <map>
#include <mutex>
#include <vector>

#include "chain.h"
#include "step_bodies.h"
#include "gather_body.h"
#include "../../include/vln_hip.h"

namespace vln {

#include "gemm_nt_body.h"
#define VLN_ATTN_FUSED_BODY_ONLY
#include "attention_fused.h"

int g_chain_mode = 0;

// ---- device side --------------------------------------------------------------------------------------------------------
#define CH_AGENT_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// Completion flags, one word per workgroup of the launch: a workgroup stores `epoch` into its word when its stores are
// released.  (Counters were tried first: ~900 agent-scope atomic adds on a handful of addresses serialise at ~90 ns each
// -- 164 us per launch against 65 us for the launches it replaced.  Flags are plain write-through stores to distinct
// words; nothing is ever read-modify-written.)  The epoch of a launch is one more than what the flags held when it
// started: every launch that uses a flag block has the same number of workgroups and sets every word, so a workgroup
// reads the previous epoch from its OWN word and no launch needs to reset anything (hipGraph replays included).
// All waves of the workgroup call this.  Wave 0 polls the producers' words; afterwards every wave invalidates what it may
// hold of the producers' outputs (agent-scope acquire).
__device__ __forceinline__ void chain_wait(const unsigned* flags, int n, unsigned epoch, unsigned* sticky) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    unsigned spins = 0;
    for (;;) {
      bool ok = true;
      for (int i = lane; i < n; i += 64) ok &= (CH_AGENT_LOAD(flags + i) == epoch);
      if (__all(ok)) break;
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 20)) {          // ~1 s: a producer never ran -- raise the sticky error, do not hang
        if (lane == 0) __hip_atomic_fetch_add(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

constexpr int kChainSmem = 2 * 36864;       // two virtual GEMM blocks (bf16: 2 buffers x 2 planes x 64 rows x 144 B each)

struct ChainWaits {          // what a workgroup waits for: producer flag ranges of its two dependencies
  const unsigned* main; int n_main; const unsigned* pre; int n_pre; unsigned epoch; unsigned* sticky;
  __device__ __forceinline__ void wait_main() const { if (n_main > 0) chain_wait(main, n_main, epoch, sticky); }
  __device__ __forceinline__ void wait_pre() const { if (n_pre > 0) chain_wait(pre, n_pre, epoch, sticky); }
};

template <typename TW, int RW, int SL, bool kBwd>
__device__ __forceinline__ void chain_attn(const unsigned char* ap, bool early, int local, unsigned char* smem, const ChainWaits& w) {
  static_assert(attn_fused_smem_bytes<TW, RW, SL>() <= kChainSmem, "LDS");
  const AttnFusedArgs& a = *reinterpret_cast<const AttnFusedArgs*>(ap);
  attn_fused_body<TW, RW, SL, kBwd>(
      a, local, smem,
      [&] {                                   // before the context block is requested
        w.wait_pre();
        if (!early) w.wait_main();
      },
      [&] { if (early) w.wait_main(); });     // before the producer's vector is read
}

template <typename TW>
__global__ __launch_bounds__(kChainThreads) void chain_kernel(ChainArgs c_by_value) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[kChainSmem];
  // The argument block is read through the kernel-argument segment pointer (constant address space: uniform scalar loads at
  // run-time offsets; indexing the by-value parameter itself would have the compiler copy all of it to scratch).
  const ChainArgs& c = *(const ChainArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  const int wg = blockIdx.x;
  unsigned* const flags = c.flags;
  const int nstages = c.nstages;
  int s = 0;
  for (int i = 1; i < nstages; ++i)
    if (wg >= c.st[i].first) s = i;
  const ChainStageDesc& d = c.st[s];
  const int local = wg - d.first;
  const unsigned char* ap = c.args + d.arg_off;
  ChainWaits w;
  w.epoch = CH_AGENT_LOAD(flags + wg) + 1u;
  w.sticky = c.sticky;
  w.main = d.dep_main >= 0 ? flags + c.st[d.dep_main].first : nullptr;
  w.n_main = d.dep_main >= 0 ? c.st[d.dep_main].nwg : 0;
  w.pre = d.dep_pre >= 0 ? flags + c.st[d.dep_pre].first : nullptr;
  w.n_pre = d.dep_pre >= 0 ? c.st[d.dep_pre].nwg : 0;
  const int tid = threadIdx.x, half = tid >> 8, t256 = tid & 255;
  auto wait_all = [&] { w.wait_pre(); w.wait_main(); };
  const long gfirst = (long)local * kChainThreads + tid, gstride = (long)d.nwg * kChainThreads;   // flat elementwise stages

  switch (d.kind) {
    case CK_GEMM_NT: {
      constexpr int BK = GemmCfg<TW>::BK;
      const GemmNTArgs& a = *reinterpret_cast<const GemmNTArgs*>(ap);
      const int nvb = d.gx * d.gy * d.gz;
      const int v0 = 2 * local, v1 = 2 * local + 1;
      const int vbi = half ? v1 : v0;
      const bool active = vbi < nvb;
      const int vc = active ? vbi : nvb - 1;
      VBlock vb{vc % d.gx, (vc / d.gx) % d.gy, vc / (d.gx * d.gy), t256, smem + half * (kChainSmem / 2)};
      int nbar = gemm_nt_nsteps(a, (v0 / d.gx) % d.gy, BK);
      if (v1 < nvb) nbar = max(nbar, gemm_nt_nsteps(a, (v1 / d.gx) % d.gy, BK));
      if (!d.early) wait_all();
      gemm_nt_body<TW, 2, true, 1>(a, vb, active, nbar, [&] { if (d.early) wait_all(); });
      break;
    }
    case CK_ATTN_FWD_0: chain_attn<TW, 10, sizeof(TW) == 2 ? 1 : 2, false>(ap, d.early != 0, local, smem, w); break;
    case CK_ATTN_FWD_1: chain_attn<TW, 2, sizeof(TW) == 2 ? 2 : 4, false>(ap, d.early != 0, local, smem, w); break;
    case CK_ATTN_FWD_2: chain_attn<TW, 2, sizeof(TW) == 2 ? 5 : 9, false>(ap, d.early != 0, local, smem, w); break;
    case CK_ATTN_FWD_3: chain_attn<TW, 5, sizeof(TW) == 2 ? 5 : 9, false>(ap, d.early != 0, local, smem, w); break;
    case CK_ATTN_BWD_0: chain_attn<TW, 10, sizeof(TW) == 2 ? 1 : 2, true>(ap, d.early != 0, local, smem, w); break;
    case CK_ATTN_BWD_1: chain_attn<TW, 2, sizeof(TW) == 2 ? 2 : 4, true>(ap, d.early != 0, local, smem, w); break;
    case CK_ATTN_BWD_2: chain_attn<TW, 2, sizeof(TW) == 2 ? 5 : 9, true>(ap, d.early != 0, local, smem, w); break;
    case CK_ATTN_BWD_3: chain_attn<TW, 5, sizeof(TW) == 2 ? 5 : 9, true>(ap, d.early != 0, local, smem, w); break;
    case CK_LSTM_PW_FWD: {
      const LstmPwFwd& a = *reinterpret_cast<const LstmPwFwd*>(ap);
      wait_all();
      // two virtual 256-thread blocks per workgroup; gy = iterations (the same for every block: uniform barriers)
      lstm_pw_fwd_body(a, 2 * local + half, d.gx, d.gy, t256, reinterpret_cast<float (*)[64]>(smem + half * 1024));
      break;
    }
    case CK_LSTM_PW_BWD: { const LstmPwBwd& a = *reinterpret_cast<const LstmPwBwd*>(ap); wait_all(); lstm_pw_bwd_body(a, gfirst, gstride); break; }
    case CK_REDUCE_EPI: { const ReduceEpiArgs& a = *reinterpret_cast<const ReduceEpiArgs*>(ap); wait_all(); reduce_epilogue_body(a, gfirst, gstride); break; }
    case CK_TANH_DROP_BWD: { const TanhDropBwdArgs& a = *reinterpret_cast<const TanhDropBwdArgs*>(ap); wait_all(); tanh_drop_bwd_body(a, gfirst, gstride); break; }
    case CK_PREP: { const PrepArgs& a = *reinterpret_cast<const PrepArgs*>(ap); wait_all(); envdrop_prep_body(a, gfirst, gstride); break; }
    case CK_PREP_BWD: { const PrepBwdArgs& a = *reinterpret_cast<const PrepBwdArgs*>(ap); wait_all(); envdrop_prep_bwd_body(a, gfirst, gstride); break; }
    case CK_GATHER_STEP: {
      const GatherStepArgs& a = *reinterpret_cast<const GatherStepArgs*>(ap);
      wait_all();
      for (int r = 2 * local + half; r < d.gx; r += 2 * d.nwg) gather_step_row<TW>(a, r, t256);
      break;
    }
    default: break;
  }

  // release: every wave's stores have reached L2, then one agent-scope write-back and the workgroup's flag
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_store(flags + wg, w.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------------
namespace {

struct Recorder {
  bool active = false;
  hipStream_t st = nullptr;
  int dtype = -1;
  int next_main = kDepPrev, next_pre = kDepNone, next_early = 0;
  double bytes = 0.0;
  int arg_used = 0, nwg_total = 0;
  ChainArgs c;
};
thread_local Recorder g_rec;

// Flag blocks: one word per workgroup of a launch, zero when first used.  A launch that is being captured into a graph gets
// a block of its own for good (the graph may be replayed on any stream, beside any other launch); direct launches use one
// block per (stream, grid size): stream order serialises them, and the epoch rule needs every user of a block to set the
// same number of words.
std::mutex g_flag_mu;
struct FlagPool { unsigned* slab = nullptr; size_t used = 0; std::map<std::pair<hipStream_t, int>, unsigned*> direct; };
std::map<int, FlagPool> g_flag_pools;
constexpr size_t kSlabWords = 1u << 20;           // 4 MiB
constexpr size_t kMaxBlockWords = 1u << 14;

unsigned* take_words(FlagPool& p, size_t n, bool may_alloc) {
  n = (n + 31) & ~size_t(31);                                 // 128-byte granules
  if (n > kMaxBlockWords) return nullptr;
  if (!p.slab || p.used + n > kSlabWords) {
    if (!may_alloc) return nullptr;
    void* m = nullptr;
    if (hipMalloc(&m, kSlabWords * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(m, 0, kSlabWords * sizeof(unsigned)) != hipSuccess) return nullptr;
    p.slab = (unsigned*)m; p.used = 0;
  }
  unsigned* r = p.slab + p.used;
  p.used += n;
  return r;
}
unsigned* flags_for(hipStream_t st, int nwg) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) cs = hipStreamCaptureStatusNone;
  std::lock_guard<std::mutex> lock(g_flag_mu);
  FlagPool& p = g_flag_pools[dev];
  // no allocation while this thread captures: chain_prime (called outside) keeps room for the blocks of a capture
  if (cs != hipStreamCaptureStatusNone) return take_words(p, (size_t)nwg, false);
  auto key = std::make_pair(st, nwg);
  auto it = p.direct.find(key);
  if (it != p.direct.end()) return it->second;
  unsigned* b = take_words(p, (size_t)nwg, true);
  if (b) p.direct[key] = b;
  return b;
}

}  // namespace

// Called by the step entry points before they (possibly) start capturing: allocations are not allowed under capture.
int chain_prime() {
  if (!g_chain_mode) return VLN_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return VLN_ERR_HIP;
  if (!sticky_dev_word()) { set_error("chained step: no host-mapped status word"); return VLN_ERR_HIP; }
  std::lock_guard<std::mutex> lock(g_flag_mu);
  FlagPool& p = g_flag_pools[dev];
  if (!p.slab || p.used + 4 * kMaxBlockWords > kSlabWords) {      // room for the chains a step call may capture
    p.slab = nullptr;
    if (!take_words(p, 32, true)) { (void)hipGetLastError(); set_error("chained step: flag slab allocation failed"); return VLN_ERR_HIP; }
  }
  return VLN_OK;
}


bool chain_recording() { return g_rec.active; }
int chain_last() { return g_rec.active ? g_rec.c.nstages - 1 : -1; }
void chain_next(int dep_main, int dep_pre, int early) {
  if (!g_rec.active) return;
  g_rec.next_main = dep_main; g_rec.next_pre = dep_pre; g_rec.next_early = early;
}

static void rec_reset() {
  g_rec.dtype = -1; g_rec.bytes = 0.0; g_rec.arg_used = 0; g_rec.nwg_total = 0; g_rec.c.nstages = 0;
  g_rec.next_main = kDepPrev; g_rec.next_pre = kDepNone; g_rec.next_early = 0;
}

int chain_flush() {
  Recorder& r = g_rec;
  if (!r.active || r.c.nstages == 0) return VLN_OK;
  const bool was = r.active;
  r.active = false;                       // the launch below goes through launch_timed -> chain_flush
  int status = VLN_OK;
  r.c.flags = flags_for(r.st, r.nwg_total);
  r.c.sticky = sticky_dev_word();
  if (!r.c.flags || !r.c.sticky) { set_error("chained step: no flag block (allocation failed, or a capture without chain_prime)"); status = VLN_ERR_HIP; }
  if (status == VLN_OK) {
    dim3 grid(r.nwg_total), block(kChainThreads);
    if (r.dtype == W_F32) launch_timed(K_CHAIN, r.bytes, chain_kernel<float>, grid, block, 0, r.st, r.c);
    else launch_timed(K_CHAIN, r.bytes, chain_kernel<bf16_raw>, grid, block, 0, r.st, r.c);
    status = check_hip(hipGetLastError(), "chain_kernel");
  }
  rec_reset();
  r.active = was;
  return status;
}

bool chain_add(hipStream_t st, int kind, int gx, int gy, int gz, const void* args, int nbytes, double algo_bytes, int dtype) {
  Recorder& r = g_rec;
  if (!r.active) return false;
  if (st != r.st) { chain_flush(); return false; }
  const int padded = (nbytes + 15) & ~15;
  if (dtype >= 0 && r.dtype >= 0 && dtype != r.dtype) chain_flush();
  if (r.c.nstages == kChainMaxStages || r.arg_used + padded > kChainArgBytes) chain_flush();
  long nwg;
  switch (kind) {
    case CK_GEMM_NT: nwg = ((long)gx * gy * gz + 1) / 2; break;
    case CK_LSTM_PW_FWD: nwg = ((long)gx + 1) / 2; break;
    case CK_GATHER_STEP: nwg = ((long)gx + 1) / 2; if (nwg > 256) nwg = 256; break;      // row pairs, grid-strided
    case CK_ATTN_FWD_0: case CK_ATTN_FWD_1: case CK_ATTN_FWD_2: case CK_ATTN_FWD_3:
    case CK_ATTN_BWD_0: case CK_ATTN_BWD_1: case CK_ATTN_BWD_2: case CK_ATTN_BWD_3: nwg = gx; break;
    default: nwg = ((long)gx + 1) / 2; break;           // flat elementwise stages: 512 threads cover two 256-thread blocks
  }
  if (nwg <= 0 || r.nwg_total + nwg > (long)kMaxBlockWords) { chain_flush(); return false; }
  ChainStageDesc& d = r.c.st[r.c.nstages];
  memset(&d, 0, sizeof(d));
  d.kind = kind; d.first = r.nwg_total; d.nwg = (int)nwg; d.gx = gx; d.gy = gy; d.gz = gz; d.arg_off = r.arg_used;
  const int prev = r.c.nstages - 1;
  d.dep_main = (r.next_main == kDepPrev) ? (prev >= 0 ? prev : kDepNone) : r.next_main;
  d.dep_pre = (r.next_pre == kDepPrev) ? (prev >= 0 ? prev : kDepNone) : r.next_pre;
  if (d.dep_main >= r.c.nstages) d.dep_main = prev >= 0 ? prev : kDepNone;      // stale index from a chain flushed meanwhile
  if (d.dep_pre >= r.c.nstages) d.dep_pre = kDepNone;
  d.early = r.next_early;
  r.next_main = kDepPrev; r.next_pre = kDepNone; r.next_early = 0;
  memcpy(r.c.args + r.arg_used, args, nbytes);
  r.arg_used += padded;
  r.nwg_total += (int)nwg;
  r.bytes += algo_bytes;
  if (dtype >= 0) r.dtype = dtype;
  r.c.nstages++;
  return true;
}

ChainScope::ChainScope(hipStream_t st, bool enable) : owner_(false) {
  if (!enable || g_rec.active) return;
  g_rec.active = true; g_rec.st = st;
  rec_reset();
  owner_ = true;
}
int ChainScope::finish() {
  if (!owner_) return VLN_OK;
  const int status = chain_flush();
  g_rec.active = false;
  owner_ = false;
  return status;
}
ChainScope::~ChainScope() {
  if (!owner_) return;
  rec_reset();                 // an error return left stages behind: they are dropped with the failed step
  g_rec.active = false;
}

}  // namespace vln

using namespace vln;

extern "C" int vln_set_chain(int mode) {
  if (mode < 0 || mode > 1) { set_error("vln_set_chain: mode must be 0 or 1"); return VLN_ERR_ARG; }
  g_chain_mode = mode;
  return VLN_OK;
}
extern "C" int vln_get_chain(void) { return g_chain_mode; }
