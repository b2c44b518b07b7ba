// Persistent packed-(bi)LSTM recurrence: ONE launch walks all L time steps (included by encoder.hip).
//
// Why: one launch per time step costs ~6.5 us + ~1.5 us of inter-kernel gap for <1 us of work (160 launches per
// EnvDrop iteration).  Here every workgroup keeps its slice of W_hh in registers and its (row, unit) cell
// and hidden state in registers for the whole sequence; only the new hidden slice crosses workgroups.
//
// Geometry: workgroup (jb, d, bb) = 16 hidden units x direction d x 16 batch rows, 4 waves = 4 gates.
// Dependency group = the Hd/16 workgroups sharing (d, bb): step s needs the h slices all of them wrote in
// step s-1.  Hand-off (cdna_hip_programming.md Guideline 16, recipe R1, counter form):
//   producer: every handed-off element is stored write-through (sc1) -> every wave `s_waitcnt vmcnt(0)` ->
//             workgroup barrier -> ONE lane adds 1 to the group's counter (agent-scope atomic);
//   consumer: wave 0 polls the counter with sc1 loads (+ s_sleep) until it reaches njb*s -> workgroup barrier
//             -> EVERY load of handed-off bytes is a `buffer_load_dwordx4 ... sc1` (bypasses this CU's L1).
// No fence is needed in this form; results do not depend on placement or timing.  All workgroups must be
// co-resident (the host only takes this path when the grid fits the 256 CUs); every spin is bounded and a
// timeout sets the status word and lets the kernel drain (wrong numbers, never a hang).
#pragma once


// A/B switch for scripts/lstm_probe.hip.  1 = the forward's bookkeeping stores are issued AFTER the arrival.  Measured
// on MI355X (same box, interleaved): 572 us vs 515 us per 80-step launch -- LATE IS SLOWER: the poll's
// `s_waitcnt vmcnt(0)` then also waits for those seven stores.  Default 0.
#ifndef VLN_FWD_LATE_STORES
#define VLN_FWD_LATE_STORES 0
#endif
#ifndef VLN_STAMP          // scripts/lstm_probe.hip defines it to record s_memrealtime at the stages of one workgroup
#define VLN_STAMP(k)
#endif

// u32x4_t, VLN_AGENT_LOAD / VLN_AGENT_STORE: common.h

// Arrival signalling.  EACH dependency group owns a 128-byte line of the sync header.  That placement is what matters:
// with the eight groups' counters packed in adjacent words (one line) every step sent 128 atomics and eight pollers
// to the same line and a step cost 6.3 us (forward) / 8.7 us (backward); one line per group: 3.4 / 2.7 us
// (scripts/lstm_probe.hip, MI355X, B=64, Hd=256, L=80).  Two forms, A/B-measured on one box:
//   VLN_SYNC_FLAGS 0 (default): one counter per group -- arrive = agent-scope atomic add, wait = poll until njb*epoch;
//   VLN_SYNC_FLAGS 1: one word per producer in the line -- arrive = write-through store of the epoch, wait = one load
//                     of the line (lane p = producer p) until every word reached it.  ~4 % slower (289 vs 278 us).
// Epochs count steps from 1; the header is zeroed before every launch.
constexpr int kSyncHeaderBytes = 8192;    // status word at byte 128, flag lines (32 groups x 128 B) from byte 256
constexpr int kRideBarWord = 1536;        // ... and (byte 6144) the two words of the gradient ride's passenger barrier (wgrad_ride.h)
#ifndef VLN_SYNC_FLAGS
#define VLN_SYNC_FLAGS 0
#endif
// Workgroup -> (hidden slice jb, direction d, row block bb).  The grid is one-dimensional; workgroups are dealt to the 8
// XCDs round-robin, so with xcd_map the 16 workgroups of one dependency group (same d, bb: they exchange hidden state
// every step) get ids congruent mod (number of groups) and land on ONE XCD when there are 8 groups: their hand-off
// loads then meet the producer's write-through stores in that XCD's L2 instead of crossing the fabric.  Placement is a
// speed matter only -- the protocol (sc1 stores, counters, sc1 loads) does not assume it.
struct PersistIdx { int jb, d, bb, nbb; };
// id: the workgroup's index among the recurrence's workgroups (blockIdx.x, or PersistRole::rid of a launch with passengers)
__device__ __forceinline__ PersistIdx persist_index(int njb, int dirs, int nbb, int xcd_map, int id) {
  const int ngrp = dirs * nbb;
  int g, jb;
  if (xcd_map) { g = id % ngrp; jb = id / ngrp; } else { jb = id % njb; g = id / njb; }
  return PersistIdx{jb, g % dirs, g / dirs, nbb};
}

// Who a workgroup of a launch WITH PASSENGERS is.  Plain form: the first `nrec` workgroups are the recurrence, the rest passengers --
// workgroups go to the 8 XCDs round-robin, so every XCD's L2 serves both kinds, and the passengers' streaming traffic slows the
// recurrence's hand-offs through that L2 (B = 64: the BPTT launch 166 us without passengers, 190 us with 64, 200 us with 96).
// PARTITIONED form (bit 2 of xcd_map; round 5): of every 8 consecutive workgroups the first 4 -- XCDs 0-3 -- are recurrence
// workgroups, the last 4 -- XCDs 4-7 -- passengers (the first `np` of them work, the others leave at once): the two kinds share no
// L2.  Recurrence workgroup r sits on XCD r % 4, so with persist_index's group = r % (dirs * row blocks) a dependency group still
// has ONE XCD whenever the group count is a multiple of 4 (two groups per XCD at B = 64: 32 workgroups, the XCD's 32 CUs).
// Placement is a speed matter only, as above.
struct PersistRole { int rid, pid; };     // rid >= 0: recurrence workgroup rid; else pid >= 0: passenger pid; else: nothing to do
__device__ __forceinline__ PersistRole persist_role(int xcd_map, int nrec, int np) {
  const int id = (int)blockIdx.x;
  if (xcd_map & 4) {
    const int xr = id & 7, ch = id >> 3;
    if (xr < 4) return PersistRole{ch * 4 + xr < nrec ? ch * 4 + xr : -1, -1};
    const int p = ch * 4 + (xr - 4);
    return PersistRole{-1, p < np ? p : -1};
  }
  if (id < nrec) return PersistRole{id, -1};
  return PersistRole{-1, id - nrec < np ? id - nrec : -1};
}

__device__ __forceinline__ void group_wait(unsigned* flags, int njb, unsigned epoch, unsigned* status, unsigned* sticky, int* s_abort) {
  if (threadIdx.x < 64 && !*s_abort) {
    const int lane = threadIdx.x;
    unsigned spins = 0;
    for (;;) {
#if VLN_SYNC_FLAGS
      const unsigned v = (lane < njb) ? VLN_AGENT_LOAD(flags + lane) : epoch;
      if (__all(v >= epoch)) break;
#else
      const unsigned v = VLN_AGENT_LOAD(flags);
      if (v >= epoch * (unsigned)njb) break;
#endif
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 24)) {            // ~1 s: a workgroup of the group is not resident / died
        if (lane == 0) {
          VLN_AGENT_STORE(status, 1u);           // this launch (zeroed with the header before the next one)
          __hip_atomic_fetch_add(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // sticky: host-mapped, never zeroed by a launch
          *s_abort = 1;
        }
        break;
      }
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void group_arrive(unsigned* flags, int jb, unsigned epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
  __syncthreads();
#if VLN_SYNC_FLAGS
  if (threadIdx.x == 0) VLN_AGENT_STORE(flags + jb, epoch);
#else
  if (threadIdx.x == 0) __hip_atomic_fetch_add(flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// resident weight fragments of one wave: NS K-steps x 32 bytes per lane
template <typename TW, int NS> struct WFrag;
template <int NS> struct WFrag<float, NS> { float4 v[NS][2]; };
template <int NS> struct WFrag<bf16_raw, NS> { bf16x8 v[NS][2]; };

template <typename TW, int NS>
__device__ __forceinline__ void load_wfrag(WFrag<TW, NS>& w, const TW* wrow, int fq) {
  constexpr int BK = RecCfg<TW>::BK, VK = RecCfg<TW>::VK;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if constexpr (sizeof(TW) == 4) {
      w.v[s][0] = *reinterpret_cast<const float4*>(wrow + s * BK + fq * VK);
      w.v[s][1] = *reinterpret_cast<const float4*>(wrow + s * BK + fq * VK + 4);
    } else {
      w.v[s][0] = *reinterpret_cast<const bf16x8*>(wrow + s * BK + fq * VK);
      w.v[s][1] = *reinterpret_cast<const bf16x8*>(wrow + s * BK + fq * VK + 8);
    }
  }
}

// acc += A[16 rows, NS K-steps] (fp32 rows in LDS, `arow` = this lane's row) * resident W fragments
template <typename TW, int NS>
__device__ __forceinline__ void mfma_resident(const float* arow, const WFrag<TW, NS>& w, int fq, f32x4& acc) {
  constexpr int BK = RecCfg<TW>::BK, VK = RecCfg<TW>::VK;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const float* p = arow + s * BK + fq * VK;
    if constexpr (sizeof(TW) == 4) {
      const float4 a0 = *reinterpret_cast<const float4*>(p), a1 = *reinterpret_cast<const float4*>(p + 4);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, w.v[s][0].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, w.v[s][0].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, w.v[s][0].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, w.v[s][0].w, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, w.v[s][1].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, w.v[s][1].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, w.v[s][1].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, w.v[s][1].w, acc, 0, 0, 0);
    } else {
      float x[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(p + q * 4);
        x[q * 4] = t.x; x[q * 4 + 1] = t.y; x[q * 4 + 2] = t.z; x[q * 4 + 3] = t.w;
      }
      bf16x8 a0, a1, l0, l1;   // activations split hi + lo: only the weight stream is quantised
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a0[j] = (__bf16)x[j];
        a1[j] = (__bf16)x[8 + j];
        l0[j] = (__bf16)(x[j] - (float)a0[j]);
        l1[j] = (__bf16)(x[8 + j] - (float)a1[j]);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0, w.v[s][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, w.v[s][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w.v[s][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, w.v[s][1], acc, 0, 0, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// forward: NS = Hd / BK
// ---------------------------------------------------------------------------------------------------------
template <typename TW, int NS>
__global__ __launch_bounds__(256) void lstm_persist_fwd_kernel(RecFwdArgs a, unsigned* counters, unsigned* status, unsigned* sticky, int xcd_map) {
  constexpr int HD = NS * RecCfg<TW>::BK;
  constexpr int LDH = HD + 4;
  __shared__ __attribute__((aligned(16))) float sh[16 * LDH];
  __shared__ float sg[4][16][17];
  __shared__ int s_abort;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 15, fq = lane >> 4;
  const PersistIdx ix = persist_index(HD / 16, a.dirs, (a.B + 15) / 16, xcd_map, (int)blockIdx.x);
  const int j0 = ix.jb * 16, d = ix.d, b0 = ix.bb * 16;
  const int B = a.B, L = a.L;
  const int njb = HD / 16;
  const int G = a.dirs * 4 * HD, Y = a.dirs * HD;
  unsigned* cnt = counters + (d * ix.nbb + ix.bb) * 32;      // this group's flag line

  WFrag<TW, NS> w;
  load_wfrag<TW, NS>(w, reinterpret_cast<const TW*>(a.w_hh) + ((long)d * 4 * HD + (long)wave * HD + j0 + fi) * HD, fq);

  const int bl = threadIdx.x >> 4, jl = threadIdx.x & 15;
  const int b = b0 + bl, j = j0 + jl;
  const bool live = b < B;
  const int len = live ? a.lengths[b] : 0;
  float hreg = 0.f, creg = 0.f;
  if (a.init && live) {      // caller-given initial state, already copied into the first time slot (plain loads)
    const long s0 = (((long)d * L + (d == 0 ? 0 : L - 1)) * B + b) * HD + j;
    hreg = a.hprev[s0]; creg = a.cprev[s0];
  }
  __amdgpu_buffer_rsrc_t hres = __builtin_amdgcn_make_buffer_rsrc(a.hprev, 0, (unsigned)((long)a.dirs * L * B * HD * 4), 0x00020000);
  __builtin_amdgcn_s_setprio(3);   // latency-critical chain: win issue arbitration against co-resident streaming work
  if (threadIdx.x == 0) s_abort = 0;
  __syncthreads();

  for (int step = 0; step < L; ++step) {
    const int t = (d == 0) ? step : (L - 1 - step);
    const long sbase = ((long)d * L + t) * B;
    const long row = (long)t * B + (live ? b : 0);
    float xi = 0.f, xf = 0.f, xg = 0.f, xo = 0.f;
    if (live) {   // plain loads: written by the projection GEMM before this launch
      const float* xp = a.xproj + row * G + (long)d * 4 * HD + j;
      xi = xp[0]; xf = xp[HD]; xg = xp[2 * HD]; xo = xp[3 * HD];
    }
    VLN_STAMP(0);
    if (step > 0 || a.init) {
      if (step > 0) group_wait(cnt, njb, (unsigned)step, status, sticky, &s_abort);
      VLN_STAMP(1);
      // h tile [16 rows x HD] of time t, written by the group's workgroups in the previous step
#pragma unroll
      for (int u = 0; u < HD / 64; ++u) {
        const int unit = threadIdx.x + u * 256;
        const int r = unit / (HD / 4), c4 = unit % (HD / 4);
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (b0 + r < B) v = __builtin_amdgcn_raw_buffer_load_b128(hres, (unsigned)(((sbase + b0 + r) * HD + c4 * 4) * 4), 0, 16);
        *reinterpret_cast<u32x4_t*>(&sh[r * LDH + c4 * 4]) = v;
      }
    } else {
      for (int i = threadIdx.x; i < 16 * LDH; i += 256) sh[i] = 0.f;
    }
    __syncthreads();
    VLN_STAMP(2);
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      mfma_resident<TW, NS>(&sh[fi * LDH], w, fq, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) sg[wave][fq * 4 + r][fi] = acc[r];
    }
    __syncthreads();
    VLN_STAMP(3);
    float si = 0.f, sf = 0.f, tg = 0.f, so = 0.f, tc = 0.f, yv = 0.f, c_in = creg;
    if (live) {
      const float pi = sg[0][bl][jl] + xi, pf = sg[1][bl][jl] + xf, pg = sg[2][bl][jl] + xg, po = sg[3][bl][jl] + xo;
      const bool valid = t < len;
      const LstmCellPw cw = lstm_cell_pw(pi, pf, pg, po, creg);
      si = cw.si; sf = cw.sf; tg = cw.tg; so = cw.so; tc = cw.tc;
      const float cn = cw.cn, hn = cw.hn;
      yv = valid ? hn : 0.f;
      const float hs = valid ? hn : hreg, cs = valid ? cn : creg;
      const int tn = (d == 0) ? t + 1 : t - 1;
      if (tn >= 0 && tn < L) {
        VLN_AGENT_STORE(a.hprev + (((long)d * L + tn) * B + b) * HD + j, hs);   // write-through: crosses workgroups
      } else {
        a.hcat[(long)b * Y + d * HD + j] = hs;
        a.ccat[(long)b * Y + d * HD + j] = cs;
      }
      hreg = hs; creg = cs;
    }
    VLN_STAMP(4);
#if VLN_FWD_LATE_STORES
    group_arrive(cnt, ix.jb, (unsigned)step + 1u);
    VLN_STAMP(5);
#endif
    if (live) {   // saved for BPTT / the layer output: read only after this launch
      float* ac = a.act + row * G + (long)d * 4 * HD + j;
      ac[0] = si; ac[HD] = sf; ac[2 * HD] = tg; ac[3 * HD] = so;
      a.tanh_c[row * Y + d * HD + j] = tc;
      a.y[row * Y + d * HD + j] = yv;
      a.cprev[(sbase + b) * HD + j] = c_in;                       // state fed into time t
      if (step == 0 && !a.init) a.hprev[(sbase + b) * HD + j] = 0.f;
    }
#if !VLN_FWD_LATE_STORES
    group_arrive(cnt, ix.jb, (unsigned)step + 1u);
    VLN_STAMP(5);
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------
// backward through time
//
// dh_t[b, j] = sum over the 4*Hd gate columns k of dgates_{t+1}[b, k] * W_hh[k, j].  Workgroup (jb, d, bb) has just
// produced dgates_{t+1} for ITS 64 gate columns (4 gates x 16 units), so it contracts those 64 columns against its
// resident [64, Hd] slice of W_hh and publishes a PARTIAL dh for all Hd units; the owner of units j0..j0+15 sums the
// njb partials.  Per step a workgroup then reads njb x 1 KB = Hd*4 bytes (like the forward kernel) instead of the
// full [16, 4*Hd] dgates rows (4x more: measured 4.5 us of an 8.7 us step, scripts/lstm_probe.hip).
// Exchange buffer (sync_ws + kSyncHeaderBytes): [d][bb][parity][producer jb][unit Hd][row 16] fp32; a lane's MFMA result (4
// consecutive rows of one unit) is ONE 16-byte write-through store, a consumer wave fetches one producer's
// [16 units][16 rows] block with ONE 16-byte sc1 load per lane.  Two parities: a producer can run at most one step
// ahead of the slowest consumer of its group (it needs everybody's arrival for step k before it may write step k+2).
// Summation order is fixed (producers wave, wave+4, ... then waves 0..3): results do not depend on timing.
// ---------------------------------------------------------------------------------------------------------
template <typename TW, int NSK> struct AFrag;
template <int NSK> struct AFrag<float, NSK> { float4 v[NSK][2]; };
template <int NSK> struct AFrag<bf16_raw, NSK> { bf16x8 hi[NSK][2], lo[NSK][2]; };

template <typename TW, int NSK>
__device__ __forceinline__ void load_afrag(AFrag<TW, NSK>& f, const float* arow, int fq) {
  constexpr int BK = RecCfg<TW>::BK, VK = RecCfg<TW>::VK;
#pragma unroll
  for (int s = 0; s < NSK; ++s) {
    const float* p = arow + s * BK + fq * VK;
    if constexpr (sizeof(TW) == 4) {
      f.v[s][0] = *reinterpret_cast<const float4*>(p);
      f.v[s][1] = *reinterpret_cast<const float4*>(p + 4);
    } else {
      float x[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(p + q * 4);
        x[q * 4] = t.x; x[q * 4 + 1] = t.y; x[q * 4 + 2] = t.z; x[q * 4 + 3] = t.w;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {     // activations split hi + lo: only the weight stream is quantised
        f.hi[s][0][j] = (__bf16)x[j];
        f.hi[s][1][j] = (__bf16)x[8 + j];
        f.lo[s][0][j] = (__bf16)(x[j] - (float)f.hi[s][0][j]);
        f.lo[s][1][j] = (__bf16)(x[8 + j] - (float)f.hi[s][1][j]);
      }
    }
  }
}
template <typename TW, int NSK>
__device__ __forceinline__ void mfma_frags(const AFrag<TW, NSK>& f, const WFrag<TW, NSK>& w, f32x4& acc) {
#pragma unroll
  for (int s = 0; s < NSK; ++s) {
    if constexpr (sizeof(TW) == 4) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.v[s][0].x, w.v[s][0].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.v[s][0].y, w.v[s][0].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.v[s][0].z, w.v[s][0].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.v[s][0].w, w.v[s][0].w, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.v[s][1].x, w.v[s][1].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.v[s][1].y, w.v[s][1].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.v[s][1].z, w.v[s][1].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.v[s][1].w, w.v[s][1].w, acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.lo[s][0], w.v[s][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.lo[s][1], w.v[s][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.hi[s][0], w.v[s][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.hi[s][1], w.v[s][1], acc, 0, 0, 0);
    }
  }
}

// floats of the exchange buffer behind the sync header
__host__ __device__ inline long persist_bwd_exchange_floats(int B, int Hd, int dirs) {
  return (long)dirs * ((B + 15) / 16) * 2 * (Hd / 16) * Hd * 16;
}

// barrier among the four EXTRA waves of a recurrence workgroup only (the forward's projection waves, the backward's weight-gradient waves;
// the workgroup barrier also counts the recurrence waves): an LDS counter
__device__ __forceinline__ void proj_waves_barrier(unsigned* cnt, unsigned& gen, int lane) {
  gen += 4u;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // this wave's LDS traffic has completed
  if (lane == 0) (void)__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  for (unsigned spins = 0; __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gen; ) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 22)) break;                               // (a projection wave died: bounded, the results are garbage either way)
  }
}
// WG (round 6; an A/B option that measured SLOWER and is off by default): the layer's OWN weight gradients d W_hh = sum_t dgates_t^T
// h_{t-1}, d W_ih = sum_t dgates_t^T x_t accumulated INSIDE the BPTT launch by FOUR EXTRA WAVES per workgroup (threads 256..511; the
// forward's projection waves are the same idea): a workgroup holds the step's dgates tile [16 rows x 64 gate columns] in LDS anyway; the
// extra waves contract it against the step's h_{t-1} and x_t rows (16 x kWgE each, staged through LDS tiles of their own) on
// v_mfma_f32_16x16x16_bf16 -- plain bf16 operands, fp32 accumulation in 128 registers per lane: the arithmetic of the packed
// contraction in its default precision (results within 5e-7 of it).  At the end every workgroup stores its [64, Hd + kWgE] partial;
// vln_lstm_wgrad_reduce adds the row blocks' partials in order.  It removes the encoder's pack (44 us) and contraction (28 us)
// launches -- and makes the BPTT launch 192 -> 392 us: every workgroup must pull 32 KB of h / x rows per step from HBM through its
// compute unit's vector-memory pipeline, which returns loads in order and keeps ~8 KB in flight; the recurrence waves' own hand-off
// loads queue behind them wherever in the step they are requested (behind the arrival barrier +1.7 us per step, two steps deep +3.5,
// behind the reduction barrier +1.7; without the rows' loads the iteration reads 1.346 ms against 1.361).  The forward's projection
// waves do not have the problem: their 16 KB per step were written a few microseconds earlier and hit in L2.  profiles/round6_notes.md.
constexpr int kWgE = 256;
template <typename TW, int NT, bool WG>   // NT = Hd / 64: output tiles (16 units each) per wave
__global__ __launch_bounds__(WG ? 512 : 256) void lstm_persist_bwd_kernel(RecBwdArgs a, unsigned* counters, unsigned* status, unsigned* sticky, float* exch, int xcd_map,
                                                               int nrec, int np, WgradRideArgs ride) {
  constexpr int HD = NT * 64;
  const PersistRole role = persist_role(xcd_map, nrec, np);
  if (role.rid < 0) {                // passengers (wgrad_ride.h): another module's parameter gradients on the CUs the recurrence leaves idle
    __shared__ float4 ride_part[64][4];
    if (WG && threadIdx.x >= 256) return;      // (the passengers' bodies are written for 256 threads)
    if (role.pid >= 0) wgrad_ride_passenger(ride, role.pid, np, status, sticky, ride_part);
    return;
  }
  constexpr int BK = RecCfg<TW>::BK, VK = RecCfg<TW>::VK;
  constexpr int NSK = 64 / BK;                 // K-steps over this workgroup's 64 gate columns
  constexpr int NJB = HD / 16;                 // producers per dependency group
  constexpr int LDT = 64 + 4;
  __shared__ __attribute__((aligned(16))) float tile[16 * LDT];   // this step's dgates: [row][gate*16 + unit]
  __shared__ __attribute__((aligned(16))) float red[4][64][4];    // per-wave sums of the incoming partial-dh blocks
  __shared__ int s_abort;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 15, fq = lane >> 4;
  // PASSES (round 6, encoder.hip: persist_passes): as in the forward kernel -- `nbb_pass` row blocks per direction in the grid, a
  // workgroup runs the L steps of row blocks bb, bb + nbb_pass, ... one after the other with the same resident W_hh slice.
  const int nbb_all = (a.B + 15) / 16, nbb_pass = a.nbb_per > 0 ? a.nbb_per : nbb_all;
  const PersistIdx ix0 = persist_index(HD / 16, a.dirs, nbb_pass, xcd_map & 1, role.rid);      // (bit 1 of xcd_map: the XCD-local hand-off may be used, see below)
  const int jb = ix0.jb, j0 = jb * 16, d = ix0.d;
  const int B = a.B, L = a.L;
  const int G = a.dirs * 4 * HD, Y = a.dirs * HD;
  // ---- the WEIGHT-GRADIENT waves (4-7) ---------------------------------------------------------------------------------------------
  constexpr int LDW = kWgE + 4;
  __shared__ __attribute__((aligned(16))) float wg_h[WG ? 16 * (HD + 4) : 4];      // h_{t-1} rows of the step (fp32, row-major)
  __shared__ __attribute__((aligned(16))) float wg_x[WG ? 16 * LDW : 4];           // x_t rows of the step
  __shared__ unsigned s_wcnt;
  if constexpr (WG) {
    if (threadIdx.x >= 256) {
      static_assert(!WG || HD == 256, "the in-launch weight gradients are instantiated for Hd = 256");
      constexpr int LDHW = HD + 4;
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      const int wtid = (int)threadIdx.x - 256, ww = wtid >> 6;          // wave ww: input-unit tiles ww * 4 .. ww * 4 + 3 of both matrices
      __builtin_amdgcn_s_setprio(0);
      f32x4 acc_h[4][4], acc_x[4][4];                                     // [gate m-tile][n-tile of this wave]
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) { acc_h[m][n] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_x[m][n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      unsigned wgen = 0;
      for (int bb = ix0.bb; bb < nbb_all; bb += nbb_pass) {
        const bool first_pass = bb == ix0.bb;
        if (!first_pass) __syncthreads();
        const int b0 = bb * 16;
        if (wtid == 0 && first_pass) s_wcnt = 0u;
        __syncthreads();                            // (the recurrence waves' barrier behind their s_abort / XCC_ID set-up)
        // the step's h_{t-1} / x_t rows (thread -> 4 float4 of each), requested a step ahead -- and WHEN matters more than how far
        // ahead: a compute unit's vector-memory pipeline returns loads in order, so these 32 KB must not be in it while the
        // recurrence waves poll their group's counter and fetch the partial sums (requested behind the arrival barrier they cost
        // 1.7 us per step, two steps deep 3.5 us); they are requested right behind the recurrence waves' reduction barrier, when
        // those waves have their exchange data and go on to arithmetic and stores
        float4 hr[4], xr[4], hr2[4], xr2[4];
        auto load_rows = [&](float4 (&hd)[4], float4 (&xd)[4], int step_) {
          const int t_ = (d == 0) ? step_ : (L - 1 - step_);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int unit = wtid + u * 256;
            const int r = unit / 64, c4 = unit % 64;                       // HD / 4 = kWgE / 4 = 64 float4 per row
            const int br = min(b0 + r, B - 1);                            // rows past the batch: a valid row (their dgates are zero)
            hd[u] = *reinterpret_cast<const float4*>(a.hprev + (((long)d * L + t_) * B + br) * HD + c4 * 4);
            xd[u] = *reinterpret_cast<const float4*>(a.x + ((long)t_ * B + br) * kWgE + c4 * 4);
          }
        };
        load_rows(hr, xr, L - 1);
        for (int step = L - 1; step >= 0; --step) {
          const int k = L - 1 - step;
          if (k > 0) {                              // the recurrence waves' group_wait barrier and their reduction barrier
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          }
          if (step > 0) load_rows(hr2, xr2, step - 1);
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the step's dgates tile is complete
          // A fragments: A[m = gate column][k = row] = tile[row][column]: lane (m = lane & 15, rows (lane >> 4) * 4 .. + 3)
          bf16x4_t af[4];
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) af[m][r] = (__bf16)tile[(fq * 4 + r) * LDT + m * 16 + fi];
          if (step > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the recurrence waves' arrival barrier (their last step has none)
          // this step's rows -> the LDS tiles of these waves, the next step's requested
          proj_waves_barrier(&s_wcnt, wgen, lane);                           // every wave has finished reading the previous tiles
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int unit = wtid + u * 256;
            const int r = unit / 64, c4 = unit % 64;
            *reinterpret_cast<float4*>(&wg_h[r * LDHW + c4 * 4]) = hr[u];
            *reinterpret_cast<float4*>(&wg_x[r * LDW + c4 * 4]) = xr[u];
          }
          proj_waves_barrier(&s_wcnt, wgen, lane);                           // the tiles are complete
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const int col = (ww * 4 + n) * 16 + fi;                          // B[k = row][n = input unit]: rows (lane >> 4) * 4 .. + 3
            bf16x4_t bh, bx;
#pragma unroll
            for (int r = 0; r < 4; ++r) { bh[r] = (__bf16)wg_h[(fq * 4 + r) * LDHW + col]; bx[r] = (__bf16)wg_x[(fq * 4 + r) * LDW + col]; }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              acc_h[m][n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af[m], bh, acc_h[m][n], 0, 0, 0);
              acc_x[m][n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af[m], bx, acc_x[m][n], 0, 0, 0);
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) { hr[u] = hr2[u]; xr[u] = xr2[u]; }     // the next step's rows (requested above)
        }
        if (a.bias_part) {                          // the recurrence waves' two barriers around the bias sums
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
      }
      // the workgroup's partial: [d][first row block][gate * HD + j0 + m][HD + kWgE]; C layout: column n = lane & 15, rows (lane >> 4) * 4 + r
      float* pp = a.wg_part + ((long)(d * nbb_pass + ix0.bb) * 4 * HD) * (HD + kWgE);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* row = pp + ((long)m * HD + j0 + fq * 4 + r) * (HD + kWgE);
            row[(ww * 4 + n) * 16 + fi] = acc_h[m][n][r];
            row[HD + (ww * 4 + n) * 16 + fi] = acc_x[m][n][r];
          }
      return;
    }
  }

  // resident slice of W_hh: rows = this workgroup's 64 gate columns, all HD input units; read from the transposed
  // shadow [unit n][4*HD] where each gate's 16 columns are contiguous
  WFrag<TW, NSK> w[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int n = (wave * NT + i) * 16 + fi;
    const TW* wrow = reinterpret_cast<const TW*>(a.w_hh_t) + ((long)d * HD + n) * 4 * HD + j0;
#pragma unroll
    for (int s = 0; s < NSK; ++s) {
      const int kk0 = s * BK + fq * VK;        // first of this lane's VK consecutive k (never straddles a gate)
      const TW* p = wrow + (long)(kk0 / 16) * HD + (kk0 % 16);
      if constexpr (sizeof(TW) == 4) {
        w[i].v[s][0] = *reinterpret_cast<const float4*>(p);
        w[i].v[s][1] = *reinterpret_cast<const float4*>(p + 4);
      } else {
        w[i].v[s][0] = *reinterpret_cast<const bf16x8*>(p);
        w[i].v[s][1] = *reinterpret_cast<const bf16x8*>(p + 8);
      }
    }
  }

  const int bl = threadIdx.x >> 4, jl = threadIdx.x & 15;
  const int j = j0 + jl;
  for (int bb = ix0.bb; bb < nbb_all; bb += nbb_pass) {
  const bool first_pass = bb == ix0.bb;
  if (!first_pass) __syncthreads();            // the dgates tile / reduction buffer of the previous pass are free
  const PersistIdx ix{jb, d, bb, nbb_all};
  const int b0 = bb * 16;
  unsigned* cnt = counters + (d * ix.nbb + ix.bb) * 32;      // this group's flag line
  const int b = b0 + bl;
  const bool live = b < B;
  const int len = live ? a.lengths[b] : 0;
  const long ci = ((long)d * B + (live ? b : 0)) * HD + j;
  float dh_pass = 0.f, dcc = 0.f;
  if (live) {
    if (a.dh_bm) {                       // the caller's [B, dirs*Hd] layout
      const long cb = ((long)b * a.dirs + d) * HD + j;
      dh_pass = a.dh_bm[cb]; dcc = a.dc_bm[cb];
    } else {
      dh_pass = a.dh_pass[ci]; dcc = a.dc_carry[ci];
    }
  }
  const long grp = (long)(d * ix.nbb + ix.bb) * 2;          // [grp + parity][producer][unit][row]
  __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
      exch, 0, (unsigned)(persist_bwd_exchange_floats(B, HD, a.dirs) * 4), 0x00020000);
  __builtin_amdgcn_s_setprio(3);   // latency-critical chain: win issue arbitration against co-resident streaming work
  if (threadIdx.x == 0 && first_pass) {
    s_abort = 0;                                          // (a timeout of an earlier pass keeps later passes from spinning again)
    if (role.rid == 0) VLN_AGENT_STORE(status, 0u);       // this launch's status word (a timeout sets it long after this store)
  }
  // XCD-LOCAL HAND-OFF (round 5).  The partial products are exchanged through memory: an `sc1` (write-through) store leaves the
  // writer's L2, and a reader on the SAME XCD then fetches the line at the cross-XCD rate (MI355X_MICROARCH.md, stores of each
  // flavour).  A plain store keeps the line in the XCD's L2, where a same-XCD reader's L1-bypassing load finds it: 2.28 -> 1.90 us
  // per step (scripts/lstm_probe).  That is only CORRECT when every workgroup of the dependency group runs on one XCD (the L2s are
  // not coherent with each other) -- which the block -> role mapping aims at but HIP does not promise.  So it is verified per
  // launch: every workgroup ORs its XCC_ID bit into a word of the group's flag line BEFORE its first arrival; after its first
  // wait a workgroup reads the word -- every member's bit is in by then -- and only if exactly one bit is set do the stores from
  // the second processed step on go plain.  Step one and any group that spans XCDs keep the write-through stores.  The word is
  // reset with the counters by the last workgroup through.
  bool xcd_local = false;
#if !VLN_SYNC_FLAGS
  if (threadIdx.x == 0 && (xcd_map & 2)) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    (void)__hip_atomic_fetch_or(cnt + 8, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // returns: performed before the arrival below
  }
#endif
  __syncthreads();

  float bs0 = 0.f, bs1 = 0.f, bs2 = 0.f, bs3 = 0.f;    // this (row, unit)'s dgates summed over the steps: the bias gradients' partial sums
  // The step's saved operands (gate activations, tanh(c), c_{t-1}, the upstream gradient: produced before this launch, plain loads)
  // do not depend on the recurrence, and their ~0.4 us of load latency sat at the top of EVERY workgroup's step, i.e. on the ring's
  // critical path (scripts/lstm_probe "loop-around").  Step s - 1's are requested in the middle of step s (after the hand-off's
  // loads, before the pointwise arithmetic) and have landed when the loop comes around.
  struct StepOps { float dyv, si, sf, tg, so, tc, cp; };
  auto load_ops = [&](int step_) {
    // UNCONDITIONAL loads (rows past the batch clamp to row 0, padded steps read what the forward stored there): a per-thread
    // branch around them would make the compiler close it with s_waitcnt vmcnt(0) and expose their latency right here; what a
    // padded step / row loaded is never used (`valid` below)
    StepOps o;
    const int t_ = (d == 0) ? step_ : (L - 1 - step_);
    const int b_ = live ? b : 0;
    const long row_ = (long)t_ * B + b_;
    o.dyv = a.dy ? a.dy[row_ * Y + d * HD + j] : 0.f;
    const float* ac = a.act + row_ * G + (long)d * 4 * HD + j;
    o.si = ac[0]; o.sf = ac[HD]; o.tg = ac[2 * HD]; o.so = ac[3 * HD];
    o.tc = a.tanh_c[row_ * Y + d * HD + j];
    o.cp = a.cprev[(((long)d * L + t_) * B + b_) * HD + j];
    return o;
  };
  StepOps nxt = load_ops(L - 1);
  for (int step = L - 1; step >= 0; --step) {
    const int k = L - 1 - step;                // steps already processed
    const int t = (d == 0) ? step : (L - 1 - step);
    const long row = (long)t * B + (live ? b : 0);
    const bool valid = live && (t < len);
    const float dyv = nxt.dyv, si = nxt.si, sf = nxt.sf, tg = nxt.tg, so = nxt.so, tc = nxt.tc, cp = nxt.cp;
    VLN_STAMP(0);
    float dh = dh_pass;
    if (k > 0) {
      group_wait(cnt, NJB, (unsigned)k, status, sticky, &s_abort);
#if !VLN_SYNC_FLAGS
      if (k == 1 && (xcd_map & 2)) {           // every member's XCC_ID bit is in (each ORed it before its first arrival)
        const unsigned m = VLN_AGENT_LOAD(cnt + 8);
        xcd_local = m != 0u && (m & (m - 1u)) == 0u;
        // cumulative tallies for the tests (vln_lstm_handoff_stats): groups that went XCD-local / that span XCDs, status line words 8, 9
        if (threadIdx.x == 0 && jb == 0) __hip_atomic_fetch_add(status + (xcd_local ? 8 : 9), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#endif
      VLN_STAMP(1);
      const unsigned rbase = (unsigned)(((grp + ((k - 1) & 1)) * NJB * HD + j0) * 16 * 4) + (unsigned)lane * 16u;
      float4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NJB / 4; ++i) {      // one producer's [16 units][16 rows] block per wave-wide load
        const unsigned p = (unsigned)(wave + 4 * i);
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(xres, rbase + p * (unsigned)(HD * 16 * 4), 0, 16);
        s4.x += __uint_as_float(v.x); s4.y += __uint_as_float(v.y); s4.z += __uint_as_float(v.z); s4.w += __uint_as_float(v.w);
      }
      *reinterpret_cast<float4*>(&red[wave][lane][0]) = s4;
      __syncthreads();
      const int q = jl * 4 + (bl >> 2), e = bl & 3;
      dh += (red[0][q][e] + red[1][q][e]) + (red[2][q][e] + red[3][q][e]);
    }
    VLN_STAMP(2);
    if (step > 0) nxt = load_ops(step - 1);          // the next step's operands: in flight under the rest of this step
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    if (valid) {
      dh += dyv;
      const float dc = dcc + dh * so * (1.f - tc * tc);
      g0 = dc * tg * si * (1.f - si);
      g1 = dc * cp * sf * (1.f - sf);
      g2 = dc * si * (1.f - tg * tg);
      g3 = dh * tc * so * (1.f - so);
      dcc = dc * sf;
      dh_pass = 0.f;
    } else {
      dh_pass = dh;
    }
    if (live) {    // consumed by the weight-gradient GEMMs after this launch: plain stores
      float* dg = a.dgates + row * G + (long)d * 4 * HD + j;
      dg[0] = g0; dg[HD] = g1; dg[2 * HD] = g2; dg[3 * HD] = g3;
    }
    bs0 += g0; bs1 += g1; bs2 += g2; bs3 += g3;     // (zeros at padded steps and rows, like the stored dgates)
    if (step == 0 && !WG) break;               // nobody consumes a partial dh of the last processed step
    float* tr = &tile[bl * LDT + jl];
    tr[0] = g0; tr[16] = g1; tr[32] = g2; tr[48] = g3;
    // LDS-only barrier: __syncthreads() would also drain the vector-memory counter, i.e. wait right here for the next step's
    // operand loads requested above (and for the dgates stores, which nobody in this launch reads)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (step == 0) break;                      // (WG: the weight-gradient waves read the last step's tile too)
    VLN_STAMP(3);
    {
      AFrag<TW, NSK> af;
      load_afrag<TW, NSK>(af, &tile[fi * LDT], fq);
      const unsigned wbase = (unsigned)(((grp + (k & 1)) * NJB + jb) * HD * 16 * 4);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mfma_frags<TW, NSK>(af, w[i], acc);
        const int n = (wave * NT + i) * 16 + fi;
        u32x4_t o = {__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3])};
        if (xcd_local) __builtin_amdgcn_raw_buffer_store_b128(o, xres, wbase + (unsigned)((n * 16 + fq * 4) * 4), 0, 0);     // stays in this XCD's L2
        else __builtin_amdgcn_raw_buffer_store_b128(o, xres, wbase + (unsigned)((n * 16 + fq * 4) * 4), 0, 16);              // sc1
      }
    }
    VLN_STAMP(4);
    group_arrive(cnt, jb, (unsigned)k + 1u);
    VLN_STAMP(5);
  }
  if (a.bias_part) {
    // the 16 rows' sums through the (now free) dgates tile, added in row order: [d][batch block][gate * HD + unit]
    __syncthreads();
    float* tr = &tile[bl * LDT + jl];
    tr[0] = bs0; tr[16] = bs1; tr[32] = bs2; tr[48] = bs3;
    __syncthreads();
    if (threadIdx.x < 64) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += tile[r * LDT + threadIdx.x];
      const int gate = threadIdx.x >> 4, u = threadIdx.x & 15;
      a.bias_part[((long)(d * ix.nbb + ix.bb) * 4 + gate) * HD + j0 + u] = t;
    }
  }
#if !VLN_SYNC_FLAGS
  // Leave the group's counter as the launch found it (zero): every workgroup of the group has passed its LAST wait on it by
  // the time it gets here, so the last one through resets it -- the host's fill launch in front of every backward
  // recurrence (4 us on the dependent chain) is not needed.  (A launch that timed out leaves garbage: it also raises the
  // sticky error, after which the library stops using these kernels.)
  if (threadIdx.x == 0) {
    // acq_rel: this workgroup's last arrival on `cnt` is ordered before its pass through here, and the last one through
    // sees every other workgroup's; the resets are release stores at agent scope (once per launch: off the step chain)
    const unsigned through = __hip_atomic_fetch_add(cnt + 24, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (through + 1u == (unsigned)NJB) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(cnt + 8, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);       // the XCC_ID mask of the XCD-local hand-off
      __hip_atomic_store(cnt + 24, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#endif
  }   // next pass
}
