// Persistent packed-(bi)LSTM recurrence, hand-off by DATA-TAGGED GRANULES (included by encoder.hip after encoder_persist.h,
// whose geometry, resident-weight fragments and MFMA helpers it shares).
//
// Round 1's protocol (encoder_persist.h: write-through payload -> every wave's vmcnt(0) -> workgroup barrier -> one atomic add on
// the group's counter; consumer: poll the counter -> barrier -> sc1 reload of the payload) cost 3.3 us (forward) / 2.7 us
// (backward) per time step: a hand-off with a separate flag is two dependent fabric round trips plus two workgroup barriers.
// Here every handed-off value travels as ONE naturally aligned 8-byte granule {fp32 value, 32-bit tag} (MI355X_MICROARCH.md,
// "handoff-1to1" / Guideline 16 form R2): the tag says which launch and which time step the value belongs to, so
//   producer:  two granules per lane-pair as ONE 16-byte write-through (sc1) store -- a 128-byte line per row is written
//              whole by one store instruction of one wave; NO vmcnt drain, NO barrier, NO counter;
//   consumer:  every wave sweeps ITS share of the tile with 16-byte sc1 loads (buffer_load_dwordx4: bypasses this CU's L1) and
//              repeats the sweep until every tag carries this step's value; the data it needs arrived with the tag.
// An 8-byte granule is written by one store and read by one load, so a reader sees either the old or the new {value, tag} --
// no ordering between granules is needed.  Two parity slots per group: a producer can only be one step ahead of the slowest
// consumer of its group (it needs every member's step-s data before it can publish step s+1, which reuses the slot of s-1).
// Tags: (per-buffer launch sequence << 8) | (step + 1); the host hands every launch of a buffer a new sequence number and
// clears the buffer when the 24-bit sequence wraps, so a stale granule can never carry a live tag.  L <= 255 on this path.
// Device-sequence form (vln_lstm_seq_fwd(device_seq >= 0)): the sequence is a 32-bit word at the end of sync_ws + the
// launch's small relative index; the caller bumps the word between iterations (vln_tick) and clears the granule regions
// before the 24-bit value wraps (runtime.DeviceClock does both).
// Spins are bounded; a timeout sets the status words (encoder.hip: vln_persistent_check) and lets the kernel drain.
// Summation orders equal the counter-protocol kernels': results are bit-identical to theirs.
#pragma once
#ifndef VLN_GRAN_ST_AUX
#define VLN_GRAN_ST_AUX 16      // cache policy of the forward hand-off's granule stores: 16 = sc1 (write-through; probe builds may override)
#endif


typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

constexpr unsigned kGranSpinLimit = 1u << 20;     // sweeps of ~1 us: ~1 s

// bytes of the two exchange regions behind the sync header (granules: 8 bytes per handed-off value)
__host__ __device__ inline long persist_g_fwd_bytes(int B, int Hd, int dirs) {
  return (long)dirs * ((B + 15) / 16) * 2 * 16 * Hd * 8;
}
__host__ __device__ inline long persist_g_bwd_bytes(int B, int Hd, int dirs) {
  return (long)dirs * ((B + 15) / 16) * 2 * (Hd / 16) * Hd * 16 * 8;
}

// bf16 weights: the h tile is split hi + lo ONCE by the thread that receives it (two bf16 planes in LDS); the four gate waves
// then read ready-made fragments.  The counter-protocol kernel converts inside every wave's MFMA loop: 4x the conversions, on
// the step's critical path.  Same values, same MFMA order -> same bits.
template <int NS>
__device__ __forceinline__ void mfma_resident_planes(const bf16_raw* hi_row, const bf16_raw* lo_row, const WFrag<bf16_raw, NS>& w,
                                                     int fq, f32x4& acc) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int k = s * 64 + fq * 16;
    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(hi_row + k), a1 = *reinterpret_cast<const bf16x8*>(hi_row + k + 8);
    const bf16x8 l0 = *reinterpret_cast<const bf16x8*>(lo_row + k), l1 = *reinterpret_cast<const bf16x8*>(lo_row + k + 8);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0, w.v[s][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, w.v[s][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w.v[s][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, w.v[s][1], acc, 0, 0, 0);
  }
}
__device__ __forceinline__ void split_store2(bf16_raw* hi, bf16_raw* lo, float x0, float x1) {
  const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
  const __bf16 l0 = (__bf16)(x0 - (float)h0), l1 = (__bf16)(x1 - (float)h1);
  *reinterpret_cast<uint32_t*>(hi) = (uint32_t)__builtin_bit_cast(bf16_raw, h0) | ((uint32_t)__builtin_bit_cast(bf16_raw, h1) << 16);
  *reinterpret_cast<uint32_t*>(lo) = (uint32_t)__builtin_bit_cast(bf16_raw, l0) | ((uint32_t)__builtin_bit_cast(bf16_raw, l1) << 16);
}

__device__ __forceinline__ void gran_timeout(unsigned* status, unsigned* sticky, int* s_abort) {
  VLN_AGENT_STORE(status, 1u);
  __hip_atomic_fetch_add(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // host-mapped word (encoder.hip)
  *s_abort = 1;
}

// ---------------------------------------------------------------------------------------------------------
// forward: NS = Hd / BK.  Exchange layout: [group][parity][row 16][unit HD] granules.
// ---------------------------------------------------------------------------------------------------------
// XP (round 6): the INPUT PROJECTION x_t W_ih^T + (b_ih + b_hh) of every step is formed INSIDE the recurrence launch (E = kInprojE input
// features) instead of by a GEMM launch over all L * B rows in front of it -- by FOUR EXTRA WAVES per workgroup (threads 256..511, one
// per SIMD beside a recurrence wave, lowest issue priority): while waves 0-3 run step s (sweep the neighbours' granules, h W_hh^T,
// pointwise, publish), waves 4-7 form the projection of step s + 1 -- its 16 rows x (4 gates x 16 units), W_ih rows resident in
// registers, the x tile prefetched two steps ahead through LDS planes of their own -- and leave it in a triple-buffered LDS tile that
// the pointwise stage reads two workgroup barriers later.  The two groups share the step's two barriers and nothing else.  (The
// first form of this, the same four waves doing both jobs, was bit-identical and NEUTRAL: the hand-off wait is a load round trip the
// wave itself issues, not idle time -- profiles/round6_notes.md section 6.)  The projection launch (31.8 us at B 64 / L 80) and the
// 42 MB it wrote and the recurrence re-read leave the iteration.  Same MFMA sequence as gemm_nt's (K-steps in order, lo x w before
// hi x w, bias added to the finished sum): bit-identical gate pre-activations.
constexpr int kInprojE = 256;
template <typename TW, int NS, bool XP>
__global__ __launch_bounds__(XP ? 512 : 256) void lstm_persist_g_fwd_kernel(RecFwdArgs a, unsigned* status, unsigned* sticky, unsigned char* exch,
                                                                 unsigned tag_base, int xcd_map, const unsigned* seq_dev, unsigned seq_rel,
                                                                 int nrec, int np_, GatherRolloutArgs ride, FetchPart fetch, RideShadows shadows) {
  // PASSENGERS: the workgroups that are not the recurrence's own `nrec` (persist_role, encoder_persist.h: past them, or -- partitioned
  // form -- the ones on XCDs 4-7) gather the rollout's feature rows (gather_body.h) on the compute units the recurrence leaves idle -- independent work (it reads the resident table and index vectors only), nothing waits
  // for it inside this launch, and the launch claims a whole CU's LDS per workgroup so that a passenger never shares a CU with
  // a recurrence workgroup (whose hand-off latency is what a co-resident streaming wave would cost, MI355X_MICROARCH.md
  // "handoff-1to1").  Rows are dealt with the passengers' stride: any number of resident passengers finishes the job.
  const PersistRole role = persist_role(xcd_map, nrec, np_);
  if (role.rid < 0) {
    int p = role.pid, np = np_;
    if (p < 0) return;
    if (XP && threadIdx.x >= 256) return;          // the passengers' bodies are written for 256 threads
    if (fetch.on) {      // the LAST passenger pulls the tail of the batch blob out of pinned host memory (PCIe-bound, ~25 us) instead of gathering
      if (p == np - 1) { host_fetch_part_body(fetch, (int)threadIdx.x); return; }
      np -= 1;
    }
    gather_ride_passenger(ride, p, np, (int)threadIdx.x);
    // then the weight shadows the ride carries (shadow_bodies.h), tiles dealt like the rows; the tile buffer is the launch's
    // dynamic LDS claim (unused otherwise: it only keeps the launch at one workgroup per CU)
    if (shadows.tiles > 0) {
      extern __shared__ __attribute__((aligned(16))) unsigned char ride_lds[];
      float (*lds)[65] = reinterpret_cast<float (*)[65]>(ride_lds);
      for (int b = p; b < shadows.tiles; b += np) {
        shadow_block<true>(shadows.jobs, b, lds);
        __syncthreads();                       // the tile buffer is written again by the next tile
      }
    }
    return;
  }
  // launch sequence in DEVICE memory (whole-iteration graphs: the launch arguments must repeat): the word is bumped by a
  // stream-ordered tick launch between iterations (vln_tick), never while a launch that reads it is in flight
  if (seq_dev) tag_base = ((*seq_dev + seq_rel) & 0xFFFFFFu) << 8;
  constexpr int HD = NS * RecCfg<TW>::BK;
  constexpr int LDH = HD + 4;
  constexpr int NLD = HD / 32;                 // 16-byte loads per thread per sweep: 16 rows x HD granules x 8 B / (256 x 16 B)
  constexpr bool kPlanes = sizeof(TW) == 2;    // bf16 weights: the tile lives in LDS as two bf16 planes (hi, lo)
  constexpr int LDP = HD + 8;                  // plane row stride in bf16: 2 HD + 16 bytes, conflict-free 16-byte fragment reads
  constexpr int kTileBytes = kPlanes ? 2 * 16 * LDP * 2 : 16 * LDH * 4;
  __shared__ __attribute__((aligned(16))) unsigned char tile_raw[kTileBytes];
  float* const sh = reinterpret_cast<float*>(tile_raw);
  bf16_raw* const ph = reinterpret_cast<bf16_raw*>(tile_raw);           // hi plane; lo plane follows
  bf16_raw* const pl = ph + 16 * LDP;
  __shared__ float sg[4][16][17];
  __shared__ int s_abort;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 15, fq = lane >> 4;
  // PASSES (round 6, encoder.hip: persist_passes): the grid holds `nbb_pass` row blocks per direction; a workgroup that has finished
  // the L steps of row block bb starts over on row block bb + nbb_pass.  Rows never interact, every (direction, row block) group has
  // its own exchange region and flag line, and the resident W_hh fragments are loaded once.
  const int nbb_all = (a.B + 15) / 16, nbb_pass = a.nbb_per > 0 ? a.nbb_per : nbb_all;
  const PersistIdx ix0 = persist_index(HD / 16, a.dirs, nbb_pass, xcd_map & 1, role.rid);
  const int j0 = ix0.jb * 16, d = ix0.d;
  const int B = a.B, L = a.L;
  const int G = a.dirs * 4 * HD, Y = a.dirs * HD;

  const int bl = (threadIdx.x & 255) >> 4, jl = threadIdx.x & 15;
  const int j = j0 + jl;
  // in-kernel input projection (XP): LDS shared by the two wave groups -- the x tile (projection waves only), the finished projection
  // of three consecutive steps, the projection waves' own barrier counter
  constexpr int NSX = XP ? kInprojE / RecCfg<TW>::BK : 1;
  constexpr int LDXH = kInprojE + 4, LDXP = kInprojE + 8;
  constexpr int kXTileBytes = XP ? (kPlanes ? 2 * 16 * LDXP * 2 : 16 * LDXH * 4) : 16;
  __shared__ __attribute__((aligned(16))) unsigned char xtile_raw[kXTileBytes];
  __shared__ float sgx[XP ? 3 : 1][4][16][17];
  __shared__ unsigned s_pcnt;
  float bx_i = 0.f, bx_f = 0.f, bx_g = 0.f, bx_o = 0.f;
  if constexpr (XP) {
    if (threadIdx.x >= 256) {
      // ---- the PROJECTION waves (4-7): wave pw forms gate pw's 16 columns; same barrier sequence as the recurrence waves below ----
      const int ptid = (int)threadIdx.x - 256, pw = ptid >> 6;
      float* const xsh = reinterpret_cast<float*>(xtile_raw);
      bf16_raw* const xph = reinterpret_cast<bf16_raw*>(xtile_raw);
      bf16_raw* const xpl = xph + 16 * LDXP;
      WFrag<TW, NSX> wx;
      load_wfrag<TW, NSX>(wx, reinterpret_cast<const TW*>(a.w_ih) + ((long)d * 4 * HD + (long)pw * HD + j0 + fi) * kInprojE, fq);
      __builtin_amdgcn_s_setprio(0);             // never ahead of the recurrence wave on the same SIMD
      unsigned pgen = 0;
      for (int bb = ix0.bb; bb < nbb_all; bb += nbb_pass) {
        const bool first_pass = bb == ix0.bb;
        if (!first_pass) __syncthreads();
        const int b0 = bb * 16;
        if (ptid == 0 && first_pass) s_pcnt = 0u;
        __syncthreads();                          // (the recurrence waves' barrier behind their s_abort / XCC_ID set-up)
        float4 xr[4], xr2[4];
        auto load_xtile = [&](float4 (&dst)[4], int step_) {
          const int t_ = (d == 0) ? step_ : (L - 1 - step_);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int unit = ptid + u * 256;
            const int r = unit / (kInprojE / 4), c4 = unit % (kInprojE / 4);
            const int br = min(b0 + r, B - 1);     // rows past the batch: any valid row (their results are never used)
            dst[u] = *reinterpret_cast<const float4*>(a.x + ((long)t_ * B + br) * kInprojE + c4 * 4);
          }
        };
        auto project = [&](int step_) {           // tile of `step_` (in xr) -> sgx[step_ % 3]; then xr <- xr2, xr2 <- tile step_ + 2
          proj_waves_barrier(&s_pcnt, pgen, lane);               // every projection wave has finished reading the previous tile
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int unit = ptid + u * 256;
            const int r = unit / (kInprojE / 4), c4 = unit % (kInprojE / 4);
            if constexpr (kPlanes) {
              split_store2(xph + r * LDXP + c4 * 4, xpl + r * LDXP + c4 * 4, xr[u].x, xr[u].y);
              split_store2(xph + r * LDXP + c4 * 4 + 2, xpl + r * LDXP + c4 * 4 + 2, xr[u].z, xr[u].w);
            } else {
              *reinterpret_cast<float4*>(&xsh[r * LDXH + c4 * 4]) = xr[u];
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) xr[u] = xr2[u];
          if (step_ + 2 < L) load_xtile(xr2, step_ + 2);
          proj_waves_barrier(&s_pcnt, pgen, lane);               // the tile is complete
          f32x4 ax = {0.f, 0.f, 0.f, 0.f};
          if constexpr (kPlanes) mfma_resident_planes<NSX>(xph + fi * LDXP, xpl + fi * LDXP, wx, fq, ax);
          else mfma_resident<TW, NSX>(&xsh[fi * LDXH], wx, fq, ax);
#pragma unroll
          for (int r = 0; r < 4; ++r) sgx[step_ % 3][pw][fq * 4 + r][fi] = ax[r];
        };
        load_xtile(xr, 0);
        if (L > 1) load_xtile(xr2, 1);
        project(0);
        for (int step = 0; step < L; ++step) {
          if (step + 1 < L) project(step + 1);    // read by the pointwise stage of step + 1, two workgroup barriers from here
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the recurrence waves' h-tile barrier (LDS-only: the prefetch stays in flight)
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // ... and their gate barrier
        }
      }
      return;
    }
    const float* bp = a.bsum + (long)d * 4 * HD + j;
    bx_i = bp[0]; bx_f = bp[HD]; bx_g = bp[2 * HD]; bx_o = bp[3 * HD];
  }
  // (recurrence waves only from here: threads 0..255)
  WFrag<TW, NS> w;
  load_wfrag<TW, NS>(w, reinterpret_cast<const TW*>(a.w_hh) + ((long)d * 4 * HD + (long)wave * HD + j0 + fi) * HD, fq);

  for (int bb = ix0.bb; bb < nbb_all; bb += nbb_pass) {
  const bool first_pass = bb == ix0.bb;
  if (!first_pass) __syncthreads();            // the tile / gate buffers of the previous pass are free
  const PersistIdx ix{ix0.jb, d, bb, nbb_all};
  const int b0 = bb * 16;
  const int b = b0 + bl;
  const bool live = b < B;
  const int len = live ? a.lengths[b] : 0;
  float hreg = 0.f, creg = 0.f;
  if (a.init && live) {      // caller-given initial state, already copied into the first time slot (plain loads)
    const long s0 = (((long)d * L + (d == 0 ? 0 : L - 1)) * B + b) * HD + j;
    hreg = a.hprev[s0]; creg = a.cprev[s0];
  }
  const unsigned grp_bytes = 2u * 16u * HD * 8u;
  __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(exch, 0, (unsigned)persist_g_fwd_bytes(B, HD, a.dirs), 0x00020000);
  const unsigned gbase = (unsigned)(d * ix.nbb + ix.bb) * grp_bytes;
  __builtin_amdgcn_s_setprio(3);   // latency-critical chain: win issue arbitration against co-resident streaming work
  // XCD-LOCAL HAND-OFF (round 5, as in the backward recurrence: encoder_persist.h).  A granule stored write-through (`sc1`) leaves
  // the writer's L2, and the group's readers on the SAME XCD fetch it at the cross-XCD rate; a plain store keeps it in that L2.
  // Correct only when the whole dependency group runs on one XCD, so it is verified per launch: every workgroup ORs its XCC_ID bit
  // into word 12 of the group's flag line before its first publish (the barrier below drains the atomic); the sweep of step 1 has
  // seen every member's first granule, hence every member's bit -- exactly one bit set switches the later publishes to plain
  // stores.  The last workgroup of the group through the end of the kernel resets the two words (word 28 counts them).
  // Bit 1 of xcd_map (tunable[14] = 1 clears it: always write-through, A/B).
  unsigned* const gw = status + 32 + (unsigned)(d * ix.nbb + ix.bb) * 32u;
  bool xcd_local = false;
  if (threadIdx.x == 0) {
    if (first_pass) {
      s_abort = 0;                                      // (a timeout of an earlier pass keeps later passes from spinning again)
      if (role.rid == 0) VLN_AGENT_STORE(status, 0u);   // this launch's status (a timeout is >= 1 s away): no fill launch in front
    }
    if (xcd_map & 2) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      (void)__hip_atomic_fetch_or(gw + 12, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();

  for (int step = 0; step < L; ++step) {
    const int t = (d == 0) ? step : (L - 1 - step);
    const long sbase = ((long)d * L + t) * B;
    const long row = (long)t * B + (live ? b : 0);
    float xi = 0.f, xf = 0.f, xg = 0.f, xo = 0.f;
    if constexpr (!XP) {
      if (live) {   // plain loads: written by the projection GEMM before this launch
        const float* xp = a.xproj + row * G + (long)d * 4 * HD + j;
        xi = xp[0]; xf = xp[HD]; xg = xp[2 * HD]; xo = xp[3 * HD];
      }
    }
    VLN_STAMP(0);
    // sweep this thread's NLD x 2 granules of the group's h tile (time t), published in step - 1
    const unsigned expect = tag_base + (unsigned)step;
    const unsigned pbase = gbase + (unsigned)((step - 1) & 1) * (16u * HD * 8u);
    u32x4_t v[NLD];
    if (step > 0) {
#pragma unroll
      for (int u = 0; u < NLD; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(xres, pbase + (unsigned)(threadIdx.x + u * 256) * 16u, 0, 16);
    }
    if (step > 0) {
      unsigned spins = 0;
      const bool dead = s_abort != 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int u = 0; u < NLD; ++u) ok = ok && (v[u].y == expect) && (v[u].w == expect);
        if (__all(ok) || dead) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kGranSpinLimit) {          // a workgroup of the group is not resident / died
          if (lane == 0) gran_timeout(status, sticky, &s_abort);
          break;
        }
#pragma unroll
        for (int u = 0; u < NLD; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(xres, pbase + (unsigned)(threadIdx.x + u * 256) * 16u, 0, 16);
      }
      VLN_STAMP(1);
      if (step == 1 && (xcd_map & 2)) {            // every member's XCC_ID bit is in: each ORed it before the granules just seen
        const unsigned m = VLN_AGENT_LOAD(gw + 12);
        xcd_local = m != 0u && (m & (m - 1u)) == 0u && s_abort == 0;
        if (threadIdx.x == 0 && ix.jb == 0) __hip_atomic_fetch_add(status + (xcd_local ? 10 : 11), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const int unit = threadIdx.x + u * 256;                 // pair index within the tile: row r, units 2*c2, 2*c2 + 1
        const int r = unit / (HD / 2), c2 = unit % (HD / 2);
        if constexpr (kPlanes) split_store2(ph + r * LDP + c2 * 2, pl + r * LDP + c2 * 2, __uint_as_float(v[u].x), __uint_as_float(v[u].z));
        else *reinterpret_cast<float2*>(&sh[r * LDH + c2 * 2]) = make_float2(__uint_as_float(v[u].x), __uint_as_float(v[u].z));
      }
    } else if (a.init) {
      VLN_STAMP(1);
      // caller-given initial state: plain rows of the first time slot
#pragma unroll
      for (int u = 0; u < HD / 64; ++u) {
        const int unit = threadIdx.x + u * 256;
        const int r = unit / (HD / 4), c4 = unit % (HD / 4);
        float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b0 + r < B) v4 = *reinterpret_cast<const float4*>(a.hprev + (sbase + b0 + r) * HD + c4 * 4);
        if constexpr (kPlanes) {
          split_store2(ph + r * LDP + c4 * 4, pl + r * LDP + c4 * 4, v4.x, v4.y);
          split_store2(ph + r * LDP + c4 * 4 + 2, pl + r * LDP + c4 * 4 + 2, v4.z, v4.w);
        } else {
          *reinterpret_cast<float4*>(&sh[r * LDH + c4 * 4]) = v4;
        }
      }
    } else {
      VLN_STAMP(1);
      for (int i = threadIdx.x; i < kTileBytes / 4; i += 256) reinterpret_cast<uint32_t*>(tile_raw)[i] = 0u;
    }
    __syncthreads();
    VLN_STAMP(2);
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if constexpr (kPlanes) mfma_resident_planes<NS>(ph + fi * LDP, pl + fi * LDP, w, fq, acc);
      else mfma_resident<TW, NS>(&sh[fi * LDH], w, fq, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) sg[wave][fq * 4 + r][fi] = acc[r];
    }
    __syncthreads();
    VLN_STAMP(3);
    float si = 0.f, sf = 0.f, tg = 0.f, so = 0.f, tc = 0.f, yv = 0.f, c_in = creg, hs = 0.f, cs = 0.f;
    if (live) {
      if constexpr (XP) {      // the projection waves' finished sums + bias: what the GEMM's epilogue stored into xproj
        const int sx = step % 3;
        xi = sgx[sx][0][bl][jl] + bx_i; xf = sgx[sx][1][bl][jl] + bx_f; xg = sgx[sx][2][bl][jl] + bx_g; xo = sgx[sx][3][bl][jl] + bx_o;
      }
      const float pi = sg[0][bl][jl] + xi, pf = sg[1][bl][jl] + xf, pg = sg[2][bl][jl] + xg, po = sg[3][bl][jl] + xo;
      const bool valid = t < len;
      const LstmCellPw cw = lstm_cell_pw(pi, pf, pg, po, creg);
      si = cw.si; sf = cw.sf; tg = cw.tg; so = cw.so; tc = cw.tc;
      const float cn = cw.cn, hn = cw.hn;
      yv = valid ? hn : 0.f;
      hs = valid ? hn : hreg; cs = valid ? cn : creg;
      hreg = hs; creg = cs;
    }
    const int tn = (d == 0) ? t + 1 : t - 1;
    if (tn >= 0 && tn < L) {
      // publish: lanes (jl even) store {h_j, tag, h_j+1, tag} = 16 bytes; the 8 stores of a row fill one 128-byte line
      const float hnb = __shfl_down(hs, 1, 64);
      if ((jl & 1) == 0) {
        const unsigned tg_ = tag_base + (unsigned)step + 1u;
        const u32x4_t o = {__float_as_uint(hs), tg_, __float_as_uint(hnb), tg_};
        const unsigned po = gbase + (unsigned)(step & 1) * (16u * HD * 8u) + (unsigned)(bl * HD + j) * 8u;
        if (xcd_local) __builtin_amdgcn_raw_buffer_store_b128(o, xres, po, 0, 0);       // stays in this XCD's L2
        else __builtin_amdgcn_raw_buffer_store_b128(o, xres, po, 0, VLN_GRAN_ST_AUX);   // sc1
      }
      if (live) a.hprev[(((long)d * L + tn) * B + b) * HD + j] = hs;     // history for BPTT / the weight gradients: plain
    } else if (live) {
      a.hcat[(long)b * Y + d * HD + j] = hs;
      a.ccat[(long)b * Y + d * HD + j] = cs;
    }
    VLN_STAMP(4);
    if (live) {   // saved for BPTT / the layer output: read only after this launch
      float* ac = a.act + row * G + (long)d * 4 * HD + j;
      ac[0] = si; ac[HD] = sf; ac[2 * HD] = tg; ac[3 * HD] = so;
      a.tanh_c[row * Y + d * HD + j] = tc;
      a.y[row * Y + d * HD + j] = yv;
      a.cprev[(sbase + b) * HD + j] = c_in;                       // state fed into time t
      if (step == 0 && !a.init) a.hprev[(sbase + b) * HD + j] = 0.f;
    }
    VLN_STAMP(5);
  }
  if (threadIdx.x == 0 && (xcd_map & 2)) {       // the last workgroup of the group through leaves the two words as it found them (zero)
    const unsigned through = __hip_atomic_fetch_add(gw + 28, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (through + 1u == (unsigned)(HD / 16)) {
      __hip_atomic_store(gw + 12, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gw + 28, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  }   // next pass
}

// ---------------------------------------------------------------------------------------------------------
// backward through time (partial-dh exchange as in encoder_persist.h, granules instead of counter + payload).
// Exchange layout: [group][parity][producer jb][unit HD][row 16] granules; a lane's MFMA result (4 consecutive rows of one
// unit) is TWO 16-byte write-through stores; the owner of 16 units fetches each producer's [16 units][16 rows] block with two
// 16-byte sc1 loads per lane.
// ---------------------------------------------------------------------------------------------------------
template <typename TW, int NT>   // NT = Hd / 64: output tiles (16 units each) per wave
__global__ __launch_bounds__(256) void lstm_persist_g_bwd_kernel(RecBwdArgs a, unsigned* status, unsigned* sticky, unsigned char* exch,
                                                                 unsigned tag_base, int xcd_map, const unsigned* seq_dev = nullptr,
                                                                 unsigned seq_rel = 0) {
  if (seq_dev) tag_base = ((*seq_dev + seq_rel) & 0xFFFFFFu) << 8;
  constexpr int HD = NT * 64;
  constexpr int BK = RecCfg<TW>::BK, VK = RecCfg<TW>::VK;
  constexpr int NSK = 64 / BK;                 // K-steps over this workgroup's 64 gate columns
  constexpr int NJB = HD / 16;                 // producers per dependency group
  constexpr int LDT = 64 + 4;
  constexpr int NPW = NJB / 4;                 // producers summed per wave
  __shared__ __attribute__((aligned(16))) float tile[16 * LDT];   // this step's dgates: [row][gate*16 + unit]
  __shared__ __attribute__((aligned(16))) float red[4][64][4];    // per-wave sums of the incoming partial-dh blocks
  __shared__ int s_abort;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 15, fq = lane >> 4;
  const PersistIdx ix = persist_index(HD / 16, a.dirs, (a.B + 15) / 16, xcd_map, (int)blockIdx.x);
  const int jb = ix.jb, j0 = jb * 16, d = ix.d, b0 = ix.bb * 16;
  const int B = a.B, L = a.L;
  const int G = a.dirs * 4 * HD, Y = a.dirs * HD;

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
  const int b = b0 + bl, j = j0 + jl;
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
  const unsigned par_bytes = (unsigned)NJB * HD * 16u * 8u;                 // one parity slot of one group
  __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(exch, 0, (unsigned)persist_g_bwd_bytes(B, HD, a.dirs), 0x00020000);
  const unsigned gbase = (unsigned)(d * ix.nbb + ix.bb) * 2u * par_bytes;
  __builtin_amdgcn_s_setprio(3);
  if (threadIdx.x == 0) s_abort = 0;
  __syncthreads();

  for (int step = L - 1; step >= 0; --step) {
    const int k = L - 1 - step;                // steps already processed
    const int t = (d == 0) ? step : (L - 1 - step);
    const long row = (long)t * B + (live ? b : 0);
    const bool valid = live && (t < len);
    float dyv = 0.f, si = 0.f, sf = 0.f, tg = 0.f, so = 0.f, tc = 0.f, cp = 0.f;
    if (valid) {   // plain loads: produced by the forward pass / upstream gradient before this launch
      if (a.dy) dyv = a.dy[row * Y + d * HD + j];
      const float* ac = a.act + row * G + (long)d * 4 * HD + j;
      si = ac[0]; sf = ac[HD]; tg = ac[2 * HD]; so = ac[3 * HD];
      tc = a.tanh_c[row * Y + d * HD + j];
      cp = a.cprev[(((long)d * L + t) * B + b) * HD + j];
    }
    VLN_STAMP(0);
    float dh = dh_pass;
    if (k > 0) {
      // producer p's block for OUR 16 units: [unit j0 .. j0+15][row 16] granules = 2 KB; lane: unit lane/4, rows (lane&3)*4 .. +3
      const unsigned expect = tag_base + (unsigned)k;
      const unsigned rbase = gbase + (unsigned)((k - 1) & 1) * par_bytes + (unsigned)(j0 * 16) * 8u + (unsigned)lane * 32u;
      u32x4_t v[NPW][2];
      unsigned spins = 0;
      const bool dead = s_abort != 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
          const unsigned po = rbase + (unsigned)(wave + 4 * i) * (unsigned)(HD * 16 * 8);
          v[i][0] = __builtin_amdgcn_raw_buffer_load_b128(xres, po, 0, 16);
          v[i][1] = __builtin_amdgcn_raw_buffer_load_b128(xres, po + 16u, 0, 16);
        }
#pragma unroll
        for (int i = 0; i < NPW; ++i)
          ok = ok && (v[i][0].y == expect) && (v[i][0].w == expect) && (v[i][1].y == expect) && (v[i][1].w == expect);
        if (__all(ok) || dead) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kGranSpinLimit) {
          if (lane == 0) gran_timeout(status, sticky, &s_abort);
          break;
        }
      }
      VLN_STAMP(1);
      float4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NPW; ++i) {          // fixed order: producers wave, wave + 4, ...
        s4.x += __uint_as_float(v[i][0].x); s4.y += __uint_as_float(v[i][0].z);
        s4.z += __uint_as_float(v[i][1].x); s4.w += __uint_as_float(v[i][1].z);
      }
      *reinterpret_cast<float4*>(&red[wave][lane][0]) = s4;
      __syncthreads();
      const int q = jl * 4 + (bl >> 2), e = bl & 3;
      dh += (red[0][q][e] + red[1][q][e]) + (red[2][q][e] + red[3][q][e]);
    }
    VLN_STAMP(2);
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
    if (step > 0) {                            // nobody consumes a partial dh of the last processed step
      float* tr = &tile[bl * LDT + jl];
      tr[0] = g0; tr[16] = g1; tr[32] = g2; tr[48] = g3;
    }
    __syncthreads();                           // tile complete; every wave is past its reads of `red`
    VLN_STAMP(3);
    if (step > 0) {
      AFrag<TW, NSK> af;
      load_afrag<TW, NSK>(af, &tile[fi * LDT], fq);
      const unsigned tg_ = tag_base + (unsigned)k + 1u;
      const unsigned wbase = gbase + (unsigned)(k & 1) * par_bytes + (unsigned)jb * (unsigned)(HD * 16 * 8);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mfma_frags<TW, NSK>(af, w[i], acc);
        const int n = (wave * NT + i) * 16 + fi;
        const unsigned o = wbase + (unsigned)(n * 16 + fq * 4) * 8u;
        const u32x4_t o0 = {__float_as_uint(acc[0]), tg_, __float_as_uint(acc[1]), tg_};
        const u32x4_t o1 = {__float_as_uint(acc[2]), tg_, __float_as_uint(acc[3]), tg_};
        __builtin_amdgcn_raw_buffer_store_b128(o0, xres, o, 0, 16);          // sc1
        __builtin_amdgcn_raw_buffer_store_b128(o1, xres, o + 16u, 0, 16);    // sc1
      }
    }
    VLN_STAMP(4);
    if (live) {    // consumed by the weight-gradient GEMMs after this launch: plain stores, off the hand-off path
      float* dg = a.dgates + row * G + (long)d * 4 * HD + j;
      dg[0] = g0; dg[HD] = g1; dg[2 * HD] = g2; dg[3 * HD] = g3;
    }
    VLN_STAMP(5);
  }
}
