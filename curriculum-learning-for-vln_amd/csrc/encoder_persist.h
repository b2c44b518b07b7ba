// Persistent packed-(bi)LSTM recurrence: ONE launch walks all L time steps (included by encoder.hip).
//
// Why: one launch per time step costs ~6.5 us + ~1.5 us of inter-kernel gap for <1 us of work (160 launches per
// EnvDrop iteration).  Here every workgroup keeps its slice of W_hh in registers and its (row, unit) cell
// and hidden state in registers for the whole sequence; only the new hidden slice crosses workgroups.
//
// Geometry: workgroup (jb, d, bb) = 16 hidden units x direction d x 16 batch rows, 4 waves = 4 gates.
// Dependency group = the Hd/16 workgroups sharing (d, bb): step s needs the h slices all of them wrote in
// step s-1.  Hand-off (cdna_hip_programming.md Guideline 16, recipe R1, counter form):
//   producer: every h element is stored write-through (`global_store_dword sc1` via a relaxed agent-scope
//             atomic store) -> every wave `s_waitcnt vmcnt(0)` -> workgroup barrier -> ONE lane adds 1 to the
//             group's counter (agent-scope atomic);
//   consumer: ONE lane polls the counter with sc1 loads (+ s_sleep) until it reaches njb*s -> workgroup barrier
//             -> EVERY load of handed-off bytes is a `buffer_load_dwordx4 ... sc1` (bypasses this CU's L1).
// No fence is needed in this form; results do not depend on placement or timing.  All workgroups must be
// co-resident (the host only takes this path when the grid fits the 256 CUs); every spin is bounded and a
// timeout sets the status word and lets the kernel drain (wrong numbers, never a hang).
#pragma once

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

#define VLN_AGENT_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define VLN_AGENT_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

__device__ __forceinline__ void group_wait(unsigned* cnt, unsigned target, unsigned* status, int* s_abort) {
  if (threadIdx.x == 0 && !*s_abort) {
    unsigned spins = 0;
    while (VLN_AGENT_LOAD(cnt) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 24)) {            // ~1 s: a workgroup of the group is not resident / died
        VLN_AGENT_STORE(status, 1u);
        *s_abort = 1;
        break;
      }
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void group_arrive(unsigned* cnt) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
__global__ __launch_bounds__(256) void lstm_persist_fwd_kernel(RecFwdArgs a, unsigned* counters, unsigned* status) {
  constexpr int HD = NS * RecCfg<TW>::BK;
  constexpr int LDH = HD + 4;
  __shared__ __attribute__((aligned(16))) float sh[16 * LDH];
  __shared__ float sg[4][16][17];
  __shared__ int s_abort;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 15, fq = lane >> 4;
  const int j0 = blockIdx.x * 16, d = blockIdx.y, b0 = blockIdx.z * 16;
  const int B = a.B, L = a.L;
  const unsigned njb = gridDim.x;
  const int G = a.dirs * 4 * HD, Y = a.dirs * HD;
  unsigned* cnt = counters + d * gridDim.z + blockIdx.z;

  WFrag<TW, NS> w;
  load_wfrag<TW, NS>(w, reinterpret_cast<const TW*>(a.w_hh) + ((long)d * 4 * HD + (long)wave * HD + j0 + fi) * HD, fq);

  const int bl = threadIdx.x >> 4, jl = threadIdx.x & 15;
  const int b = b0 + bl, j = j0 + jl;
  const bool live = b < B;
  const int len = live ? a.lengths[b] : 0;
  float hreg = 0.f, creg = 0.f;
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
    if (step > 0) {
      group_wait(cnt, njb * (unsigned)step, status, &s_abort);
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
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      mfma_resident<TW, NS>(&sh[fi * LDH], w, fq, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) sg[wave][fq * 4 + r][fi] = acc[r];
    }
    __syncthreads();
    if (live) {
      const float pi = sg[0][bl][jl] + xi, pf = sg[1][bl][jl] + xf, pg = sg[2][bl][jl] + xg, po = sg[3][bl][jl] + xo;
      const bool valid = t < len;
      const float si = sigmoidf_(pi), sf = sigmoidf_(pf), tg = tanhf(pg), so = sigmoidf_(po);
      const float cn = sf * creg + si * tg, tc = tanhf(cn), hn = so * tc;
      float* ac = a.act + row * G + (long)d * 4 * HD + j;
      ac[0] = si; ac[HD] = sf; ac[2 * HD] = tg; ac[3 * HD] = so;
      a.tanh_c[row * Y + d * HD + j] = tc;
      a.y[row * Y + d * HD + j] = valid ? hn : 0.f;
      a.cprev[(sbase + b) * HD + j] = creg;                       // state fed into time t (for BPTT)
      if (step == 0) a.hprev[(sbase + b) * HD + j] = 0.f;
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
    group_arrive(cnt);
  }
}

// ---------------------------------------------------------------------------------------------------------
// backward through time
// ---------------------------------------------------------------------------------------------------------
template <typename TW, int NS>
__global__ __launch_bounds__(256) void lstm_persist_bwd_kernel(RecBwdArgs a, unsigned* counters, unsigned* status) {
  constexpr int HD = NS * RecCfg<TW>::BK;
  constexpr int LDD = 4 * HD + 4;
  // ALL LDS of this kernel is one dynamic array (base stays 16-byte aligned, Guideline 17):
  //   sd [16][LDD] dgates tile of the later time step | sp [4][16][17] partial dh | abort flag
  extern __shared__ __attribute__((aligned(16))) float sd[];
  float (*sp)[16][17] = reinterpret_cast<float (*)[16][17]>(sd + 16 * LDD);
  int& s_abort = *reinterpret_cast<int*>(sd + 16 * LDD + 4 * 16 * 17);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 15, fq = lane >> 4;
  const int j0 = blockIdx.x * 16, d = blockIdx.y, b0 = blockIdx.z * 16;
  const int B = a.B, L = a.L;
  const unsigned njb = gridDim.x;
  const int G = a.dirs * 4 * HD, Y = a.dirs * HD;
  unsigned* cnt = counters + d * gridDim.z + blockIdx.z;

  WFrag<TW, NS> w;   // rows j of W_hh^T, this wave's gate block of the contraction
  load_wfrag<TW, NS>(w, reinterpret_cast<const TW*>(a.w_hh_t) + ((long)d * HD + j0 + fi) * 4 * HD + (long)wave * HD, fq);

  const int bl = threadIdx.x >> 4, jl = threadIdx.x & 15;
  const int b = b0 + bl, j = j0 + jl;
  const bool live = b < B;
  const int len = live ? a.lengths[b] : 0;
  const long ci = ((long)d * B + (live ? b : 0)) * HD + j;
  float dh_pass = live ? a.dh_pass[ci] : 0.f, dcc = live ? a.dc_carry[ci] : 0.f;
  __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(a.dgates, 0, (unsigned)((long)L * B * G * 4), 0x00020000);
  __builtin_amdgcn_s_setprio(3);   // latency-critical chain: win issue arbitration against co-resident streaming work
  if (threadIdx.x == 0) s_abort = 0;
  __syncthreads();

  for (int step = L - 1; step >= 0; --step) {
    const int t = (d == 0) ? step : (L - 1 - step);
    const bool first = (step == L - 1);
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
    if (!first) {
      group_wait(cnt, njb * (unsigned)(L - 1 - step), status, &s_abort);
      const int tl = (d == 0) ? t + 1 : t - 1;
#pragma unroll 4
      for (int u = 0; u < HD / 16; ++u) {
        const int unit = threadIdx.x + u * 256;
        const int r = unit / HD, c4 = unit % HD;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (b0 + r < B)
          v = __builtin_amdgcn_raw_buffer_load_b128(gres, (unsigned)((((long)tl * B + b0 + r) * G + (long)d * 4 * HD + c4 * 4) * 4), 0, 16);
        *reinterpret_cast<u32x4_t*>(&sd[r * LDD + c4 * 4]) = v;
      }
    }
    __syncthreads();
    if (!first) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      mfma_resident<TW, NS>(&sd[fi * LDD + wave * HD], w, fq, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) sp[wave][fq * 4 + r][fi] = acc[r];
    }
    __syncthreads();
    if (live) {
      float dh = dh_pass;
      if (!first) dh += sp[0][bl][jl] + sp[1][bl][jl] + sp[2][bl][jl] + sp[3][bl][jl];
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
      float* dg = a.dgates + row * G + (long)d * 4 * HD + j;      // write-through: the group reads it next step
      VLN_AGENT_STORE(dg, g0); VLN_AGENT_STORE(dg + HD, g1); VLN_AGENT_STORE(dg + 2 * HD, g2); VLN_AGENT_STORE(dg + 3 * HD, g3);
    }
    group_arrive(cnt);
  }
}
