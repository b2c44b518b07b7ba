// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wavefront = 64 lanes everywhere; no warp-32 idioms, no multi-arch paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_raw;

#define VLN_WAVE 64

// cross-workgroup hand-offs (persistent recurrence, split attention): relaxed agent-scope accesses = `sc1` loads / stores
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#define VLN_AGENT_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define VLN_AGENT_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// status codes returned by every C-ABI entry point
#define VLN_OK 0
#define VLN_ERR_ARG 1
#define VLN_ERR_HIP 2

__device__ __forceinline__ float bf16_bits_to_f32(bf16_raw v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_raw f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(bf16_raw, h);
}

// element loaders for the streamed operand type (float or bf16 bits)
template <typename T> struct Elt;
template <> struct Elt<float> {
  static constexpr int kVec = 4;  // elements per 16-byte access
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
  __device__ static __forceinline__ void ld4(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  __device__ static __forceinline__ void st4(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
  // one 16-byte access -> kVec floats
  __device__ static __forceinline__ void ld16(const float* p, float* o) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
};
template <> struct Elt<bf16_raw> {
  static constexpr int kVec = 8;
  __device__ static __forceinline__ float ld(const bf16_raw* p) { return bf16_bits_to_f32(*p); }
  __device__ static __forceinline__ void st(bf16_raw* p, float v) { *p = f32_to_bf16_bits(v); }
  // 4 consecutive elements (8 bytes)
  __device__ static __forceinline__ void ld4(const bf16_raw* p, float (&o)[4]) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  __device__ static __forceinline__ void st4(bf16_raw* p, const float (&o)[4]) {
    uint2 v;
    v.x = (uint32_t)f32_to_bf16_bits(o[0]) | ((uint32_t)f32_to_bf16_bits(o[1]) << 16);
    v.y = (uint32_t)f32_to_bf16_bits(o[2]) | ((uint32_t)f32_to_bf16_bits(o[3]) << 16);
    *reinterpret_cast<uint2*>(p) = v;
  }
  __device__ static __forceinline__ void ld16(const bf16_raw* p, float* o) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// The fused LSTM cell's pointwise part with the roundings spelled out (no compiler-chosen FMA contraction): the per-step,
// the counter-protocol and the granule-protocol recurrence kernels must produce bit-identical states.
struct LstmCellPw { float si, sf, tg, so, cn, tc, hn; };
__device__ __forceinline__ LstmCellPw lstm_cell_pw(float pi, float pf, float pg, float po, float c_prev) {
  LstmCellPw o;
  o.si = sigmoidf_(pi); o.sf = sigmoidf_(pf); o.tg = tanhf(pg); o.so = sigmoidf_(po);
  o.cn = __fmaf_rn(o.sf, c_prev, __fmul_rn(o.si, o.tg));
  o.tc = tanhf(o.cn);
  o.hn = __fmul_rn(o.so, o.tc);
  return o;
}

// ---------------------------------------------------------------------------
// Philox4x32-10 counter RNG: dropout masks are a pure function of
// (seed, stream offset, element index) so backward regenerates them instead
// of storing them, and tests can export the exact mask the kernels used.
// ---------------------------------------------------------------------------
struct Philox4 { uint32_t x, y, z, w; };
__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) {
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}
__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint64_t seed, uint64_t offset, uint32_t idx) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = idx, c1 = 0u, c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}
// Keep-masks (already scaled by 1/(1-p)).  Element e of a site draws the 16-bit lane (e & 7) of Philox call (e >> 3): word
// (e & 7) >> 1 of the call's four 32-bit outputs, low half for even e, high half for odd e; kept iff u16 / 65536 >= p (the
// keep probability is exact to 2^-16).  ONE call serves 8 consecutive elements: the 32-bit integer multiplies of Philox run
// at a quarter of the vector rate on gfx950, and with a call per 4 elements the feature gather (5.8 M elements per decoder
// step), the context dropout and the embedding kernels were bound by them, not by memory (round 3: gather 14.2 -> see
// profiles/round3_notes.md).  dropout_scale8 / 4 / 1 all follow this one mapping, so vln_dropout_mask exports what every
// kernel uses.
__host__ __device__ __forceinline__ void dropout_scale8(uint64_t seed, uint64_t offset, uint32_t idx8, float p, float (&m)[8]) {
  const Philox4 r = philox4x32_10(seed, offset, idx8);
  const float inv = 1.0f / (1.0f - p);
  const float u = 1.0f / 65536.0f;
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    m[2 * k] = ((float)(w[k] & 0xFFFFu) * u >= p) ? inv : 0.0f;
    m[2 * k + 1] = ((float)(w[k] >> 16) * u >= p) ? inv : 0.0f;
  }
}
// 4 consecutive elements idx4*4 .. idx4*4+3 (half of call idx4 >> 1)
__host__ __device__ __forceinline__ void dropout_scale4(uint64_t seed, uint64_t offset, uint32_t idx4, float p,
                                                        float (&m)[4]) {
  const Philox4 r = philox4x32_10(seed, offset, idx4 >> 1);
  const float inv = 1.0f / (1.0f - p);
  const float u = 1.0f / 65536.0f;
  const uint32_t w0 = (idx4 & 1u) ? r.z : r.x, w1 = (idx4 & 1u) ? r.w : r.y;
  m[0] = ((float)(w0 & 0xFFFFu) * u >= p) ? inv : 0.0f;
  m[1] = ((float)(w0 >> 16) * u >= p) ? inv : 0.0f;
  m[2] = ((float)(w1 & 0xFFFFu) * u >= p) ? inv : 0.0f;
  m[3] = ((float)(w1 >> 16) * u >= p) ? inv : 0.0f;
}
// single element
__host__ __device__ __forceinline__ float dropout_scale1(uint64_t seed, uint64_t offset, uint32_t e, float p) {
  if (p <= 0.0f) return 1.0f;
  float m[4];
  dropout_scale4(seed, offset, e >> 2, p, m);
  return m[e & 3];
}

// A matrix that still lies in `n` split-K partial slabs: value(r, c) = sum_{s < n} p[s*stride + r*ld + c], summed in
// the fixed order s = 0 .. n-1.  The consumer of a GEMM result adds the partials while it loads them, so a skinny
// product can be split over all 256 CUs without a reduce launch in the dependent chain (n == 1: a plain matrix).
struct SlabVec {
  const float* p; long ld; int n; long stride;
  const float* bias = nullptr;     // nullable: a per-column vector added to the sum (a Linear's bias that no reduce launch applied)
  // Partials are fetched four at a time (independent loads in flight, no branch between them) and added in slab order.
  __device__ __forceinline__ float at(long r, long c) const {
    const float* q = p + r * ld + c;
    float v = 0.f;
    int s = 0;
    for (; s + 3 < n; s += 4) {
      const float t0 = q[(long)s * stride], t1 = q[(long)(s + 1) * stride], t2 = q[(long)(s + 2) * stride], t3 = q[(long)(s + 3) * stride];
      v += t0; v += t1; v += t2; v += t3;
    }
    for (; s < n; ++s) v += q[(long)s * stride];
    if (bias) v += bias[c];
    return v;
  }
  __device__ __forceinline__ float4 at4(long r, long c) const {      // 16-byte aligned column group
    const float* q = p + r * ld + c;
    auto ld4 = [&](int s) { return *reinterpret_cast<const float4*>(q + (long)s * stride); };
    auto add = [](float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 3 < n; s += 4) {
      const float4 t0 = ld4(s), t1 = ld4(s + 1), t2 = ld4(s + 2), t3 = ld4(s + 3);
      add(v, t0); add(v, t1); add(v, t2); add(v, t3);
    }
    for (; s < n; ++s) add(v, ld4(s));
    if (bias) add(v, *reinterpret_cast<const float4*>(bias + c));
    return v;
  }
  __host__ __device__ SlabVec shifted(long cols) const { return SlabVec{p + cols, ld, n, stride, bias ? bias + cols : nullptr}; }
};
static inline SlabVec plain_vec(const float* p, long ld) { return SlabVec{p, ld, 1, 0}; }

// x[0..3] . t in ONE fixed association.  The candidate-logit kernels (attn_dot_kernel, attn_dot_multi_kernel, cand_sample_kernel) must
// agree bit for bit (per-step, rollout-wide and in-step forms of the same logits): left to the compiler, the contraction of
// `a * b + c * d + ...` into fused multiply-adds may differ from one kernel to the next.
__device__ __forceinline__ float dot4(const float* x, const float4& t) { return fmaf(x[3], t.w, fmaf(x[2], t.z, fmaf(x[1], t.y, x[0] * t.x))); }

struct DropSpec {      // one dropout site; p == 0 disables it
  uint64_t seed;
  uint64_t offset;     // Philox offset of the site -- or, when `step` is set, the site index k of offset = *step * 8 + k
  float p;
  // Optional: the per-step offset lives in DEVICE memory (written before the step's launches), so the launch
  // arguments of a step do not change from call to call and the whole step can be replayed as one hipGraph.
  const unsigned long long* step = nullptr;
  __device__ __forceinline__ uint64_t off() const { return step ? (uint64_t)(*step) * 8ull + offset : offset; }
};
// C-ABI form: `offset_base_dev` (nullable) = a device word; the site's Philox offset is then (*offset_base_dev) * 8 + offset,
// read on the device, so the launch arguments repeat from iteration to iteration (whole-iteration graphs, runtime.DeviceClock)
static inline DropSpec drop_spec(uint64_t seed, uint64_t offset, float p, const uint64_t* offset_base_dev) {
  return DropSpec{seed, offset, p, reinterpret_cast<const unsigned long long*>(offset_base_dev)};
}
