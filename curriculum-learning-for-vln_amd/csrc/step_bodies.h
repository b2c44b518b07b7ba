// Workgroup bodies of the decoder steps' elementwise stages (included inside namespace vln by pointwise.hip, envdrop.hip,
// gemm.hip and chain.hip).  Every body runs either as a kernel of its own or as a stage of the chained step kernel
// (chain.hip): the work decomposition is passed in, results do not depend on it.
#pragma once
#include "vln_internal.h"

namespace vln {

// ---- LSTM cell pointwise (torch.nn.LSTMCell semantics, gate order i,f,g,o) ------------------------------------------
// 256 threads = 64 (row, unit) pairs x 4 gates: each thread sums the split-K slabs of ONE gate (4 independent
// load streams per wave instead of 4*nsplit dependent loads per thread), the gates meet in LDS (`sg`: 4 x 64 floats).
// Virtual block `vb` of `nvb` takes the 64-pair groups vb, vb + nvb, ...; `iters` is the same for every block of a
// workgroup (the loop contains workgroup barriers).
__device__ __forceinline__ void lstm_pw_fwd_body(const LstmPwFwd& a, int vb, int nvb, int iters, int tid, float (*sg)[64]) {
  const long total = (long)a.B * a.H;
  const int H = a.H;
  const int pl = tid & 63, q = tid >> 6;
  for (int it = 0; it < iters; ++it) {
    const long base = ((long)vb + (long)it * nvb) * 64;
    const long e = base + pl;
    const bool ok = e < total && vb < nvb;
    const int b = ok ? (int)(e / H) : 0, j = ok ? (int)(e % H) : 0;
    {
      const int col = q * H + j;
      float v0 = 0.f, v1 = 0.f;
      if (ok) {
        if (a.bias_a) v0 += a.bias_a[col];
        if (a.bias_b) v1 += a.bias_b[col];
        const float* gp = a.gates + (long)b * 4 * H + col;
        int s = 0;
        for (; s + 3 < a.nsplit; s += 4) {        // four partials in flight, two accumulation chains
          const float t0 = gp[(long)s * a.slab_stride], t1 = gp[(long)(s + 1) * a.slab_stride];
          const float t2 = gp[(long)(s + 2) * a.slab_stride], t3 = gp[(long)(s + 3) * a.slab_stride];
          v0 += t0; v1 += t1; v0 += t2; v1 += t3;
        }
        for (; s + 1 < a.nsplit; s += 2) {
          v0 += gp[(long)s * a.slab_stride];
          v1 += gp[(long)(s + 1) * a.slab_stride];
        }
        if (s < a.nsplit) v0 += gp[(long)s * a.slab_stride];
      }
      sg[q][pl] = v0 + v1;
    }
    __syncthreads();
    if (q == 0 && ok) {
      float g[4] = {sg[0][pl], sg[1][pl], sg[2][pl], sg[3][pl]};
      const float si = sigmoidf_(g[0]), sf = sigmoidf_(g[1]), tg = tanhf(g[2]), so = sigmoidf_(g[3]);
      const float c0 = a.c0[(long)b * a.ldc0 + j];
      const float c1 = sf * c0 + si * tg;
      const float tc = tanhf(c1);
      const float h1 = so * tc;
      a.h1[(long)b * a.ldh1 + j] = h1;
      a.c1[(long)b * a.ldc1 + j] = c1;
      if (a.act) {
        float* p = a.act + (long)b * 4 * H + j;
        p[0] = si; p[H] = sf; p[2 * H] = tg; p[3 * H] = so;
      }
      if (a.tanh_c1) a.tanh_c1[e] = tc;
      if (a.h1_drop) a.h1_drop[(long)b * a.ldh1d + j] = h1 * dropout_scale1(a.drop.seed, a.drop.off(), (uint32_t)e, a.drop.p);
    }
    __syncthreads();   // sg is rewritten by the next iteration
  }
}

__device__ __forceinline__ void lstm_pw_bwd_body(const LstmPwBwd& a, long first, long stride) {
  const long total = (long)a.B * a.H;
  const int H = a.H;
  for (long e = first; e < total; e += stride) {
    const int b = (int)(e / H), j = (int)(e % H);
    float dh = a.dh1_a ? a.dh1_a[(long)b * a.ld_a + j] : 0.f;
    if (a.dh1_b.p) {
      float v = a.dh1_b.at(b, j);
      if (a.dh1_b2.p) v += a.dh1_b2.at(b, j);
      dh += v * dropout_scale1(a.drop.seed, a.drop.off(), (uint32_t)e, a.drop.p);
    }
    const float* act = a.act + (long)b * 4 * H + j;
    const float si = act[0], sf = act[H], tg = act[2 * H], so = act[3 * H];
    const float tc = a.tanh_c1[e];
    const float c0 = a.c0[(long)b * a.ldc0 + j];
    float dc = (a.dc1 ? a.dc1[(long)b * a.lddc1 + j] : 0.f) + dh * so * (1.f - tc * tc);
    float* dg = a.dgates + (long)b * a.lddg + j;
    dg[0] = dc * tg * si * (1.f - si);
    dg[H] = dc * c0 * sf * (1.f - sf);
    dg[2 * H] = dc * si * (1.f - tg * tg);
    dg[3 * H] = dh * tc * so * (1.f - so);
    a.dc0[(long)b * a.lddc0 + j] = dc * sf;
  }
}

// ---- out = act(sum_s slabs[s] + bias); optional second output out2 = out * dropout mask -------------------------------
struct ReduceEpiArgs {
  const float* slabs; int nsplit; long slab_stride; long lds;
  float* out; long ldo; int M, N;
  const float* bias; int act;
  float* out2; long ldo2; DropSpec drop;
};
__device__ __forceinline__ void reduce_epilogue_body(const ReduceEpiArgs& a, long first, long stride) {
  const long total = (long)a.M * a.N;
  const int N = a.N, nsplit = a.nsplit;
  for (long e = first; e < total; e += stride) {
    const int r = (int)(e / N), c = (int)(e % N);
    float v = a.bias ? a.bias[c] : 0.0f;
    const float* sp = a.slabs + (long)r * a.lds + c;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int s = 0;
    for (; s + 3 < nsplit; s += 4) {   // 4 independent load streams
      v += sp[(long)s * a.slab_stride];
      v1 += sp[(long)(s + 1) * a.slab_stride];
      v2 += sp[(long)(s + 2) * a.slab_stride];
      v3 += sp[(long)(s + 3) * a.slab_stride];
    }
    for (; s < nsplit; ++s) v += sp[(long)s * a.slab_stride];
    v += (v1 + v2) + v3;
    if ((a.act & 3) == ACT_TANH) v = tanhf(v);
    else if ((a.act & 3) == ACT_RELU) v = fmaxf(v, 0.0f);
    if (a.act & ACT_ACCUM) v += a.out[(long)r * a.ldo + c];
    a.out[(long)r * a.ldo + c] = v;
    if (a.out2) a.out2[(long)r * a.ldo2 + c] = v * dropout_scale1(a.drop.seed, a.drop.off(), (uint32_t)e, a.drop.p);
  }
}

// ---- EnvDrop step backward: act embedding + the two uses of h_tilde_prev (policy.py:224,234 differentiated) -----------
struct PrepBwdArgs {
  SlabVec dxcat; const float* e; SlabVec dhq;      // dxcat [B, AE+F+H] and dhq [B,H] may still lie in split-K slabs
  float* s_de; float* dhtp;
  int B, AE, F, H;
  DropSpec d_act, d_h;
};
// The three element formulas of these stages with every rounding spelled out: the stand-alone kernels and the chained kernel (which
// does two of the stages in one launch) must produce the same bits, and the compiler's FMA contraction (-ffp-contract=fast is the
// HIP default; __fmul_rn / __fadd_rn are plain operators on AMD) decides differently in different kernels.  Contraction is off
// inside these helpers and the fused multiply-adds that ARE wanted are written as such.
__device__ __forceinline__ float prep_bwd_de(float dx, float mask, float e) {           // d act-embedding pre-activation
#pragma clang fp contract(off)
  const float t = __builtin_fmaf(-e, e, 1.f);
  return (dx * mask) * t;
}
__device__ __forceinline__ float prep_bwd_dhtp(float dx_tail, float dhq, float mask) {   // d h_tilde_prev: its two uses
#pragma clang fp contract(off)
  return __builtin_fmaf(dhq, mask, dx_tail);
}
__device__ __forceinline__ float tanh_drop_dz(float gs, float mask, float dht, bool has_dht, float h) {
#pragma clang fp contract(off)
  const float g = has_dht ? __builtin_fmaf(gs, mask, dht) : gs * mask;
  const float t = __builtin_fmaf(-h, h, 1.f);
  return g * t;
}

__device__ __forceinline__ void envdrop_prep_bwd_body(const PrepBwdArgs& p, long first, long stride) {
  const long ne = (long)p.B * p.AE, nh = (long)p.B * p.H;
  for (long i = first; i < ne + nh; i += stride) {
    if (i < ne) {
      const int b = (int)(i / p.AE), j = (int)(i % p.AE);
      p.s_de[i] = prep_bwd_de(p.dxcat.at(b, j), dropout_scale1(p.d_act.seed, p.d_act.off(), (uint32_t)i, p.d_act.p), p.e[i]);
    } else {
      const long k = i - ne;
      const int b = (int)(k / p.H), j = (int)(k % p.H);
      p.dhtp[k] = prep_bwd_dhtp(p.dxcat.at(b, p.AE + p.F + j), p.dhq.at(b, j), dropout_scale1(p.d_h.seed, p.d_h.off(), (uint32_t)k, p.d_h.p));
    }
  }
}

// Chained steps (round 4): the prep backward of step t and the tanh / dropout backward of step t-1 -- the FIRST stage of the next
// step's backward, whose d h_tilde is exactly this stage's dhtp -- as one launch: dhtp is stored AND used in place of `dht_ext`.
struct PrepTanhBwdArgs { PrepBwdArgs p; SlabVec dhtd; const float* dhtd2; const float* ht; float* dz; DropSpec d; };
__device__ __forceinline__ void envdrop_prep_tanh_bwd_body(const PrepTanhBwdArgs& a, long first, long stride) {
  const PrepBwdArgs& p = a.p;
  const long ne = (long)p.B * p.AE, nh = (long)p.B * p.H;
  for (long i = first; i < ne + nh; i += stride) {
    if (i < ne) {
      const int b = (int)(i / p.AE), j = (int)(i % p.AE);
      p.s_de[i] = prep_bwd_de(p.dxcat.at(b, j), dropout_scale1(p.d_act.seed, p.d_act.off(), (uint32_t)i, p.d_act.p), p.e[i]);
    } else {
      const long k = i - ne;
      const int b = (int)(k / p.H), j = (int)(k % p.H);
      const float v = prep_bwd_dhtp(p.dxcat.at(b, p.AE + p.F + j), p.dhq.at(b, j), dropout_scale1(p.d_h.seed, p.d_h.off(), (uint32_t)k, p.d_h.p));
      p.dhtp[k] = v;
      float gs = a.dhtd.at(b, j);                    // tanh_drop_bwd_body of the next step, with dht_ext = v
      if (a.dhtd2) gs += a.dhtd2[k];
      a.dz[k] = tanh_drop_dz(gs, dropout_scale1(a.d.seed, a.d.off(), (uint32_t)k, a.d.p), v, true, a.ht[k]);
    }
  }
}

// dz = ((dhtd + dhtd2) * mask + dht_ext) * (1 - ht^2)      (dhtd [B,H] may still lie in split-K slabs; dhtd2 nullable:
// the second consumer of the logits when the rollout-wide logit branch already covered the first)
struct TanhDropBwdArgs {
  SlabVec dhtd; const float* dhtd2; const float* dht_ext; const float* ht; float* dz; int B, H; DropSpec d;
};
__device__ __forceinline__ void tanh_drop_bwd_body(const TanhDropBwdArgs& a, long first, long stride) {
  const long n = (long)a.B * a.H;
  for (long i = first; i < n; i += stride) {
    float gs = a.dhtd.at(i / a.H, i % a.H);
    if (a.dhtd2) gs += a.dhtd2[i];
    a.dz[i] = tanh_drop_dz(gs, dropout_scale1(a.d.seed, a.d.off(), (uint32_t)i, a.d.p), a.dht_ext ? a.dht_ext[i] : 0.f, a.dht_ext != nullptr, a.ht[i]);
  }
}

}  // namespace vln
