// MLPwithBN (reference units.py:210-242) forward and backward as ONE C call each: BatchNorm1d, then per hidden layer Linear,
// BatchNorm1d, Dropout, ReLU.  The Self-Monitor agent runs it twice per decoder step (previous action [B, F], candidates
// [B*C, F]: policy.py:146-149).  The launches are the ones `functional.BnMlpFn` used to drive from Python -- vln_bn_fwd / bwd
// (row-chunked for the tall candidate input), the skinny GEMMs, one grouped weight-gradient and one grouped bias-gradient
// launch -- issued here back to back with a single Python -> C crossing.
#include "vln_internal.h"
#include "../../include/vln_hip.h"

namespace vln {
namespace {

inline long r64(long n) { return (n + 63) & ~63L; }
inline long lin_ws(long ws_floats, long M, long N) {
  long cap = 16 * M * N;
  if (cap > (1L << 24)) cap = 1L << 24;
  return ws_floats < cap ? ws_floats : cap;
}

int check_mlp(const vln_bn_mlp* m) {
  if (!m || m->R <= 0 || m->D0 <= 0 || m->nl <= 0 || m->nl > VLN_BN_MLP_MAX_LAYERS || (m->D0 & 3) || m->R1 < 0 || m->R1 >= m->R) { set_error("bn_mlp: bad dims"); return VLN_ERR_ARG; }
  for (int i = 0; i < m->nl; ++i)
    if (m->layer[i].out <= 0 || (m->layer[i].out & 3) || !m->layer[i].w || !m->layer[i].w_t) { set_error("bn_mlp: bad layer %d", i); return VLN_ERR_ARG; }
  if (m->x2 && m->R1 <= 0) { set_error("bn_mlp: x2 (the second batch's own array) needs R1 > 0"); return VLN_ERR_ARG; }
  return VLN_OK;
}

struct SavedLayout { long y0, s0, z[VLN_BN_MLP_MAX_LAYERS], s[VLN_BN_MLP_MAX_LAYERS], y[VLN_BN_MLP_MAX_LAYERS], total; };
SavedLayout saved_layout(const vln_bn_mlp* m) {
  SavedLayout L{};
  long off = 0;
  auto take = [&](long n) { long o = off; off += r64(n); return o; };
  const long R = m->R;
  const long ns = m->R1 > 0 ? 4 : 2;                 // per BatchNorm: (mean, rstd) per segment
  L.y0 = take(R * m->D0); L.s0 = take(ns * m->D0);
  for (int i = 0; i < m->nl; ++i) {
    const long D = m->layer[i].out;
    L.z[i] = take(R * D); L.s[i] = take(ns * D); L.y[i] = take(R * D);
  }
  L.total = off;
  return L;
}

}  // namespace
}  // namespace vln

using namespace vln;
#define RUN(x) do { int _s = (x); if (_s != VLN_OK) return _s; } while (0)

extern "C" int64_t vln_bn_mlp_saved_floats(const vln_bn_mlp* m) { return check_mlp(m) == VLN_OK ? saved_layout(m).total : -1; }
extern "C" int64_t vln_bn_mlp_out_offset(const vln_bn_mlp* m) { return check_mlp(m) == VLN_OK ? saved_layout(m).y[m->nl - 1] : -1; }
extern "C" int64_t vln_bn_mlp_ws_floats(const vln_bn_mlp* m) {
  if (check_mlp(m) != VLN_OK) return -1;
  // the grouped weight-gradient launch's packed operands (both bf16 planes of dy and x in fragment order) + the slabs of its
  // row split (gemm.hip; the same sizing as ops.WgradBatch)
  const long MS = (m->R + 31) / 32;
  long area = 0, dmax = m->D0, in = m->D0, tiles = 0, nk = 0;
  for (int i = 0; i < m->nl; ++i) {
    const long out = m->layer[i].out;
    area += 2 * ((out + 15) / 16 + (in + 15) / 16) * MS * 1024 / 4;
    tiles += ((out + 127) / 128) * ((in + 127) / 128);
    nk += out * in;
    if (out > dmax) dmax = out;
    in = out;
  }
  long msplit = 1;
  if (tiles < 256) { msplit = 256 / (tiles > 0 ? tiles : 1); if (msplit > MS / 4) msplit = MS / 4; if (msplit < 1) msplit = 1; }
  if (msplit > 1) area += msplit * nk;
  long n = area;
  const long bn_part = (long)((m->R + 127) / 128 + 1) * 2 * dmax;  // chunked BatchNorm partials (+1: two segments round up separately)
  const long slabs = 16L * m->R * dmax;                            // split-K slabs of the skinny products
  if (bn_part > n) n = bn_part;
  if (slabs > n) n = slabs;
  if (n < (1L << 22)) n = 1L << 22;
  return n;
}
extern "C" int64_t vln_bn_mlp_bwd_scratch_floats(const vln_bn_mlp* m) {
  if (check_mlp(m) != VLN_OK) return -1;
  long n = 0, in = m->D0;
  for (int i = 0; i < m->nl; ++i) { n += r64((long)m->R * m->layer[i].out) + r64((long)m->R * in); in = m->layer[i].out; }
  return n;
}

extern "C" int vln_bn_mlp_fwd(const vln_bn_mlp* m, const float* x, int64_t ldx, float* saved, float* ws, int64_t ws_floats, vln_stream_t s) {
  RUN(check_mlp(m));
  if (!x || !saved || !ws) { set_error("vln_bn_mlp_fwd: null pointer"); return VLN_ERR_ARG; }
  DropBaseScope drop_scope(m->offset_base_dev);
  hipStream_t st = (hipStream_t)s;
  const SavedLayout L = saved_layout(m);
  const int R = m->R, tr = m->training;
  float* y = saved + L.y0;
  float* st0 = saved + L.s0;
  const int R1 = m->R1;
  RUN(bn_fwd_seg(x, ldx, y, m->D0, m->bn0.gamma, m->bn0.beta, m->bn0.run_mean, m->bn0.run_var, tr ? m->bn0.nbt : nullptr,
                 tr ? st0 : nullptr, tr ? st0 + m->D0 : nullptr, R, R1, 2L * m->D0, m->D0, m->eps, m->momentum, tr, 0, 0, 0, 0, 0.f, nullptr,
                 ws, ws_floats, s, m->x2, m->ldx2));
  int in = m->D0;
  for (int i = 0; i < m->nl; ++i) {
    const vln_bn_mlp_layer& l = m->layer[i];
    float* z = saved + L.z[i];
    float* si = saved + L.s[i];
    float* yi = saved + L.y[i];
    // split-K scratch bounded like ops.linear_fwd bounds it: the same K split, hence the same bits, as the operator path.
    // VLN_F32S (fp32-streamed layer of the bf16 mode): the FORWARD product runs in the fp32-GRADE six-product form (W_F32X) -- a
    // ReLU follows (behind the BatchNorm), and a unit whose pre-activation moves across zero changes a whole gradient term: the
    // 2^-16 of the three-product form flipped ~10 of the 1.2 M units at BASELINE config 2 (2.5e-2 of the weight gradient's range,
    // profiles/round4_notes.md), the six-product form none beyond what exact fp32 itself flips against fp64 (the exact fp32 MFMA
    // cost 92 us per call at M = 1152).  The backward's products (no ReLU decision in them) keep the three-product form.
    // a ReLU decision follows this product: an fp32-streamed layer multiplies in the fp32-GRADE six-product form (W_F32X), not the 2^-16 one
    RUN(gemm_nt(st, y, in, l.w, m->wtype == W_F32S ? (int)W_F32X : m->wtype, in, z, l.out, R, l.out, in, l.b, ACT_NONE, ws, lin_ws(ws_floats, R, l.out), nullptr));
    const bool last = (i == m->nl - 1);
    RUN(bn_fwd_seg(z, l.out, yi, l.out, l.bn.gamma, l.bn.beta, l.bn.run_mean, l.bn.run_var, tr ? l.bn.nbt : nullptr, tr ? si : nullptr,
                   tr ? si + l.out : nullptr, R, R1, 2L * l.out, l.out, m->eps, m->momentum, tr, 1, l.seed, l.offset, l.offset2,
                   tr ? l.p_drop : 0.f, last ? m->row_zero : nullptr, ws, ws_floats, s));
    y = yi; in = l.out;
  }
  return VLN_OK;
}

extern "C" int vln_bn_mlp_bwd(const vln_bn_mlp* m, const float* x, int64_t ldx, const float* saved, const float* dy, int64_t lddy,
                              float* dx, int64_t lddx, const vln_bn_mlp_grads* g, float* ws, int64_t ws_floats, vln_stream_t s) {
  RUN(check_mlp(m));
  if (!x || !saved || !dy || !g || !ws || !g->scratch || g->scratch_floats < vln_bn_mlp_bwd_scratch_floats(m)) {
    set_error("vln_bn_mlp_bwd: null pointer or scratch too small");
    return VLN_ERR_ARG;
  }
  if (m->x2 && dx) { set_error("vln_bn_mlp_bwd: an input gradient of the two-array form is not provided (pass the batches as one array)"); return VLN_ERR_ARG; }
  DropBaseScope drop_scope(m->offset_base_dev);
  hipStream_t st = (hipStream_t)s;
  const SavedLayout L = saved_layout(m);
  const int R = m->R, tr = m->training, nl = m->nl;
  float* sc = g->scratch;
  long so = 0;
  auto take = [&](long n) { float* p = sc + so; so += r64(n); return p; };
  // the input BatchNorm's d gamma / d beta from the first layer's weight gradient (vln_bn0_grads_from_wgrad): nothing below the
  // first layer's BatchNorm backward is formed here
  const bool skip0 = g->bn0_from_wgrad != 0;
  if (skip0 && (dx || !tr || !g->defer)) { set_error("vln_bn_mlp_bwd: bn0_from_wgrad needs training mode, no input gradient and deferred parameter jobs"); return VLN_ERR_ARG; }
  vln_wgrad_job wj[VLN_BN_MLP_MAX_LAYERS];
  vln_colsum_job cj[VLN_BN_MLP_MAX_LAYERS];
  int nw = 0, nc = 0;
  const float* gcur = dy;
  long ldg = lddy;
  for (int i = nl - 1; i >= 0; --i) {
    const vln_bn_mlp_layer& l = m->layer[i];
    const int in = (i == 0) ? m->D0 : m->layer[i - 1].out;
    const float* z = saved + L.z[i];
    const float* si = saved + L.s[i];
    const float* yi = saved + L.y[i];
    const float* yprev = (i == 0) ? saved + L.y0 : saved + L.y[i - 1];
    float* dz = take((long)R * l.out);
    const bool last = (i == nl - 1);
    RUN(bn_bwd_seg(z, l.out, gcur, ldg, yi, l.out, l.bn.gamma, tr ? si : l.bn.run_mean, tr ? si + l.out : l.bn.run_var, dz, l.out,
                   g->layer[i].g_gamma, g->layer[i].g_beta, R, m->R1, 2L * l.out, l.out, m->eps, tr, 1, g->layer[i].acc_bn, l.seed, l.offset,
                   l.offset2, tr ? l.p_drop : 0.f, last ? m->row_zero : nullptr, ws, ws_floats, s));
    if (g->layer[i].g_w) wj[nw++] = vln_wgrad_job{dz, yprev, g->layer[i].g_w, l.out, in, in, l.out, in, g->layer[i].acc_w, 0};
    if (g->layer[i].g_b && l.b) cj[nc++] = vln_colsum_job{dz, g->layer[i].g_b, nullptr, l.out, l.out, g->layer[i].acc_b};
    if (skip0 && i == 0) break;
    float* gi = take((long)R * in);
    RUN(gemm_nt(st, dz, l.out, l.w_t, m->wtype, l.out, gi, in, R, in, l.out, nullptr, ACT_NONE, ws, lin_ws(ws_floats, R, in), nullptr));
    gcur = gi; ldg = in;
  }
  // Mt is the same for every product of the call: one grouped launch each for weights and biases.
  // Weight gradients BEHIND a BatchNorm: dz has cancelling column sums (sum_r dz = 0 by construction), so rounding its rows to
  // plain bf16 (precision 2) leaves an absolute error that the small true sums do not hide (3e-2 of the tensor's maximum against
  // the same-weights oracle, round 3).  The plain form is promoted to split operands (hi + lo planes, three MFMAs) here.
  const int prec = g->precision == 2 ? 1 : g->precision;
  if (g->defer) {        // the caller forms them once per rollout (vln_param_jobs)
    for (int i = 0; i < nw; ++i) g->defer->w[i] = wj[i];
    for (int i = 0; i < nc; ++i) g->defer->c[i] = cj[i];
    g->defer->nw = nw; g->defer->nc = nc; g->defer->rows = R; g->defer->precision = prec;
  } else {
    if (nw) RUN(wgrad_grouped(st, wj, nw, R, prec, ws, ws_floats));
    if (nc) RUN(colsum_grouped(st, cj, nc, R, ws, ws_floats));
  }
  if (skip0) return VLN_OK;
  const float* s0 = saved + L.s0;
  RUN(bn_bwd_seg(x, ldx, gcur, ldg, nullptr, 0, m->bn0.gamma, tr ? s0 : m->bn0.run_mean, tr ? s0 + m->D0 : m->bn0.run_var, dx, lddx,
                 g->g_gamma0, g->g_beta0, R, m->R1, 2L * m->D0, m->D0, m->eps, tr, 0, g->acc0, 0, 0, 0, 0.f, nullptr, ws, ws_floats, s,
                 m->x2, m->ldx2));
  return VLN_OK;
}


// ---- the input BatchNorm's parameter gradients from the first layer's weight gradient (include/vln_hip.h) ------------------------------
namespace vln {
struct Bn0FromW {
  const float* dW; const float* db; const float* W; long ldw; const float* gamma; const float* beta;
  float* gW; float* gb; float* gg; float* gbeta; float* part; int N, K, nch, per, acc_w, acc_b, acc_bn; unsigned* sticky;
};
// grid (ceil(K / 256), nch): thread = column k, workgroup row = a chunk of `per` rows n; partial sums -> part[chunk][2][K]
__global__ __launch_bounds__(256) void bn0_from_wgrad_part_kernel(Bn0FromW a) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= a.K) return;
  const int n0 = blockIdx.y * a.per, n1 = min(a.N, n0 + a.per);
  const float g = a.gamma[k], bt = a.beta[k];
  const bool bad = g == 0.f;
  // |gamma_k| far below |beta_k|: (dW - beta db) cancels to ~|gamma| / |beta| of its terms and the quotient amplifies dW's rounding
  // by |beta| / |gamma|.  Above VLN_BN0_MAX_AMPLIFICATION the result is still written but the launch REPORTS it
  // (sticky word 4) instead of handing on a silently degraded d gamma: the caller switches to the direct path.
  const bool ill = !bad && fabsf(g) * VLN_BN0_MAX_AMPLIFICATION < fabsf(bt);
  if ((bad || ill) && blockIdx.y == 0 && a.sticky) atomicAdd(a.sticky + 4, 1u);
  const float inv = bad ? 0.f : 1.f / g;
  float sg = 0.f, sb = 0.f;
  for (int n = n0; n < n1; ++n) {
    const float d = a.dW[(long)n * a.K + k], w = a.W[(long)n * a.ldw + k], s = a.db[n];
    sg += (d - bt * s) * inv * w;
    sb += s * w;
    float* o = a.gW + (long)n * a.K + k;
    *o = a.acc_w ? *o + d : d;
  }
  a.part[((long)blockIdx.y * 2 + 0) * a.K + k] = sg;
  a.part[((long)blockIdx.y * 2 + 1) * a.K + k] = sb;
}
// the chunks' partials in chunk order -> d gamma / d beta; the layer's bias gradient handed on
__global__ __launch_bounds__(256) void bn0_from_wgrad_finish_kernel(Bn0FromW a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.K) {
    float sg = 0.f, sb = 0.f;
    for (int c = 0; c < a.nch; ++c) { sg += a.part[((long)c * 2 + 0) * a.K + i]; sb += a.part[((long)c * 2 + 1) * a.K + i]; }
    if (a.gg) a.gg[i] = a.acc_bn ? a.gg[i] + sg : sg;
    if (a.gbeta) a.gbeta[i] = a.acc_bn ? a.gbeta[i] + sb : sb;
  }
  if (i < a.N && a.gb) a.gb[i] = a.acc_b ? a.gb[i] + a.db[i] : a.db[i];
}
}  // namespace vln

extern "C" int vln_bn0_grads_from_wgrad(const float* dW, const float* db, const float* W, int64_t ldw, const float* gamma, const float* beta,
                                        float* gW, float* gb, float* g_gamma, float* g_beta, int N, int K, int acc_w, int acc_b, int acc_bn,
                                        float* ws, int64_t ws_floats, vln_stream_t s) {
  if (!dW || !db || !W || !gamma || !beta || !gW || !ws || N <= 0 || K <= 0 || ldw < K) { set_error("vln_bn0_grads_from_wgrad: bad args"); return VLN_ERR_ARG; }
  int nch = 32;
  if (nch > N) nch = N;
  if ((int64_t)nch * 2 * K > ws_floats) { set_error("vln_bn0_grads_from_wgrad: workspace below %d * 2 * K floats", nch); return VLN_ERR_ARG; }
  Bn0FromW a{dW, db, W, (long)ldw, gamma, beta, gW, gb, g_gamma, g_beta, ws, N, K, nch, (N + nch - 1) / nch, acc_w, acc_b, acc_bn, sticky_dev_word()};
  a.nch = (N + a.per - 1) / a.per;
  hipStream_t st = (hipStream_t)s;
  VLN_LAUNCH(bn0_from_wgrad_part_kernel, dim3((K + 255) / 256, a.nch), dim3(256), 0, st, a);
  const int most = K > N ? K : N;
  VLN_LAUNCH(bn0_from_wgrad_finish_kernel, dim3((most + 255) / 256), dim3(256), 0, st, a);
  VLN_CHECK_LAUNCH("bn0_grads_from_wgrad");
  return VLN_OK;
}
