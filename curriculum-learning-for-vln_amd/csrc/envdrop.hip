// EnvDropDecoder.forward (reference policy.py:208-246) and its hand-derived backward as ONE C call each:
// the whole step is issued as a fixed chain of launches on the caller's stream with no host sync and a single
// Python->C crossing.  Weight gradients are NOT formed here: backward only produces the dX chain and drops the
// dY operand of every Linear into the rollout stash; the caller forms dW once per optimizer step with one
// contraction over (steps x batch) per weight (vln_linear_wgrad).
//
// Dropout sites (Philox offset = step_offset*8 + site):
//   0 act-embedding (policy.py:224)   1 h_tilde_prev (:234)   2 h_1 (:240)   3 h_tilde (:243)
//   4 image features (:228)           5 candidate features (:230)
#include <string.h>

#include <mutex>
#include <unordered_map>

#include "vln_internal.h"
#include "graph_cache.h"
#include "envdrop_prep.h"
#include "step_bodies.h"
#include "../../include/vln_hip.h"

namespace vln {

__global__ __launch_bounds__(256) void envdrop_prep_kernel(PrepArgs p) {
  envdrop_prep_body(p, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

__global__ __launch_bounds__(256) void envdrop_prep_bwd_kernel(PrepBwdArgs p) {
  envdrop_prep_bwd_body(p, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}
__global__ __launch_bounds__(256) void tanh_drop_bwd_kernel(TanhDropBwdArgs a) {
  tanh_drop_bwd_body(a, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}
__global__ __launch_bounds__(256) void envdrop_prep_tanh_bwd_kernel(PrepTanhBwdArgs a) {
  envdrop_prep_tanh_bwd_body(a, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

static inline int nblocks(long n, int cap = 2048) {
  long b = (n + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// Per-step scratch.  Every skinny product of the step is split over K until ~256 workgroups are in flight and leaves
// its partial slabs in one of four regions; the kernel that consumes the result adds the partials while loading
// (SlabVec), so no reduce launch sits in the dependent chain.  Regions are reused once their consumer has run:
//   s1 [S][B,F]        fwd: visual query, then linear_out scratch, then candidate query      bwd: -
//   s2 [S][B,max(4H,XK)]  fwd: LSTM gate pre-activations                                     bwd: d xcat
//   s3 [S][B,H]        fwd: text query                                                       bwd: d drop(h_tilde), d drop(h1) part 2, d hq
//   s4 [S][B,2H]       fwd: -                                                                bwd: [d weighted ctx | d drop(h1)]
constexpr int kMaxSlabs = 16;
struct Ws {
  float *s1, *s2, *s3, *s4, *dots, *dtcat, *tt;
  long n1, n2, n3, n4;
};
static long ws_layout(const vln_envdrop_dims& d, float* base, Ws* w) {
  const long B = d.B, F = d.IMG + d.ANG, XK = d.AE + F + d.H, H = d.H;
  long off = 0;
  auto take = [&](long n) { long o = off; off += (n + 63) & ~63L; return o; };
  const long n1 = kMaxSlabs * B * F, n2 = kMaxSlabs * B * (4 * H > XK ? 4 * H : XK), n3 = 2 * kMaxSlabs * B * H, n4 = kMaxSlabs * B * 2 * H;
  long o1 = take(n1), o2 = take(n2), o3 = take(n3), o4 = take(n4);
  long smax = d.V; if (d.L > smax) smax = d.L; if (d.C > smax) smax = d.C;
  long o_dots = take(B * smax), o_dtcat = take(B * 2 * H), o_tt = take(B * H);
  if (w) {
    w->s1 = base + o1; w->s2 = base + o2; w->s3 = base + o3; w->s4 = base + o4;
    w->dots = base + o_dots; w->dtcat = base + o_dtcat; w->tt = base + o_tt;
    w->n1 = n1; w->n2 = n2; w->n3 = n3; w->n4 = n4;
  }
  return off;
}

// Dropout site k of this step.  With io->offset_dev the step's offset is read from device memory by the kernels (the
// launch arguments then repeat from call to call, which is what lets the step replay as a hipGraph).
static inline DropSpec site(const vln_envdrop_step* io, int k, float p) {
  if (io->offset_base_dev)      // (*base + offset) * 8 + k
    return DropSpec{io->seed, (uint64_t)k + io->offset * 8, p, reinterpret_cast<const unsigned long long*>(io->offset_base_dev)};
  if (io->offset_dev) return DropSpec{io->seed, (uint64_t)k, p, reinterpret_cast<const unsigned long long*>(io->offset_dev)};
  return DropSpec{io->seed, io->offset * 8 + (uint64_t)k, p};
}
__global__ void set_u64_kernel(unsigned long long* p, unsigned long long v) { *p = v; }

// ---- chained steps (vln_envdrop_step.chain) ------------------------------------------------------------------------------------------
// Two elementwise stages of a step only exist to hand a [B,H] block to the NEXT call on the same stream: the forward's last launch
// (sum of linear_out's split-K slabs, tanh, dropout -> h_tilde, drop(h_tilde)) feeds the next step's first launch (h_tilde_prev
// copy + dropout), and the backward's last launch (prep backward -> d h_tilde_prev) feeds the previous step's first backward
// launch (tanh / dropout backward).  With the chain bits set a step leaves that last stage PENDING (per stream, below) and the
// next step's first launch does both -- one dependent launch less per step and direction.  A pending stage that the next call
// does not consume (other pointers, another entry point, the end of the rollout) is issued on its own first: vln_envdrop_flush.
struct PendFwd { bool on; const float* slabs; int n; long stride; float* ht; float* htd; DropSpec drop; int B, H; };
struct PendBwd { bool on; PrepBwdArgs pa; };
struct PendState { PendFwd f; PendBwd b; };
static std::mutex g_pend_mu;
static std::unordered_map<hipStream_t, PendState> g_pend;
static PendState& pend_of(hipStream_t st) { return g_pend[st]; }      // callers hold g_pend_mu

static int issue_pending_fwd(hipStream_t st, const PendFwd& f) {
  return reduce_epilogue(st, f.slabs, f.n, f.stride, f.H, f.ht, f.H, f.B, f.H, nullptr, ACT_TANH, f.htd, f.H, f.drop);
}
static int issue_pending_bwd(hipStream_t st, const PendBwd& b) {
  VLN_LAUNCH(envdrop_prep_bwd_kernel, dim3(nblocks((long)b.pa.B * (b.pa.AE + b.pa.H))), dim3(256), 0, st, b.pa);
  VLN_CHECK_LAUNCH("envdrop_prep_bwd");
  return VLN_OK;
}

static int check_dims(const vln_envdrop_dims* d) {
  if (!d || d->B <= 0 || d->L <= 0 || d->V <= 0 || d->C <= 0 || d->H <= 0 || d->IMG <= 0 || d->ANG <= 0 || d->AE <= 0) {
    set_error("envdrop: bad dims");
    return VLN_ERR_ARG;
  }
  if ((d->H & 3) || (d->IMG & 3) || (d->ANG & 3) || (d->AE & 3)) {
    set_error("envdrop: H, IMG, ANG, AE must be multiples of 4");
    return VLN_ERR_ARG;
  }
  return VLN_OK;
}

}  // namespace vln

using namespace vln;

#define RUN(x) do { int _s = (x); if (_s != VLN_OK) return _s; } while (0)

extern "C" int64_t vln_envdrop_ws_floats(const vln_envdrop_dims* d) {
  if (check_dims(d) != VLN_OK) return -1;
  return ws_layout(*d, nullptr, nullptr);
}

static int step_fwd_issue(hipStream_t st, const vln_envdrop_dims* d, const vln_envdrop_weights* w, const vln_envdrop_step* io,
                          const PendFwd& use = PendFwd{}) {
  const int B = d->B, H = d->H, F = d->IMG + d->ANG, AE = d->AE, XK = AE + F + H;
  if (io->ws_floats < ws_layout(*d, nullptr, nullptr)) { set_error("envdrop fwd: workspace too small"); return VLN_ERR_ARG; }
  Ws ws; ws_layout(*d, io->ws, &ws);
  const bool lp = (d->ctype == VLN_BF16);
  const auto wt = [&](int bit) { return ((w->f32_mask >> bit) & 1) ? (int)W_F32S : d->wtype; };     // per-matrix fp32 override
  if (lp && (!io->img_lp || !io->cand_lp || !io->ctx_lp)) { set_error("envdrop fwd: bf16 stream copies missing"); return VLN_ERR_ARG; }
  const float pf = io->already_dropfeat ? 0.f : io->p_feat;

  {
    const uintptr_t al = (uintptr_t)io->a_prev | (uintptr_t)io->h_tilde_prev | (uintptr_t)io->hq | (uintptr_t)io->xcat |
                         (uintptr_t)io->a_stash | (uintptr_t)w->act_w;
    if (al & 15) { set_error("envdrop fwd: a_prev, h_tilde_prev, hq, xcat, a_stash and act_w must be 16-byte aligned"); return VLN_ERR_ARG; }
  }
  // (1) act embedding, h_tilde_prev copy + dropout            policy.py:224,234
  PrepArgs pa{io->a_prev, w->act_w, w->act_b, io->h_tilde_prev, io->e, io->xcat, XK, io->hq,
              B, d->ANG, AE, F, H, site(io, 0, io->p_drop), site(io, 1, io->p_drop),
              io->a_stash == io->a_prev ? nullptr : io->a_stash,
              nullptr, 0, 0, nullptr, nullptr, DropSpec{0, 0, 0.f}};
  if (use.on) {       // the previous step's linear_out epilogue rides in this step's first launch (chained steps)
    pa.pend_slabs = use.slabs; pa.pend_n = use.n; pa.pend_stride = use.stride; pa.pend_ht = use.ht; pa.pend_htd = use.htd; pa.pend_drop = use.drop;
  }
  GatherStepArgs ga{};
  if (io->g_table) {
    // (1)+(2) in ONE launch: the step gathers its own feature rows from the resident table (dropout sites 4 / 5 on the way)
    // and the prep work rides along as extra workgroups (features.hip)
    ga = GatherStepArgs{io->g_table, io->g_angle_table, (const long long*)io->g_rows, io->g_vidx, (const long long*)io->g_crows,
                        io->g_cviews, io->g_chead, io->g_celev, lp ? nullptr : io->img, lp ? (bf16_raw*)io->img_lp : nullptr,
                        lp ? nullptr : io->cand, lp ? (bf16_raw*)io->cand_lp : nullptr, B, d->V, d->C, d->IMG, d->ANG,
                        site(io, 4, pf), site(io, 5, pf), gather_check(io->g_table)};
    if (!lp && (!io->img || !io->cand)) { set_error("envdrop fwd: gathered features need img / cand buffers"); return VLN_ERR_ARG; }
    RUN(gather_step_prep(st, ga, io->g_ttype, pa));
  } else {
    VLN_LAUNCH(envdrop_prep_kernel, dim3(nblocks((long)B * AE * 8 + (long)B * (H + (pa.a_stash ? d->ANG : 0)) / 4)), dim3(256), 0, st, pa);
    VLN_CHECK_LAUNCH("envdrop_prep");
    // (2) environmental feature dropout, in place                policy.py:226-231
    // (skipped entirely when the caller already dropped the features and filled the bf16 copies: vln_gather_*)
    const bool need_copy = lp && !io->lp_ready;
    RUN(feat_dropout_inplace(st, io->img, W_F32, (long)B * d->V, d->IMG, d->ANG, site(io, 4, pf), need_copy ? io->img_lp : nullptr));
    RUN(feat_dropout_inplace(st, io->cand, W_F32, (long)B * d->C, d->IMG, d->ANG, site(io, 5, pf), need_copy ? io->cand_lp : nullptr));
  }
  const void* img = lp ? (const void*)io->img_lp : (const void*)io->img;
  const void* cand = lp ? (const void*)io->cand_lp : (const void*)io->cand;
  const void* ctx = lp ? io->ctx_lp : (const void*)io->ctx;
  // (3) visual attention (context-only SoftDot)                policy.py:235, units.py:106-118
  int n1 = 1, n2 = 1, n3 = 1;
  RUN(gemm_nt(st, io->hq, H, w->w_vin, wt(0), H, nullptr, 0, B, F, H, nullptr, ACT_NONE, ws.s1, ws.n1, &n1));
  RUN(attn_fwd_rows_sv(st, img, d->ctype, SlabVec{ws.s1, F, n1, (long)B * F}, nullptr, 0, nullptr, io->alpha_v, io->xcat + AE, XK,
                       ws.dots, B, d->V, F, io->attn_sync, io->attn_sync_bytes));
  // (4) LSTM cell on [drop(e) | visual | h_tilde_prev]         policy.py:237-238
  RUN(gemm_nt(st, io->xcat, XK, w->w_cat, wt(1), XK, nullptr, 0, B, 4 * H, XK, nullptr, ACT_NONE, ws.s2, ws.n2, &n2));
  LstmPwFwd pw{};
  pw.gates = ws.s2; pw.nsplit = n2; pw.slab_stride = (long)B * 4 * H;
  pw.bias_a = w->b_ih; pw.bias_b = w->b_hh; pw.c0 = io->c0; pw.ldc0 = H;
  pw.h1 = io->h1; pw.ldh1 = H; pw.c1 = io->c1; pw.ldc1 = H; pw.act = io->gate_act; pw.tanh_c1 = io->tanh_c1;
  pw.h1_drop = io->tcat + H; pw.ldh1d = 2 * H; pw.drop = site(io, 2, io->p_drop); pw.B = B; pw.H = H;
  if (io->kctx) {
    // (4b)+(5) in ONE launch: the cell's pointwise stage and the text attention on the projected context K = ctx W_in
    // (logits = K . drop(h1): no per-step query product; attention_textk.h)
    RUN(attn_textk_fwd(st, ctx, d->ctype, io->kctx, io->ctx_mask, io->alpha_t, io->tcat, 2 * H, pw, B, d->L, H, io->attn_sync,
                       io->attn_sync_bytes));
  } else {
    RUN(lstm_pointwise_fwd(st, pw));
    // (5) text attention (full SoftDot)                           policy.py:240-241, units.py:106-121
    RUN(gemm_nt(st, io->tcat + H, 2 * H, w->w_tin, wt(2), H, nullptr, 0, B, H, H, nullptr, ACT_NONE, ws.s3, ws.n3, &n3));
    RUN(attn_fwd_rows_sv(st, ctx, d->ctype, SlabVec{ws.s3, H, n3, (long)B * H}, io->tt, H, io->ctx_mask, io->alpha_t, io->tcat, 2 * H,
                         ws.dots, B, d->L, H, io->attn_sync, io->attn_sync_bytes));
  }
  if (io->chain & 1) {        // chained: the slabs stay in ws.s1, the NEXT call's first launch (or vln_envdrop_flush) finishes them
    int nl = 1;
    RUN(gemm_nt(st, io->tcat, 2 * H, w->w_tout, wt(3), 2 * H, nullptr, 0, B, H, 2 * H, nullptr, ACT_NONE, ws.s1, ws.n1, &nl));
    if (nl != gemm_nt_slabs(B, H, 2 * H, wt(3), ws.n1)) { set_error("envdrop fwd: chained epilogue: slab count mismatch"); return VLN_ERR_ARG; }
    return VLN_OK;
  }
  RUN(gemm_nt_fused(st, io->tcat, 2 * H, w->w_tout, wt(3), 2 * H, io->h_tilde, H, B, H, 2 * H, nullptr, ACT_TANH, io->htd, H,
                    site(io, 3, io->p_drop), ws.s1, ws.n1));
  // (6) candidate logits                                        policy.py:243-244,199-206
  if (io->defer_logits) return VLN_OK;      // formed for the whole rollout at once by the caller (vln_attn_dot_multi)
  RUN(gemm_nt(st, io->htd, H, w->w_c, wt(4), H, nullptr, 0, B, F, H, nullptr, ACT_NONE, ws.s1, ws.n1, &n1));
  if (io->s_probs)        // the sampled-action branch rides on the logits' launch (mask, softmax, draw, log-prob, entropy, action to the host)
    return cand_logits_sample(st, cand, d->ctype, SlabVec{ws.s1, F, n1, (long)B * F}, io->logit, io->s_cand_mask, io->s_action_in,
                              io->s_action_out, io->s_action_host, io->s_probs, io->s_logp, io->s_ent, io->s_seed, io->s_offset,
                              io->s_offset_base_dev, B, d->C, F);
  RUN(attn_dot_sv(st, cand, d->ctype, SlabVec{ws.s1, F, n1, (long)B * F}, io->logit, B, d->C, F));
  return VLN_OK;
}

static int step_bwd_issue(hipStream_t st, const vln_envdrop_dims* d, const vln_envdrop_weights* w, const vln_envdrop_step* io,
                          const vln_envdrop_grads* g, const PendBwd& use = PendBwd{}) {
  const int B = d->B, H = d->H, F = d->IMG + d->ANG, AE = d->AE, XK = AE + F + H;
  if (io->ws_floats < ws_layout(*d, nullptr, nullptr)) { set_error("envdrop bwd: workspace too small"); return VLN_ERR_ARG; }
  Ws ws; ws_layout(*d, io->ws, &ws);
  const bool lp = (d->ctype == VLN_BF16);
  const void* img = lp ? (const void*)io->img_lp : (const void*)io->img;
  const void* cand = lp ? (const void*)io->cand_lp : (const void*)io->cand;
  const void* ctx = lp ? io->ctx_lp : (const void*)io->ctx;
  const auto wt = [&](int bit) { return ((w->f32_mask >> bit) & 1) ? (int)W_F32S : d->wtype; };

  // (6') logits -> d(cand query) -> d(drop(h_tilde))
  int n2 = 1, n3 = 1, n3b = 1, n4 = 1;
  SlabVec dhtd{ws.s3, H, 1, (long)B * H};
  const float* dhtd2 = nullptr;
  if (g->dhtd_ext && g->dlogit) {
    // The rollout-wide logit branch (losses.RolloutCE -> logit_branch_backward) covered ITS d logits; `dlogit` is what the
    // OTHER consumers of the same logits sent (sampled log-probs / entropy, a per-step loss, envdrop.py:173-195): their
    // branch runs here and is added -- to the d(cand query) rows of the stash (d cand_attn.weight) and to d drop(h_tilde).
    RUN(rows_wsum(st, cand, d->ctype, g->dlogit, ws.s1, F, B, d->C, F));
    RUN(add_inplace(st, g->s_dtc, F, ws.s1, F, B, F));
    RUN(gemm_nt(st, ws.s1, F, w->w_c_t, wt(4), F, nullptr, 0, B, H, F, nullptr, ACT_NONE, ws.s3, ws.n3, &n3));
    dhtd.n = n3;
    dhtd2 = g->dhtd_ext;
  } else if (g->dhtd_ext) {     // the logit branch of the whole rollout was formed up front (vln_rows_wsum_multi + one GEMM)
    dhtd = SlabVec{g->dhtd_ext, H, 1, (long)B * H};
  } else if (g->dlogit) {
    RUN(rows_wsum(st, cand, d->ctype, g->dlogit, g->s_dtc, F, B, d->C, F));
    RUN(gemm_nt(st, g->s_dtc, F, w->w_c_t, wt(4), F, nullptr, 0, B, H, F, nullptr, ACT_NONE, ws.s3, ws.n3, &n3));
    dhtd.n = n3;
  } else {
    RUN(fill_f32(st, g->s_dtc, (long)B * F, 0.f));
    RUN(fill_f32(st, ws.s3, (long)B * H, 0.f));
  }
  // h_tilde = tanh(.) with dropout on the way to the logits and the external grad on h_tilde itself
  if (use.on) {      // chained: the NEXT step's prep backward (its d h_tilde_prev IS this step's d h_tilde) in the same launch
    const PrepTanhBwdArgs pt{use.pa, dhtd, dhtd2, io->h_tilde, g->s_dz, site(io, 3, io->p_drop)};
    VLN_LAUNCH(envdrop_prep_tanh_bwd_kernel, dim3(nblocks((long)B * (AE + H))), dim3(256), 0, st, pt);
    VLN_CHECK_LAUNCH("envdrop_prep_tanh_bwd");
  } else {
    const TanhDropBwdArgs ta{dhtd, dhtd2, g->dh_tilde, io->h_tilde, g->s_dz, B, H, site(io, 3, io->p_drop)};
    VLN_LAUNCH(tanh_drop_bwd_kernel, dim3(nblocks((long)B * H)), dim3(256), 0, st, ta);
    VLN_CHECK_LAUNCH("tanh_drop_bwd");
  }
  // (5') linear_out -> [d weighted ctx | d drop(h1)], still in slabs (s4)
  RUN(gemm_nt(st, g->s_dz, H, w->w_tout_t, wt(3), H, nullptr, 0, B, 2 * H, H, nullptr, ACT_NONE, ws.s4, ws.n4, &n4));
  const SlabVec dtcat{ws.s4, 2 * H, n4, (long)B * 2 * H};
  // The context gradient is either accumulated in place per step (g->dctx: T read-modify-write sweeps over [B,L,H]) or
  // deferred: this step only leaves d logits (g->s_dl) and d weighted ctx (g->s_dtcat[:, :H]) behind and the caller
  // forms dctx once per rollout with vln_attn_dctx_deferred.
  // (4') the LSTM cell's pointwise backward: its arguments (the K-mode text attention runs it inside its own launch)
  LstmPwBwd pb{};
  pb.dh1_a = g->dh1; pb.ld_a = H;
  pb.drop = site(io, 2, io->p_drop); pb.dc1 = g->dc1; pb.lddc1 = H;
  pb.act = io->gate_act; pb.tanh_c1 = io->tanh_c1; pb.c0 = io->c0; pb.ldc0 = H;
  pb.dgates = g->s_dgates; pb.lddg = 4 * H; pb.dc0 = g->dc0; pb.lddc0 = H; pb.B = B; pb.H = H;
  if (io->kctx) {
    if (g->dctx) { set_error("envdrop bwd: the projected-context step (kctx) needs the deferred context gradient (dctx == NULL)"); return VLN_ERR_ARG; }
    // (5')+(4') in ONE launch: softmax backward, dq = sum dl ctx (the dY rows of d W_in), d drop(h1) = sum dl K + linear_out's
    // share, the cell's pointwise backward
    RUN(attn_textk_bwd(st, ctx, d->ctype, io->kctx, io->alpha_t, dtcat, g->s_dtcat, 2 * H, g->s_dtt, H, g->s_dl, pb, B, d->L, H,
                       io->attn_sync, io->attn_sync_bytes));
  } else {
    if (g->dctx) {
      RUN(reduce_epilogue(st, ws.s4, n4, (long)B * 2 * H, 2 * H, ws.dtcat, 2 * H, B, 2 * H, nullptr, ACT_NONE, nullptr, 0, DropSpec{0, 0, 0.f}));
      RUN(attn_dot(st, ctx, d->ctype, ws.dtcat, 2 * H, ws.dots, B, d->L, H));
      RUN(attn_bwd(st, ctx, d->ctype, io->alpha_t, ws.dots, nullptr, ws.dtcat, 2 * H, io->tt, H, g->s_dtt, H, g->dctx, g->s_dl, B, d->L, H));
    } else {
      RUN(attn_bwd_rows_sv(st, ctx, d->ctype, io->alpha_t, dtcat, g->s_dtcat, 2 * H, nullptr, g->s_dtt, H, g->s_dl, ws.dots, B, d->L, H,
                           io->attn_sync, io->attn_sync_bytes));
    }
    RUN(gemm_nt(st, g->s_dtt, H, w->w_tin_t, wt(2), H, nullptr, 0, B, H, H, nullptr, ACT_NONE, ws.s3, ws.n3, &n3b));
    pb.dh1_b = dtcat.shifted(H); pb.dh1_b2 = SlabVec{ws.s3, H, n3b, (long)B * H};
    RUN(lstm_pointwise_bwd(st, pb));
  }
  RUN(gemm_nt(st, g->s_dgates, 4 * H, w->w_cat_t, wt(1), 4 * H, nullptr, 0, B, XK, 4 * H, nullptr, ACT_NONE, ws.s2, ws.n2, &n2));
  const SlabVec dxcat{ws.s2, XK, n2, (long)B * XK};
  // (3') visual attention: features carry no gradient, only the query does
  RUN(attn_bwd_rows_sv(st, img, d->ctype, io->alpha_v, dxcat.shifted(AE), nullptr, 0, nullptr, g->s_dtv, F, nullptr, ws.dots, B, d->V, F,
                       io->attn_sync, io->attn_sync_bytes));
  RUN(gemm_nt(st, g->s_dtv, F, w->w_vin_t, wt(0), F, nullptr, 0, B, H, F, nullptr, ACT_NONE, ws.s3, ws.n3, &n3));
  // (1') act embedding + the two uses of h_tilde_prev
  if (io->chain & 2) {                       // chained: left pending (bwd_pending below names the same arguments)
    if (n2 != gemm_nt_slabs(B, XK, 4 * H, wt(1), ws.n2) || n3 != gemm_nt_slabs(B, H, F, wt(0), ws.n3)) {
      set_error("envdrop bwd: chained prep backward: slab count mismatch");
      return VLN_ERR_ARG;
    }
    return VLN_OK;
  }
  PrepBwdArgs pa{dxcat, io->e, SlabVec{ws.s3, H, n3, (long)B * H}, g->s_de, g->dh_tilde_prev, B, AE, F, H,
                 site(io, 0, io->p_drop), site(io, 1, io->p_drop)};
  VLN_LAUNCH(envdrop_prep_bwd_kernel, dim3(nblocks((long)B * (AE + H))), dim3(256), 0, st, pa);
  VLN_CHECK_LAUNCH("envdrop_prep_bwd");
  return VLN_OK;
}

// the prep backward a chained step leaves pending: the arguments step_bwd_issue would have launched it with
static PendBwd bwd_pending(const vln_envdrop_dims* d, const vln_envdrop_weights* w, const vln_envdrop_step* io, const vln_envdrop_grads* g) {
  const int B = d->B, H = d->H, F = d->IMG + d->ANG, AE = d->AE, XK = AE + F + H;
  Ws ws; ws_layout(*d, io->ws, &ws);
  const auto wt = [&](int bit) { return ((w->f32_mask >> bit) & 1) ? (int)W_F32S : d->wtype; };
  const int n2 = gemm_nt_slabs(B, XK, 4 * H, wt(1), ws.n2), n3 = gemm_nt_slabs(B, H, F, wt(0), ws.n3);
  const SlabVec dxcat{ws.s2, XK, n2, (long)B * XK};
  return PendBwd{true, PrepBwdArgs{dxcat, io->e, SlabVec{ws.s3, H, n3, (long)B * H}, g->s_de, g->dh_tilde_prev, B, AE, F, H,
                                  site(io, 0, io->p_drop), site(io, 1, io->p_drop)}};
}


// ---- C entry points: one call per decoder step ---------------------------------------------------------------------
// With io->offset_dev the ~13 (forward) / ~15 (backward) launches of a step are a pure function of the argument block:
// the first call with a given block is stream-captured, later calls replay it with ONE hipGraphLaunch (the host cost of
// the launches, not their GPU time, bounds the rollout at B = 64).  Only the 8-byte offset changes per call; it is
// written to device memory by a one-thread launch ahead of the graph.
namespace {
struct StepKey {
  vln_envdrop_dims d; vln_envdrop_weights w; vln_envdrop_step io; vln_envdrop_grads g; int bwd;
  int tun[8];          // the launch plan depends on the run-time tunables: a changed tunable never replays an old graph
  // chained steps: the pending stage of the previous call that this call's first launch carries (all zero: none)
  const void* pend_p[6]; long pend_v[6];
};
}  // namespace

// The stage a chained step left pending on stream `s`, if any, as a launch of its own (end of a rollout's forward / backward,
// or before anything outside vln_envdrop_step_* reads h_tilde / drop(h_tilde) / d h_tilde_prev / the act-embedding gradient rows).
extern "C" int vln_envdrop_flush(vln_stream_t s) {
  hipStream_t st = (hipStream_t)s;
  std::lock_guard<std::mutex> lock(g_pend_mu);
  auto it = g_pend.find(st);
  if (it == g_pend.end()) return VLN_OK;
  PendState& ps = it->second;
  if (ps.f.on) { ps.f.on = false; RUN(issue_pending_fwd(st, ps.f)); }
  if (ps.b.on) { ps.b.on = false; RUN(issue_pending_bwd(st, ps.b)); }
  return VLN_OK;
}

// Forget what a chained step left pending on stream `s` WITHOUT issuing it: the rollout it belongs to was abandoned (an iteration
// that raised), its buffers may be gone.  Returns the number of stages dropped (0, 1 or 2).
extern "C" int vln_envdrop_drop_pending(vln_stream_t s) {
  std::lock_guard<std::mutex> lock(g_pend_mu);
  auto it = g_pend.find((hipStream_t)s);
  if (it == g_pend.end()) return 0;
  const int n = (it->second.f.on ? 1 : 0) + (it->second.b.on ? 1 : 0);
  it->second.f.on = false; it->second.b.on = false;
  return n;
}

extern "C" int vln_envdrop_step_fwd(const vln_envdrop_dims* d, const vln_envdrop_weights* w, vln_envdrop_step* io,
                                    vln_stream_t s) {
  RUN(check_dims(d));
  if (!w || !io) { set_error("vln_envdrop_step_fwd: null pointer"); return VLN_ERR_ARG; }
  hipStream_t st = (hipStream_t)s;
  if (io->s_probs && io->defer_logits) { set_error("vln_envdrop_step_fwd: the in-step sampler needs the step's logits (defer_logits = 0)"); return VLN_ERR_ARG; }
  if ((io->chain & 1) && !io->defer_logits) { set_error("vln_envdrop_step_fwd: chain needs defer_logits (nothing may read h_tilde before the next step)"); return VLN_ERR_ARG; }
  // chained steps: what the previous call left pending on this stream
  PendFwd use{};
  {
    std::lock_guard<std::mutex> lock(g_pend_mu);
    PendState& ps = pend_of(st);
    if (ps.b.on) { ps.b.on = false; RUN(issue_pending_bwd(st, ps.b)); }        // a backward stage left over: not this call's business
    if (ps.f.on) {
      ps.f.on = false;
      if (ps.f.ht == io->h_tilde_prev && ps.f.B == d->B && ps.f.H == d->H) { use = ps.f; use.on = true; }     // consumed by this step's first launch
      else RUN(issue_pending_fwd(st, ps.f));
    }
    // A step whose h_tilde_prev did NOT come out of a chained step (the head of a rollout: the encoder's state) must finish its own
    // backward: whoever consumes its d h_tilde_prev is not a chained step (autograd may even COPY that buffer the moment the
    // step's backward returns -- an AccumulateGrad of a leaf -- long before any flush).
    if (!use.on && (io->chain & 1)) io->chain &= ~2;      // (chain == 2: the caller named the steps that follow another one)
    if (io->chain & 1) {
      Ws ws; ws_layout(*d, io->ws, &ws);
      const int wt3 = ((w->f32_mask >> 3) & 1) ? (int)W_F32S : d->wtype;
      ps.f = PendFwd{true, ws.s1, gemm_nt_slabs(d->B, d->H, 2 * d->H, wt3, ws.n1), (long)d->B * d->H, io->h_tilde, io->htd,
                     site(io, 3, io->p_drop), d->B, d->H};
    }
  }
  if (!io->offset_dev && !io->offset_base_dev) return step_fwd_issue(st, d, w, io, use);
  if (!io->offset_base_dev) {
    VLN_LAUNCH(set_u64_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<unsigned long long*>(io->offset_dev),
                       (unsigned long long)io->offset);
    VLN_CHECK_LAUNCH("envdrop step offset");
  }
  static StepKey key;                     // zero-initialised once: padding bytes stay zero, fields are overwritten
  static std::mutex mu;
  static GraphCache cache;
  std::lock_guard<std::mutex> lock(mu);
  key.d = *d; key.w = *w; key.io = *io; key.bwd = 0;
  if (!io->offset_base_dev) key.io.offset = 0;      // per-step word: the value is not a launch argument
  memset(&key.g, 0, sizeof(key.g));
  memcpy(key.tun, g_tunable, sizeof(key.tun));
  memset(key.pend_p, 0, sizeof(key.pend_p)); memset(key.pend_v, 0, sizeof(key.pend_v));
  if (use.on) {
    key.pend_p[0] = use.slabs; key.pend_p[1] = use.ht; key.pend_p[2] = use.htd; key.pend_p[3] = use.drop.step;
    key.pend_v[0] = use.n; key.pend_v[1] = use.stride; key.pend_v[2] = (long)use.drop.seed; key.pend_v[3] = (long)use.drop.offset;
    memcpy(&key.pend_v[4], &use.drop.p, sizeof(float));
  }
  return cache.run(st, &key, sizeof(key), [&](hipStream_t cs) { return step_fwd_issue(cs, d, w, io, use); });
}

extern "C" int vln_envdrop_step_bwd(const vln_envdrop_dims* d, const vln_envdrop_weights* w, vln_envdrop_step* io,
                                    vln_envdrop_grads* g, vln_stream_t s) {
  RUN(check_dims(d));
  if (!w || !io || !g) { set_error("vln_envdrop_step_bwd: null pointer"); return VLN_ERR_ARG; }
  hipStream_t st = (hipStream_t)s;
  PendBwd use{};
  {
    std::lock_guard<std::mutex> lock(g_pend_mu);
    PendState& ps = pend_of(st);
    if (ps.f.on) { ps.f.on = false; RUN(issue_pending_fwd(st, ps.f)); }        // the rollout's last forward stage (normally flushed by the caller)
    if (ps.b.on) {
      ps.b.on = false;
      // consumed when this step's incoming d h_tilde IS the pending stage's output and this step's first launch reads no
      // workspace slabs of its own (the rollout-wide logit branch covered its d logits: dhtd_ext without dlogit)
      if (g->dh_tilde && ps.b.pa.dhtp == g->dh_tilde && ps.b.pa.B == d->B && ps.b.pa.H == d->H && g->dhtd_ext && !g->dlogit) { use = ps.b; use.on = true; }
      else RUN(issue_pending_bwd(st, ps.b));
    }
    if (io->chain & 2) ps.b = bwd_pending(d, w, io, g);
  }
  if (!io->offset_dev && !io->offset_base_dev) return step_bwd_issue(st, d, w, io, g, use);
  static StepKey key;
  static std::mutex mu;
  static GraphCache cache;
  std::lock_guard<std::mutex> lock(mu);
  key.d = *d; key.w = *w; key.io = *io; key.g = *g; key.bwd = 1;
  if (!io->offset_base_dev) key.io.offset = 0;
  memcpy(key.tun, g_tunable, sizeof(key.tun));
  memset(key.pend_p, 0, sizeof(key.pend_p)); memset(key.pend_v, 0, sizeof(key.pend_v));
  if (use.on) {
    const PrepBwdArgs& q = use.pa;
    key.pend_p[0] = q.dxcat.p; key.pend_p[1] = q.dhq.p; key.pend_p[2] = q.e; key.pend_p[3] = q.s_de; key.pend_p[4] = q.dhtp; key.pend_p[5] = q.d_act.step;
    key.pend_v[0] = q.dxcat.n; key.pend_v[1] = q.dhq.n; key.pend_v[2] = (long)q.d_act.seed; key.pend_v[3] = (long)q.d_act.offset;
    key.pend_v[4] = (long)q.d_h.offset; memcpy(&key.pend_v[5], &q.d_act.p, sizeof(float));
  }
  return cache.run(st, &key, sizeof(key), [&](hipStream_t cs) { return step_bwd_issue(cs, d, w, io, g, use); });
}
