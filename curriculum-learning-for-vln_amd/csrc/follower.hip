// AttnDecoderLSTM.forward (reference policy.py:37-60) + ActionScoring (units.py:163-185) and the hand-derived backward as ONE C
// call each: the launch sequence of functional.FollowerCoreFn (17 forward / ~30 backward launches) issued by the library instead
// of one ctypes call per launch from Python.
//
//   forward   alpha_v, pano = VisualSoftDot(h0, img): logits_v = img_v . (W_v^T (W_h h0 + b_h)) [= (W_v img_v + b_v) . query up to a
//             per-episode constant], pano = sum_v alpha_v img_v
//             x = drop([a_prev | pano]);  h1, c1 = LSTMCell(x, (h0, c0))                       policy.py:46-52
//             grounded, alpha_c = SoftDot(drop(h1), ctx, ctx_mask)                             policy.py:54-55
//             logit = w_out . ((W_act cands + b_act) (.) (W_hid grounded + b_hid)) + b_out     units.py:175-184
//   backward  the mirrored chain.  The two projections over many rows (36 views, C candidates) never form their [B*S, dot]
//             gradient: dW_v = tq^T (sum_v dl_v img_v), dW_act = q^T (sum_c dlogit_c cand_c); the eight weight gradients of
//             the step in ONE grouped launch over the B rows, the bias / head gradients in another.
// Dropout sites: `off` (the LSTM input row) and `off + 1` (h1), seed `seed`.
#include "vln_internal.h"
#include "../../include/vln_hip.h"

namespace vln {
int copy_blocks2(hipStream_t st, int rows, const float* s0, long lds0, float* d0, long ldd0, int cols0, const float* s1, long lds1,
                 float* d1, long ldd1, int cols1);      // monitor.hip

static int check_follower_dims(const vln_follower_dims* d) {
  if (!d || d->B <= 0 || d->L <= 0 || d->V <= 0 || d->C <= 0 || d->H <= 0 || d->F <= 0 || d->A <= 0 || d->D <= 0) {
    set_error("follower step: bad dims");
    return VLN_ERR_ARG;
  }
  if ((d->H & 3) || (d->F & 3) || (d->A & 3) || (d->D & 3)) { set_error("follower step: H, F, A and D must be multiples of 4"); return VLN_ERR_ARG; }
  return VLN_OK;
}
}  // namespace vln

using namespace vln;

#define RUN(x) do { int _s = (x); if (_s != VLN_OK) return _s; } while (0)

extern "C" int64_t vln_follower_bwd_scratch_floats(const vln_follower_dims* d) {
  if (check_follower_dims(d) != VLN_OK) return -1;
  const long B = d->B, L = d->L, V = d->V, C = d->C, H = d->H, F = d->F, A = d->A, D = d->D, XK = A + F + H;
  long n = 0;
  auto take = [&](long k) { n += (k + 63) & ~63L; };
  take(B * D); take(B * D); take(B * D); take(B * A); take(B * D); take(B);          // dq, dtarget, Zo, rc, qs, sl
  take(B * H); take(B * H); take(B * 2 * H); take(B * H); take(B * L); take(B * H); take(B * H);   // dgr, dz, dtcat, dtq2, dl_t, t1, dhd
  take(B * 4 * H); take(B * XK); take(B * V); take(B * V); take(B * F); take(B * D); take(B * D); take(B * H);   // dg, dxcat, dalpha, dl_v, rv, dtq, tqs, t2
  take(B * C);                                                                      // zero d logits
  return n;
}

extern "C" int vln_follower_step_fwd(const vln_follower_dims* d, const vln_follower_weights* w, vln_follower_step* io, vln_stream_t s) {
  RUN(check_follower_dims(d));
  if (!w || !io || !io->img || !io->a_prev || !io->cands || !io->h0 || !io->c0 || !io->ctx || !io->ws || !w->w_v_t || !io->keys) {
    set_error("vln_follower_step_fwd: null pointer");
    return VLN_ERR_ARG;
  }
  DropBaseScope drop_scope(io->offset_base_dev);
  hipStream_t st = (hipStream_t)s;
  const int B = d->B, L = d->L, V = d->V, C = d->C, H = d->H, F = d->F, A = d->A, D = d->D, XK = A + F + H, wt = d->wtype;
  // (1) panorama attention (units.py:144-159): logits_v = (W_v img_v + b_v) . tq with tq = W_h h0 + b_h.  Taken as img_v . (W_v^T tq):
  // b_v . tq is one constant per episode under the softmax, and the product over the B * V view rows ([2304, 2176] x [256, 2176]^T
  // per step at BASELINE config 0's model) becomes a B-row one -- the keys are never formed.  Weighted sum over the UN-projected views.
  RUN(gemm_nt(st, io->h0, H, w->w_h, wt, H, io->tq, D, B, D, H, w->b_h, ACT_NONE, io->ws, io->ws_floats, nullptr));
  {
    SlabArea ar{io->ws, (long)io->ws_floats};
    SlabVec vq;
    RUN(gemm_nt_to_consumer(st, ar, io->tq, D, w->w_v_t, wt, D, io->keys, F, B, F, D, nullptr, &vq));
    RUN(attn_fwd_rows_sv(st, io->img, W_F32, vq, nullptr, 0, nullptr, io->view_w, io->xcat + A, XK, io->dots, B, V, F, io->attn_sync,
                         io->attn_sync_bytes));
  }
  {   // xcat = [drop(a_prev) | drop(pano), in place | h0]: the dropout over cat(a_prev, pano) (policy.py:49-51) in the launch that
      // copies the two blocks (they were a launch each)
    const DropSpec dr = tls_drop(io->seed, io->off, io->p_drop);
    AddNSvJob aj[3] = {{io->xcat, XK, B, A, 1, {plain_vec(io->a_prev, A), SlabVec{}, SlabVec{}, SlabVec{}}, dr, A + F, 0},
                       {io->xcat + A, XK, B, F, 1, {plain_vec(io->xcat + A, XK), SlabVec{}, SlabVec{}, SlabVec{}}, dr, A + F, A},
                       {io->xcat + A + F, XK, B, H, 1, {plain_vec(io->h0, H), SlabVec{}, SlabVec{}, SlabVec{}}}};
    RUN(add_n_sv_multi(st, aj, 3));
  }
  // (2) LSTM cell; drop(h1) lands in its tcat block
  // the product's K-chunks stay split-K slabs in the workspace: the pointwise launch sums them while it loads (no reduce launch)
  int gate_slabs = 1;
  if (!io->ws || io->ws_floats < (int64_t)B * 4 * H) { set_error("vln_follower_step_fwd: the gate product's slabs need a workspace of >= B * 4H floats"); return VLN_ERR_ARG; }
  RUN(gemm_nt(st, io->xcat, XK, w->w_cat, wt, XK, nullptr, 0, B, 4 * H, XK, nullptr, ACT_NONE, io->ws, io->ws_floats, &gate_slabs));
  {
    LstmPwFwd a{};
    a.gates = io->ws; a.nsplit = gate_slabs; a.slab_stride = (long)B * 4 * H; a.bias_a = w->b_ih; a.bias_b = w->b_hh;
    a.c0 = io->c0; a.ldc0 = H; a.h1 = io->h1; a.ldh1 = H; a.c1 = io->c1; a.ldc1 = H; a.act = io->act; a.tanh_c1 = io->tanh_c1;
    a.h1_drop = io->tcat + H; a.ldh1d = 2 * H; a.drop = tls_drop(io->seed, io->off + 1, io->p_drop); a.B = B; a.H = H;
    RUN(lstm_pointwise_fwd(st, a));
  }
  // (3) text attention + tanh(W_out [wc ; drop(h1)])
  RUN(gemm_nt(st, io->tcat + H, 2 * H, w->w_tin, wt, H, io->tq2, H, B, H, H, nullptr, ACT_NONE, io->ws, io->ws_floats, nullptr));
  RUN(attn_fwd_rows_sv(st, io->ctx, W_F32, plain_vec(io->tq2, H), nullptr, 0, io->ctx_mask, io->word_w, io->tcat, 2 * H, io->dots, B, L, H,
                       io->attn_sync, io->attn_sync_bytes));
  RUN(gemm_nt(st, io->tcat, 2 * H, w->w_tout, wt, 2 * H, io->grounded, H, B, H, 2 * H, nullptr, ACT_TANH, io->ws, io->ws_floats, nullptr));
  // (4) candidate scores: logit = context . (target (.) w_out) + b_out
  RUN(gemm_nt(st, io->grounded, H, w->w_hid, wt, H, io->target, D, B, D, H, w->b_hid, ACT_NONE, io->ws, io->ws_floats, nullptr));
  if (!io->context_ready)     // (formed for the whole rollout up front by the caller when it is teacher-forced: vln_follower_step.context_ready)
    RUN(gemm_nt(st, io->cands, A, w->w_act, wt, A, io->context, D, B * C, D, A, w->b_act, ACT_NONE, io->ws, io->ws_floats, nullptr));
  // q = target (.) w_out and the + b_out inside the dot launch (they were a launch each); q written back for the backward
  RUN(attn_dot_sv(st, io->context, W_F32, plain_vec(io->target, D), io->logit, B, C, D, io->q, D, w->w_out, w->b_out));
  return VLN_OK;
}

extern "C" int vln_follower_step_bwd(const vln_follower_dims* d, const vln_follower_weights* w, const vln_follower_step* io,
                                     const vln_follower_grads* g, vln_stream_t s) {
  RUN(check_follower_dims(d));
  if (!w || !io || !g || !g->scratch || !g->dh0 || !g->dc0) { set_error("vln_follower_step_bwd: null pointer"); return VLN_ERR_ARG; }
  if (g->scratch_floats < vln_follower_bwd_scratch_floats(d)) { set_error("vln_follower_step_bwd: scratch too small"); return VLN_ERR_ARG; }
  DropBaseScope drop_scope(io->offset_base_dev);
  hipStream_t st = (hipStream_t)s;
  const int B = d->B, L = d->L, V = d->V, C = d->C, H = d->H, F = d->F, A = d->A, D = d->D, XK = A + F + H, wt = d->wtype;
  float* p = g->scratch;
  auto take = [&](long k) { float* r = p; p += (k + 63) & ~63L; return r; };
  float *dq = take((long)B * D), *dtarget = take((long)B * D), *Zo = take((long)B * D), *rc = take((long)B * A), *qs = take((long)B * D), *sl = take(B);
  float *dgr = take((long)B * H), *dz = take((long)B * H), *dtcat = take((long)B * 2 * H), *dtq2 = take((long)B * H), *dl_t = take((long)B * L);
  float *t1 = take((long)B * H), *dhd = take((long)B * H);
  float *dg = take((long)B * 4 * H), *dxcat = take((long)B * XK), *dalpha = take((long)B * V), *dl_v = take((long)B * V), *rv = take((long)B * F);
  float *dtq = take((long)B * D), *tqs = take((long)B * D), *t2 = take((long)B * H), *zlogit = take((long)B * C);
  const float* dlogit = g->dlogit;
  if (!dlogit) { RUN(fill_f32(st, zlogit, (long)B * C, 0.f)); dlogit = zlogit; }
  // (4) scores: logit = context . q + b_out, q = target (.) w_out, context = W_act cands + b_act
  RUN(rows_wsum(st, io->context, W_F32, dlogit, dq, D, B, C, D));
  {   // four independent row-wise forms of dq / d logits in ONE launch (they were four)
    const EwJob ej[4] = {{0, dq, D, w->w_out, 0, 0, dtarget, D, B, D},
                         {0, dq, D, io->target, D, 0, Zo, D, B, D},                  // colsum -> d w_out
                         {3, io->q, D, dlogit, C, C, qs, D, B, D},                   // colsum -> d b_act
                         {3, nullptr, 0, dlogit, C, C, sl, 1, B, 1}};                // row sums; colsum -> d b_out
    RUN(ew_multi(st, ej, 4));
  }
  RUN(rows_wsum(st, io->cands, W_F32, dlogit, rc, A, B, C, A));                     // sum_c dlogit_c cand_c -> d W_act = q^T rc
  RUN(gemm_nt(st, dtarget, D, w->w_hid_t, wt, D, dgr, H, B, H, D, nullptr, ACT_NONE, io->ws, io->ws_floats, nullptr));
  // (3) grounded = tanh(W_out tcat)
  RUN(vln_ew(2, dgr, H, io->grounded, H, 0, dz, H, B, H, s));
  RUN(gemm_nt(st, dz, H, w->w_tout_t, wt, H, dtcat, 2 * H, B, 2 * H, H, nullptr, ACT_NONE, io->ws, io->ws_floats, nullptr));   // -> wc | drop(h1)
  RUN(attn_bwd_rows_sv(st, io->ctx, W_F32, io->word_w, plain_vec(dtcat, 2 * H), nullptr, 0, g->dww_ext, dtq2, H, dl_t, io->dots, B, L, H,
                       io->attn_sync, io->attn_sync_bytes));
  if (g->dctx_term) {
    *g->dctx_term = vln_dctx_term{io->word_w, dl_t, dtcat, io->tq2, 2L * H, H, 0, 0, 0.f, 0.f};
  } else if (g->dctx) {
    const float* al[1] = {io->word_w};
    const float* dl[1] = {dl_t};
    const float* gg[1] = {dtcat};
    const float* qq[1] = {io->tq2};
    RUN(attn_dctx_deferred(st, al, dl, gg, 2 * H, qq, H, 1, g->dctx, B, L, H, g->dctx_accumulate));
  }
  RUN(gemm_nt(st, dtq2, H, w->w_tin_t, wt, H, t1, H, B, H, H, nullptr, ACT_NONE, io->ws, io->ws_floats, nullptr));
  // (2) LSTM cell; d drop(h1) = dtcat[:, H:] + t1, added by the pointwise launch (it was a launch of its own)
  (void)dhd;
  {
    LstmPwBwd a{};
    a.dh1_a = g->dh1; a.ld_a = H; a.dh1_b = plain_vec(dtcat + H, 2 * H); a.dh1_b2 = plain_vec(t1, H);
    a.drop = tls_drop(io->seed, io->off + 1, io->p_drop); a.dc1 = g->dc1; a.lddc1 = H; a.act = io->act; a.tanh_c1 = io->tanh_c1;
    a.c0 = io->c0; a.ldc0 = H; a.dgates = dg; a.lddg = 4 * H; a.dc0 = g->dc0; a.lddc0 = H; a.B = B; a.H = H;
    RUN(lstm_pointwise_bwd(st, a));
  }
  {   // d xcat -> a_prev | pano | h0: the product's split-K slabs are summed AND the input dropout's mask applied by one launch
      // (they were a reduce launch and a dropout launch)
    // The previous action's block of d xcat (columns [0, A): 2176 of 4608) is wanted only when a_prev carries a gradient -- it is a
    // feature row in the reference's agents -- so the product starts at column c0 = A then (rows c0.. of the transposed weight).
    SlabArea ar{io->ws, (long)io->ws_floats};
    SlabVec s_dxcat;
    const int c0 = g->da_prev ? 0 : A;
    const char* wct = static_cast<const char*>(w->w_cat_t) + (size_t)c0 * 4 * H * (wt == W_BF16 ? 2 : 4);
    RUN(gemm_nt_to_consumer(st, ar, dg, 4 * H, wct, wt, 4 * H, dxcat + c0, XK, B, XK - c0, 4 * H, nullptr, &s_dxcat));
    if (s_dxcat.p != dxcat + c0 || io->p_drop > 0.f) {
      AddNSvJob aj[2] = {{dxcat + c0, XK, B, A + F - c0, 1, {s_dxcat, SlabVec{}, SlabVec{}, SlabVec{}}, tls_drop(io->seed, io->off, io->p_drop), A + F, c0},
                         {dxcat + A + F, XK, B, H, 1, {s_dxcat.shifted(A + F - c0), SlabVec{}, SlabVec{}, SlabVec{}}}};
      RUN(add_n_sv_multi(st, aj, s_dxcat.p != dxcat + c0 ? 2 : 1));
    }
  }
  // (1) panorama attention: pano = sum_v alpha_v img_v, alpha = softmax(keys . tq); rv = sum_v dl_v img_v comes out of the same pass
  // (round 6: the dots and the softmax backward in ONE launch, on four workgroups per episode when the exchange buffer is given; the
  // two-launch pair of rounds 1-5 when the shape fits neither one-launch form)
  (void)dalpha;
  RUN(attn_bwd_rows_sv(st, io->img, W_F32, io->view_w, plain_vec(dxcat + A, XK), nullptr, 0, g->dvw_ext, rv, F, dl_v, io->dots, B, V, F,
                       io->attn_sync, io->attn_sync_bytes));
  // logits_v = img_v . (W_v^T tq): d(W_v^T tq) = sum_v dl_v img_v = rv, so d tq = rv W_v^T (a B-row product) and d W_v = tq^T rv
  // (below); d b_v is exactly 0 (b_v . tq shifts every view's logit of an episode alike)
  (void)tqs;
  RUN(gemm_nt(st, rv, F, w->w_v, wt, F, dtq, D, B, D, F, nullptr, ACT_NONE, io->ws, io->ws_floats, nullptr));
  RUN(gemm_nt(st, dtq, D, w->w_h_t, wt, D, t2, H, B, H, D, nullptr, ACT_NONE, io->ws, io->ws_floats, nullptr));
  {   // d h0 and (optionally) d a_prev, one launch
    const AddNJob aj[2] = {{g->dh0, H, B, H, 2, {dxcat + A + F, t2, nullptr, nullptr}, {XK, H, 0, 0}},
                           {g->da_prev, A, B, A, 1, {dxcat, nullptr, nullptr, nullptr}, {XK, 0, 0, 0}}};
    RUN(add_n_multi(st, aj, g->da_prev ? 2 : 1));
  }
  if (g->g_bv && !g->acc[3]) RUN(fill_f32(st, g->g_bv, D, 0.f));      // d b_v = 0 exactly (see the panorama attention above)
  // parameter gradients: eight products over the same B rows -> one grouped launch; the biases and the head -> another
  {
    vln_wgrad_job jobs[8];
    int n = 0;
    auto add = [&](const float* dy, long ldy, const float* x, long ldx, float* dw, long ldw, int N, int K, int acc) {
      if (dw) jobs[n++] = vln_wgrad_job{dy, x, dw, ldy, ldx, ldw, N, K, acc, 0};
    };
    add(dtq, D, io->xcat + A + F, XK, g->g_wh, H, D, H, g->acc[0]);      // X = h0, from its copy inside xcat (see vln_param_jobs)
    add(io->tq, D, rv, F, g->g_wv, F, D, F, g->acc[2]);
    add(dg, 4 * H, io->xcat, XK, g->g_ih, A + F, 4 * H, A + F, g->acc[4]);
    add(dg, 4 * H, io->xcat + A + F, XK, g->g_hh, H, 4 * H, H, g->acc[5]);
    add(dtq2, H, io->tcat + H, 2 * H, g->g_tin, H, H, H, g->acc[8]);
    add(dz, H, io->tcat, 2 * H, g->g_tout, 2 * H, H, 2 * H, g->acc[9]);
    add(io->q, D, rc, A, g->g_wact, A, D, A, g->acc[10]);
    add(dtarget, D, io->grounded, H, g->g_whid, H, D, H, g->acc[12]);
    if (g->defer) { for (int i = 0; i < n; ++i) g->defer->w[i] = jobs[i]; g->defer->nw = n; g->defer->rows = B; g->defer->precision = g->precision; }
    else if (n) RUN(wgrad_grouped(st, jobs, n, B, g->precision, io->ws, io->ws_floats));
  }
  {
    vln_colsum_job jobs[8];
    int n = 0;
    auto add = [&](const float* Am, long lda, float* o1, float* o2, int cols, int acc) {
      if (o1) jobs[n++] = vln_colsum_job{Am, o1, o2, lda, cols, acc};
    };
    add(dtq, D, g->g_bh, nullptr, D, g->acc[1]);
    add(qs, D, g->g_bact, nullptr, D, g->acc[11]);
    add(dtarget, D, g->g_bhid, nullptr, D, g->acc[13]);
    add(Zo, D, g->g_wout, nullptr, D, g->acc[14]);
    if (g->g_bih && g->g_bhh && g->acc[6] == g->acc[7]) add(dg, 4 * H, g->g_bih, g->g_bhh, 4 * H, g->acc[6]);
    else { add(dg, 4 * H, g->g_bih, nullptr, 4 * H, g->acc[6]); add(dg, 4 * H, g->g_bhh, nullptr, 4 * H, g->acc[7]); }
    if (g->defer) { for (int i = 0; i < n; ++i) g->defer->c[i] = jobs[i]; g->defer->nc = n; }
    else if (n) RUN(colsum_grouped(st, jobs, n, B, io->ws, io->ws_floats));
    if (g->g_bout) RUN(colsum(st, sl, 1, g->g_bout, B, 1, g->acc[15], io->ws, io->ws_floats));
  }
  return VLN_OK;
}
