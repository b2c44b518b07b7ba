// MonitorDecoder.forward (reference policy.py:132-166: co-grounding step + progress monitor) after its BN-MLP, and the
// hand-derived backward, as ONE C call each: a fixed chain of launches on the caller's stream, no host sync, a single
// Python->C crossing (round 1 drove the same ~20 + ~25 launches from Python, one ctypes call each: the Self-Monitor agent at
// B = 128 was bound by that Python, 9.2 ms per iteration for 6.9 ms of kernels).
//
//   forward   positioned context pctx = dropout(ctx + pe)                       units.py:188-207, policy.py:152
//             words, word_w = SoftDot_ctx_only(h0, pctx, ctx_mask)              policy.py:153, units.py:106-118
//             moves, move_w = VisualSoftDot(h0, cand_rep, cand_mask)            policy.py:155, units.py:144-159
//             h1, c1 = LSTMCell([prev_rep | moves | words], (h0, c0))           policy.py:157-158
//             logit = cand_rep . (W_a [words ; drop(h1)] + b_a)                 policy.py:108-117,160
//             prog = tanh(w_c . [word_w ; drop(sigmoid(W_m [h0 ; moves] + b_m) * tanh(c1))] + b_c)   policy.py:119-130,162
//   backward  the mirrored chain; the six weight gradients of the step in ONE grouped launch (B rows each), the bias / head
//             gradients in another, accumulated into the caller's gradient buffers.
// Dropout sites: pe (seed_pe, off_pe), h1 (seed, off_h1), progress-monitor memory (seed, off_mem).
#include "vln_internal.h"
#include "../../include/vln_hip.h"

namespace vln {

struct CopyJobs { const float* src[4]; float* dst[4]; long lds[4], ldd[4]; int cols[4]; int n, rows; };
// up to four [rows, cols] row-block copies in one launch (the concatenated operands of the step: [prev | moves | words | h0] ...)
__global__ __launch_bounds__(256) void copy_blocks_kernel(CopyJobs j) {
  for (int k = 0; k < j.n; ++k) {
    const int c4 = j.cols[k] >> 2;
    const long total = (long)j.rows * c4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
      const long r = e / c4, c = (e % c4) * 4;
      *reinterpret_cast<float4*>(j.dst[k] + r * j.ldd[k] + c) = *reinterpret_cast<const float4*>(j.src[k] + r * j.lds[k] + c);
    }
  }
}
static int copy_blocks(hipStream_t st, const CopyJobs& j) {
  long most = 0;
  for (int k = 0; k < j.n; ++k) {
    if ((j.cols[k] & 3) || (j.lds[k] & 3) || (j.ldd[k] & 3) || ((uintptr_t)j.src[k] & 15) || ((uintptr_t)j.dst[k] & 15)) {
      set_error("monitor step: row blocks must be 16-byte aligned with widths that are multiples of 4");
      return VLN_ERR_ARG;
    }
    const long t = (long)j.rows * (j.cols[k] >> 2);
    if (t > most) most = t;
  }
  int blocks = (int)((most + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  VLN_LAUNCH(copy_blocks_kernel, dim3(blocks), dim3(256), 0, st, j);
  VLN_CHECK_LAUNCH("copy_blocks");
  return VLN_OK;
}

// two row-block copies (the second optional) -- shared with follower.hip
int copy_blocks2(hipStream_t st, int rows, const float* s0, long lds0, float* d0, long ldd0, int cols0, const float* s1, long lds1,
                 float* d1, long ldd1, int cols1) {
  CopyJobs j{};
  j.rows = rows; j.n = s1 ? 2 : 1;
  j.src[0] = s0; j.lds[0] = lds0; j.dst[0] = d0; j.ldd[0] = ldd0; j.cols[0] = cols0;
  j.src[1] = s1; j.lds[1] = lds1; j.dst[1] = d1; j.ldd[1] = ldd1; j.cols[1] = cols1;
  return copy_blocks(st, j);
}

static int check_monitor_dims(const vln_monitor_dims* d) {
  if (!d || d->B <= 0 || d->L <= 0 || d->C <= 0 || d->H <= 0 || d->M <= 0) { set_error("monitor step: bad dims"); return VLN_ERR_ARG; }
  if ((d->H & 3) || (d->M & 3) || (d->L & 3)) { set_error("monitor step: H, M and L must be multiples of 4"); return VLN_ERR_ARG; }
  return VLN_OK;
}

}  // namespace vln

using namespace vln;

#define RUN(x) do { int _s = (x); if (_s != VLN_OK) return _s; } while (0)

// floats of the backward's temporaries (vln_monitor_grads.scratch)
extern "C" int64_t vln_monitor_bwd_scratch_floats(const vln_monitor_dims* d) {
  if (check_monitor_dims(d) != VLN_OK) return -1;
  const long B = d->B, L = d->L, C = d->C, H = d->H, M = d->M, XK = 2 * M + 2 * H;
  long n = 0;
  auto take = [&](long k) { n += (k + 63) & ~63L; };
  take(B * H); take(B * H); take(B * L); take(B * (L + H)); take(B);          // dmg, dc1_t, dww, Z, dpre
  take(B * (H + M)); take(B * M); take(B * 2 * H); take(B * 4 * H); take(B * XK);   // dhm, daq, dtcat, dg, dxcat
  take(B * M); take(B * H); take(B * M); take(B * C); take(B * H);            // dmoves, dwords, dvq, dl_v, dh0_v
  take(B * H); take(B * L); take(B * H); take(B * C);                        // dtq, dl_t, dh0_t, zero d logits
  return n;
}

// floats of the step's workspace (vln_monitor_step.ws): the backward keeps the slabs of its five skinny products live at once
// (gemm_nt_to_consumer); a smaller workspace still works -- a product that does not fit is reduced by its own launch.
extern "C" int64_t vln_monitor_ws_floats(const vln_monitor_dims* d) {
  if (check_monitor_dims(d) != VLN_OK) return -1;
  const int B = d->B, H = d->H, M = d->M, XK = 2 * M + 2 * H;
  const int nk[5][2] = {{H + M, H}, {2 * H, M}, {XK, 4 * H}, {H, M}, {H, H}};
  long n = 0;
  for (int i = 0; i < 5; ++i) {
    int most = 1;
    for (int wt : {(int)W_F32, (int)W_BF16, (int)W_F32S}) {
      const int k = gemm_nt_slabs(B, nk[i][0], nk[i][1], wt, 1L << 40);
      if (k > most) most = k;
    }
    n += ((long)most * B * nk[i][0] + 63) & ~63L;
  }
  return n < (1L << 22) ? (1L << 22) : n;
}

extern "C" int vln_monitor_step_fwd(const vln_monitor_dims* d, const vln_monitor_weights* w, vln_monitor_step* io, vln_stream_t s) {
  RUN(check_monitor_dims(d));
  if (!w || !io || !io->prev_rep || !io->cand_rep || !io->h0 || !io->c0 || !io->ctx || !io->ctx_mask || !io->cand_mask || !io->ws) {
    set_error("vln_monitor_step_fwd: null pointer");
    return VLN_ERR_ARG;
  }
  DropBaseScope drop_scope(io->offset_base_dev);
  hipStream_t st = (hipStream_t)s;
  const int B = d->B, L = d->L, C = d->C, H = d->H, M = d->M, XK = 2 * M + 2 * H;
  const auto wt = [&](int bit) { return ((w->f32_mask >> bit) & 1) ? (int)W_F32S : d->wtype; };     // per-matrix fp32 override
  // (1) positioned, dropped context (fresh mask per step, units.py:205-207)
  RUN(vln_pe_dropout(io->ctx, w->pe, io->pctx, B, L, H, io->seed_pe, io->off_pe, io->p_pe, s));
  // (2) the row blocks that need no computing: xcat = [prev_rep | . | . | h0], hm = [h0 | .]
  {
    CopyJobs j{};
    j.n = 3; j.rows = B;
    j.src[0] = io->prev_rep; j.lds[0] = M; j.dst[0] = io->xcat; j.ldd[0] = XK; j.cols[0] = M;
    j.src[1] = io->h0; j.lds[1] = H; j.dst[1] = io->xcat + 2 * M + H; j.ldd[1] = XK; j.cols[1] = H;
    j.src[2] = io->h0; j.lds[2] = H; j.dst[2] = io->hm; j.ldd[2] = H + M; j.cols[2] = H;
    RUN(copy_blocks(st, j));
  }
  // The products of the step go to consumers that sum the split-K slabs themselves (gemm_nt_to_consumer: no reduce launch between
  // a product and its reader; the reader writes the summed vector back where the backward wants it).
  SlabArea ar{io->ws, (long)io->ws_floats};
  const auto area_reset = [&]() { ar = SlabArea{io->ws, (long)io->ws_floats}; };
  SlabVec sv;
  // (3) text attention over the positioned context, weighted context straight into its xcat block
  RUN(gemm_nt_to_consumer(st, ar, io->h0, H, w->w_tin, wt(0), H, io->tq, H, B, H, H, nullptr, &sv));
  RUN(attn_fwd_rows_sv(st, io->pctx, W_F32, sv, sv.p != io->tq ? io->tq : nullptr, H, io->ctx_mask, io->word_w, io->xcat + 2 * M, XK,
                       io->dots, B, L, H));
  area_reset();
  // (4) attention over the projected candidates (padded slots masked)
  RUN(gemm_nt_to_consumer(st, ar, io->h0, H, w->w_vh, wt(1), H, io->vq, M, B, M, H, w->b_vh, &sv));
  RUN(attn_fwd_rows_sv(st, io->cand_rep, W_F32, sv, sv.p != io->vq ? io->vq : nullptr, M, io->cand_mask, io->move_w, io->xcat + M, XK, io->dots,
                       B, C, M));
  area_reset();
  // (5) LSTM cell on [prev_rep | moves | words | h0]; drop(h1) lands in its tcat block
  // the product's K-chunks stay split-K slabs in the workspace: the pointwise launch sums them while it loads (no reduce launch)
  int gate_slabs = 1;
  if (!io->ws || io->ws_floats < (int64_t)B * 4 * H) { set_error("vln_monitor_step_fwd: the gate product's slabs need a workspace of >= B * 4H floats"); return VLN_ERR_ARG; }
  RUN(gemm_nt(st, io->xcat, XK, w->w_cat, wt(2), XK, nullptr, 0, B, 4 * H, XK, nullptr, ACT_NONE, io->ws, io->ws_floats, &gate_slabs));
  {
    LstmPwFwd a{};
    a.gates = io->ws; a.nsplit = gate_slabs; a.slab_stride = (long)B * 4 * H; a.bias_a = w->b_ih; a.bias_b = w->b_hh;
    a.c0 = io->c0; a.ldc0 = H; a.h1 = io->h1; a.ldh1 = H; a.c1 = io->c1; a.ldc1 = H; a.act = io->act; a.tanh_c1 = io->tanh_c1;
    a.h1_drop = io->tcat + H; a.ldh1d = 2 * H; a.drop = tls_drop(io->seed, io->off_h1, io->p_drop); a.B = B; a.H = H;
    RUN(lstm_pointwise_fwd(st, a));
  }
  {
    CopyJobs j{};
    j.n = 2; j.rows = B;
    j.src[0] = io->xcat + 2 * M; j.lds[0] = XK; j.dst[0] = io->tcat; j.ldd[0] = 2 * H; j.cols[0] = H;          // words
    j.src[1] = io->xcat + M; j.lds[1] = XK; j.dst[1] = io->hm + H; j.ldd[1] = H + M; j.cols[1] = M;            // moves
    RUN(copy_blocks(st, j));
  }
  // (6) action logits
  area_reset();
  RUN(gemm_nt_to_consumer(st, ar, io->tcat, 2 * H, w->w_a, wt(3), 2 * H, io->aq, M, B, M, 2 * H, w->b_a, &sv));
  RUN(attn_dot_sv(st, io->cand_rep, W_F32, sv, io->logit, B, C, M, sv.p != io->aq ? io->aq : nullptr, M));
  // (7) progress monitor
  area_reset();
  RUN(gemm_nt_to_consumer(st, ar, io->hm, H + M, w->w_m, wt(4), H + M, io->mg, H, B, H, H + M, w->b_m, &sv));
  if (sv.p != io->mg)
    RUN(monitor_head_fwd_sv(st, sv, io->mg, io->c1, io->word_w, w->w_c, w->b_c, io->mem, io->prog, B, L, H, io->seed, io->off_mem, io->p_drop));
  else
    RUN(vln_monitor_head_fwd(io->mg, io->c1, io->word_w, w->w_c, w->b_c, io->mem, io->prog, B, L, H, io->seed, io->off_mem, io->p_drop, s));
  return VLN_OK;
}

extern "C" int vln_monitor_step_bwd(const vln_monitor_dims* d, const vln_monitor_weights* w, const vln_monitor_step* io,
                                    const vln_monitor_grads* g, vln_stream_t s) {
  RUN(check_monitor_dims(d));
  if (!w || !io || !g || !g->scratch || !g->dh0 || !g->dc0 || !g->dprev_rep) { set_error("vln_monitor_step_bwd: null pointer"); return VLN_ERR_ARG; }
  if (g->scratch_floats < vln_monitor_bwd_scratch_floats(d)) { set_error("vln_monitor_step_bwd: scratch too small"); return VLN_ERR_ARG; }
  DropBaseScope drop_scope(io->offset_base_dev);
  hipStream_t st = (hipStream_t)s;
  const int B = d->B, L = d->L, C = d->C, H = d->H, M = d->M, XK = 2 * M + 2 * H;
  const auto wt = [&](int bit) { return ((w->f32_mask >> bit) & 1) ? (int)W_F32S : d->wtype; };
  float* q = g->scratch;
  auto take = [&](long k) { float* p = q; q += (k + 63) & ~63L; return p; };
  float *dmg = take((long)B * H), *dc1_t = take((long)B * H), *dww = take((long)B * L), *Z = take((long)B * (L + H)), *dpre = take(B);
  float *dhm = take((long)B * (H + M)), *daq = take((long)B * M), *dtcat = take((long)B * 2 * H), *dg = take((long)B * 4 * H);
  float *dxcat = take((long)B * XK), *dmoves = take((long)B * M), *dwords = take((long)B * H), *dvq = take((long)B * M);
  float *dl_v = take((long)B * C), *dh0_v = take((long)B * H), *dtq = take((long)B * H), *dl_t = take((long)B * L);
  float *dh0_t = take((long)B * H), *zlogit = take((long)B * C);
  // progress head (policy.py:126-130)
  RUN(vln_monitor_head_bwd(io->mg, io->c1, io->word_w, w->w_c, io->mem, io->prog, g->dprog, g->dc1, g->dww_ext, dmg, dc1_t, dww, Z,
                           dpre, B, L, H, io->seed, io->off_mem, io->p_drop, s));
  // The five skinny products of the backward stay in split-K slabs inside the workspace (bump-allocated: all five are live until
  // the last sum) and their readers -- the cell's pointwise backward and the two add_n launches -- sum the slabs themselves.
  SlabArea ar{io->ws, (long)io->ws_floats};
  SlabVec s_dhm, s_dtcat, s_dxcat, s_dh0v, s_dh0t;
  RUN(gemm_nt_to_consumer(st, ar, dmg, H, w->w_m_t, wt(4), H, dhm, H + M, B, H + M, H, nullptr, &s_dhm));   // -> h0 | moves
  // action logits (policy.py:108-117): logit = cand_rep . aq
  const float* dlogit = g->dlogit;
  if (!dlogit) { RUN(fill_f32(st, zlogit, (long)B * C, 0.f)); dlogit = zlogit; }
  RUN(rows_wsum(st, io->cand_rep, W_F32, dlogit, daq, M, B, C, M));
  RUN(gemm_nt_to_consumer(st, ar, daq, M, w->w_a_t, wt(3), M, dtcat, 2 * H, B, 2 * H, M, nullptr, &s_dtcat));  // -> words | drop(h1)
  // LSTM cell
  {
    LstmPwBwd a{};
    a.dh1_a = g->dh1; a.ld_a = H; a.dh1_b = s_dtcat.shifted(H); a.dh1_b2 = plain_vec(nullptr, 0);
    a.drop = tls_drop(io->seed, io->off_h1, io->p_drop); a.dc1 = dc1_t; a.lddc1 = H; a.act = io->act; a.tanh_c1 = io->tanh_c1;
    a.c0 = io->c0; a.ldc0 = H; a.dgates = dg; a.lddg = 4 * H; a.dc0 = g->dc0; a.lddc0 = H; a.B = B; a.H = H;
    RUN(lstm_pointwise_bwd(st, a));
  }
  RUN(gemm_nt_to_consumer(st, ar, dg, 4 * H, w->w_cat_t, wt(2), 4 * H, dxcat, XK, B, XK, 4 * H, nullptr, &s_dxcat));   // -> prev | moves | words | h0
  {   // d moves and d words: two sums of the same producers, one launch
    const AddNSvJob aj[2] = {{dmoves, M, B, M, 2, {s_dhm.shifted(H), s_dxcat.shifted(M), SlabVec{}, SlabVec{}}},
                             {dwords, H, B, H, 2, {s_dtcat, s_dxcat.shifted(2 * M), SlabVec{}, SlabVec{}}}};
    RUN(add_n_sv_multi(st, aj, 2));
  }
  // candidates: d cand_rep = move_w (x) dmoves + dl_v (x) vq + dlogit (x) aq
  RUN(attn_bwd_rows(st, io->cand_rep, W_F32, io->move_w, dmoves, M, g->dmw_ext, dvq, M, dl_v, io->dots, B, C, M));
  if (g->dcand_rep) {
    const float* al[2] = {io->move_w, dlogit};
    const float* dl[2] = {dl_v, nullptr};
    const float* gg[2] = {dmoves, io->aq};
    const float* qq[2] = {io->vq, nullptr};
    RUN(attn_dctx_deferred(st, al, dl, gg, M, qq, M, 2, g->dcand_rep, B, C, M, 0));
  }
  RUN(gemm_nt_to_consumer(st, ar, dvq, M, w->w_vh_t, wt(1), M, dh0_v, H, B, H, M, nullptr, &s_dh0v));
  // words: d ctx = (word_w (x) dwords + dl_t (x) tq) * this step's pe-dropout mask
  RUN(attn_bwd_rows(st, io->pctx, W_F32, io->word_w, dwords, H, dww, dtq, H, dl_t, io->dots, B, L, H));
  if (g->dctx_term) {
    *g->dctx_term = vln_dctx_term{io->word_w, dl_t, dwords, io->tq, H, H, io->seed_pe, io->off_pe, io->p_pe, 0.f};
  } else if (g->dctx) {
    const float* al[1] = {io->word_w};
    const float* dl[1] = {dl_t};
    const float* gg[1] = {dwords};
    const float* qq[1] = {io->tq};
    const uint64_t ds[1] = {io->seed_pe}, dof[1] = {io->off_pe};
    const float dp[1] = {io->p_pe};
    RUN(attn_dctx_deferred(st, al, dl, gg, H, qq, H, 1, g->dctx, B, L, H, g->dctx_accumulate, ds, dof, dp));
  }
  RUN(gemm_nt_to_consumer(st, ar, dtq, H, w->w_tin_t, wt(0), H, dh0_t, H, B, H, H, nullptr, &s_dh0t));
  {   // d h0 (four contributions) and d prev_rep (a column block of d xcat), one launch
    const AddNSvJob aj[2] = {{g->dh0, H, B, H, 4, {s_dhm, s_dxcat.shifted(2 * M + H), s_dh0v, s_dh0t}},
                             {g->dprev_rep, M, B, M, 1, {s_dxcat, SlabVec{}, SlabVec{}, SlabVec{}}}};
    RUN(add_n_sv_multi(st, aj, 2));
  }
  // parameter gradients: six products over the same B rows -> one grouped launch; biases and the head -> another
  {
    vln_wgrad_job jobs[VLN_PARAM_JOBS_MAX];
    int n = 0;
    auto add = [&](const float* dy, long ldy, const float* x, long ldx, float* dw, long ldw, int N, int K, int acc) {
      if (dw) jobs[n++] = vln_wgrad_job{dy, x, dw, ldy, ldx, ldw, N, K, acc, 0};
    };
    // X = h0: read from its copy inside xcat (a slot of the step's saved block, so the rollout-level form finds every step's)
    add(dtq, H, io->xcat + 2 * M + H, XK, g->g_tin, H, H, H, g->acc[0]);
    add(dvq, M, io->xcat + 2 * M + H, XK, g->g_vh, H, M, H, g->acc[1]);
    add(dg, 4 * H, io->xcat, XK, g->g_ih, 2 * M + H, 4 * H, 2 * M + H, g->acc[3]);
    add(dg, 4 * H, io->xcat + 2 * M + H, XK, g->g_hh, H, 4 * H, H, g->acc[4]);
    add(daq, M, io->tcat, 2 * H, g->g_a, 2 * H, M, 2 * H, g->acc[7]);
    add(dmg, H, io->hm, H + M, g->g_m, H + M, H, H + M, g->acc[9]);
    if (g->defer) { for (int i = 0; i < n; ++i) g->defer->w[i] = jobs[i]; g->defer->nw = n; g->defer->rows = B; g->defer->precision = g->precision; }
    else if (n) RUN(wgrad_grouped(st, jobs, n, B, g->precision, io->ws, io->ws_floats));
  }
  {
    vln_colsum_job jobs[VLN_PARAM_JOBS_MAX];
    int n = 0;
    auto add = [&](const float* A, long lda, float* o1, float* o2, int cols, int acc) {
      if (o1) jobs[n++] = vln_colsum_job{A, o1, o2, lda, cols, acc};
    };
    add(dvq, M, g->g_bvh, nullptr, M, g->acc[2]);
    add(daq, M, g->g_ba, nullptr, M, g->acc[8]);
    add(dmg, H, g->g_bm, nullptr, H, g->acc[10]);
    add(Z, L + H, g->g_wc, nullptr, L + H, g->acc[11]);
    if (g->g_bih && g->g_bhh && g->acc[5] == g->acc[6]) add(dg, 4 * H, g->g_bih, g->g_bhh, 4 * H, g->acc[5]);
    else { add(dg, 4 * H, g->g_bih, nullptr, 4 * H, g->acc[5]); add(dg, 4 * H, g->g_bhh, nullptr, 4 * H, g->acc[6]); }
    if (g->defer) { for (int i = 0; i < n; ++i) g->defer->c[i] = jobs[i]; g->defer->nc = n; }
    else if (n) RUN(colsum_grouped(st, jobs, n, B, io->ws, io->ws_floats));
    if (g->g_bc) RUN(colsum(st, dpre, 1, g->g_bc, B, 1, g->acc[12], io->ws, io->ws_floats));
  }
  return VLN_OK;
}
