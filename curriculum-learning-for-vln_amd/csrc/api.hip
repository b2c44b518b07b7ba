// C-ABI wrappers (include/vln_hip.h) around the internal launchers + error plumbing.
#include <stdarg.h>
#include <stdio.h>

#include <vector>

#include <cstring>
#include "vln_internal.h"
#include "prologue_bodies.h"
#include "../../include/vln_hip.h"

namespace vln {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }
int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return VLN_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return VLN_ERR_HIP;
}
int g_graphs_enabled = 1;
long long g_graph_stats[3] = {0, 0, 0};
// [0] = 256: interleaved A/B (scripts/ab_bench.py) shows 256/512/768 within noise in time; 256 halves the split-K
// slab traffic (PMC), so it wins on bytes
// [2] = 1, [3] = 512: narrow outputs (N <= 1024) with a SHORT contraction (K <= 512) use the 16-column kernel with the K
// split inside the workgroup.  Interleaved A/B, GPU-bound iteration (scripts/ab_bench.py): all K 3.075 ms, K <= 1024
// 2.890, K <= 512 2.861, off 2.878 -- every 16-column workgroup streams the whole X, which loses for K = 2176.
const unsigned long long*& drop_base_tls() {
  static thread_local const unsigned long long* base = nullptr;
  return base;
}
int g_tunable[16] = {384, 1, 1, 512, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0};
// ---- per-kernel event timers -------------------------------------------------------------------------
unsigned g_prof_mask = 0;
namespace {
struct ProfSlot { hipEvent_t a, b; };
struct ProfState {
  std::vector<ProfSlot> pool;   // created lazily, reused across reads
  size_t used = 0;
  double bytes = 0.0;
};
ProfState g_prof[K_COUNT];
const char* kKernelNames[K_COUNT] = {"gemm_nt", "gemm_tn", "attn_dot", "attn_wsum", "attn_bwd", "lstm_rec_fwd",
                                     "lstm_rec_bwd", "feat_dropout", "lstm_pointwise", "reduce_epilogue"};
}  // namespace
void prof_begin(hipStream_t st, int kid, double algo_bytes) {
  ProfState& p = g_prof[kid];
  if (p.used == p.pool.size()) {
    ProfSlot s;
    if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return;
    p.pool.push_back(s);
  }
  p.bytes += algo_bytes;
  (void)hipEventRecord(p.pool[p.used].a, st);
}
bool prof_slot(int kid, double algo_bytes, hipEvent_t* a, hipEvent_t* b) {
  ProfState& p = g_prof[kid];
  if (p.used == p.pool.size()) {
    ProfSlot s;
    if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return false;
    p.pool.push_back(s);
  }
  p.bytes += algo_bytes;
  *a = p.pool[p.used].a; *b = p.pool[p.used].b;
  p.used++;
  return true;
}
void prof_end(hipStream_t st, int kid) {
  ProfState& p = g_prof[kid];
  if (p.used < p.pool.size()) {
    (void)hipEventRecord(p.pool[p.used].b, st);
    p.used++;
  }
}
}  // namespace vln

using namespace vln;

extern "C" int vln_set_graphs(int on) { g_graphs_enabled = on ? 1 : 0; return VLN_OK; }
extern "C" int vln_graph_stats(int64_t out[3]) {
  if (!out) { set_error("vln_graph_stats: null pointer"); return VLN_ERR_ARG; }
  for (int i = 0; i < 3; ++i) out[i] = g_graph_stats[i];
  return VLN_OK;
}
extern "C" int vln_set_tunable(int id, int value) {
  if (id < 0 || id >= 16) { set_error("vln_set_tunable: bad id"); return VLN_ERR_ARG; }
  g_tunable[id] = value;
  return VLN_OK;
}
extern "C" int vln_prof_enable(int kernel_id, int on) {
  if (kernel_id < 0 || kernel_id >= K_COUNT) { set_error("vln_prof_enable: bad kernel id"); return VLN_ERR_ARG; }
  if (on) g_prof_mask |= (1u << kernel_id); else g_prof_mask &= ~(1u << kernel_id);
  return VLN_OK;
}
extern "C" const char* vln_prof_kernel_name(int kernel_id) {
  return (kernel_id >= 0 && kernel_id < K_COUNT) ? kKernelNames[kernel_id] : nullptr;
}
// Sums and clears the recorded event pairs of one kernel (synchronises on them).
extern "C" int vln_prof_read(int kernel_id, int64_t* launches, double* total_ms, double* total_bytes) {
  if (kernel_id < 0 || kernel_id >= K_COUNT || !launches || !total_ms || !total_bytes) { set_error("vln_prof_read: bad args"); return VLN_ERR_ARG; }
  ProfState& p = g_prof[kernel_id];
  double ms = 0.0;
  for (size_t i = 0; i < p.used; ++i) {
    float t = 0.f;
    if (hipEventSynchronize(p.pool[i].b) != hipSuccess || hipEventElapsedTime(&t, p.pool[i].a, p.pool[i].b) != hipSuccess) {
      set_error("vln_prof_read: event query failed");
      return VLN_ERR_HIP;
    }
    ms += t;
  }
  *launches = (int64_t)p.used; *total_ms = ms; *total_bytes = p.bytes;
  p.used = 0; p.bytes = 0.0;
  return VLN_OK;
}

extern "C" int vln_abi_version(void) { return 19; }
extern "C" int64_t vln_struct_size(const char* name) {
  if (!name) return -1;
#define VLN_SZ(T) if (std::strcmp(name, #T) == 0) return (int64_t)sizeof(T)
  VLN_SZ(vln_tick_item);
  VLN_SZ(vln_wgrad_job);
  VLN_SZ(vln_colsum_job);
  VLN_SZ(vln_param_jobs);
  VLN_SZ(vln_shadow_job);
  VLN_SZ(vln_wsum_step);
  VLN_SZ(vln_dot_step);
  VLN_SZ(vln_ce_step);
  VLN_SZ(vln_select_step);
  VLN_SZ(vln_monitor_loss_step);
  VLN_SZ(vln_cat_step);
  VLN_SZ(vln_monitor_dims);
  VLN_SZ(vln_monitor_weights);
  VLN_SZ(vln_monitor_step);
  VLN_SZ(vln_monitor_grads);
  VLN_SZ(vln_follower_dims);
  VLN_SZ(vln_follower_weights);
  VLN_SZ(vln_follower_step);
  VLN_SZ(vln_follower_grads);
  VLN_SZ(vln_bn_affine);
  VLN_SZ(vln_bn_mlp_layer);
  VLN_SZ(vln_bn_mlp);
  VLN_SZ(vln_bn_mlp_grad_layer);
  VLN_SZ(vln_bn_mlp_grads);
  VLN_SZ(vln_gather_rollout_step);
  VLN_SZ(vln_gather_ride);
  VLN_SZ(vln_envdrop_dims);
  VLN_SZ(vln_envdrop_weights);
  VLN_SZ(vln_envdrop_step);
  VLN_SZ(vln_envdrop_grads);
  VLN_SZ(vln_dctx_term);
#undef VLN_SZ
  return -1;
}
extern "C" const char* vln_last_error_string(void) { return get_error(); }

// The row tiling a tall product Y[M,N] = X W^T takes (csrc/gemm_rows.h) on a device of `cus` compute units: host arithmetic only
// (no device call).  Returns 1 and (n_big, rb_big, tiles) -- row tiles [0, n_big) of rb_big 16-row blocks, the others of
// rb_big - 1, tiles = row tiles x ceil(N / 64) workgroups -- or 0 when the product keeps the 64-row tiles.
extern "C" int vln_gemm_rows_tiling(int M, int N, int cus, int* n_big, int* rb_big, int* tiles) {
  return gemm_rows_tiling(M, N, cus, n_big, rb_big, tiles);
}

extern "C" int vln_linear_fwd_slabs(const float* X, int64_t ldx, const void* W, int wtype, int64_t ldw, int M, int N, int K, float* ws,
                                    int64_t ws_floats, int* n_slabs, vln_stream_t s) {
  if (!X || !W || !ws || !n_slabs || M <= 0 || N <= 0 || K <= 0 || ws_floats < (int64_t)M * N) {
    set_error("vln_linear_fwd_slabs: null pointer, bad dims, or a workspace below one [M, N] slab"); return VLN_ERR_ARG;
  }
  return gemm_nt((hipStream_t)s, X, ldx, W, wtype, ldw, nullptr, 0, M, N, K, nullptr, ACT_NONE, ws, ws_floats, n_slabs);
}
extern "C" int vln_linear_fwd(const float* X, int64_t ldx, const void* W, int wtype, int64_t ldw, float* Y,
                              int64_t ldy, int M, int N, int K, const float* bias, int act, float* ws,
                              int64_t ws_floats, vln_stream_t s) {
  if (!X || !W || !Y) { set_error("vln_linear_fwd: null pointer"); return VLN_ERR_ARG; }
  return gemm_nt((hipStream_t)s, X, ldx, W, wtype, ldw, Y, ldy, M, N, K, bias, act, ws, ws_floats, nullptr);
}
extern "C" int vln_linear_wgrad(const float* A, int64_t lda, const float* X, int64_t ldx, float* D, int64_t ldd,
                                int Mt, int N, int K, int accumulate, float* ws, int64_t ws_floats, vln_stream_t s) {
  if (!A || !X || !D) { set_error("vln_linear_wgrad: null pointer"); return VLN_ERR_ARG; }
  return gemm_tn((hipStream_t)s, A, lda, X, ldx, D, ldd, Mt, N, K, accumulate, ws, ws_floats);
}
extern "C" int vln_linear_wgrad_p(const float* A, int64_t lda, const float* X, int64_t ldx, float* D, int64_t ldd,
                                  int Mt, int N, int K, int accumulate, int precision, float* ws, int64_t ws_floats,
                                  vln_stream_t s) {
  if (!A || !X || !D) { set_error("vln_linear_wgrad_p: null pointer"); return VLN_ERR_ARG; }
  if (precision != 0 && precision != 1) { set_error("vln_linear_wgrad_p: precision must be 0 (fp32) or 1 (split bf16)"); return VLN_ERR_ARG; }
  return gemm_tn((hipStream_t)s, A, lda, X, ldx, D, ldd, Mt, N, K, accumulate, ws, ws_floats, precision);
}
extern "C" int vln_wgrad_grouped(const vln_wgrad_job* jobs, int n_jobs, int Mt, int precision, float* ws, int64_t ws_floats,
                                 vln_stream_t s) {
  if (!jobs || n_jobs <= 0) { set_error("vln_wgrad_grouped: bad args"); return VLN_ERR_ARG; }
  if (precision < 0 || precision > 2) { set_error("vln_wgrad_grouped: precision must be 0 (fp32), 1 (split bf16) or 2 (plain bf16)"); return VLN_ERR_ARG; }
  return wgrad_grouped((hipStream_t)s, jobs, n_jobs, Mt, precision, ws, ws_floats);
}
extern "C" int vln_wgrad_grouped_seg(const vln_wgrad_job* jobs, const int64_t* dy_seg_stride, const int64_t* x_seg_stride, int n_jobs,
                                     int seg_rows, int n_seg, int precision, float* ws, int64_t ws_floats, vln_stream_t s) {
  if (!jobs || n_jobs <= 0) { set_error("vln_wgrad_grouped_seg: bad args"); return VLN_ERR_ARG; }
  if (precision < 0 || precision > 2) { set_error("vln_wgrad_grouped_seg: precision must be 0 (fp32), 1 (split bf16) or 2 (plain bf16)"); return VLN_ERR_ARG; }
  return wgrad_grouped_seg((hipStream_t)s, jobs, n_jobs, seg_rows, n_seg, dy_seg_stride, x_seg_stride, precision, ws, ws_floats);
}
extern "C" int vln_colsum_grouped_seg(const vln_colsum_job* jobs, const int64_t* seg_stride, int n_jobs, int seg_rows, int n_seg,
                                      float* ws, int64_t ws_floats, vln_stream_t s) {
  if (!jobs || !seg_stride || n_jobs <= 0 || seg_rows <= 0 || n_seg <= 0) { set_error("vln_colsum_grouped_seg: bad args"); return VLN_ERR_ARG; }
  for (int i = 0; i < n_jobs; ++i)
    if (seg_stride[i] & 3) { set_error("vln_colsum_grouped_seg: segment strides must be multiples of 4 floats"); return VLN_ERR_ARG; }
  return colsum_grouped((hipStream_t)s, jobs, n_jobs, seg_rows * n_seg, ws, ws_floats, seg_rows, seg_stride);
}
extern "C" int64_t vln_wgrad_grouped_ws_floats(const vln_wgrad_job* jobs, int n_jobs, int Mt) {
  if (!jobs || n_jobs <= 0 || Mt <= 0) return -1;
  return wgrad_grouped_ws_floats(jobs, n_jobs, Mt);
}
extern "C" int vln_colsum(const float* A, int64_t lda, float* out, int rows, int cols, int accumulate, float* ws,
                          int64_t ws_floats, vln_stream_t s) {
  if (!A || !out || cols <= 0) { set_error("vln_colsum: bad args"); return VLN_ERR_ARG; }
  return colsum((hipStream_t)s, A, lda, out, rows, cols, accumulate, ws, ws_floats);
}
extern "C" int vln_colsum_grouped(const vln_colsum_job* jobs, int n_jobs, int rows, float* ws, int64_t ws_floats, vln_stream_t s) {
  if (!jobs) { set_error("vln_colsum_grouped: null pointer"); return VLN_ERR_ARG; }
  return colsum_grouped((hipStream_t)s, jobs, n_jobs, rows, ws, ws_floats);
}
extern "C" int vln_transpose_cast(const float* W, int64_t ldw, void* Wt, int out_type, int64_t ldt, int N, int K,
                                  vln_stream_t s) {
  if (!W || !Wt || N <= 0 || K <= 0) { set_error("vln_transpose_cast: bad args"); return VLN_ERR_ARG; }
  return transpose_cast((hipStream_t)s, W, ldw, Wt, out_type, ldt, N, K);
}
extern "C" int vln_cast_copy(const float* W, int64_t ldw, void* out, int out_type, int64_t ldo, int rows, int cols,
                             vln_stream_t s) {
  if (!W || !out || rows <= 0 || cols <= 0) { set_error("vln_cast_copy: bad args"); return VLN_ERR_ARG; }
  return cast_copy((hipStream_t)s, W, ldw, out, out_type, ldo, rows, cols);
}
extern "C" int vln_attn_dot(const void* ctx, int ctype, const float* vec, int64_t ldv, float* dots, int B, int S,
                            int D, vln_stream_t s) {
  if (!ctx || !vec || !dots) { set_error("vln_attn_dot: null pointer"); return VLN_ERR_ARG; }
  return attn_dot((hipStream_t)s, ctx, ctype, vec, ldv, dots, B, S, D);
}
extern "C" int vln_attn_softmax_wsum(const void* ctx, int ctype, const float* logits, const uint8_t* mask,
                                     float* attn, float* out, int64_t ldo, int B, int S, int D, vln_stream_t s) {
  if (!ctx || !logits || !out) { set_error("vln_attn_softmax_wsum: null pointer"); return VLN_ERR_ARG; }
  return attn_softmax_wsum((hipStream_t)s, ctx, ctype, logits, mask, attn, out, ldo, B, S, D);
}
extern "C" int vln_rows_wsum(const void* ctx, int ctype, const float* w, float* out, int64_t ldo, int B, int S,
                             int D, vln_stream_t s) {
  if (!ctx || !w || !out) { set_error("vln_rows_wsum: null pointer"); return VLN_ERR_ARG; }
  return rows_wsum((hipStream_t)s, ctx, ctype, w, out, ldo, B, S, D);
}
extern "C" int vln_attn_dot_multi(const vln_dot_step* steps, int T, int ctype, int B, int D, int64_t ldv, vln_stream_t s) {
  return attn_dot_multi((hipStream_t)s, steps, T, ctype, B, D, (long)ldv);
}
namespace vln {
struct SelectMulti { const float* src[VLN_CE_MAX_STEPS]; const long long* index[VLN_CE_MAX_STEPS]; float* out[VLN_CE_MAX_STEPS]; int C[VLN_CE_MAX_STEPS]; int T, B, F; unsigned* bad; };
__global__ __launch_bounds__(256) void select_rows_multi_kernel(SelectMulti m) {
  const int t = (int)blockIdx.x / m.B, b = (int)blockIdx.x % m.B;
  long i = m.index[t][b];
  if (i < 0) i += m.C[t];
  const bool ok = i >= 0 && i < m.C[t];
  if (!ok && threadIdx.x == 0 && m.bad) __hip_atomic_fetch_add(m.bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const float4* src = reinterpret_cast<const float4*>(m.src[t] + ((long)b * m.C[t] + (ok ? i : 0)) * m.F);
  float4* dst = reinterpret_cast<float4*>(m.out[t] + (long)b * m.F);
  for (int c = threadIdx.x; c < m.F / 4; c += 256) dst[c] = ok ? src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
}
}  // namespace vln
extern "C" int vln_select_rows_multi(const vln_select_step* steps, int T, int B, int F, vln_stream_t s) {
  if (!steps || T <= 0 || T > VLN_CE_MAX_STEPS || B <= 0 || F <= 0 || (F & 3)) { set_error("vln_select_rows_multi: bad dims (F %% 4 == 0, T <= %d)", VLN_CE_MAX_STEPS); return VLN_ERR_ARG; }
  SelectMulti m{};
  m.T = T; m.B = B; m.F = F;
  unsigned* w = sticky_dev_word();
  m.bad = w ? w + 1 : nullptr;
  for (int t = 0; t < T; ++t) {
    if (!steps[t].src || !steps[t].index || !steps[t].out || steps[t].C <= 0 || ((uintptr_t)steps[t].src & 15) || ((uintptr_t)steps[t].out & 15)) {
      set_error("vln_select_rows_multi: bad step %d", t);
      return VLN_ERR_ARG;
    }
    m.src[t] = steps[t].src; m.index[t] = (const long long*)steps[t].index; m.out[t] = steps[t].out; m.C[t] = steps[t].C;
  }
  VLN_LAUNCH(select_rows_multi_kernel, dim3(T * B), dim3(256), 0, (hipStream_t)s, m);
  VLN_CHECK_LAUNCH("select_rows_multi");
  return VLN_OK;
}
extern "C" int vln_rows_wsum_multi(const vln_wsum_step* steps, int T, int ctype, int B, int D, int64_t ldo, float ce_scale,
                                   const float* ce_dloss, int64_t ignore_index, vln_stream_t s) {
  return rows_wsum_multi((hipStream_t)s, steps, T, ctype, B, D, (long)ldo, ce_scale, ce_dloss, (long)ignore_index);
}
extern "C" int vln_attn_bwd(const void* ctx, int ctype, const float* attn, const float* dalpha,
                            const float* dattn_ext, const float* dwc, int64_t lddwc, const float* vec,
                            int64_t ldvec, float* dvec, int64_t lddvec, float* dctx, float* dl_out, int B, int S,
                            int D, vln_stream_t s) {
  if (!ctx || !attn) { set_error("vln_attn_bwd: null pointer"); return VLN_ERR_ARG; }
  return attn_bwd((hipStream_t)s, ctx, ctype, attn, dalpha, dattn_ext, dwc, lddwc, vec, ldvec, dvec, lddvec, dctx,
                  dl_out, B, S, D);
}
extern "C" int vln_attn_fwd_rows(const void* ctx, int ctype, const float* vec, int64_t ldv, const uint8_t* mask, float* attn,
                                 float* out, int64_t ldo, float* dots_scratch, int B, int S, int D, void* sync,
                                 int64_t sync_bytes, vln_stream_t s) {
  if (!ctx || !vec || !out) { set_error("vln_attn_fwd_rows: null pointer"); return VLN_ERR_ARG; }
  return attn_fwd_rows_sv((hipStream_t)s, ctx, ctype, plain_vec(vec, ldv), nullptr, 0, mask, attn, out, ldo, dots_scratch, B, S, D, sync, sync_bytes);
}
extern "C" int vln_attn_bwd_rows(const void* ctx, int ctype, const float* attn, const float* dwc, int64_t lddwc,
                                 const float* dattn_ext, float* dvec, int64_t lddvec, float* dl_out, float* dots_scratch,
                                 int B, int S, int D, void* sync, int64_t sync_bytes, vln_stream_t s) {
  if (!ctx || !attn || !dwc || !dvec) { set_error("vln_attn_bwd_rows: null pointer"); return VLN_ERR_ARG; }
  return attn_bwd_rows_sv((hipStream_t)s, ctx, ctype, attn, plain_vec(dwc, lddwc), nullptr, 0, dattn_ext, dvec, lddvec, dl_out, dots_scratch, B, S, D,
                          sync, sync_bytes);
}
extern "C" int vln_attn_dctx_deferred(const float* const* alpha, const float* const* dl, const float* const* g, int64_t ldg,
                                      const float* const* q, int64_t ldq, int T, float* dctx, int B, int S, int D,
                                      int accumulate, float* dk, vln_stream_t s) {
  if (!alpha || !dl || !g || !q || !dctx) { set_error("vln_attn_dctx_deferred: null pointer"); return VLN_ERR_ARG; }
  return attn_dctx_deferred((hipStream_t)s, alpha, dl, g, ldg, q, ldq, T, dctx, B, S, D, accumulate, nullptr, nullptr, nullptr, dk);
}
extern "C" int vln_attn_dctx_deferred_drop(const float* const* alpha, const float* const* dl, const float* const* g, int64_t ldg,
                                           const float* const* q, int64_t ldq, int T, float* dctx, int B, int S, int D,
                                           int accumulate, const uint64_t* drop_seed, const uint64_t* drop_off,
                                           const float* drop_p, const uint64_t* offset_base_dev, vln_stream_t s) {
  if (!alpha || !dl || !g || !q || !dctx) { set_error("vln_attn_dctx_deferred_drop: null pointer"); return VLN_ERR_ARG; }
  DropBaseScope drop_scope(offset_base_dev);
  return attn_dctx_deferred((hipStream_t)s, alpha, dl, g, ldg, q, ldq, T, dctx, B, S, D, accumulate, drop_seed, drop_off, drop_p);
}
extern "C" int vln_lstm_pointwise_fwd(const float* gates, int nsplit, int64_t slab_stride, const float* b_ih,
                                      const float* b_hh, const float* c0, float* h1, float* c1, float* act,
                                      float* tanh_c1, float* h1_drop, uint64_t seed, uint64_t offset, float p, int B,
                                      int H, vln_stream_t s) {
  if (!gates || !c0 || !h1 || !c1 || B <= 0 || H <= 0 || nsplit < 1) { set_error("vln_lstm_pointwise_fwd: bad args"); return VLN_ERR_ARG; }
  LstmPwFwd a{};
  a.gates = gates; a.nsplit = nsplit; a.slab_stride = slab_stride; a.bias_a = b_ih; a.bias_b = b_hh;
  a.c0 = c0; a.ldc0 = H; a.h1 = h1; a.ldh1 = H; a.c1 = c1; a.ldc1 = H; a.act = act; a.tanh_c1 = tanh_c1;
  a.h1_drop = h1_drop; a.ldh1d = H; a.drop = DropSpec{seed, offset, p}; a.B = B; a.H = H;
  return lstm_pointwise_fwd((hipStream_t)s, a);
}
extern "C" int vln_lstm_pointwise_bwd(const float* dh1, const float* dh1_drop, const float* dc1, uint64_t seed,
                                      uint64_t offset, float p, const float* act, const float* tanh_c1,
                                      const float* c0, float* dgates, float* dc0, int B, int H, vln_stream_t s) {
  if (!act || !tanh_c1 || !c0 || !dgates || !dc0 || B <= 0 || H <= 0) { set_error("vln_lstm_pointwise_bwd: bad args"); return VLN_ERR_ARG; }
  LstmPwBwd a{};
  a.dh1_a = dh1; a.ld_a = H; a.dh1_b = plain_vec(dh1_drop, H); a.dh1_b2 = plain_vec(nullptr, 0);
  a.drop = DropSpec{seed, offset, p}; a.dc1 = dc1; a.lddc1 = H; a.act = act; a.tanh_c1 = tanh_c1;
  a.c0 = c0; a.ldc0 = H; a.dgates = dgates; a.lddg = 4 * H; a.dc0 = dc0; a.lddc0 = H; a.B = B; a.H = H;
  return lstm_pointwise_bwd((hipStream_t)s, a);
}
extern "C" int vln_attn_textk_fwd(const void* ctx, int ctype, const float* kctx, const uint8_t* mask, const float* gates, int nsplit,
                                  int64_t slab_stride, const float* b_ih, const float* b_hh, const float* c0, float* h1, float* c1,
                                  float* act, float* tanh_c1, float* tcat, float* alpha, uint64_t seed, uint64_t offset, float p,
                                  int B, int S, int H, void* sync, int64_t sync_bytes, vln_stream_t s) {
  if (!ctx || !kctx || !gates || !c0 || !h1 || !c1 || !tcat || !alpha || B <= 0 || S <= 0 || H <= 0) { set_error("vln_attn_textk_fwd: bad args"); return VLN_ERR_ARG; }
  LstmPwFwd a{};
  a.gates = gates; a.nsplit = nsplit; a.slab_stride = slab_stride; a.bias_a = b_ih; a.bias_b = b_hh;
  a.c0 = c0; a.ldc0 = H; a.h1 = h1; a.ldh1 = H; a.c1 = c1; a.ldc1 = H; a.act = act; a.tanh_c1 = tanh_c1;
  a.h1_drop = tcat + H; a.ldh1d = 2 * H; a.drop = DropSpec{seed, offset, p}; a.B = B; a.H = H;
  return attn_textk_fwd((hipStream_t)s, ctx, ctype, kctx, mask, alpha, tcat, 2 * H, a, B, S, H, sync, (long)sync_bytes);
}
extern "C" int vln_attn_textk_bwd(const void* ctx, int ctype, const float* kctx, const float* alpha, const float* dtcat, int nsplit,
                                  int64_t slab_stride, float* dwc_out, float* dq, float* dl, const float* dh1, const float* dc1,
                                  const float* act, const float* tanh_c1, const float* c0, float* dgates, float* dc0, uint64_t seed,
                                  uint64_t offset, float p, int B, int S, int H, void* sync, int64_t sync_bytes, vln_stream_t s) {
  if (!ctx || !kctx || !alpha || !dtcat || nsplit < 1 || !dq || B <= 0 || S <= 0 || H <= 0) { set_error("vln_attn_textk_bwd: bad args"); return VLN_ERR_ARG; }
  LstmPwBwd a{};
  a.dh1_a = dh1; a.ld_a = H; a.drop = DropSpec{seed, offset, p}; a.dc1 = dc1; a.lddc1 = H; a.act = act; a.tanh_c1 = tanh_c1;
  a.c0 = c0; a.ldc0 = H; a.dgates = dgates; a.lddg = 4 * H; a.dc0 = dc0; a.lddc0 = H; a.B = B; a.H = H;
  return attn_textk_bwd((hipStream_t)s, ctx, ctype, kctx, alpha, SlabVec{dtcat, 2L * H, nsplit, (long)slab_stride}, dwc_out, 2 * H, dq, H,
                        dl, a, B, S, H, sync, (long)sync_bytes);
}
extern "C" int vln_dropout_mask(float* out, int64_t n, uint64_t seed, uint64_t offset, float p, vln_stream_t s) {
  if (!out || n < 0) { set_error("vln_dropout_mask: bad args"); return VLN_ERR_ARG; }
  return export_dropout_mask((hipStream_t)s, out, n, DropSpec{seed, offset, p});
}
extern "C" int vln_scale_dropout(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int cols,
                                 uint64_t seed, uint64_t offset, float p, const uint64_t* offset_base_dev, vln_stream_t s) {
  if (!x || !y) { set_error("vln_scale_dropout: null pointer"); return VLN_ERR_ARG; }
  return scale_dropout((hipStream_t)s, x, ldx, y, ldy, rows, cols, drop_spec(seed, offset, p, offset_base_dev));
}
extern "C" int vln_feat_dropout_inplace(void* x, int xtype, int64_t rows, int img, int angle, uint64_t seed,
                                        uint64_t offset, float p, void* copy_bf16, const uint64_t* offset_base_dev, vln_stream_t s) {
  if (!x) { set_error("vln_feat_dropout_inplace: null pointer"); return VLN_ERR_ARG; }
  return feat_dropout_inplace((hipStream_t)s, x, xtype, rows, img, angle, drop_spec(seed, offset, p, offset_base_dev), copy_bf16);
}

// ---- device-resident counters (runtime.DeviceClock) ----------------------------------------------------------------
namespace vln {
__global__ void tick_kernel(TickArgs a) { tick_body(a, (int)threadIdx.x); }
int tick_args(const vln_tick_item* items, int n, TickArgs* a) {
  if (!items || n <= 0 || n > VLN_TICK_MAX) { set_error("vln_tick: 1..%d items", VLN_TICK_MAX); return VLN_ERR_ARG; }
  *a = TickArgs{};
  a->n = n;
  for (int i = 0; i < n; ++i) {
    if (!items[i].word || (items[i].width != 4 && items[i].width != 8)) { set_error("vln_tick: item %d: null word or width not 4 / 8", i); return VLN_ERR_ARG; }
    if (items[i].width == 8) { a->w64[i] = (unsigned long long*)items[i].word; a->inc64[i] = items[i].inc; }
    else { a->w32[i] = (unsigned*)items[i].word; a->inc32[i] = (unsigned)items[i].inc; }
  }
  return VLN_OK;
}
}  // namespace vln
extern "C" int vln_tick(const vln_tick_item* items, int n, vln_stream_t s) {
  TickArgs a;
  int r = tick_args(items, n, &a); if (r) return r;
  VLN_LAUNCH(tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, a);
  VLN_CHECK_LAUNCH("tick");
  return VLN_OK;
}

// ---- the batch hand-over: the GPU PULLS the packed batch out of pinned host memory ------------------------------------------------
// The reference marshals every batch on the host (agent/base.py:114-178) and copies it with `.to(device)`.  Here the trainer packs the
// batch's small tensors (tokens, masks, per-step index vectors, targets: ~0.4 MB) into ONE pinned host blob and stores the blob's
// address in a pinned SLOT word; this kernel -- the first node of the iteration graph -- reads the slot and copies the blob into the
// fixed device buffers the captured iteration reads.  No copy API call sits between two graph replays (a stream-ordered
// hipMemcpyAsync in front of the graph cost 130 us per iteration on MI355X: profiles/round4_notes.md), the launch arguments repeat,
// and the host's share per iteration is one store.
namespace vln {
__global__ __launch_bounds__(256) void host_fetch_kernel(FetchArgs f) { host_fetch_body(f, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x); }
int fetch_args(const uint64_t* slots_dev, int ring, uint64_t* seq, uint32_t* done, void* dst, int64_t nbytes, FetchArgs* f, int* blocks) {
  if (!slots_dev || !seq || !done || !dst || ring < 1 || nbytes <= 0 || (nbytes & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) {
    set_error("vln_host_fetch: null pointer, empty ring, or size / destination not a multiple of 16 bytes");
    return VLN_ERR_ARG;
  }
  const long n16 = nbytes / 16;
  int b = (int)((n16 + 255) / 256);
  if (b > 512) b = 512;
  *f = FetchArgs{reinterpret_cast<const unsigned long long*>(slots_dev), ring, reinterpret_cast<unsigned long long*>(seq),
                 reinterpret_cast<unsigned*>(done), static_cast<u32x4*>(dst), n16};
  *blocks = b;
  return VLN_OK;
}
__global__ __launch_bounds__(256) void host_fetch_part_kernel(FetchPart f) { host_fetch_part_body(f, (int)threadIdx.x); }
int fetch_part_args(const ::vln_gather_ride& r, FetchPart* f) {
  *f = FetchPart{};
  if (!r.fetch_slots) return VLN_OK;
  if (!r.fetch_seq || !r.fetch_dst || r.fetch_ring < 1 || r.fetch_bytes <= 0 || (r.fetch_bytes & 15) || (r.fetch_offset & 15) || r.fetch_offset < 0 ||
      (reinterpret_cast<uintptr_t>(r.fetch_dst) & 15)) {
    set_error("vln_gather_ride: the batch-tail fetch needs a ring, a sequence word and 16-byte multiples for offset / size / destination");
    return VLN_ERR_ARG;
  }
  *f = FetchPart{reinterpret_cast<const unsigned long long*>(r.fetch_slots), reinterpret_cast<const unsigned long long*>(r.fetch_seq),
                 static_cast<u32x4*>(r.fetch_dst), (long)(r.fetch_offset / 16), (long)(r.fetch_bytes / 16), r.fetch_ring, 1};
  return VLN_OK;
}
int launch_fetch_part(hipStream_t st, const FetchPart& f) {
  if (!f.on) return VLN_OK;
  VLN_LAUNCH(host_fetch_part_kernel, dim3(1), dim3(256), 0, st, f);
  VLN_CHECK_LAUNCH("host_fetch_part");
  return VLN_OK;
}
}  // namespace vln
extern "C" int vln_host_device_pointer(const void* host, void** dev) {
  if (!host || !dev) { set_error("vln_host_device_pointer: null pointer"); return VLN_ERR_ARG; }
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, const_cast<void*>(host), 0) != hipSuccess || !d) {
    (void)hipGetLastError();
    set_error("vln_host_device_pointer: %p is not pinned host memory mapped into the device's address space", host);
    return VLN_ERR_ARG;
  }
  *dev = d;
  return VLN_OK;
}
// ---- the device WAITS for the host inside a launch sequence (round 5) -------------------------------------------------------------------
// A rollout whose next step needs the HOST (envdrop.py:196-206: the simulator takes the sampled action and answers with the next
// observation) used to be one graph launch per step: launch latency and the wake-up of the stream stand between two steps.  Here the
// whole iteration is ONE graph and the host's turn is a one-wave kernel between two steps that spins on a word of pinned host memory
// until it holds the value of a device word (`want`: the device clock's word of THIS iteration, so the flag of an earlier iteration
// never matches).  The host polls the action words the previous step stored to pinned memory, does its work, writes the next step's
// inputs and then the flag.  Bounded: a host that never answers raises the sticky word instead of hanging the queue.
namespace vln {
// limit > 0: a number of polls; limit < 0: -limit ticks of the 100 MHz constant clock (s_memrealtime: a wall-clock bound that does not
// depend on how fast this wave polls).  A flag equal to VLN_HOST_WAIT_POISON means the host has given the iteration up (an exception in
// its turn): the wait ends at once, the sticky word is raised so that the next vln_persistent_check reports the iteration as invalid.
__device__ __forceinline__ void host_wait_spin(const unsigned long long* flag, const unsigned long long* want, unsigned* sticky, long long limit) {
  const unsigned long long w = *want;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  bool bad = false;
  for (unsigned long long i = 0;; ++i) {
    const unsigned long long f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (f == w) break;
    if (f == VLN_HOST_WAIT_POISON) { bad = true; break; }
    if (limit > 0 ? (long long)i > limit : (long long)(__builtin_amdgcn_s_memrealtime() - t0) > -limit) { bad = true; break; }
    __builtin_amdgcn_s_sleep(2);
  }
  if (bad && sticky) __hip_atomic_fetch_add(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __atomic_thread_fence(__ATOMIC_ACQUIRE);      // what the host wrote before the flag is visible to the launches behind this one
}
// `ack` (nullable, pinned host memory): once the wait is over -- and, in the fetching form, the host's bytes have been read -- the
// launch stores *want there: the host may start the SAME turn of the NEXT iteration (rewrite that turn's mailbox and flag) only after
// it has seen this iteration's value, so a host that runs ahead of the device cannot overwrite a flag the device has yet to see.
__global__ void host_wait_kernel(const unsigned long long* flag, const unsigned long long* want, unsigned* sticky, long long limit,
                                 unsigned long long* ack) {
  if (threadIdx.x != 0) return;
  host_wait_spin(flag, want, sticky, limit);
  if (ack) __hip_atomic_store(ack, *want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ... and the same wait followed, in the SAME launch, by the pull of what the host left for the next step (the observation's packed
// index vectors, a few KB in pinned host memory): one workgroup, the flag first, then 16-byte reads through PCIe.
__global__ __launch_bounds__(256) void host_wait_fetch_kernel(const unsigned long long* flag, const unsigned long long* want, unsigned* sticky,
                                                              long long limit, const u32x4* src, u32x4* dst, long n16, unsigned long long* ack) {
  if (threadIdx.x == 0) host_wait_spin(flag, want, sticky, limit);
  __syncthreads();
  for (long i = threadIdx.x; i < n16; i += 256) dst[i] = __builtin_nontemporal_load(src + i);
  __syncthreads();                              // every read of the mailbox has returned (a load's data is needed for its store)
  if (ack && threadIdx.x == 0) __hip_atomic_store(ack, *want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
static long long host_wait_limit(int64_t spin_limit) {
  if (spin_limit > 0) return (long long)spin_limit;                                   // polls
  const long long us = spin_limit < 0 ? -(long long)spin_limit : (long long)VLN_HOST_WAIT_DEFAULT_US;
  return -(us * 100ll);                                                                // 100 MHz ticks
}
}  // namespace vln
extern "C" int vln_host_wait(const uint64_t* flag_dev, const uint64_t* want_dev, int64_t spin_limit, uint64_t* ack_dev, vln_stream_t s) {
  if (!flag_dev || !want_dev) { set_error("vln_host_wait: null pointer"); return VLN_ERR_ARG; }
  unsigned* sticky = sticky_dev_word();
  VLN_LAUNCH(host_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, reinterpret_cast<const unsigned long long*>(flag_dev),
             reinterpret_cast<const unsigned long long*>(want_dev), sticky ? sticky + 3 : nullptr, host_wait_limit(spin_limit),
             reinterpret_cast<unsigned long long*>(ack_dev));
  VLN_CHECK_LAUNCH("host_wait");
  return VLN_OK;
}
extern "C" int vln_host_wait_fetch(const uint64_t* flag_dev, const uint64_t* want_dev, int64_t spin_limit, const void* src_dev, void* dst,
                                   int64_t nbytes, uint64_t* ack_dev, vln_stream_t s) {
  if (!flag_dev || !want_dev || !src_dev || !dst || nbytes <= 0 || (nbytes & 15) || (reinterpret_cast<uintptr_t>(dst) & 15) ||
      (reinterpret_cast<uintptr_t>(src_dev) & 15)) {
    set_error("vln_host_wait_fetch: null pointer, or size / source / destination not a multiple of 16 bytes");
    return VLN_ERR_ARG;
  }
  unsigned* sticky = sticky_dev_word();
  VLN_LAUNCH(host_wait_fetch_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, reinterpret_cast<const unsigned long long*>(flag_dev),
             reinterpret_cast<const unsigned long long*>(want_dev), sticky ? sticky + 3 : nullptr, host_wait_limit(spin_limit),
             static_cast<const u32x4*>(src_dev), static_cast<u32x4*>(dst), (long)(nbytes / 16), reinterpret_cast<unsigned long long*>(ack_dev));
  VLN_CHECK_LAUNCH("host_wait_fetch");
  return VLN_OK;
}
// What a step hands BACK to the host without a copy call: `nbytes` (multiple of 8) from device memory to pinned host memory
// (device-visible address), stored by one workgroup with system-scope stores -- the teacher-forced rollout's chosen actions
// (envdrop.py:198 `a_t.cpu()`), which the host polls while the captured iteration goes on (the sampled rollout's draw stores its
// actions itself: cand_sample_kernel).
namespace vln {
__global__ __launch_bounds__(256) void store_to_host_kernel(const unsigned long long* src, unsigned long long* dst, long n8) {
  for (long i = threadIdx.x; i < n8; i += 256) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace vln
extern "C" int vln_store_to_host(const void* src, void* dst_dev, int64_t nbytes, vln_stream_t s) {
  if (!src || !dst_dev || nbytes <= 0 || (nbytes & 7) || (reinterpret_cast<uintptr_t>(src) & 7) || (reinterpret_cast<uintptr_t>(dst_dev) & 7)) {
    set_error("vln_store_to_host: null pointer, or size / addresses not multiples of 8 bytes");
    return VLN_ERR_ARG;
  }
  VLN_LAUNCH(store_to_host_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, static_cast<const unsigned long long*>(src),
             static_cast<unsigned long long*>(dst_dev), (long)(nbytes / 8));
  VLN_CHECK_LAUNCH("store_to_host");
  return VLN_OK;
}
extern "C" int vln_host_fetch(const uint64_t* slots_dev, int ring, uint64_t* seq, uint32_t* done, void* dst, int64_t nbytes, vln_stream_t s) {
  FetchArgs f; int blocks = 0;
  int r = fetch_args(slots_dev, ring, seq, done, dst, nbytes, &f, &blocks); if (r) return r;
  VLN_LAUNCH(host_fetch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, f);
  VLN_CHECK_LAUNCH("host_fetch");
  return VLN_OK;
}

// ---- measurement only: a chain of trivial DEPENDENT launches (what does a kernel boundary cost inside THIS process / graph?) ----
namespace vln {
__global__ __launch_bounds__(256) void debug_trivial_kernel(const float4* in, float4* out, int n4, int shift) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n4) return;
  int jr = j + shift; if (jr >= n4) jr -= n4;
  const float4 v = in[jr];
  out[j] = make_float4(v.x * .5f, v.y * .5f, v.z * .5f, v.w * .5f + 1.f);
}
}  // namespace vln
// A kernel that only OCCUPIES compute units for a while: `workgroups` x 1024 threads, `lds_bytes` of LDS each, resident for
// `micros` microseconds (wall clock) -- the stand-in for a communication kernel (RCCL) that is resident on another stream while
// the persistent recurrence needs its own workgroups co-resident (tests/test_hip_dp_rccl.py).
namespace vln {
__global__ __launch_bounds__(1024) void debug_occupy_kernel(unsigned long long ticks, int* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char occ_lds[];
  occ_lds[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (sink && occ_lds[(threadIdx.x * 7) & 1023] == 255 && ticks == 0) *sink = 1;     // keeps the LDS allocation alive
}
}  // namespace vln
extern "C" int vln_debug_occupy(int workgroups, int lds_bytes, int micros, vln_stream_t s) {
  if (workgroups < 1 || workgroups > 1024 || lds_bytes < 1024 || lds_bytes > 160 * 1024 || micros < 0 || micros > 100000) {
    set_error("vln_debug_occupy: 1..1024 workgroups, 1 KB..160 KB of LDS, <= 100 ms");
    return VLN_ERR_ARG;
  }
  if (lds_bytes > 64 * 1024 &&
      check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(debug_occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes),
                "vln_debug_occupy: hipFuncSetAttribute") != VLN_OK) return VLN_ERR_HIP;
  VLN_LAUNCH(debug_occupy_kernel, dim3(workgroups), dim3(1024), (unsigned)lds_bytes, (hipStream_t)s, (unsigned long long)micros * 100ull, (int*)nullptr);
  VLN_CHECK_LAUNCH("debug_occupy");
  return VLN_OK;
}

extern "C" int vln_debug_trivial_chain(float* a, float* b, int n_floats, int launches, int shift, vln_stream_t s) {
  if (!a || !b || n_floats < 4 || (n_floats & 3) || launches < 1) { set_error("vln_debug_trivial_chain: bad args"); return VLN_ERR_ARG; }
  const int n4 = n_floats / 4;
  for (int k = 0; k < launches; ++k) {
    VLN_LAUNCH(debug_trivial_kernel, dim3((n4 + 255) / 256), dim3(256), 0, (hipStream_t)s, (const float4*)((k & 1) ? b : a),
               (float4*)((k & 1) ? a : b), n4, shift);
  }
  VLN_CHECK_LAUNCH("debug_trivial_chain");
  return VLN_OK;
}

extern "C" int vln_linear_fwd_post(const float* x, int64_t ldx, const void* w, int wtype, int64_t ldw, float* y, int64_t ldy, int M, int N, int K) {
  return linear_fwd_post(x, (long)ldx, w, wtype, (long)ldw, y, (long)ldy, M, N, K);
}
extern "C" int vln_linear_fwd_post_flush(float* ws, int64_t ws_floats, vln_stream_t s) {
  return linear_fwd_post_flush((hipStream_t)s, ws, (long)ws_floats);
}
extern "C" int vln_layout_post(int kind, const float* src, float* dst, void* dst_bf16, int B, int L, int W, uint64_t seed, uint64_t offset, float p,
                               const uint64_t* offset_base_dev) {
  return layout_post(kind, src, dst, dst_bf16, B, L, W, drop_spec(seed, offset, p, offset_base_dev));
}
extern "C" int vln_layout_post_flush(vln_stream_t s) { return layout_post_flush((hipStream_t)s); }
extern "C" int vln_posted_drop(void) { return posted_drop(); }
extern "C" int vln_colsum_post(const vln_colsum_job* jobs, int n_jobs, int rows) { return colsum_post(jobs, n_jobs, rows); }
extern "C" int vln_colsum_post_flush(float* ws, int64_t ws_floats, vln_stream_t s) { return colsum_post_flush((hipStream_t)s, ws, (long)ws_floats); }
extern "C" int vln_shadow_refresh(const vln_shadow_job* jobs, int n_jobs, vln_stream_t s) {
  if (!jobs || n_jobs <= 0) { set_error("vln_shadow_refresh: bad args"); return VLN_ERR_ARG; }
  return shadow_refresh((hipStream_t)s, jobs, n_jobs);
}
