// Internal C++ launcher interface shared by the .hip translation units.
// Everything here is host-side glue; the public boundary is include/vln_hip.h.
#pragma once
#include <hip/hip_ext.h>

#include "common.h"

struct vln_shadow_job;   // include/vln_hip.h
struct vln_gather_ride;
struct vln_wsum_step;    // include/vln_hip.h
struct vln_dot_step;     // include/vln_hip.h
struct vln_wgrad_job;
struct vln_colsum_job;

namespace vln {

// ACT_ACCUM (a flag on top of the activation): the finished result is ADDED to what the output holds (Y += act(X W^T + bias))
enum Act { ACT_NONE = 0, ACT_TANH = 1, ACT_RELU = 2, ACT_ACCUM = 8 };
// W_F32S (round 4): the weights are an fp32 array in memory like W_F32, but the product runs on the bf16 matrix pipe with BOTH
// operands split hi + lo (x_hi w_hi + x_lo w_hi + x_hi w_lo; the dropped lo*lo term is 2^-16 relative, fp32 accumulate): what the
// bf16 mode uses for the matrices it streams in fp32 all the same (EnvDropDecoder / MonitorDecoder `fp32_weights`).  Same bytes as
// W_F32, 3/16 of its matrix-pipe time -- it matters for the tall products (the BN-MLP's 1152 rows), not for the 64-row ones.
enum WType { W_F32 = 0, W_BF16 = 1, W_F32S = 2, W_F32X = 3 };
struct f32x_raw { float f; };          // W_F32X: fp32 in memory, three bf16 pieces per operand, six products (gemm_nt_body.h)
struct f32s_raw { float f; };          // element type tag of W_F32S operands in the kernels' templates

// thread-local last error text (vln_last_error_string)
void set_error(const char* fmt, ...);
const char* get_error();
int check_hip(hipError_t e, const char* what);

// run-time tunables (vln_set_tunable; defaults in api.hip) -- A/B switches, every setting computes the same results:
//   [0] gemm_nt split-K target (workgroups in flight, 384: 24-round A/B 1.794 vs 1.802 ms at 256)       [1] no-split rule for wide shallow products with a fused epilogue
//   [2] 16-column GEMM for narrow outputs on/off, [3] its largest K
//   [4] 1: two-kernel attention path instead of the one-launch rows
//   [5] gemm_nt: 0 fast form with depth-2 prefetch, 1 same, 2 bounds-checked form, 4 depth-4 prefetch
//   [6] weight gradients of precision 1: 0 packed grouped form, 1 exact fp32 form, 2 LDS-staged grouped form
//   [7] 1: persistent LSTM workgroups in dispatch order instead of one XCD per dependency group
//   [8] gemm_nt launches with >= 8 row tiles (the encoder's M = B * L products): 1 XCD-aware tile order (sharers of an X block behind one L2), 0 grid order
//   [9] 1: products handed to slab-summing consumers (gemm_nt_to_consumer: the Self-Monitor / Follower steps) take the plain call + reduce launch instead (A/B, same bits)
//   [10] 1: a pending gradient ride (vln_wgrad_ride_post) is always issued as its own launches (A/B)
//   [12] 1: tall products keep gemm_nt's 64-row tiles instead of the 16-row-block tiling of gemm_rows.h (A/B, same bits)
//   [14] 1: the backward recurrence's hand-off stores are always write-through (sc1), also when its group verified that it runs on one XCD (A/B)
//   [15] 1: a recurrence launch with passengers interleaves the two kinds of workgroup over all 8 XCDs (round 4) instead of keeping the
//        recurrence on XCDs 0-3 and the passengers on XCDs 4-7 (persist_role, encoder_persist.h) (A/B)
//   [11] >= 8: at most this many passenger workgroups carry a gradient ride (A/B; default: every idle CU up to the recurrence's own count)
// The EnvDrop step's graph key includes [0..7] (the step has one row tile: [8] never applies), so a changed tunable never replays a stale graph.
extern int g_tunable[16];

// ---- device-resident dropout offsets of the struct-driven steps -----------------------------------------------------------------
// vln_monitor_step / vln_follower_step / vln_bn_mlp carry `offset_base_dev`: while such a call issues its launches, every dropout
// site it builds (also inside the entry points it calls: vln_pe_dropout, vln_monitor_head_*, vln_bn_fwd / vln_bn_bwd) takes the
// device word as its base -- offset = (*base) * 8 + the struct's offset field (runtime.DeviceClock; see vln_embed_fwd).  The scope
// lives on the calling thread for the duration of the call only.
const unsigned long long*& drop_base_tls();
struct DropBaseScope {
  const unsigned long long* prev;
  explicit DropBaseScope(const void* base) : prev(drop_base_tls()) { drop_base_tls() = static_cast<const unsigned long long*>(base); }
  ~DropBaseScope() { drop_base_tls() = prev; }
};
inline DropSpec tls_drop(uint64_t seed, uint64_t offset, float p) { return DropSpec{seed, offset, p, drop_base_tls()}; }

// ---- optional per-kernel HIP-event timers (bench.py roofline leg; zero cost when disabled) -----------
enum KernelId {
  K_GEMM_NT = 0, K_GEMM_TN, K_ATTN_DOT, K_ATTN_WSUM, K_ATTN_BWD, K_LSTM_REC_FWD, K_LSTM_REC_BWD, K_FEAT_DROPOUT,
  K_LSTM_PW, K_REDUCE_EPI, K_COUNT
};
extern unsigned g_prof_mask;
void prof_begin(hipStream_t st, int kid, double algo_bytes);
void prof_end(hipStream_t st, int kid);
struct ProfScope {   // brackets the launches issued in its scope with an event pair on the SAME stream
  hipStream_t st; int kid; bool on;
  ProfScope(hipStream_t s, int k, double bytes) : st(s), kid(k), on((g_prof_mask >> k) & 1u) {
    if (on) prof_begin(st, kid, bytes);
  }
  ~ProfScope() { if (on) prof_end(st, kid); }
};

// Single-kernel timing: the event pair rides on the dispatch itself (hipExtLaunchKernelGGL), so the elapsed time
// is the kernel's own begin->end on the device, the figure rocprofv3 --kernel-trace reports, with no
// launch gap or event-record packet inside the bracket.
bool prof_slot(int kid, double algo_bytes, hipEvent_t* a, hipEvent_t* b);
// BatchNorm of two independent row segments in one launch pair (pointwise.hip; R1 == 0: one segment = vln_bn_fwd / vln_bn_bwd)
int bn_fwd_seg(const float* x, int64_t ldx, float* y, int64_t ldy, const float* gamma, const float* beta, float* running_mean,
               float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_rstd, int R, int R1, int64_t stat2, int D,
               float eps, float momentum, int training, int relu, uint64_t seed, uint64_t offset, uint64_t offset2, float p_drop,
               const uint8_t* row_zero, float* ws, int64_t ws_floats, void* s, const float* x2 = nullptr, int64_t ldx2 = 0);
int bn_bwd_seg(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* gamma,
               const float* mean, const float* rstd_or_var, float* dx, int64_t lddx, float* dgamma, float* dbeta, int R, int R1,
               int64_t stat2, int D, float eps, int training, int relu, int accumulate, uint64_t seed, uint64_t offset, uint64_t offset2,
               float p_drop, const uint8_t* row_zero, float* ws, int64_t ws_floats, void* s, const float* x2 = nullptr, int64_t ldx2 = 0);
// (x2, ldx2: nullable -- the SECOND segment's rows read from their own array, row r of the segment at x2 + r * ldx2)
// up to four row-wise elementwise forms (the ops of vln_ew) in one launch (pointwise.hip)
struct EwJob { int op; const float* a; long lda; const float* b; long ldb; int nb; float* y; long ldy; int rows, cols; };
int ew_multi(hipStream_t st, const EwJob* jobs, int n);
// up to four independent `out = sum of <= 4 strided matrices` in one launch (pointwise.hip)
struct AddNJob { float* out; long ldo; int rows, cols, n; const float* src[4]; long ld[4]; };
int add_n_multi(hipStream_t st, const AddNJob* jobs, int n);
// the same with sources still in split-K slabs (each source's slabs summed in slab order, then the sources in order)
// drop (p > 0): the sum times a dropout mask indexed r * drop_cols + drop_col0 + c (a column block of a wider dropped row)
struct AddNSvJob { float* out; long ldo; int rows, cols, n; SlabVec src[4]; DropSpec drop = DropSpec{0, 0, 0.f}; int drop_cols = 0, drop_col0 = 0; };
int add_n_sv_multi(hipStream_t st, const AddNSvJob* jobs, int n);
// vln_monitor_head_fwd with the gate product W_m [h0 ; moves] still in split-K slabs (+ bias); the sum is written to mg_out
int monitor_head_fwd_sv(hipStream_t st, SlabVec mg, float* mg_out, const float* c1, const float* word_w, const float* wc, const float* bc,
                        float* mem, float* prog, int B, int L, int H, uint64_t seed, uint64_t offset, float p);
struct GatherCheck;
GatherCheck gather_check(const void* table);   // features.hip: the registered extent of a feature table (vln_feature_table_extent)
extern int g_split_attn_enabled;  // encoder.hip: 0 after a four-workgroup attention exchange timed out (vln_persistent_check)
unsigned* sticky_dev_word();      // encoder.hip: host-mapped word of the current device that bounded waits raise on a timeout
int device_cus();                 // encoder.hip: CU count of the current device (queried once), 0 if unknown
// every launch in the library goes through launch_timed or VLN_LAUNCH
#define VLN_LAUNCH(kernel, grid, block, lds, st, ...) hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__)

template <typename F, typename... A>
inline void launch_timed(int kid, double algo_bytes, F kernel, dim3 grid, dim3 block, unsigned lds, hipStream_t st,
                         A... args) {
  hipEvent_t ea, eb;
  if (((g_prof_mask >> kid) & 1u) && prof_slot(kid, algo_bytes, &ea, &eb))
    hipExtLaunchKernelGGL(kernel, grid, block, lds, st, ea, eb, 0u, args...);
  else
    hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
}

#define VLN_CHECK_LAUNCH(what)                                \
  do {                                                           \
    int _st = vln::check_hip(hipGetLastError(), what);           \
    if (_st != VLN_OK) return _st;                               \
  } while (0)

// ---- gemm.hip -------------------------------------------------------------
// Y[M,N] = act(X[M,K] * W[N,K]^T + bias).  W is the streamed operand (fp32 or
// bf16 bits), X/Y fp32.  When the problem is split over K (nsplit > 1) partial
// slabs go to `ws` (nsplit*M*N floats) and a reduce+epilogue pass finishes.
// `ws_floats` bounds the split.  If `slabs_out`/`nsplit_out` are given the
// reduce pass is skipped and the caller's consumer kernel sums the slabs
// (slab s at ws + s*M*N, dense ld = N).
int gemm_nt_slabs(int M, int N, int K, int wtype, long ws_floats);   // slabs gemm_nt(..., nsplit_out) leaves for this product (gemm.hip)
int gemm_nt(hipStream_t st, const float* X, long ldx, const void* W, int wtype, long ldw, float* Y, long ldy,
            int M, int N, int K, const float* bias, int act, float* ws, long ws_floats, int* nsplit_out);
// (nsplit_out != nullptr  =>  raw partial sums ALWAYS go to ws, even for nsplit == 1; no bias/act applied)

// a product handed to a slab-summing consumer (gemm.hip): `ar` = what is left of the step's workspace
struct SlabArea { float* base; long left; };
int gemm_nt_plain_slabs(int M, int N, int K, int wtype, long ws_floats);
int gemm_rows_tiling(int M, int N, int cus, int* n_big, int* rb_big, int* tiles);   // gemm_rows.h's plan (host arithmetic only)
int gemm_nt_to_consumer(hipStream_t st, SlabArea& ar, const float* X, long ldx, const void* W, int wtype, long ldw, float* Y, long ldy,
                        int M, int N, int K, const float* bias, SlabVec* out);

// Y = act(X W^T + bias) and optionally Y2 = Y * dropout mask, finished output in the fewest launches
int gemm_nt_fused(hipStream_t st, const float* X, long ldx, const void* W, int wtype, long ldw, float* Y, long ldy, int M,
                  int N, int K, const float* bias, int act, float* Y2, long ldy2, DropSpec drop, float* ws, long ws_floats);

// D[N,K] (+)= A[Mt,N]^T * X[Mt,K]   (weight gradients; contraction over rows)
// precision 0: exact fp32 MFMA; 1: both operands split into bf16 hi + lo planes, three bf16 MFMAs, fp32 accumulation
int gemm_tn(hipStream_t st, const float* A, long lda, const float* X, long ldx, float* D, long ldd, int Mt,
            int N, int K, int accumulate, float* ws, long ws_floats, int precision = 0);

// every job's dW[N,K] (+)= dy[Mt,N]^T x[Mt,K] in one launch (precision 1) or one launch each (precision 0)
int wgrad_grouped(hipStream_t st, const ::vln_wgrad_job* jobs, int n, int Mt, int precision, float* ws, long ws_floats);
int wgrad_grouped_seg(hipStream_t st, const ::vln_wgrad_job* jobs, int n, int seg_rows, int n_seg, const int64_t* dy_seg,
                      const int64_t* x_seg, int precision, float* ws, long ws_floats);
int64_t wgrad_grouped_ws_floats(const ::vln_wgrad_job* jobs, int n, int Mt);

// out[c] (+)= sum_r A[r*lda + c]
int colsum(hipStream_t st, const float* A, long lda, float* out, int rows, int cols, int accumulate, float* ws,
           long ws_floats);

// every bias gradient of a module in one launch (two when the rows are split)
struct WgradRideArgs;              // wgrad_ride.h: a module's grouped weight / bias gradients as passengers of the backward recurrence launch
bool wgrad_ride_prepare(const ::vln_wgrad_job* jobs, int n, int Mt, int precision, const ::vln_colsum_job* cjobs, int nc, float* ws,
                        long ws_floats, WgradRideArgs* out);
int colsum_grouped(hipStream_t st, const ::vln_colsum_job* jobs, int n, int rows, float* ws, long ws_floats, int seg_rows = 0,
                   const int64_t* seg_stride = nullptr);

// out = act(sum_s slabs[s] + bias); optional second output out2 = out * dropout mask
int reduce_epilogue(hipStream_t st, const float* slabs, int nsplit, long slab_stride, long lds, float* out,
                    long ldo, int M, int N, const float* bias, int act, float* out2, long ldo2, DropSpec drop);

// Wt[K,N] = W[N,K]^T (fp32 or bf16 out);  Wc = cast(W)
int transpose_cast(hipStream_t st, const float* W, long ldw, void* Wt, int out_type, long ldt, int N, int K);
int cast_copy(hipStream_t st, const float* W, long ldw, void* out, int out_type, long ldo, int rows, int cols);
int linear_fwd_post(const float* X, long ldx, const void* W, int wtype, long ldw, float* Y, long ldy, int M, int N, int K);   // gemm.hip
int linear_fwd_post_flush(hipStream_t st, float* ws, long ws_floats);
int colsum_post(const ::vln_colsum_job* jobs, int n, int rows);
int layout_post(int kind, const float* src, float* dst, void* dst_lp, int B, int L, int W, DropSpec dr);     // gemm.hip (kind 0: tm -> bm, 1: bm -> tm)
int layout_post_flush(hipStream_t st);
int posted_drop();
int colsum_post_flush(hipStream_t st, float* ws, long ws_floats);
int shadow_refresh(hipStream_t st, const ::vln_shadow_job* jobs, int n);   // all shadows of a module in one launch

// ---- attention.hip --------------------------------------------------------
// dots[b,s] = ctx[b,s,:] . vec[b,:]   (ctx streamed: fp32 or bf16)
int attn_dot(hipStream_t st, const void* ctx, int ctype, const float* vec, long ldv, float* dots, int B, int S,
             int D);
// Forward weighted sum: attn = softmax(mask(logits)); out[b,:] = sum_s attn[b,s] ctx[b,s,:]
int attn_softmax_wsum(hipStream_t st, const void* ctx, int ctype, const float* logits, const uint8_t* mask,
                      float* attn, float* out, long ldo, int B, int S, int D);
// Backward: dl = attn*(dalpha - sum(attn*dalpha)) [+ attn*(dattn_ext - ...)]; dvec[b,:] = sum_s dl ctx;
// if dctx != null: dctx[b,s,:] += attn[b,s]*dwc[b,:] + dl[b,s]*vec[b,:]
int attn_bwd(hipStream_t st, const void* ctx, int ctype, const float* attn, const float* dalpha,
             const float* dattn_ext, const float* dwc, long lddwc, const float* vec, long ldvec, float* dvec,
             long lddvec, float* dctx, float* dl_out, int B, int S, int D);
// One launch per attention row block (attention_fused.h) with the two-kernel path as the fallback for shapes that
// do not fit the register-resident configurations; `dots_scratch` [B,S] is used by the fallback only.
int attn_fwd_rows(hipStream_t st, const void* ctx, int ctype, const float* vec, long ldv, const uint8_t* mask, float* attn,
                  float* out, long ldo, float* dots_scratch, int B, int S, int D);
int attn_bwd_rows(hipStream_t st, const void* ctx, int ctype, const float* attn, const float* dwc, long lddwc,
                  const float* dattn_ext, float* dvec, long lddvec, float* dl_out, float* dots_scratch, int B, int S, int D);
// same, with the vector operand still in split-K slabs (SlabVec) and an optional write-back of the summed vector
// vec_out (nullable): the summed vector (+ its bias, x vmul) written back, [B, D]; vmul (nullable, [D]): column-wise multiplier of
// the vector; add0 (nullable, [1]): a scalar added to every dot
int attn_dot_sv(hipStream_t st, const void* ctx, int ctype, SlabVec vec, float* dots, int B, int S, int D, float* vec_out = nullptr,
                long ldvo = 0, const float* vmul = nullptr, const float* add0 = nullptr);
// `sync` / `sync_bytes` (nullable): the caller's zero-initialised exchange buffer of attn_split_sync_floats(B) floats; with it
// (and B * 4 <= the device's CU count) a row's block is split over FOUR workgroups (attention_split.h)
int attn_fwd_rows_sv(hipStream_t st, const void* ctx, int ctype, SlabVec vec, float* vec_out, long ldvo, const uint8_t* mask,
                     float* attn, float* out, long ldo, float* dots_scratch, int B, int S, int D, void* sync = nullptr, long sync_bytes = 0);
int attn_bwd_rows_sv(hipStream_t st, const void* ctx, int ctype, const float* attn, SlabVec dwc, float* dwc_out, long lddo,
                     const float* dattn_ext, float* dvec, long lddvec, float* dl_out, float* dots_scratch, int B, int S, int D,
                     void* sync = nullptr, long sync_bytes = 0);
// The EnvDrop step's text attention on the PROJECTED context K = ctx W_in with the LSTM cell's pointwise stage in the same launch
// (attention_textk.h; four workgroups per row only: attn_textk_ok says whether the shape / device / exchange buffer allow it)
struct LstmPwFwd; struct LstmPwBwd;
bool attn_textk_ok(int ctype, int B, int S, int D, const void* sync, long sync_bytes);
int attn_textk_fwd(hipStream_t st, const void* ctx, int ctype, const float* kctx, const uint8_t* mask, float* alpha, float* out,
                   long ldo, const LstmPwFwd& pw, int B, int S, int D, void* sync, long sync_bytes);
int attn_textk_bwd(hipStream_t st, const void* ctx, int ctype, const float* kctx, const float* alpha, SlabVec dwc, float* dwc_out,
                   long lddo, float* dq, long lddq, float* dl_out, const LstmPwBwd& pb, int B, int S, int D, void* sync,
                   long sync_bytes);
// candidate logits of a step + the sampled-action branch on them (mask, softmax, draw / given action, log-prob, entropy; the
// action also to a host-mapped word) in ONE launch: envdrop.hip's step (6) when the caller attached a sampler
int cand_logits_sample(hipStream_t st, const void* cand, int ctype, SlabVec q, float* logits, const uint8_t* mask, const int64_t* action_in,
                       int64_t* action_out, int64_t* action_host, float* probs, float* logp, float* ent, uint64_t seed, uint64_t offset,
                       const uint64_t* offset_base_dev, int B, int C, int D);
// dctx[b,s,:] (+)= sum_t alpha_t[b,s] g_t[b,:] + dl_t[b,s] q_t[b,:]   (host arrays of T device pointers)
int attn_dctx_deferred(hipStream_t st, const float* const* alpha, const float* const* dl, const float* const* g, long ldg,
                       const float* const* q, long ldq, int T, float* dctx, int B, int S, int D, int accumulate,
                       const uint64_t* drop_seed = nullptr, const uint64_t* drop_off = nullptr, const float* drop_p = nullptr,
                       float* dk = nullptr);      // dk: the (dl, q) half written to its own [B,S,D] tensor instead of into dctx
// dvec[b,:] = sum_c w[b,c] ctx[b,c,:]  (plain weighted sum, no softmax)
int rows_wsum(hipStream_t st, const void* ctx, int ctype, const float* w, float* out, long ldo, int B, int S,
              int D);
int rows_wsum_multi(hipStream_t st, const vln_wsum_step* steps, int T, int ctype, int B, int D, long ldo, float ce_scale,
                    const float* ce_dloss, long ignore_index);
int attn_dot_multi(hipStream_t st, const vln_dot_step* steps, int T, int ctype, int B, int D, long ldv);

// ---- features.hip ---------------------------------------------------------
struct GatherRolloutArgs;
int gather_ride_args(const ::vln_gather_ride& r, int t0, GatherRolloutArgs* a);      // steps [t0, t0 + 12) as a kernel argument block
int gather_ride_launch(hipStream_t st, const ::vln_gather_ride& r);                  // the ride as its own launch(es)

// ---- pointwise.hip --------------------------------------------------------
struct LstmPwFwd {
  const float* gates; int nsplit; long slab_stride;  // pre-activation slabs [nsplit][B,4H]
  const float* bias_a; const float* bias_b;          // b_ih, b_hh (nullable)
  const float* c0; long ldc0;
  float* h1; long ldh1; float* c1; long ldc1;
  float* act;          // [B,4H] saved sigma(i),sigma(f),tanh(g),sigma(o)
  float* tanh_c1;      // [B,H] saved
  float* h1_drop; long ldh1d; DropSpec drop;         // optional dropped copy of h1
  int B, H;
};
int lstm_pointwise_fwd(hipStream_t st, const LstmPwFwd& a);
struct LstmPwBwd {
  const float* dh1_a; long ld_a;   // external grad on h1 (nullable)
  SlabVec dh1_b;                   // grad flowing from the dropped copy (p nullable); multiplied by the mask
  SlabVec dh1_b2;                  // second contribution to the dropped copy's grad (p nullable)
  DropSpec drop;
  const float* dc1; long lddc1;    // external grad on c1 (nullable)
  const float* act; const float* tanh_c1; const float* c0; long ldc0;
  float* dgates; long lddg;        // [B,4H]
  float* dc0; long lddc0;
  int B, H;
};
int lstm_pointwise_bwd(hipStream_t st, const LstmPwBwd& a);

// y = x * dropout ; generic small elementwise helpers
int scale_dropout(hipStream_t st, const float* x, long ldx, float* y, long ldy, int rows, int cols, DropSpec d);
// feature dropout in place on x[..., :img] of rows of length img+angle (policy.py:228-231)
int feat_dropout_inplace(hipStream_t st, void* x, int xtype, long rows, int img, int angle, DropSpec d,
                         void* copy_bf16);
int export_dropout_mask(hipStream_t st, float* out, long n, DropSpec d);
int fill_f32(hipStream_t st, float* p, long n, float v);
int add_inplace(hipStream_t st, float* y, long ldy, const float* x, long ldx, int rows, int cols);

}  // namespace vln
