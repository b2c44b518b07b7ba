/* vln_hip.h -- C ABI of the MI355X (gfx950) training hot path for the R2R navigation agents.
 *
 * The reference (IMNearth/Curriculum-Learning-For-VLN, tasks/R2R-judy) has no FFI layer: its boundary for
 * this path is the nn.Module surface of src/model/units.py and src/model/policy.py.  This library is what a
 * torch.autograd.Function inside a drop-in nn.Module binds to (ctypes).  Every entry
 * point takes raw DEVICE pointers, explicit sizes/strides and a hipStream_t, returns an int status
 * (0 = VLN_OK) and never allocates or frees CALLER memory.  State the library keeps on its own behalf (all of it
 * process-wide, none of it changes results): a thread-local error string; the hipGraph caches of memoised launch chains
 * (vln_set_graphs / vln_graph_stats); the A/B tunables and mode switches (vln_set_tunable, vln_set_persistent); the optional
 * per-kernel timers (vln_prof_*); a per-buffer launch sequence for the recurrence's data-tagged hand-offs; one host-mapped
 * status line per device that a timed-out bounded wait or an out-of-range gather index raises (vln_persistent_check); the
 * registered extents of feature tables (vln_feature_table_extent).  Tensors are row-major; "ld" = row
 * stride in elements.
 *
 * dtype codes: 0 = fp32, 1 = bf16 (raw uint16 bits).  Activations/gradients are fp32; the streamed
 * operands (weights, feature/context tensors) may be bf16 shadows with fp32 accumulation.
 */
#ifndef VLN_HIP_H
#define VLN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* vln_stream_t; /* hipStream_t */

#define VLN_OK 0
#define VLN_ERR_ARG 1
#define VLN_ERR_HIP 2
#define VLN_F32 0
#define VLN_BF16 1
/* Weight operands only (ABI v13): an fp32 array in memory like VLN_F32, multiplied on the bf16 matrix pipe with both operands
 * split hi + lo (x_hi w_hi + x_lo w_hi + x_hi w_lo, fp32 accumulate; 2^-16 relative) -- what the bf16 mode uses for the matrices
 * it streams in fp32 all the same (vln_envdrop_weights.f32_mask, vln_monitor_weights.f32_mask, vln_bn_mlp.wtype). */
#define VLN_F32S 2
/* Weight operands only (ABI v14): fp32 in memory, both operands as THREE bf16 pieces, six products (every pair above 2^-24 of the
 * full product): fp32-grade results at 6/16 of the exact fp32 MFMA's time.  vln_bn_mlp_fwd uses it for the forward Linear of a
 * VLN_F32S layer (a ReLU decision follows the product); vln_linear_fwd takes it like any weight type. */
#define VLN_F32X 3
#define VLN_ACT_NONE 0
#define VLN_ACT_TANH 1
#define VLN_ACT_RELU 2
#define VLN_ACT_ACCUM 8   /* flag, OR-ed onto an activation: the finished result is ADDED to the output (vln_linear_fwd: Y += ...) */

int vln_abi_version(void);     /* 19 */
/* sizeof(struct vln_<name>) as THIS library was compiled, -1 for an unknown name: a binding checks its struct mirrors against
 * it when it loads the library (a mirror that is one field short makes the kernels read wild pointers). */
int64_t vln_struct_size(const char* name);
const char* vln_last_error_string(void);

/* Optional per-kernel timers (measurement only; the reference has no counterpart): when enabled for a kernel id,
 * every launch of that kernel is bracketed by a hipEvent pair on the launch stream.  vln_prof_read sums and clears
 * them and returns the algorithmic bytes of those launches. */
/* Launch chains (one per LSTM time step / per decoder step) are memoised as hipGraphs keyed by their argument
 * block (csrc/graph_cache.h).  vln_set_graphs(0) forces plain launches; results are identical. */
int vln_set_graphs(int on);
/* The batch hand-over (reference: agent/base.py:114-178 marshals every batch on the host, `.to(device)` copies it).  The caller
 * packs a batch's small tensors into ONE pinned host blob (a multiple of 16 bytes, 16-byte aligned) and stores the blob's DEVICE
 * address (vln_host_device_pointer) into a ring of `ring` pinned 8-byte slots; vln_host_fetch launches a kernel that copies
 * `nbytes` from the blob named by slot (*seq % ring) to `dst` and then bumps *seq (a zero-initialised DEVICE word; `done` = a
 * zero-initialised device scratch word).  Inside a captured iteration its arguments repeat: a new batch costs the host one store
 * into the next slot -- it may run ahead of the device by up to `ring` - 1 batches.  `slots_dev` = the ring's device address.
 * A blob must not be rewritten, nor its slot reused, before the fetch that reads it has run. */
int vln_host_device_pointer(const void* host, void** dev);
int vln_host_fetch(const uint64_t* slots_dev, int ring, uint64_t* seq, uint32_t* done, void* dst, int64_t nbytes, vln_stream_t s);

/* Device-resident counters: ONE tiny launch adds inc to every listed word (width 8: uint64, width 4: uint32) -- the dropout
 * clock (`offset_base_dev`) and the recurrence's launch sequence (vln_lstm_sync_seq_offset) of a whole-iteration graph. */
#define VLN_TICK_MAX 8
typedef struct vln_tick_item { void* word; uint64_t inc; int32_t width; int32_t pad_; } vln_tick_item;
int vln_tick(const vln_tick_item* items, int n /* 1..VLN_TICK_MAX */, vln_stream_t s);
/* hipGraph memoisation counters since load: out[0] replays, out[1] captures (= misses), out[2] times a chain's capturing was paused (2 x capacity captures without one replay) */
int vln_graph_stats(int64_t out[3]);
/* performance / A-B tunables, ids 0..15 (never change results beyond summation order; documented in
 * csrc/vln_internal.h): 0 = gemm split-K workgroup target (256), 1 = keep wide shallow fused-epilogue products unsplit,
 * 2/3 = 16-column GEMM on / its largest K, 4 = two-kernel attention, 5 = gemm_nt form, 6 = weight-gradient form,
 * 7 = persistent-LSTM workgroup order, 8 = XCD-aware tile order of gemm_nt launches with many row tiles (1), 9 = spare, 10 = 1: gradient rides never travel as passengers, 11 = cap on a ride's passenger workgroups */
int vln_set_tunable(int id, int value);
/* Measurement only: `launches` dependent launches of a trivial kernel (each reads what the one before wrote, rotated by `shift`
 * float4 elements so the bytes come from another XCD) ping-ponging between a and b [n_floats]: the price of a kernel boundary
 * inside the caller's own stream / captured graph (scripts/boundary_probe.hip is the stand-alone form). */
/* Test only: `workgroups` x 1024 threads that hold `lds_bytes` of LDS each and stay resident for `micros` microseconds -- a
 * stand-in for a communication kernel resident on another stream while the persistent recurrence runs. */
int vln_debug_occupy(int workgroups, int lds_bytes, int micros, vln_stream_t s);
int vln_debug_trivial_chain(float* a, float* b, int n_floats, int launches, int shift, vln_stream_t s);
int vln_prof_enable(int kernel_id, int on);
const char* vln_prof_kernel_name(int kernel_id);   /* NULL past the last id */
int vln_prof_read(int kernel_id, int64_t* launches, double* total_ms, double* total_bytes);

/* ---- generic operators ---------------------------------------------------------------------------- */

/* nn.Linear forward / dX product: Y[M,N] = act(X[M,K] W[N,K]^T + bias).  Replaces the F.linear calls inside
 * units.py:69,106,120,144,146,181-184 and policy.py:115,124,189,204.  ws: split-K scratch (may be NULL). */
int vln_linear_fwd(const float* X, int64_t ldx, const void* W, int wtype, int64_t ldw, float* Y, int64_t ldy,
                   int M, int N, int K, const float* bias, int act, float* ws, int64_t ws_floats, vln_stream_t s);
/* ABI v16, host arithmetic only (no device call): the row tiling a tall product (M >= 256 rows, fp32 / split-fp32 weights, not
 * split over K) takes on a device of `cus` compute units -- csrc/gemm_rows.h: row tiles [0, n_big) of rb_big 16-row blocks, the
 * others of rb_big - 1, `tiles` = row tiles x ceil(N / 64) workgroups chosen to fill whole rounds of the CUs.  Returns 1 and the
 * tiling, or 0 when the product keeps the 64 x 64 tiles.  The results do not depend on the tiling (bit-identical). */
int vln_gemm_rows_tiling(int M, int N, int cus, int* n_big, int* rb_big, int* tiles);
/* The same product left as its split-K partial slabs (ABI v14): slabs [*n_slabs][M,N] in ws, Y = their sum in slab order -- for a
 * consumer that adds the partials while it loads them (vln_lstm_pointwise_fwd's `nsplit`), so that no reduce launch sits between the
 * two.  ws_floats >= M * N; more lets the contraction split over more workgroups. */
int vln_linear_fwd_slabs(const float* X, int64_t ldx, const void* W, int wtype, int64_t ldw, int M, int N, int K, float* ws,
                         int64_t ws_floats, int* n_slabs, vln_stream_t s);
/* weight gradient: D[N,K] (+)= A[Mt,N]^T X[Mt,K] (autograd of the same Linear layers, batched over steps) */
int vln_linear_wgrad(const float* A, int64_t lda, const float* X, int64_t ldx, float* D, int64_t ldd, int Mt,
                     int N, int K, int accumulate, float* ws, int64_t ws_floats, vln_stream_t s);
/* bias gradient: out[c] (+)= sum_r A[r, c]; ws (nullable) lets long contractions split over workgroups */
/* Same contraction with a choice of arithmetic: precision 0 = exact fp32 MFMA (what vln_linear_wgrad does),
 * 1 = both operands split into bf16 hi + lo planes, three bf16 MFMAs per product with fp32 accumulation
 * (relative error 2^-16 per product; the bf16 compute mode's weight gradients). */
int vln_linear_wgrad_p(const float* A, int64_t lda, const float* X, int64_t ldx, float* D, int64_t ldd, int Mt,
                       int N, int K, int accumulate, int precision, float* ws, int64_t ws_floats, vln_stream_t s);
/* All weight gradients of a module in one launch: job i forms dw[N,K] (+)= dy[Mt,N]^T x[Mt,K]; every job contracts
 * over the same Mt rows (the rollout stash: steps x batch).  precision as in vln_linear_wgrad_p; with precision 0, or
 * operands the grouped kernel does not take, it is one vln_linear_wgrad_p launch per job. */
#define VLN_WGRAD_MAX_JOBS 16
typedef struct vln_wgrad_job {
  const float* dy; const float* x; float* dw;
  int64_t ld_dy, ld_x, ld_dw;
  int N, K, accumulate;
  int rows;      /* 0: the call's Mt rows.  > 0 (ABI v15, gradient rides only: vln_wgrad_ride_add): this product contracts over its
                  * FIRST `rows` rows and its operands have no more than that (rows past them count as zeros) */
} vln_wgrad_job;
int vln_wgrad_grouped(const vln_wgrad_job* jobs, int n_jobs, int Mt, int precision, float* ws, int64_t ws_floats, vln_stream_t s);
/* (ABI v17) POST a plain product y[M,N] = x[M,K] w[N,K]^T (vln_linear_fwd without bias / activation; w_type VLN_F32 or VLN_BF16) for the
 * NEXT vln_wgrad_grouped call of this thread to issue as extra workgroups of its pack launch -- the encoder backward's d x = dgates W_ih
 * beside the pack of the same dgates (model.py:9-66 backward).  Same tiles and arithmetic as vln_linear_fwd: bit-identical.  ALWAYS
 * follow the wgrad call with vln_linear_fwd_post_flush, which issues a post nobody took (another weight-gradient form, a split
 * product) as its own launch; a second post while one is pending is an error. */
int vln_linear_fwd_post(const float* x, int64_t ldx, const void* w, int w_type, int64_t ldw, float* y, int64_t ldy, int M, int N, int K);
int vln_linear_fwd_post_flush(float* ws, int64_t ws_floats, vln_stream_t s);
/* the same for grouped column sums (vln_colsum_grouped's jobs over `rows` rows: the layer's bias gradients): they ride in the launch
 * that carries a posted product; vln_colsum_post_flush issues what nobody took */
int vln_colsum_post(const vln_colsum_job* jobs, int n_jobs, int rows);
/* POST a layout change -- kind 0 = vln_tm_to_bm's arguments ([L,B,W] -> [B,L,W] (+ bf16 copy) with dropout), kind 1 = vln_bm_to_tm's -- for
 * the NEXT narrow product of this thread (a vln_linear_fwd / vln_linear_fused that takes the 16-column kernel: N <= 1024, M <= 64 ...) to
 * issue as extra workgroups of its launch; vln_layout_post_flush issues a post nobody took.  Same results as the stand-alone calls. */
int vln_layout_post(int kind, const float* src, float* dst, void* dst_bf16 /* kind 0 only, nullable */, int B, int L, int W, uint64_t seed,
                    uint64_t offset, float p, const uint64_t* offset_base_dev);
int vln_layout_post_flush(vln_stream_t s);
/* Forget every post of this thread that was neither taken nor flushed (a caller that raised between its post and its flush; its buffers
 * may be gone); returns how many.  The drop-in modules call it at the top of the calls that post. */
int vln_posted_drop(void);
int vln_colsum_post_flush(float* ws, int64_t ws_floats, vln_stream_t s);
int vln_colsum(const float* A, int64_t lda, float* out, int rows, int cols, int accumulate, float* ws,
               int64_t ws_floats, vln_stream_t s);
/* weight shadows (transposed and/or bf16 copies), refreshed once per optimizer step */
/* All bias gradients of a module in one launch: out1[c] (and out2[c], nullable: an LSTM's b_ih and b_hh receive the
 * same sum) (+)= sum_r A[r*lda + c], every job over the same `rows`.  cols and lda multiples of 4, A 16-byte aligned. */
#define VLN_COLSUM_MAX_JOBS 12
typedef struct vln_colsum_job {
  const float* A; float* out1; float* out2;
  int64_t lda;
  int cols, accumulate;
} vln_colsum_job;
int vln_colsum_grouped(const vln_colsum_job* jobs, int n_jobs, int rows, float* ws, int64_t ws_floats, vln_stream_t s);
/* Gradient rides (ABI v14): a module's grouped weight / bias gradients as PASSENGER workgroups of the next backward recurrence
 * launch on the same stream.  The encoder's BPTT (vln_lstm_seq_bwd) keeps half of the chip's CUs busy with latency-bound
 * hand-offs; the DECODER's parameter gradients depend on nothing it produces.  vln_wgrad_ride_post leaves the jobs (what
 * vln_wgrad_grouped + vln_colsum_grouped would take: at most 8 products and 4 column sums over the same `rows`, precision as in
 * vln_wgrad_grouped) pending on stream s; the next vln_lstm_seq_bwd on s carries them when it can -- counter-protocol
 * persistent kernel, >= 8 idle CUs, bf16-mode precision (1 / 2), operands the packed kernels take without a reduce launch --
 * and otherwise issues them as their own launches in front of the recurrence; vln_wgrad_ride_flush issues a still-pending ride
 * (no recurrence followed).  Results are bit-identical to vln_wgrad_grouped + vln_colsum_grouped.  The operands, outputs and
 * ws must stay valid and unwritten by other work until the stream has passed the carrying (or flushing) call.
 * vln_wgrad_ride_stats: out[0] rides carried as passengers, out[1] rides issued as their own launches, since load.
 * Replaces nothing in the reference: it is a schedule of autograd's parameter gradients (trainer.py:421-427). */
int vln_wgrad_ride_post(const vln_wgrad_job* jobs, int n_jobs, const vln_colsum_job* cjobs /*nullable*/, int n_cjobs, int rows,
                        int precision, float* ws, int64_t ws_floats, vln_stream_t s);
int vln_wgrad_ride_flush(vln_stream_t s);
/* Append ONE more product, over its own `rows` <= the pending ride's rows, to the ride pending on stream s (ABI v15): the
 * instruction encoder's head (`enc2dec`: 64 rows against the decoder's steps x batch) joins the decoder's ride instead of standing
 * as its own launch in front of the BPTT.  Returns 1 when it was appended (same precision, a free slot, the pack area holds it),
 * 0 when not -- the caller then forms the gradient itself. */
int vln_wgrad_ride_add(const vln_wgrad_job* job, int rows, int precision, vln_stream_t s);
int vln_wgrad_ride_drop(vln_stream_t s);     /* forget a pending ride without issuing it (its iteration was abandoned); 1 if one was pending */
int vln_wgrad_ride_stats(int64_t out[2]);

/* Parameter gradients of the per-step C calls (vln_monitor_step_bwd, vln_follower_step_bwd, vln_bn_mlp_bwd) once per ROLLOUT
 * (ABI v11).  A per-step call that accumulates into p.grad reads and rewrites every weight gradient of the module per decoder
 * step (Self-Monitor: 22 pack + 22 contraction + 22 bias launches per iteration, 0.7 ms).  With `defer` set in its grads
 * struct the call SKIPS those launches and instead writes the jobs it would have run -- this step's operand pointers -- into
 * the caller's vln_param_jobs.  When every step's saved-activation block and backward scratch are slots of two arenas
 * (constant distance between consecutive steps), operand i of all T steps is ONE segmented matrix: rows
 * [t * seg_rows, (t + 1) * seg_rows) at base_i + t * seg_stride_i.  The *_seg calls contract / sum over all T * seg_rows rows
 * in one pack + one contraction (+ one reduce) / one launch; operands the packed kernel does not take are handled segment by
 * segment.  functional.RolloutWgrads drives this from the end of autograd's backward pass. */
#define VLN_PARAM_JOBS_MAX 8
typedef struct vln_param_jobs {
  vln_wgrad_job w[VLN_PARAM_JOBS_MAX]; vln_colsum_job c[VLN_PARAM_JOBS_MAX];
  int32_t nw, nc, rows, precision;
} vln_param_jobs;
int vln_wgrad_grouped_seg(const vln_wgrad_job* jobs, const int64_t* dy_seg_stride, const int64_t* x_seg_stride, int n_jobs,
                          int seg_rows, int n_seg, int precision, float* ws, int64_t ws_floats, vln_stream_t s);
int vln_colsum_grouped_seg(const vln_colsum_job* jobs, const int64_t* seg_stride, int n_jobs, int seg_rows, int n_seg, float* ws,
                           int64_t ws_floats, vln_stream_t s);
int64_t vln_wgrad_grouped_ws_floats(const vln_wgrad_job* jobs, int n_jobs, int Mt);   /* workspace the one-contraction form wants */
int vln_transpose_cast(const float* W, int64_t ldw, void* Wt, int out_type, int64_t ldt, int N, int K, vln_stream_t s);
int vln_cast_copy(const float* W, int64_t ldw, void* out, int out_type, int64_t ldo, int rows, int cols, vln_stream_t s);
/* All weight shadows of a module in ONE launch (they are refreshed once per optimizer step): job = fp32 matrix src
 * [N,K] (+ src2, same layout, nullable: the two LSTM biases are summed) -> dst [N,K] (nullable) and/or dst_t [K,N]
 * (nullable) in out_type (VLN_F32 / VLN_BF16); ld_* in elements. */
#define VLN_SHADOW_MAX_JOBS 24
typedef struct vln_shadow_job {
  const float* src; const float* src2; void* dst; void* dst_t;
  int64_t ld_src, ld_dst, ld_dst_t;
  int N, K, out_type, pad_;
} vln_shadow_job;
int vln_shadow_refresh(const vln_shadow_job* jobs, int n_jobs, vln_stream_t s);
/* The three launches at the top of a training iteration -- vln_host_fetch (skipped when slots_dev is NULL), vln_tick (n_ticks items)
 * and vln_shadow_refresh (n_jobs <= VLN_SHADOW_MAX_JOBS jobs: the weight shadows of every module whose parameters the last optimizer
 * step changed) -- as ONE launch: none depends on another (the pull is PCIe-bound, the refresh HBM-bound).  Same results as the
 * three calls. */
int vln_prologue(const uint64_t* slots_dev, int ring, uint64_t* seq, uint32_t* done, void* dst, int64_t nbytes,
                 const vln_tick_item* ticks, int n_ticks, const vln_shadow_job* jobs, int n_jobs, vln_stream_t s);

/* out_t[b,:] = sum_s w_t[b,s] ctx_t[b,s,:] for T steps in one launch (ctx_t [B,S_t,D] contiguous, w_t [B,S_t], out_t rows of
 * leading dimension ldo): the d(query) of the candidate logits of a whole rollout (policy.py:199-206 backward). */
typedef struct vln_wsum_step {
  const void* ctx; const float* w; float* out; int S;
  /* optional (w == NULL): the weights are the cross-entropy gradient formed on the fly, w[b,s] = g * (probs[b,s] - 1[s ==
   * target[b]]) (0 for rows whose target is ignore_index) -- the d logits of vln_masked_ce_multi_fwd without materialising them */
  const float* probs; const int64_t* target;
} vln_wsum_step;
/* ce_scale / ce_dloss ([1] device scalar) / ignore_index: only read for steps given as (probs, target); g = *ce_dloss * ce_scale */
int vln_rows_wsum_multi(const vln_wsum_step* steps, int T, int ctype, int B, int D, int64_t ldo, float ce_scale,
                        const float* ce_dloss, int64_t ignore_index, vln_stream_t s);

/* ABI v19: out_t[b,:] = src_t[b, index_t[b], :] for T steps in ONE launch -- the previous-action rows of a TEACHER-FORCED rollout
 * (a_t_prev = a_t_cand[arange, a_t], monitor.py:191 / follower.py:164: with teacher forcing a_t is the batch's target, every step's
 * row is known when the rollout starts; the loop issued one indexing launch per step on its dependent chain).  src_t [B, C_t, F]
 * contiguous fp32, index_t [B] int64 (a negative index counts from the end, as torch's; an index outside [-C_t, C_t) yields a zero
 * row and raises sticky word 1 of vln_persistent_check), out_t [B, F] dense; F % 4 == 0, T <= VLN_CE_MAX_STEPS. */
typedef struct vln_select_step { const float* src; const int64_t* index; float* out; int C; } vln_select_step;
int vln_select_rows_multi(const vln_select_step* steps, int T, int B, int F, vln_stream_t s);

/* dots_t[b,s] = ctx_t[b,s,:] . vec_t[b,:] for T steps in one launch (ctx_t [B,S_t,D] contiguous, vec_t rows of leading
 * dimension ldv, dots_t [B,S_t] dense): the candidate logits of a whole teacher-forced rollout (policy.py:199-206), whose
 * queries W_c drop(h_tilde_t) come from ONE GEMM over (steps x batch) rows -- see vln_envdrop_step.defer_logits. */
typedef struct vln_dot_step { const void* ctx; const float* vec; float* dots; int S; } vln_dot_step;
int vln_attn_dot_multi(const vln_dot_step* steps, int T, int ctype, int B, int D, int64_t ldv, vln_stream_t s);

/* SoftDotAttention / VisualSoftDotAttention pieces (units.py:100-122, 138-160):
 *   dots[b,s] = ctx[b,s,:] . vec[b,:]                          torch.bmm(context, target)
 *   attn = softmax(mask(logits)); out[b,:] = sum_s attn ctx     masked_fill_ + Softmax + torch.bmm(attn3, context)
 *   backward of both, with optional in-place dctx accumulation */
int vln_attn_dot(const void* ctx, int ctype, const float* vec, int64_t ldv, float* dots, int B, int S, int D, vln_stream_t s);
int vln_attn_softmax_wsum(const void* ctx, int ctype, const float* logits, const uint8_t* mask, float* attn,
                          float* out, int64_t ldo, int B, int S, int D, vln_stream_t s);
int vln_rows_wsum(const void* ctx, int ctype, const float* w, float* out, int64_t ldo, int B, int S, int D, vln_stream_t s);
/* One launch per attention (dots -> softmax -> weighted sum with the [S,D] block of a batch row resident in the
 * workgroup's registers); shapes that do not fit fall back to the two launches above and need dots_scratch [B,S].
 * Forward = SoftDotAttention / VisualSoftDotAttention core (units.py:106-118,150-158); backward returns d(query) and
 * optionally d(logits) (dl_out), the in-place dctx update being replaced by vln_attn_dctx_deferred.
 * sync / sync_bytes (nullable): a zero-initialised scratch of vln_attn_sync_bytes(B) bytes that the caller keeps for these
 * calls alone -- a row's block is then split over FOUR workgroups (see vln_envdrop_step.attn_sync). */
int vln_attn_fwd_rows(const void* ctx, int ctype, const float* vec, int64_t ldv, const uint8_t* mask /*nullable*/,
                      float* attn /*nullable*/, float* out, int64_t ldo, float* dots_scratch /*nullable*/, int B, int S, int D,
                      void* sync /*nullable*/, int64_t sync_bytes, vln_stream_t s);
int vln_attn_bwd_rows(const void* ctx, int ctype, const float* attn, const float* dwc, int64_t lddwc,
                      const float* dattn_ext /*nullable*/, float* dvec, int64_t lddvec, float* dl_out /*nullable*/,
                      float* dots_scratch /*nullable*/, int B, int S, int D, void* sync /*nullable*/, int64_t sync_bytes,
                      vln_stream_t s);
/* dctx[b,s,:] (+)= sum_t alpha[t][b,s] * g[t][b,:] + dl[t][b,s] * q[t][b,:]  -- the context gradient of a whole rollout
 * (T decoder steps) in one pass; alpha/dl/g/q are HOST arrays of T device pointers; a step may carry only its (alpha, g) or
 * only its (dl, q) pair (the other two NULL).  dk (nullable, ABI v15): the (dl, q) half is written to THIS [B,S,D] tensor
 * instead of into dctx -- with vln_envdrop_step.kctx the q operands are the steps' drop(h_1) rows and
 * dctx = sum_t alpha_t g_t + dk W_in^T (one vln_linear_fwd with VLN_ACT_ACCUM onto dctx). */
int vln_attn_dctx_deferred(const float* const* alpha, const float* const* dl, const float* const* g, int64_t ldg,
                           const float* const* q, int64_t ldq, int T, float* dctx, int B, int S, int D, int accumulate,
                           float* dk, vln_stream_t s);
/* Same, with step t's contribution multiplied by the dropout mask (drop_seed[t], drop_off[t], drop_p[t]; mask index = flat
 * [B,S,D] index) the attended tensor went through in that step, and with (dl[t], q[t]) allowed to be NULL together
 * (a pure outer-product term).  HOST arrays of T entries.  The Self-Monitor agent's context and candidate gradients. */
int vln_attn_dctx_deferred_drop(const float* const* alpha, const float* const* dl, const float* const* g, int64_t ldg,
                                const float* const* q, int64_t ldq, int T, float* dctx, int B, int S, int D, int accumulate,
                                const uint64_t* drop_seed, const uint64_t* drop_off, const float* drop_p,
                                const uint64_t* offset_base_dev /*nullable, see vln_embed_fwd*/, vln_stream_t s);
int vln_attn_bwd(const void* ctx, int ctype, const float* attn, const float* dalpha, const float* dattn_ext,
                 const float* dwc, int64_t lddwc, const float* vec, int64_t ldvec, float* dvec, int64_t lddvec,
                 float* dctx, float* dl_out, int B, int S, int D, vln_stream_t s);

/* nn.LSTMCell pointwise stage (policy.py:30,96,192): gates are nsplit pre-activation slabs [B,4H] */
int vln_lstm_pointwise_fwd(const float* gates, int nsplit, int64_t slab_stride, const float* b_ih, const float* b_hh,
                           const float* c0, float* h1, float* c1, float* act, float* tanh_c1, float* h1_drop,
                           uint64_t seed, uint64_t offset, float p, int B, int H, vln_stream_t s);
int vln_lstm_pointwise_bwd(const float* dh1, const float* dh1_drop, const float* dc1, uint64_t seed, uint64_t offset,
                           float p, const float* act, const float* tanh_c1, const float* c0, float* dgates, float* dc0,
                           int B, int H, vln_stream_t s);

/* nn.Dropout replacements: Philox4x32-10 keyed by (seed, offset, element index) */
int vln_dropout_mask(float* out_scaled_mask, int64_t n, uint64_t seed, uint64_t offset, float p, vln_stream_t s);
int vln_scale_dropout(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int cols, uint64_t seed,
                      uint64_t offset, float p, const uint64_t* offset_base_dev /*nullable, see vln_embed_fwd*/, vln_stream_t s);
/* EnvDropDecoder feature dropout, in place on x[..., :img] (policy.py:226-231); optional bf16 copy of x */
int vln_feat_dropout_inplace(void* x, int xtype, int64_t rows, int img, int angle, uint64_t seed, uint64_t offset,
                             float p, void* copy_bf16, const uint64_t* offset_base_dev /*nullable, see vln_embed_fwd*/, vln_stream_t s);

/* ---- optimizer step over flat buffers (engine/trainer.py:380-381,423-427): per-group clip_grad_norm (max_norms: HOST array
 * of ngroups norms, torch semantics, 0 = that group is not clipped -- the reference clips encoder and decoder at 40 and leaves
 * the critic alone, trainer.py:425-426; NULL = no clipping) + torch.optim.RMSprop update (alpha, eps, no momentum, not centered).  group_begin: ngroups+1
 * element offsets (multiples of 4); partial: vln_rmsprop_partial_floats() floats of scratch; norms_out[ngroups] nullable;
 * grad_scale multiplies every gradient first (e.g. 1/world); grad_scale < 0: scale by |grad_scale| and CLEAR the gradient
 * buffer in the same pass (the update is its last reader: the next iteration's zero_grad costs nothing). */
int64_t vln_rmsprop_partial_floats(const int64_t* group_begin, int ngroups);
int vln_rmsprop_clip_step(float* params, float* grads, float* square_avg, const int64_t* group_begin, int ngroups,
                          float* partial, float* norms_out, float lr, float alpha, float eps, const float* max_norms,
                          float grad_scale, vln_stream_t s);
/* Same contract for the other two entries of the reference's optim_switcher (trainer.py:17-21), torch defaults:
 * Adam (betas, eps, bias correction with step = 1, 2, ...; no weight decay / amsgrad) and plain SGD. */
int vln_adam_clip_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* group_begin,
                       int ngroups, float* partial, float* norms_out, float lr, float beta1, float beta2, float eps,
                       int64_t step, const int64_t* step_dev /* nullable: the step count lives in a DEVICE word (bumped by vln_tick
                       between iterations) and `step` is added to it -- bias corrections then form on the device and the launch can be
                       replayed from a graph */, const float* max_norms, float grad_scale, vln_stream_t s);
int vln_sgd_clip_step(float* params, float* grads, const int64_t* group_begin, int ngroups, float* partial,
                      float* norms_out, float lr, const float* max_norms, float grad_scale, vln_stream_t s);

/* ---- loss / action-selection stage of the rollouts (follower.py:123-139, envdrop.py:173-195, monitor.py:146-176):
 * logits.masked_fill_(cand_mask, -inf) [in place when write_mask], CrossEntropyLoss(ignore_index, reduction="none"),
 * softmax probabilities and, for a given action, Categorical(probs).log_prob / .entropy() -- one wave per row.
 * Backward of the CE: dlogits = dloss[b] * (probs - onehot(target)), 0 for ignored rows. */
int vln_masked_ce_fwd(float* logits, int64_t ld, const int64_t* target /*nullable*/, const uint8_t* cand_mask /*nullable*/,
                      float* loss /*[B] nullable*/, float* loss_sum /*[1] nullable: sum over rows, same launch (reduction="sum")*/,
                      float* probs /*[B,C] nullable*/, const int64_t* action /*nullable*/,
                      float* logp /*[B]*/, float* entropy /*[B] nullable*/, int B, int C, int64_t ignore_index, int write_mask,
                      vln_stream_t s);
/* dloss_stride: 1 = one upstream gradient per row, 0 = one scalar for all rows (backward of the fused sum) */
int vln_masked_ce_bwd(const float* probs, const int64_t* target, const float* dloss, int64_t dloss_stride, float* dlogits,
                      int B, int C, int64_t ignore_index, vln_stream_t s);
/* ABI v16: reduction = "mean" (nn.CrossEntropyLoss(ignore_index)'s default, follower.py:62) in the same launch: mean_out[0] = the
 * mean over the rows with a target, mean_out[1] = 1 / their count (0 rows: nan, as torch); the backward multiplies it in. */
#define VLN_CE_MEAN_ONE_LAUNCH_MAX 16384     /* B * C up to which the mean is formed by ONE workgroup in one launch (the decoders' B x <= 16 logits) */
/* rows_scratch (nullable, [B] floats): larger problems (the speaker's [B * Lw, vocab] word logits) take a wave per row over the whole
 * chip + a finishing launch when it is given; without it they run on the one workgroup all the same (slow, correct). */
int vln_masked_ce_mean_fwd(float* logits, int64_t ld, const int64_t* target, const uint8_t* cand_mask /*nullable*/, float* mean_out /*[2]*/,
                           float* probs /*[B,C]*/, int B, int C, int64_t ignore_index, float* rows_scratch, vln_stream_t s);
int vln_masked_ce_mean_bwd(const float* probs, const int64_t* target, const float* dloss /*[1]*/, const float* inv_count /*mean_out + 1*/,
                           float* dlogits, int B, int C, int64_t ignore_index, vln_stream_t s);

/* The IL loss of a WHOLE rollout, ml_loss = sum_t CrossEntropy(mask(logits_t), target_t) (envdrop.py:173-179 summed over
 * the steps), in one launch after the last decoder step, and all the d logits_t in one launch at the start of backward:
 * the loss is on nobody's dependent chain, the 2T per-step launches only lengthen the stream.  Steps may differ in C.
 * probs_t [B,C_t] is written forward and read backward; dlogits_t [B,C_t] dense.  T <= VLN_CE_MAX_STEPS per call
 * (`accumulate` adds to *loss_sum for longer rollouts).  `scale` multiplies the sum (and the gradients): the agents'
 * `ml_loss * ML_WEIGHT / batch_size` (envdrop.py:268) without two more elementwise launches each way. */
#define VLN_CE_MAX_STEPS 40
typedef struct vln_ce_step {
  float* logits; int64_t ld; const int64_t* target; const uint8_t* cand_mask /*nullable*/; float* probs; float* dlogits /*bwd only*/;
  int C;
} vln_ce_step;
int vln_masked_ce_multi_fwd(const vln_ce_step* steps, int T, int B, int64_t ignore_index, float scale,
                            float* loss_sum /*[1]: total over steps and episodes*/,
                            float* loss_rows /*[B]: per episode (reduction="none" summed over the steps, what SELF-PACE weighs,
                                               curriculum.py:296); exactly one of the two*/,
                            int accumulate,
                            float* inv_counts /*nullable (ABI v19) [T]: the MEAN PER STEP (nn.CrossEntropyLoss(ignore_index)'s default reduction on
                                                every step's batch, summed over the steps: follower.py:62,123-139) -- loss_sum = scale *
                                                sum_t (sum_b CE_tb / n_t); inv_counts[t] = 1 / n_t is written for the backward; loss_sum only,
                                                T * B <= 8192*/,
                            vln_stream_t s);
/* dloss_stride 0: one upstream scalar; 1: one per episode; inv_counts (nullable): what the forward of the mean per step left */
int vln_masked_ce_multi_bwd(const vln_ce_step* steps, int T, int B, int64_t ignore_index, float scale, const float* dloss,
                            int64_t dloss_stride, const float* inv_counts, vln_stream_t s);

/* The Self-Monitor agent's step loss (monitor.py:146-165) in one launch each way, no host round trip: action CE on the masked
 * logits, the progress target from the distances (monitor.py:155-157: (start_dist - cur_dist) / start_dist, 1 where
 * cur_dist <= 3, the prediction itself where `ended`), MSE(progress, target), and the mix: t == 0 -> CE, t > 0 ->
 * lam * MSE + (1 - lam) * CE.  per_sample 0: nn.CrossEntropyLoss(ignore_index) / nn.MSELoss() means, out [1];
 * per_sample 1: reduction="none" terms (the curriculum criteria, monitor.py:151,162), out [B].
 * stats [2] = {mean progress MSE (what the agent logs as progress_loss), number of rows with a target}. */
int vln_monitor_loss_fwd(float* logits, int64_t ld, const int64_t* target, const uint8_t* cand_mask /*nullable*/,
                         const float* progress, int64_t ldp, const float* start_dist, const float* cur_dist, const uint8_t* ended,
                         int t, float lam, int per_sample, float* probs /*[B,C] out*/, float* prog_target /*[B] out*/,
                         float* out, float* stats, int B, int C, int64_t ignore_index, vln_stream_t s);
int vln_monitor_loss_bwd(const float* probs, const int64_t* target, const float* progress, int64_t ldp, const float* prog_target,
                         const float* stats, const float* dloss, int64_t dloss_stride /*0: one scalar, 1: per episode*/, int t,
                         float lam, int per_sample, float* dlogits /*[B,C]*/, float* dprogress /*[B]*/, int B, int C,
                         int64_t ignore_index, vln_stream_t s);
/* ABI v19: the same loss for a WHOLE ROLLOUT, sum over the steps t0 .. t0 + T - 1 of cur_loss_t (the non-curriculum means), in one
 * launch each way: out [1] (+= when accumulate), stats [T][2] = per step {mean progress MSE, rows with a target} (read again by the
 * backward); T <= VLN_MONITOR_LOSS_MAX_STEPS per call (t0 = the first step's index: only step 0 is the plain CE), T * B <= 4096. */
#define VLN_MONITOR_LOSS_MAX_STEPS 16
typedef struct vln_monitor_loss_step {
  float* logits; int64_t ld; const int64_t* target; const uint8_t* cand_mask /*nullable*/; float* probs /*[B,C] out*/;
  const float* progress; int64_t ldp; const float* start_dist; const float* cur_dist; const uint8_t* ended;
  float* prog_target /*[B] out*/; float* dlogits /*[B,C] bwd*/; float* dprogress /*[B] bwd*/;
  int C;
} vln_monitor_loss_step;
int vln_monitor_loss_multi_fwd(const vln_monitor_loss_step* steps, int T, int B, int t0, float lam, int64_t ignore_index, float* out,
                               float* stats, int accumulate, vln_stream_t s);
int vln_monitor_loss_multi_bwd(const vln_monitor_loss_step* steps, int T, int B, int t0, float lam, int64_t ignore_index,
                               const float* stats, const float* dloss /*[1]*/, vln_stream_t s);

/* The sampled-action branch of a rollout step (envdrop.py:186-195) as one launch: probs = softmax(logits masked with
 * -inf where cand_mask), action ~ Categorical(probs) unless action_in is given (then action_out may be NULL), logp =
 * log pi(action) and the entropy with torch.distributions' clamp_probs.  The draw uses the Philox word (seed, offset, b).
 * Backward: dlogits from the upstream gradients on logp and entropy (either may be NULL). */
int vln_categorical_fwd(const float* logits, int64_t ld, const uint8_t* cand_mask /*nullable*/, const int64_t* action_in /*nullable*/,
                        int64_t* action_out /*nullable*/, float* probs, float* logp, float* entropy, int B, int C, uint64_t seed,
                        uint64_t offset, const uint64_t* offset_base_dev /*nullable, see vln_embed_fwd (ABI v14)*/, vln_stream_t s);
int vln_categorical_bwd(const float* probs, const int64_t* action, const float* dlogp /*nullable*/, const float* dent /*nullable*/,
                        float* dlogits, int B, int C, vln_stream_t s);
/* The backward of EVERY step of a sampled rollout in one launch (losses.RolloutSampler): dlogp / dent [T,B], row t = step t. */
typedef struct vln_cat_step { const float* probs; const int64_t* action; float* dlogits; int C; } vln_cat_step;
int vln_categorical_multi_bwd(const vln_cat_step* steps, int T /* <= VLN_CE_MAX_STEPS */, int B, const float* dlogp /*nullable*/,
                              const float* dent /*nullable*/, vln_stream_t s);

/* Row-wise elementwise forms of the Speaker-Follower step: ActionScoring's `context * target` folded into the query
 * (units.py:180-184: logit = linear_out(context * target) = context . (target (.) w_out) + b_out) and the tanh backward.
 *   op 0 VLN_EW_MUL        y[r,c] = a[r,c] * b[r*ldb + c]      (ldb = 0: b is one row vector)
 *   op 1 VLN_EW_ADD_SCALAR y[r,c] = a[r,c] + b[0]
 *   op 2 VLN_EW_TANH_GRAD  y[r,c] = a[r,c] * (1 - b[r,c]^2)
 *   op 3 VLN_EW_MUL_ROWSUM y[r,c] = a[r,c] * sum_{j<nb} b[r*ldb + j]   (a NULL: the row sums themselves) */
int vln_ew(int op, const float* a, int64_t lda, const float* b, int64_t ldb, int nb, float* y, int64_t ldy, int rows, int cols,
           vln_stream_t s);

/* Pieces of the Self-Monitor decoder step (policy.py:119-166) besides GEMMs, attention rows and the BN-MLP:
 *   vln_pe_dropout        out = dropout(ctx + pe[:L])  [B,L,H]                               (units.py:188-207)
 *   vln_monitor_head_fwd  mem = dropout(sigmoid(mg) * tanh(c1)); prog = tanh(wc . [word_w ; mem] + bc)   (policy.py:126-130)
 *   vln_monitor_head_bwd  from dprog: dmg, dc1 (= dc1_ext + its own), dww (= dww_ext + its own), Z[b,:] = dpre[b] * [word_w ; mem]
 *                         (colsum(Z) = d wc, colsum(dpre) = d bc)
 *   vln_add_n             out (+)= s0 + s1 + s2 + s3 (NULL sources skipped): the gradient contributions of one tensor */
int vln_pe_dropout(const float* ctx, const float* pe, float* out, int B, int L, int H, uint64_t seed, uint64_t offset, float p,
                   vln_stream_t s);
int vln_monitor_head_fwd(const float* mg, const float* c1, const float* word_w, const float* wc, const float* bc, float* mem,
                         float* prog, int B, int L, int H, uint64_t seed, uint64_t offset, float p, vln_stream_t s);
int vln_monitor_head_bwd(const float* mg, const float* c1, const float* word_w, const float* wc, const float* mem, const float* prog,
                         const float* dprog, const float* dc1_ext, const float* dww_ext, float* dmg, float* dc1, float* dww, float* Z,
                         float* dpre, int B, int L, int H, uint64_t seed, uint64_t offset, float p, vln_stream_t s);
int vln_add_n(float* out, int64_t ldo, int rows, int cols, const float* s0, int64_t ld0, const float* s1, int64_t ld1,
              const float* s2, int64_t ld2, const float* s3, int64_t ld3, int accumulate, vln_stream_t s);

/* MonitorDecoder.forward after its BN-MLP (policy.py:132-166) and the backward of that, ONE call each (csrc/monitor.hip).
 * All tensors fp32 row-major and dense unless a leading dimension is named; weights are the streamed copies in `wtype`
 * ([N,K] and the transposed [K,N]); every buffer is caller-owned; the step's saved activations (pctx .. tanh_c1) must stay
 * alive until vln_monitor_step_bwd has been issued.  M = width of the projected candidates (mlp_dims[-1]). */
typedef struct vln_monitor_dims { int B, L, C, H, M, wtype; } vln_monitor_dims;
typedef struct vln_monitor_weights {
  const void *w_tin, *w_tin_t;                       /* text_attn.linear_in.weight        [H, H]            */
  const void *w_vh, *w_vh_t; const float* b_vh;      /* visual_attn.linear_in_h           [M, H], [M]       */
  const void *w_cat, *w_cat_t; const float *b_ih, *b_hh; /* [lstm.weight_ih | weight_hh]  [4H, 2M+H+H]      */
  const void *w_a, *w_a_t; const float* b_a;         /* action_linear                     [M, 2H], [M]      */
  const void *w_m, *w_m_t; const float* b_m;         /* monitor_linear                    [H, H+M], [H]     */
  const float *w_c, *b_c;                            /* critic.0 (progress head), fp32    [L+H], [1]        */
  const float* pe;                                   /* position.pe                       [L, H]            */
  int32_t f32_mask, pad_;                            /* ABI v13: bit i set = matrix i (w_tin, w_vh, w_cat, w_a, w_m) and its transpose are
                                                      * fp32 arrays whatever vln_monitor_dims.wtype says (per-matrix override of the bf16 mode) */
} vln_monitor_weights;
typedef struct vln_monitor_step {
  const float *prev_rep /*[B,M]*/, *cand_rep /*[B,C,M]*/, *h0, *c0 /*[B,H]*/, *ctx /*[B,L,H]*/;
  const uint8_t *ctx_mask /*[B,L]*/, *cand_mask /*[B,C]*/;                                  /* 1 = masked */
  float *logit /*[B,C]*/, *prog /*[B]*/, *h1, *c1 /*[B,H]*/, *word_w /*[B,L]*/, *move_w /*[B,C]*/;      /* outputs */
  float *pctx /*[B,L,H]*/, *tq /*[B,H]*/, *vq /*[B,M]*/, *xcat /*[B,2M+2H]*/, *tcat /*[B,2H]*/, *aq /*[B,M]*/, *hm /*[B,H+M]*/,
        *mg /*[B,H]*/, *mem /*[B,H]*/, *act /*[B,4H]*/, *tanh_c1 /*[B,H]*/;                 /* saved for the backward */
  float *gates /*[B,4H]: unused since ABI v14 -- the gate product stays as split-K slabs in ws (>= B * 4H floats)*/, *dots /*[B,max(L,C)]*/;   /* scratch of the call */
  float* ws; int64_t ws_floats;                                                             /* split-K / grouped-launch scratch */
  uint64_t seed_pe, off_pe; float p_pe;              /* dropout on the positioned context (units.py:207) */
  uint64_t seed, off_h1, off_mem; float p_drop;      /* dropout on h_1 (policy.py:160) and on the monitor memory (:128) */
  const uint64_t* offset_base_dev;                   /* nullable: every site's offset is (*offset_base_dev) * 8 + its field (see vln_embed_fwd) */
} vln_monitor_step;
/* One decoder step's share of the rollout's context gradient (ABI v16), reported instead of launched:
 *   dctx[b,s,:] += (alpha[b,s] g[b,:] + dl[b,s] q[b,:]) * dropout mask(seed, offset, p over the flat [B,S,D] index; p = 0: none)
 * The caller keeps the four arrays alive and forms every step's term with ONE vln_attn_dctx_deferred(_drop) launch per rollout. */
typedef struct vln_dctx_term { const float *alpha, *dl, *g, *q; int64_t ldg, ldq; uint64_t seed, offset; float p, pad_; } vln_dctx_term;
typedef struct vln_monitor_grads {
  const float *dlogit, *dprog, *dh1, *dc1, *dww_ext, *dmw_ext;        /* upstream gradients, each nullable */
  float *dprev_rep /*[B,M]*/, *dcand_rep /*[B,C,M], nullable*/, *dh0, *dc0 /*[B,H]*/, *dctx /*[B,L,H], nullable*/;
  int dctx_accumulate;                               /* 1: add this step's term to dctx (one buffer for the rollout) */
  /* parameter gradients in the order W_tin, W_vh, b_vh, W_ih, W_hh, b_ih, b_hh, W_a, b_a, W_m, b_m, w_c, b_c; each
   * nullable; acc[i] = 1 adds to the buffer's contents (e.g. the optimizer's flat gradient views) */
  float *g_tin, *g_vh, *g_bvh, *g_ih, *g_hh, *g_bih, *g_bhh, *g_a, *g_ba, *g_m, *g_bm, *g_wc, *g_bc;
  int acc[13];
  int precision;                                     /* weight gradients: 0 exact fp32 MFMA, 1 split-bf16 (three bf16 MFMAs) */
  float* scratch; int64_t scratch_floats;            /* >= vln_monitor_bwd_scratch_floats(dims) */
  vln_param_jobs* defer;                             /* nullable (ABI v11): see vln_param_jobs; b_c keeps its own one-column launch */
  vln_dctx_term* dctx_term;                          /* nullable (ABI v16, out): the step's context-gradient term is REPORTED here and not
                                                      * launched (dctx is then ignored); its arrays live in `scratch` and the step's saved block */
} vln_monitor_grads;
int64_t vln_monitor_bwd_scratch_floats(const vln_monitor_dims* d);
/* ABI v16: floats of vln_monitor_step.ws with which every skinny product of the step reaches its consumer as split-K slabs (the
 * consumers sum them: no reduce launches; the backward keeps five products' slabs live at once).  A smaller workspace computes the
 * same numbers with a reduce launch for the products that do not fit. */
int64_t vln_monitor_ws_floats(const vln_monitor_dims* d);
int vln_monitor_step_fwd(const vln_monitor_dims* d, const vln_monitor_weights* w, vln_monitor_step* io, vln_stream_t s);
int vln_monitor_step_bwd(const vln_monitor_dims* d, const vln_monitor_weights* w, const vln_monitor_step* io,
                         const vln_monitor_grads* g, vln_stream_t s);

/* AttnDecoderLSTM.forward + ActionScoring (policy.py:37-60, units.py:163-185) and the backward of that, ONE call each
 * (csrc/follower.hip).  Same conventions as the Self-Monitor step above.  F = view feature size, A = candidate / previous-action
 * feature size, D = dot size of the panorama attention and of ActionScoring (256). */
typedef struct vln_follower_dims { int B, L, V, C, H, F, A, D, wtype; } vln_follower_dims;
typedef struct vln_follower_weights {
  const void *w_h, *w_h_t; const float* b_h;         /* visual_attn.linear_in_h           [D, H], [D]       */
  const void* w_v; const float* b_v;                 /* visual_attn.linear_in_v           [D, F], [D]       */
  const void *w_cat, *w_cat_t; const float *b_ih, *b_hh; /* [lstm.weight_ih | weight_hh]  [4H, A+F+H]       */
  const void *w_tin, *w_tin_t;                       /* text_attn.linear_in.weight        [H, H]            */
  const void *w_tout, *w_tout_t;                     /* text_attn.linear_out.weight       [H, 2H]           */
  const void* w_act; const float* b_act;             /* decode_action.linear_act          [D, A], [D]       */
  const void *w_hid, *w_hid_t; const float* b_hid;   /* decode_action.linear_hid          [D, H], [D]       */
  const float *w_out, *b_out;                        /* decode_action.linear_out, fp32    [D], [1]          */
  const void* w_v_t;                                 /* ABI v16: linear_in_v.weight transposed [F, D]: the view logits are img . (W_v^T tq) --
                                                      * the [B*V, D] keys W_v img + b_v are never formed (b_v . tq is one constant per episode
                                                      * under the softmax: d b_v is exactly 0) */
} vln_follower_weights;
typedef struct vln_follower_step {
  const float *img /*[B,V,F]*/, *a_prev /*[B,A]*/, *cands /*[B,C,A]*/, *h0, *c0 /*[B,H]*/, *ctx /*[B,L,H]*/;
  const uint8_t* ctx_mask;                           /* [B,L], 1 = masked; nullable */
  float *logit /*[B,C]*/, *h1, *c1 /*[B,H]*/, *word_w /*[B,L]*/, *view_w /*[B,V]*/;                         /* outputs */
  float *tq /*[B,D]*/, *keys /*[B,F] since ABI v16: the projected query W_v^T tq (was the [B*V,D] keys)*/, *vlog /*[B,V]: unused since ABI v16*/, *xcat /*[B,A+F+H]*/, *act /*[B,4H]*/, *tanh_c1 /*[B,H]*/, *tq2 /*[B,H]*/,
        *tcat /*[B,2H]*/, *grounded /*[B,H]*/, *target /*[B,D]*/, *q /*[B,D]*/, *context /*[B*C,D]*/;      /* saved for the backward */
  float *gates /*[B,4H]: unused since ABI v14 -- the gate product stays as split-K slabs in ws (>= B * 4H floats)*/, *dots /*[B,max(L,V,C)]*/;   /* scratch of the call */
  float* ws; int64_t ws_floats;
  uint64_t seed, off; float p_drop;                  /* dropout sites `off` (LSTM input row, policy.py:49) and `off + 1` (h_1, :54) */
  const uint64_t* offset_base_dev;                   /* nullable: offsets relative to a device word (see vln_embed_fwd) */
  void* attn_sync; int64_t attn_sync_bytes;          /* nullable (ABI v19): vln_attn_sync_bytes(B) of zero-initialised scratch: the two attentions
                                                      * of the step run on FOUR workgroups per episode (attention_split.h), forward and backward */
  int context_ready;                                 /* 1 (ABI v19): `context` = W_act cands + b_act was formed by the caller -- for all steps of a
                                                      * teacher-forced rollout at once (it depends on the batch only) -- and is only read here */
} vln_follower_step;
typedef struct vln_follower_grads {
  const float *dlogit, *dh1, *dc1, *dww_ext, *dvw_ext;                /* upstream gradients, each nullable */
  float *da_prev /*[B,A], nullable*/, *dh0, *dc0 /*[B,H]*/, *dctx /*[B,L,H], nullable*/;
  int dctx_accumulate;
  /* parameter gradients in the order W_h, b_h, W_v, b_v, W_ih, W_hh, b_ih, b_hh, W_tin, W_tout, W_act, b_act, W_hid, b_hid,
   * w_out, b_out; each nullable; acc[i] = 1 adds to the buffer's contents */
  float *g_wh, *g_bh, *g_wv, *g_bv, *g_ih, *g_hh, *g_bih, *g_bhh, *g_tin, *g_tout, *g_wact, *g_bact, *g_whid, *g_bhid, *g_wout, *g_bout;
  int acc[16];
  int precision;
  float* scratch; int64_t scratch_floats;            /* >= vln_follower_bwd_scratch_floats(dims) */
  vln_param_jobs* defer;                             /* nullable (ABI v11): see vln_param_jobs; b_out keeps its own one-column launch */
  vln_dctx_term* dctx_term;                          /* nullable (ABI v16, out): as in vln_monitor_grads */
} vln_follower_grads;
int64_t vln_follower_bwd_scratch_floats(const vln_follower_dims* d);
int vln_follower_step_fwd(const vln_follower_dims* d, const vln_follower_weights* w, vln_follower_step* io, vln_stream_t s);
int vln_follower_step_bwd(const vln_follower_dims* d, const vln_follower_weights* w, const vln_follower_step* io,
                          const vln_follower_grads* g, vln_stream_t s);

/* BatchNorm1d (+ fused ReLU): the BN-MLP of the Self-Monitor agent (units.py:210-242; `bn_mlp` =
 * vln_bn_fwd / vln_linear_fwd / vln_bn_fwd(relu)).  Training: batch statistics, running statistics updated in place with
 * `momentum` and the unbiased variance, *num_batches_tracked += 1, save_mean / save_rstd [D] kept for backward.  Eval:
 * running statistics.  Backward: training form with the saved statistics, eval form with (running_mean, running_var).
 * Optional epilogue in the BN-MLP's layer order (BatchNorm, Dropout, ReLU): a Philox dropout (seed, offset, p_drop) between
 * the normalisation and the ReLU, and row_zero[r] != 0 forcing output row r to 0 (padded candidate slots). */
int vln_bn_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, const float* gamma, const float* beta, float* running_mean,
               float* running_var, int64_t* num_batches_tracked /*nullable*/, float* save_mean, float* save_rstd, int R, int D,
               float eps, float momentum, int training, int relu, uint64_t seed, uint64_t offset, float p_drop /*0: none*/,
               const uint8_t* row_zero /*nullable [R]*/,
               float* ws /*nullable: ceil(R/128) * 2 * D floats; with it, inputs of R >= 512 rows take the row-chunked two-launch form*/,
               int64_t ws_floats, vln_stream_t s);
int vln_bn_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* y /*relu only*/, int64_t ldy,
               const float* gamma, const float* mean, const float* rstd_or_var, float* dx /*nullable*/, int64_t lddx,
               float* dgamma /*nullable*/, float* dbeta /*nullable*/, int R, int D, float eps, int training, int relu,
               int accumulate, uint64_t seed, uint64_t offset, float p_drop, const uint8_t* row_zero,
               float* ws /*nullable, as in vln_bn_fwd*/, int64_t ws_floats, vln_stream_t s);

/* MLPwithBN (units.py:210-242: BatchNorm1d, then per hidden layer Linear, BatchNorm1d, Dropout, ReLU -- the Self-Monitor agent's
 * proj_navigable_mlp, applied to the previous action [B, F] and to the candidates [B*C, F] every step, policy.py:146-149) as ONE
 * C call each way (the 5 forward / 7 backward launcher calls of a two-layer MLP used to be one ctypes call each, ~24 per decoder
 * step: the Self-Monitor iteration was bound by that Python).  Same launches, same numbers as the vln_bn_* / vln_linear_*
 * sequence.  `saved` (vln_bn_mlp_saved_floats) receives every intermediate the backward reads; the output y [R, out of the last
 * layer] is the LAST block of `saved` (vln_bn_mlp_out_offset).  row_zero [R] (nullable) zeroes output rows (padded
 * candidate slots, policy.py:148-149).  Backward: dx nullable (features carry no gradient); parameter gradients are written to
 * (acc = 0) or accumulated into (acc = 1) the given buffers; all weight gradients in one grouped launch, all bias gradients in
 * another. */
#define VLN_BN_MLP_MAX_LAYERS 4
typedef struct vln_bn_affine { const float* gamma; const float* beta; float* run_mean; float* run_var; int64_t* nbt; } vln_bn_affine;
typedef struct vln_bn_mlp_layer {
  const void* w; const void* w_t; const float* w_f32; const float* b;   /* Linear [out, in]: streamed shadow, its transpose, fp32 master (unused), bias (nullable) */
  vln_bn_affine bn; int32_t out, pad_; float p_drop, padf_; uint64_t seed, offset;
  uint64_t offset2;                                  /* ABI v12: the dropout offset of the SECOND segment's rows (vln_bn_mlp.R1 > 0) */
} vln_bn_mlp_layer;
typedef struct vln_bn_mlp {
  int32_t R, D0, nl, wtype, training;
  int32_t R1;     /* ABI v12.  > 0: rows [0, R1) and [R1, R) are TWO independent batches (the Self-Monitor projects the previous action, B rows,
                   * and the candidates, B*C rows, with the same MLP: policy.py:140-149) -- one call: every BatchNorm normalises each
                   * segment with its own statistics (running statistics updated twice, in order), the Linear layers run over all R rows;
                   * row_zero then has R - R1 entries (segment 1's rows), each layer's `offset2` is segment 1's dropout offset, and the
                   * saved statistics hold segment 1's after segment 0's.  Same results as two calls up to summation order. */
  float eps, momentum;
  vln_bn_affine bn0;
  vln_bn_mlp_layer layer[VLN_BN_MLP_MAX_LAYERS];
  const uint8_t* row_zero;
  const uint64_t* offset_base_dev;                   /* nullable: the layers' dropout offsets relative to a device word (see vln_embed_fwd) */
  const float* x2; int64_t ldx2;                     /* ABI v16, nullable (R1 > 0 only): the SECOND batch's rows live in their own array -- row r of
                                                      * the batch at x2 + r * ldx2 -- and `x` holds the first batch's R1 rows: the previous action and
                                                      * the candidates (policy.py:146-149) are read where they are, no concatenated copy.  No dx then. */
} vln_bn_mlp;
typedef struct vln_bn_mlp_grad_layer { float* g_w; float* g_b; float* g_gamma; float* g_beta; int32_t acc_w, acc_b, acc_bn, pad_; } vln_bn_mlp_grad_layer;
typedef struct vln_bn_mlp_grads {
  float* g_gamma0; float* g_beta0; int32_t acc0, pad0_;
  vln_bn_mlp_grad_layer layer[VLN_BN_MLP_MAX_LAYERS];
  int32_t precision;                /* weight gradients: 0 fp32, 1 split bf16, 2 plain bf16 (vln_wgrad_grouped) */
  int32_t bn0_from_wgrad;           /* ABI v16.  1 (training, dx == NULL, `defer` set): the input BatchNorm's d gamma / d beta are NOT formed
                                     * here -- the call skips the product dz W of the first layer and the BatchNorm backward over the
                                     * [R, D0] input that would only feed them; the caller derives them from the first layer's own weight
                                     * gradient with vln_bn0_grads_from_wgrad (below) */
  float* scratch; int64_t scratch_floats;     /* vln_bn_mlp_bwd_scratch_floats */
  vln_param_jobs* defer;                      /* nullable (ABI v11): the Linear weight / bias gradients as jobs, see vln_param_jobs */
} vln_bn_mlp_grads;
/* The INPUT BatchNorm's parameter gradients of a BN-MLP whose input carries no gradient (the Self-Monitor projects raw features,
 * policy.py:146-149), from the first Linear layer's weight gradient instead of a pass over the rows (ABI v16).  With y0 = gamma * xhat + beta
 * the layer's input, dW[n,k] = sum_r dz[r,n] y0[r,k] = gamma_k A[n,k] + beta_k db[n] where A = dz^T xhat and db = sum_r dz[r,:], so
 *   d gamma_k = sum_r (dz W)[r,k] xhat[r,k] = sum_n A[n,k] W[n,k] = sum_n (dW[n,k] - beta_k db[n]) / gamma_k * W[n,k]
 *   d beta_k  = sum_r (dz W)[r,k]           = sum_n db[n] W[n,k]
 * -- the [R, K] product dz W and the BatchNorm backward over the [R, K] input are never formed (at BASELINE config 2: 5.1 GFLOP and two
 * passes over 10 MB per decoder step).  dW [N,K] / db [N] hold THIS rollout's sums (not yet added to the accumulated gradients); the call
 * also adds them to gW / gb (acc_w / acc_b = 0: stores them).  g_gamma / g_beta: acc_bn = 1 adds.  ws: >= 64 * K floats.  A gamma_k of
 * exactly 0 has no quotient: the launch raises the sticky status word 4 (vln_persistent_check) and leaves d gamma_k = 0.  A gamma_k with
 * |gamma_k| * VLN_BN0_MAX_AMPLIFICATION < |beta_k| (ABI v18) is ill-conditioned -- (dW - beta db) cancels and the quotient amplifies dW's
 * rounding by |beta| / |gamma| --: d gamma_k is still written, and the same sticky word reports it, so a degraded gradient is never
 * handed on silently (the caller takes the direct path: vln_bn_mlp_bwd without skip). */
#define VLN_BN0_MAX_AMPLIFICATION 64   /* split-bf16 dW (2^-16) x 64 = 1e-3, a tenth of the bf16 bound; fp32 dW x 64 = 4e-6 */
int vln_bn0_grads_from_wgrad(const float* dW, const float* db, const float* W /*fp32 master [N,K]*/, int64_t ldw, const float* gamma,
                             const float* beta, float* gW, float* gb, float* g_gamma, float* g_beta, int N, int K, int acc_w, int acc_b,
                             int acc_bn, float* ws, int64_t ws_floats, vln_stream_t s);
int64_t vln_bn_mlp_saved_floats(const vln_bn_mlp* m);
int64_t vln_bn_mlp_out_offset(const vln_bn_mlp* m);
int64_t vln_bn_mlp_ws_floats(const vln_bn_mlp* m);            /* forward and backward workspace (split-K slabs, chunked-BN partials) */
int64_t vln_bn_mlp_bwd_scratch_floats(const vln_bn_mlp* m);
int vln_bn_mlp_fwd(const vln_bn_mlp* m, const float* x, int64_t ldx, float* saved, float* ws, int64_t ws_floats, vln_stream_t s);
int vln_bn_mlp_bwd(const vln_bn_mlp* m, const float* x, int64_t ldx, const float* saved, const float* dy, int64_t lddy,
                   float* dx /*nullable*/, int64_t lddx, const vln_bn_mlp_grads* g, float* ws, int64_t ws_floats, vln_stream_t s);

/* A2C sweep of the EnvDrop rollout (envdrop.py:235-264) as one launch.  All step tensors are stacked [T,B]:
 * logp = log pi(a_t), ent = entropies (NULL with ent_coef unused: feedback != "sample"), val = critic values (with
 * grad), reward, mask (1 = episode still running at t), last_value [B] (detached), ended [B].
 *   R_t = gamma R_{t+1} + r_t,  R_T = ended ? 0 : last_value;   loss_b[b] = sum_t m (-logp (R-V) + (R-V)^2 / 2 - c ent)
 * with (R-V) a constant in the first term.  dlogp / dval / dent [T,B] receive the partial derivatives, total (nullable)
 * the sum of the masks (the reference's RL_NORMALIZE == 'total' divisor).  Backward: g* = dloss_b[b] * d*. */
int vln_a2c_loss_fwd(const float* logp, const float* ent, const float* val, const float* reward, const uint8_t* mask,
                     const float* last_value, const uint8_t* ended, int T, int B, float gamma, float ent_coef, float* loss_b,
                     float* dlogp, float* dval, float* dent, float* total, vln_stream_t s);
int vln_a2c_loss_bwd(const float* dloss_b, int64_t dloss_stride, const float* dlogp, const float* dval, const float* dent, int T,
                     int B, float* glogp, float* gval, float* gent, vln_stream_t s);

/* ---- per-step feature marshalling on the device (agent/base.py:141-157, common_env.py:307-308) -----------------
 * The ResNet feature table [N_viewpoints, V, IMG] (fp32 or bf16) lives in HBM; a step ships indices only.
 * vln_gather_pano : out[b,v,:] = [ table[rows[b], v, :] | angle_table[view_index[b], v, :] ]        (BasicR2RAgent._feature_variable)
 * vln_gather_cands: out[r,:]   = [ table[rows[r], views[r], :] | make_angle_feat(heading[r], elevation[r]) ], rows[r] < 0 ->
 *                   all-zero STOP/padding slot                                                       (BasicR2RAgent._candidate_variable)
 * Both apply the EnvDrop feature dropout on the image part when p_feat > 0 (policy.py:226-231; call the decoder with
 * already_dropfeat=True then) and optionally emit the bf16 copy of the row; `out` (fp32) may be NULL when only the bf16
 * copy is wanted (a bf16 decoder never reads the fp32 rows: that halves the pass's HBM writes). */
/* Every step of a TEACHER-FORCED rollout in one launch (the path, hence every step's viewpoint / candidate rows, is known when
 * the rollout starts: base.py:141-157 driven by the ground-truth actions): per step the operands of vln_gather_step. */
typedef struct vln_gather_rollout_step {
  const int64_t* rows; const int32_t* view_index; const int64_t* crows; const int32_t* cviews; const float* heading; const float* elevation;
  float* out; void* out_bf16; float* cout; void* cout_bf16;
  uint64_t offset_pano, offset_cand;
} vln_gather_rollout_step;
/* The same rollout-wide gather as a description (vln_gather_rollout's arguments), for entry points that run it beside their own
 * work: vln_lstm_seq_fwd(ride = ...) executes it as PASSENGER workgroups of the persistent recurrence launch, on the compute
 * units that launch leaves idle (at B = 64 the recurrence holds 128 of 256 CUs for ~180 us). */
typedef struct vln_gather_ride {
  const void* table; const float* angle_table; const vln_gather_rollout_step* steps /* HOST array of T */;
  int32_t ttype, T, B, V, C, IMG, ANG, pad_;
  uint64_t seed; float p_feat; float padf_;
  const uint64_t* offset_base_dev;     /* nullable, see vln_embed_fwd */
  /* optional (ABI v14; fetch_slots NULL = none): bytes [fetch_offset, fetch_offset + fetch_bytes) of the batch blob that the LAST
   * vln_host_fetch / vln_prologue on this ring pulled the head of -- slot (*fetch_seq - 1) % fetch_ring -- copied to fetch_dst by ONE
   * passenger workgroup: the part of a packed batch that only the decoder reads (angle features, masks, targets) crosses PCIe under
   * the recurrence instead of in front of it.  Multiples of 16 bytes; without room for passengers it is its own small launch. */
  const uint64_t* fetch_slots; const uint64_t* fetch_seq; void* fetch_dst; int64_t fetch_offset, fetch_bytes; int32_t fetch_ring, pad2_;
  /* optional (ABI v17; NULL / 0 = none): weight-shadow jobs (vln_shadow_refresh's, HOST array) of modules the carrier launch does
   * not read itself -- the decoder's, whose first use comes after the instruction encoder -- refreshed by the passengers once their
   * rows are gathered: 69 of the 79 MB the iteration's prologue launch moved at B = 64 leave the dependent chain.  Same results as
   * vln_shadow_refresh(shadow_jobs, n_shadow_jobs); more than 8 jobs, or no room for passengers: that call is issued instead. */
  const vln_shadow_job* shadow_jobs; int32_t n_shadow_jobs, pad3_;
} vln_gather_ride;
/* Index range checks (ABI v10).  A caller that registers its table's extent -- N viewpoint rows, and the number of view
 * indices the angle table holds (36) -- gets every gather of that table (all entry points below, the in-step gather of
 * vln_envdrop_step_fwd, the passengers of vln_lstm_seq_fwd) range-checked on the device: a viewpoint row outside [0, N), a
 * panorama view index outside [0, n_angle_views) or a candidate view outside [0, V) reads NOTHING (the output row is all
 * zeros, like an empty candidate slot) and is counted in the device's host-mapped sticky word; the next
 * vln_persistent_check() -- every later vln_lstm_seq_*, the optimizer step, graphs.IterationGraph.replay -- reports the count
 * ONCE as VLN_ERR_ARG.  Candidate rows < 0 stay the legitimate empty slot.  n_rows = 0 removes the registration (unchecked). */
int vln_feature_table_extent(const void* table, int64_t n_rows, int n_angle_views);
int vln_gather_rollout(const void* table, int ttype, const float* angle_table, const vln_gather_rollout_step* steps, int T, int B, int V,
                       int C, int IMG, int ANG, uint64_t seed, float p_feat, const uint64_t* offset_base_dev /*nullable, see vln_embed_fwd*/,
                       vln_stream_t s);
int vln_gather_pano(const void* table, int ttype, const int64_t* rows, const int32_t* view_index, const float* angle_table,
                    float* out, void* out_bf16, int B, int V, int IMG, int ANG, uint64_t seed, uint64_t offset, float p_feat,
                    vln_stream_t s);
int vln_gather_cands(const void* table, int ttype, const int64_t* rows, const int32_t* views, const float* heading,
                     const float* elevation, float* out, void* out_bf16, int BC, int V, int IMG, int ANG, uint64_t seed,
                     uint64_t offset, float p_feat, vln_stream_t s);
/* Both gathers of a decoder step in ONE launch (same outputs, same dropout indexing: site offsets offset_pano /
 * offset_cand are what vln_gather_pano / vln_gather_cands would have been given). */
int vln_gather_step(const void* table, int ttype, const float* angle_table, const int64_t* rows, const int32_t* view_index,
                    const int64_t* crows, const int32_t* cviews, const float* heading, const float* elevation, float* out,
                    void* out_bf16, float* cout, void* cout_bf16, int B, int V, int C, int IMG, int ANG, uint64_t seed,
                    uint64_t offset_pano, uint64_t offset_cand, float p_feat, vln_stream_t s);

/* ---- EncoderLSTM pieces (units.py:48-74) ------------------------------------------------------------
 * Internal layout is TIME-major: row (t*B + b).  nn.Embedding + Dropout -> vln_embed_fwd; the input projection
 * of all steps is one vln_linear_fwd (M = L*B, bias = b_ih + b_hh); the packed recurrence of one layer (both
 * directions per launch, L launches) is vln_lstm_seq_fwd; pad_packed_sequence + Dropout -> vln_tm_to_bm. */
/* `offset_base_dev` (nullable, here and wherever it appears): a DEVICE word; the site's Philox offset is then
 * (*offset_base_dev) * 8 + offset, read by the kernel, so the launch arguments repeat from iteration to iteration and the
 * launch can be captured into a whole-iteration hipGraph (the word is bumped between iterations: vln_tick).  NULL: `offset`
 * is the Philox offset itself. */
int vln_embed_fwd(const int64_t* tokens /*[B,L]*/, const float* E, float* out_tm /*[L*B,D]*/, int B, int L, int D,
                  uint64_t seed, uint64_t offset, float p, const uint64_t* offset_base_dev, vln_stream_t s);
int vln_embed_bwd(const int64_t* tokens, const int32_t* lengths, const float* dx_tm, float* dE /* += */, int B, int L,
                  int D, int64_t padding_idx, uint64_t seed, uint64_t offset, float p, const uint64_t* offset_base_dev, vln_stream_t s);
/* Same gradient without float atomics (opt-in: ~5x the time): one workgroup per vocabulary row V adds its tokens'
 * gradient rows in a fixed (t, b) order, so the result is reproducible bit for bit (D <= 1024). */
int vln_embed_bwd_det(const int64_t* tokens, const int32_t* lengths, const float* dx_tm, float* dE /* += */, int B, int L,
                      int D, int V, int64_t padding_idx, uint64_t seed, uint64_t offset, float p, const uint64_t* offset_base_dev,
                      vln_stream_t s);
int vln_tm_to_bm(const float* tm /*[L,B,W]*/, float* bm /*[B,L,W]*/, void* bm_bf16 /*nullable*/, int B, int L, int W,
                 uint64_t seed, uint64_t offset, float p, const uint64_t* offset_base_dev, vln_stream_t s);
int vln_bm_to_tm(const float* bm, float* tm, int B, int L, int W, uint64_t seed, uint64_t offset, float p,
                 const uint64_t* offset_base_dev, vln_stream_t s);
/* xproj [L*B, dirs*4Hd]; w_hh [dirs][4Hd,Hd] (wtype); hprev/cprev [dirs][L][B][Hd] (state fed into time t, written
 * here); y_tm [L*B, dirs*Hd]; act [L*B, dirs*4Hd]; tanh_c [L*B, dirs*Hd]; hcat/ccat [B, dirs*Hd] final states */
/* sync_ws (nullable): device scratch of vln_lstm_sync_ws_bytes(B, Hd, dirs) bytes, 16-byte aligned: an 8 KB header
 * (word 32 = status: 0 ok, 1 a bounded spin timed out; one 128-byte arrival line per dependency group from byte 256) followed by the backward's
 * exchange buffer.
 * When given (and large enough) and the grid of (Hd/16) x dirs x ceil(B/16) workgroups is co-resident (<= 256) with
 * Hd in {128,256,512}, the whole sequence runs as ONE persistent launch: W_hh slices and cell state stay in registers;
 * forward exchanges hidden slices, backward exchanges partial dh blocks, both with write-through stores + a
 * per-group arrival counter.  Otherwise L launches (hipGraph-memoised).  The forward needs only the header.
 * INITIAL CONTENTS: the caller zero-fills sync_ws once, when it allocates it.  The granule exchange (data-tagged values) reads
 * stale tags as "not yet written" only if they are older launches' or zero; the arrival counters of the header are zeroed by
 * the library in front of the first counter-protocol launch on a buffer and whenever another protocol (vln_set_persistent)
 * or the counter-protocol forward touched it since -- afterwards the backward kernel leaves them zero itself.
 * The library remembers that by the buffer's ADDRESS: memory that was freed and handed out again at an address the library has
 * seen, or a header the caller wrote to, must be announced with vln_lstm_sync_ws_forget() before the next launch on it (the
 * header then gets its fill again); EncoderLSTM does so for every buffer it has not used before. */
int64_t vln_lstm_sync_ws_bytes(int B, int Hd, int dirs);
int vln_lstm_sync_ws_forget(const void* sync_ws);
/* The granule hand-off tags every exchanged value with a per-buffer LAUNCH SEQUENCE.  device_seq < 0: the library counts the
 * launches of a sync_ws on the host (and clears the exchange when the 24-bit count wraps).  device_seq >= 0: the sequence is
 * the 32-bit device word at byte vln_lstm_sync_seq_offset() of sync_ws PLUS device_seq, read by the kernel -- the launch
 * arguments then repeat from iteration to iteration (whole-iteration hipGraphs); the caller gives every launch between two
 * bumps a different device_seq, bumps the word between iterations by more than the largest one (vln_tick) and zeroes the
 * byte range vln_lstm_sync_granule_range() before the low 24 bits of the word wrap (runtime.DeviceClock does all three). */
int64_t vln_lstm_sync_seq_offset(int B, int Hd, int dirs);
int vln_lstm_sync_granule_range(int B, int Hd, int dirs, int64_t* offset, int64_t* bytes);
int vln_lstm_seq_fwd(const float* xproj, const void* w_hh, int wtype, const int32_t* lengths, float* hprev, float* cprev,
                     float* y_tm, float* act, float* tanh_c, float* hcat, float* ccat, int B, int L, int Hd, int dirs,
                     const float* h0, const float* c0 /* [dirs][B][Hd] initial state, nullable = zeros (no gradient flows
                     back into it: vln_lstm_seq_bwd starts from the final states only) */,
                     void* sync_ws, int64_t sync_ws_bytes, int64_t device_seq,
                     const vln_gather_ride* ride /* nullable: a rollout's feature gather done by this call -- as passenger
                     workgroups of the persistent launch when that path is taken (T <= 12), else as its own launch first */,
                     vln_stream_t s);
/* The same forward recurrence with the INPUT PROJECTION formed inside the persistent launch (ABI v18, round 6): instead of xproj =
 * x W_ih^T + (b_ih + b_hh) from a GEMM launch over all L * B rows, the call takes x [L*B, E] (time-major, fp32), w_ih [dirs*4Hd, E] in
 * the recurrence's weight type and bsum [dirs*4Hd]; four EXTRA waves of every recurrence workgroup form its 16 rows x 64 gate columns
 * of step s + 1 from register-resident W_ih rows while the workgroup's first four waves run step s.  Same MFMA sequence as
 * vln_linear_fwd: bit-identical results.  Only where vln_lstm_inproj_ok(...) returns 1 (the granule-protocol persistent launch,
 * Hd 256, E = 256); elsewhere
 * the caller forms xproj itself and calls vln_lstm_seq_fwd. */
int vln_lstm_inproj_ok(int B, int L, int Hd, int dirs, int E, const void* sync_ws, int64_t sync_ws_bytes);
int vln_lstm_seq_fwd_x(const float* x, int E, const void* w_ih, const float* bsum, const void* w_hh, int wtype,
                       const int32_t* lengths, float* hprev, float* cprev, float* y_tm, float* act, float* tanh_c, float* hcat,
                       float* ccat, int B, int L, int Hd, int dirs, const float* h0, const float* c0, void* sync_ws,
                       int64_t sync_ws_bytes, int64_t device_seq, const vln_gather_ride* ride, vln_stream_t s);
/* The backward recurrence with the layer's OWN weight gradients accumulated inside the persistent launch (ABI v18, round 6): four extra
 * waves per recurrence workgroup contract the step's dgates tile against the step's h_{t-1} and x_t rows (plain bf16 operands, fp32
 * accumulation: the packed contraction's default precision) and leave per-workgroup partial sums in wg_part
 * (vln_lstm_wgrad_part_floats); vln_lstm_wgrad_reduce then adds the row blocks' partials in order into out_hh[d] [4Hd, Hd] and
 * out_ih[d] [4Hd, E] (acc_*[d] = 1: added to what is there; a NULL output is skipped).  d W_hh = sum_t dgates_t^T h_{t-1},
 * d W_ih = sum_t dgates_t^T x_t as vln_wgrad_grouped forms them over the same rows, in another summation order.  Only where
 * vln_lstm_wgrad_inlaunch_ok(...) returns 1 (the counter-protocol persistent launch, bf16 weights, Hd 256, E 256, precision 2). */
int vln_lstm_wgrad_inlaunch_ok(int B, int L, int Hd, int dirs, int E, int wtype, int precision, const void* sync_ws, int64_t sync_ws_bytes);
int64_t vln_lstm_wgrad_part_floats(int B, int Hd, int dirs, int E);
int vln_lstm_seq_bwd_w(const float* dy_tm, const void* w_hh_t, int wtype, const int32_t* lengths, const float* act, const float* tanh_c,
                       const float* cprev, float* dgates, float* dh_pass, float* dc_carry, const float* dh_init_bm, const float* dc_init_bm,
                       int B, int L, int Hd, int dirs, void* sync_ws, int64_t sync_ws_bytes, int64_t device_seq, float* bias_partials,
                       const float* x, int E, const float* hprev, float* wg_part, int64_t wg_part_floats, vln_stream_t s);
int vln_lstm_wgrad_reduce(const float* part, int B, int Hd, int dirs, int E, float* const* out_hh, float* const* out_ih,
                          const int* acc_hh, const int* acc_ih, vln_stream_t s);
int vln_set_persistent(int on);   /* 0 = per-step launch chain; 1 (default) = persistent kernels, granule hand-off forward, counter
                                   * hand-off backward; 2 = counter both ways (round 1); 3 = granules both ways; identical results */
/* The persistent recurrence spins (bounded) on its neighbour workgroups; the host only launches it when the grid fits the
 * device's CU count (queried).  If a wait still times out the kernels count it in a sticky word that every launch copies to
 * pinned host memory; this call -- made by every later vln_lstm_seq_* and by the optimizer step -- reports it ONCE as
 * VLN_ERR_HIP (the affected iteration's numbers are invalid) and switches the process to per-step launches. */
int vln_persistent_check(void);
/* Test hook: raise the current device's sticky word from the host as a timed-out recurrence wait (word 0), an out-of-range
 * gather index (word 1) or a timed-out four-workgroup attention exchange (word 2) would -- so that callers' fallback paths can
 * be exercised on a healthy device. */
int vln_debug_raise_sticky(int word);
/* The four-workgroups-per-row attention (units.py:77-160 on csrc/attention_split.h) spins (bounded) on its three sibling
 * workgroups.  A timeout is counted in its OWN sticky word: vln_persistent_check reports it once as VLN_ERR_HIP and clears
 * this switch, after which every attention runs on one workgroup per row.  1 = allowed (default), 0 = off. */
/* Test hook (ABI v16): cumulative tallies, kept in a recurrence sync workspace's header, of the backward recurrence's per-launch
 * hand-off decisions -- dependency groups that verified they run on ONE XCD (their partial products stay in that XCD's L2) and groups
 * that span XCDs (write-through stores).  Synchronous. */
int vln_lstm_handoff_stats(const void* sync_ws, uint32_t* xcd_local, uint32_t* spanning);
/* (ABI v17) the same tallies for the FORWARD recurrence's granule hand-off (csrc/encoder_persist_g.h) */
int vln_lstm_fwd_handoff_stats(const void* sync_ws, uint32_t* xcd_local, uint32_t* spanning);
int vln_set_split_attention(int on);
int vln_get_split_attention(void);
/* dy_tm grad of y_tm (nullable); w_hh_t [dirs][Hd,4Hd]; dgates [L*B, dirs*4Hd] out; dh_pass/dc_carry [dirs][B][Hd]
 * in: grads of the final states, clobbered.  dh_init_bm / dc_init_bm (both or neither): the same initial gradients in the
 * caller's [B, dirs*Hd] layout (hcat / ccat, units.py:63-67) -- dh_pass / dc_carry are then scratch only and the caller's two
 * transposing copies are not needed. */
int vln_lstm_seq_bwd(const float* dy_tm, const void* w_hh_t, int wtype, const int32_t* lengths, const float* act,
                     const float* tanh_c, const float* cprev, float* dgates, float* dh_pass, float* dc_carry,
                     const float* dh_init_bm /*nullable*/, const float* dc_init_bm /*nullable*/, int B, int L,
                     int Hd, int dirs, void* sync_ws, int64_t sync_ws_bytes, int64_t device_seq,
                     float* bias_partials /* nullable out (ABI v14): [dirs][ceil(B / 16)][4 * Hd] = dgates' columns of each direction summed
                     over the L steps and the 16 rows of a batch block -- the LSTM's bias gradients (b_ih and b_hh receive the same sum,
                     policy: autograd of nn.LSTM) are the sum over the batch blocks, a [ceil(B / 16), 4 * Hd] column sum instead of a pass
                     over the [L * B, 4 * Hd] dgates; the persistent kernel's own threads accumulate them, other forms add one launch */,
                     vln_stream_t s);

/* ---- EnvDropDecoder.forward as one call (policy.py:208-246) and its backward ---------------------- */

typedef struct vln_envdrop_dims {
  int B, L, V, C;        /* batch, instruction length, views (36), candidates (+STOP) */
  int H, IMG, ANG, AE;   /* hidden, image feat (2048), angle feat (128), action embedding (64) */
  int wtype;             /* dtype of the weight shadows (VLN_F32 / VLN_BF16) */
  int ctype;             /* dtype of the streamed img/cand/ctx tensors the attention kernels read */
} vln_envdrop_dims;

typedef struct vln_envdrop_weights {
  const float* act_w;    /* act_embed.0.weight [AE,ANG] */
  const float* act_b;    /* act_embed.0.bias   [AE]     */
  const void* w_vin;     /* visual_attn.linear_in.weight [F,H]           (wtype) */
  const void* w_vin_t;   /* its transpose [H,F]                                    */
  const void* w_cat;     /* [lstm.weight_ih | lstm.weight_hh]  [4H, AE+F+H]        */
  const void* w_cat_t;   /* transpose [AE+F+H, 4H]                                 */
  const float* b_ih;     /* lstm.bias_ih [4H] */
  const float* b_hh;     /* lstm.bias_hh [4H] */
  const void* w_tin;     /* text_attn.linear_in.weight [H,H]   */
  const void* w_tin_t;
  const void* w_tout;    /* text_attn.linear_out.weight [H,2H] */
  const void* w_tout_t;  /* [2H,H] */
  const void* w_c;       /* cand_attn.weight [F,H] */
  const void* w_c_t;     /* [H,F] */
  /* Per-matrix override of dims.wtype: bit k set = that matrix AND its transpose are fp32 although the step streams bf16
   * (bit 0 w_vin, 1 w_cat, 2 w_tin, 3 w_tout, 4 w_c) -- the matrices whose 2^-9 rounding a caller does not want in front of a
   * softmax (EnvDropDecoder.fp32_weights; the measured trade is in profiles/round3_notes.md). */
  int32_t f32_mask; int32_t pad_;
} vln_envdrop_weights;

typedef struct vln_envdrop_step {
  /* inputs (forward) */
  const float* a_prev;   /* [B,ANG] */
  float* img;            /* [B,V,F] fp32, MUTATED in place by the feature dropout */
  float* cand;           /* [B,C,F] fp32, MUTATED in place */
  void* img_lp;          /* [B,V,F] bf16 copy written by fwd when ctype==BF16 (else NULL) */
  void* cand_lp;         /* [B,C,F] bf16 copy */
  const float* h_tilde_prev; /* [B,H] */
  const float* c0;       /* [B,H] */
  const float* ctx;      /* [B,L,H] fp32 */
  const void* ctx_lp;    /* [B,L,H] bf16 copy when ctype==BF16 (else NULL) */
  const uint8_t* ctx_mask; /* [B,L] 1 = masked, may be NULL */
  /* outputs */
  float* logit;          /* [B,C] */
  float* h1;             /* [B,H] */
  float* c1;             /* [B,H] */
  float* h_tilde;        /* [B,H] */
  /* saved for backward; e/xcat/hq/tcat/h1d/htd double as the X operands of the deferred weight-grad GEMMs */
  float* e;              /* [B,AE]   tanh(act_embed) before dropout */
  float* xcat;           /* [B,AE+F+H] LSTM input [drop(e) | visual | h_tilde_prev] */
  float* hq;             /* [B,H]    drop(h_tilde_prev) */
  float* alpha_v;        /* [B,V] */
  float* gate_act;       /* [B,4H] */
  float* tanh_c1;        /* [B,H] */
  float* tcat;           /* [B,2H]  [weighted ctx | drop(h1)] */
  float* tt;             /* [B,H]   text attention query */
  float* alpha_t;        /* [B,L] */
  float* htd;            /* [B,H]   drop(h_tilde) */
  float* a_stash;        /* [B,ANG] nullable: fwd copies a_prev here (X operand of the deferred act_embed weight grad) */
  /* dropout */
  uint64_t seed, offset; /* site k of this step uses Philox offset = offset*8 + k */
  float p_drop, p_feat;
  int already_dropfeat;
  int lp_ready;          /* 1: img_lp / cand_lp were filled by the caller (e.g. vln_gather_*): skip the copy pass */
  /* scratch */
  float* ws; int64_t ws_floats;
  /* optional (nullable): 8 bytes of device memory owned by THIS step (kept until its backward has run).  When given,
   * forward stores `offset` there and every kernel reads the step's dropout offset from it, so the launch arguments
   * repeat from call to call and the step is replayed as one hipGraph (captured on first use per argument block). */
  uint64_t* offset_dev;
  /* optional (nullable, ABI v3): a device word the CALLER wrote, holding a base offset shared by many steps; the step's
   * offset is then *offset_base_dev + offset (offset = the step's position relative to the base).  Nothing but that
   * small relative number varies between the steps of a rollout, no per-step write is launched, and the n-th step of
   * every iteration has the same argument block (replayed as one hipGraph).  The word must stay unchanged until the
   * step's backward has run.  Takes precedence over offset_dev. */
  const uint64_t* offset_base_dev;
  /* ABI v4.  1: forward leaves `logit` UNWRITTEN (step (6), the cand_attn projection + candidate dots, is skipped): with
   * teacher forcing nothing reads the logits before the loss, so the caller forms them for all steps of the rollout at
   * once -- one GEMM over (steps x batch) rows of the `htd` stash + vln_attn_dot_multi -- right before the rollout's loss
   * launch (EnvDropDecoder.defer_logits + losses.RolloutCE).  Sampling / greedy rollouts need the logits per step: 0. */
  int defer_logits; int pad_;
  /* ABI v6, optional (g_table nullable).  When given, the step reads its features from the HBM-resident ResNet table itself
   * (staging.DeviceFeatureStore; reference marshalling agent/base.py:141-157): panorama rows table[g_rows[b], v, :] ++ the
   * static angle embedding of (g_vidx[b], v), candidate rows table[g_crows[b,c], g_cviews[b,c], :] ++ the angle features of
   * (g_chead, g_celev); rows with g_crows < 0 (STOP slot, padding) are zero.  The feature dropout of policy.py:226-231 is
   * applied on the way (the step's own Philox sites 4 / 5, as when it drops caller-given tensors in place) and the rows
   * land in img / cand (fp32 compute) or img_lp / cand_lp (bf16 compute).  The gather and the step's first launch (act
   * embedding, h_tilde_prev copy / dropout) are ONE launch. */
  const void* g_table; const float* g_angle_table;
  const int64_t* g_rows; const int32_t* g_vidx;                       /* [B], [B] */
  const int64_t* g_crows; const int32_t* g_cviews; const float* g_chead; const float* g_celev;   /* [B,C] each */
  int g_ttype; int pad2_;                                             /* table element type: VLN_F32 / VLN_BF16 */
  /* Optional: zero-initialised device scratch of vln_attn_sync_bytes(B) bytes, 16-byte aligned, owned by the caller for the
   * life of the module and touched by nothing else.  With it (and B * 4 <= the device's CU count) the step's two attentions
   * -- 36 x 2176 panorama, <= 80 x 512 instruction context -- run on FOUR workgroups per episode (csrc/attention_split.h: the
   * block's columns are split, the partial row dots exchanged once as data-tagged granules), forward and backward.  NULL:
   * one workgroup per episode. */
  void* attn_sync; int64_t attn_sync_bytes;
  /* ABI v13, CHAINED STEPS (teacher forcing, opt-in; needs defer_logits).  bit 0: forward leaves the last stage of the step -- the
   * sum of linear_out's split-K slabs, tanh, dropout -> h_tilde, drop(h_tilde) (policy.py:241-243) -- PENDING; the next
   * vln_envdrop_step_fwd on the same stream whose h_tilde_prev is this step's h_tilde does it inside its own first launch (which
   * reads exactly that block), anything else gets it issued first (vln_envdrop_flush).  bit 1: backward leaves its last stage --
   * the act-embedding / h_tilde_prev backward (policy.py:224,234) that produces d h_tilde_prev -- pending for the first launch of
   * the next vln_envdrop_step_bwd whose d h_tilde is that buffer.  One dependent launch less per step and direction.  The CALLER
   * promises that nothing outside these two entry points reads h_tilde / htd (forward) or d h_tilde_prev / the `de` stash rows
   * (backward) before the next step call or a vln_envdrop_flush on that stream.  vln_envdrop_step_fwd CLEARS bit 1 in place for a
   * step that did not consume a pending stage itself (the head of a rollout: what consumes its d h_tilde_prev is not a chained step).
   * chain == 2 (bit 1 alone, ABI v15: sampled rollouts, whose forward cannot be chained -- every step's logits are read): the
   * caller decides which steps follow another one (h_tilde_prev IS the previous step's h_tilde and nothing else consumes it) and
   * sets the bit for exactly those; the library leaves it as given. */
  int chain; int pad3_;
  /* ABI v15, optional (nullable): K = ctx W_in, [B,L,H] fp32 -- the instruction context projected ONCE per rollout through
   * text_attn.linear_in (units.py:106-109: ctx . (W_in h) = (ctx W_in) . h; the caller forms it with one vln_linear_fwd over the
   * B * L rows of ctx against w_tin_t).  With it the step's text attention scores K . drop(h_1) directly: the per-step query
   * product W_in drop(h_1) and, in the backward, its transpose leave the step, and the LSTM cell's pointwise stage (forward) /
   * pointwise backward run inside the text-attention launch (csrc/attention_textk.h) -- two dependent launches less per step and
   * direction.  Needs attn_sync and a shape vln_attn_textk_ok accepts (else the call fails), and in the backward the DEFERRED
   * context gradient (vln_envdrop_grads.dctx == NULL): `tt` is NOT written, the caller forms
   *   dctx = sum_t alpha_t g_t + (sum_t dl_t hd_t) W_in^T,   hd_t = tcat[:, H:] of step t
   * with vln_attn_dctx_deferred (dl / q = hd pairs -> [B,L,H]), one vln_linear_fwd against w_tin and a second
   * vln_attn_dctx_deferred (alpha / g pairs, accumulate).  d W_in still comes from the s_dtt rows. */
  const float* kctx;
  /* ABI v15, optional (s_probs nullable): the SAMPLED-ACTION branch of envdrop.py:173,186-195 on this step's logits inside the
   * step's last launch (candidate dots + mask + softmax + draw + log-prob + entropy: what vln_attn_dot + vln_categorical_fwd + a
   * device-to-host copy of the action did as three launches).  s_cand_mask [B,C] 1 = not a candidate (nullable); s_action_in
   * [B] a given action (nullable: the kernel draws with Philox (s_seed, s_offset [+ *s_offset_base_dev * 8]), row b's word);
   * s_action_out [B] int64 device (nullable when s_action_in is given); s_action_host [B] int64 in HOST-MAPPED pinned memory
   * (device-visible address, nullable): the action lands there with a system-scope store -- the host that steps the simulator
   * (envdrop.py:196-206) polls it; s_probs [B,C], s_logp [B], s_ent [B] outputs.  `logit` still receives the raw logits.
   * Not with defer_logits. */
  const uint8_t* s_cand_mask; const int64_t* s_action_in; int64_t* s_action_out; int64_t* s_action_host;
  float* s_probs; float* s_logp; float* s_ent;
  uint64_t s_seed, s_offset; const uint64_t* s_offset_base_dev;
} vln_envdrop_step;

typedef struct vln_envdrop_grads {
  const float* dlogit;   /* [B,C] nullable */
  const float* dh1;      /* [B,H] nullable */
  const float* dc1;      /* [B,H] nullable */
  const float* dh_tilde; /* [B,H] nullable */
  float* dh_tilde_prev;  /* [B,H] out */
  float* dc0;            /* [B,H] out */
  float* dctx;           /* [B,L,H] accumulated in place (+=), nullable */
  /* dY operands of the deferred weight-gradient GEMMs (row blocks of the rollout stash) */
  float* s_dtc;          /* [B,F]  -> cand_attn.weight */
  float* s_dz;           /* [B,H]  -> text_attn.linear_out.weight */
  float* s_dtt;          /* [B,H]  -> text_attn.linear_in.weight */
  float* s_dgates;       /* [B,4H] -> lstm.weight_ih / weight_hh / biases */
  float* s_dtv;          /* [B,F]  -> visual_attn.linear_in.weight */
  float* s_de;           /* [B,AE] -> act_embed.0.{weight,bias} */
  /* deferred context gradient (ABI v2; both nullable).  With dctx == NULL the step leaves these behind instead of
   * sweeping [B,L,H]; the caller forms dctx once per rollout: vln_attn_dctx_deferred(alpha_t, s_dl, s_dtcat, tt). */
  float* s_dl;           /* [B,L]  d(text attention logits) of this step */
  float* s_dtcat;        /* [B,2H] rows; the step writes d(weighted ctx) into columns [0,H) (ld 2H): the g operand of the deferred dctx */
  /* ABI v4: the candidate-logit branch of the backward (logit = cand . (W_c drop(h_tilde)), policy.py:199-206,243-244) precomputed
   * for ALL steps of a rollout -- vln_rows_wsum_multi fills every step's s_dtc, one GEMM over (steps x batch) rows maps them
   * through W_c -- because it depends on no other step's backward.  Given (non-NULL), dlogit is ignored, s_dtc is left as the
   * caller filled it, and this [B,H] block is used as d drop(h_tilde) from the logits. */
  const float* dhtd_ext;
} vln_envdrop_grads;

/* The two launches vln_envdrop_step.kctx puts into the step, callable on their own (csrc/attention_textk.h; reference
 * policy.py:237-241 + units.py:106-117).  Forward: the LSTM cell's pointwise stage on the gate pre-activations `gates`
 * ([nsplit][B,4H] split-K slabs, slab s at gates + s * slab_stride; order i,f,g,o; + b_ih + b_hh, nullable) -> h1, c1 [B,H],
 * act [B,4H] / tanh_c1 [B,H] (saved, nullable), tcat[:, H:2H) = dropout(h1) (Philox (seed, offset), p); then
 * alpha [B,S] = softmax(mask(kctx . dropout(h1))) and tcat[:, 0:H) = sum_s alpha ctx (tcat [B,2H]).  ctx [B,S,H] of `ctype`
 * (VLN_F32 / VLN_BF16), kctx [B,S,H] fp32 = ctx W_in.  Backward: dtcat = d tcat ([nsplit][B,2H] slabs) -> d alpha = ctx . d wc,
 * dl (d logits, [B,S], nullable), dq [B,H] = sum_s dl ctx (the dY rows of d W_in), dwc_out (nullable, [B,2H] rows: columns
 * [0,H) receive the summed d wc), and the cell's pointwise backward with d drop(h1) = dtcat[:, H:] + sum_s dl kctx:
 * dgates [B,4H], dc0 [B,H] (dh1 / dc1: external gradients on h1 / c1, nullable).  `sync` as vln_envdrop_step.attn_sync; the call
 * fails unless vln_attn_textk_ok(ctype, B, S, H, sync, sync_bytes). */
int vln_attn_textk_fwd(const void* ctx, int ctype, const float* kctx, const uint8_t* mask, const float* gates, int nsplit,
                       int64_t slab_stride, const float* b_ih, const float* b_hh, const float* c0, float* h1, float* c1, float* act,
                       float* tanh_c1, float* tcat, float* alpha, uint64_t seed, uint64_t offset, float p, int B, int S, int H,
                       void* sync, int64_t sync_bytes, vln_stream_t s);
int vln_attn_textk_bwd(const void* ctx, int ctype, const float* kctx, const float* alpha, const float* dtcat, int nsplit,
                       int64_t slab_stride, float* dwc_out, float* dq, float* dl, const float* dh1, const float* dc1,
                       const float* act, const float* tanh_c1, const float* c0, float* dgates, float* dc0, uint64_t seed,
                       uint64_t offset, float p, int B, int S, int H, void* sync, int64_t sync_bytes, vln_stream_t s);
/* The device waits for the HOST between two launches of a sequence (ABI v15): a one-wave launch that spins until the 8-byte word
 * at flag_dev (pinned host memory, device-visible address: vln_host_device_pointer) equals the device word *want_dev, then
 * lets the stream go on.  For rollouts whose next step needs the host (the simulator step of envdrop.py:196-206) inside ONE
 * captured iteration: `want` = the device clock's word (it changes every iteration, so an old flag never matches).  The wait is
 * bounded (ABI v18): spin_limit > 0 = that many polls; 0 = VLN_HOST_WAIT_DEFAULT_US of wall clock (the 100 MHz constant clock);
 * < 0 = -spin_limit microseconds.  On a timeout -- or when the host stores VLN_HOST_WAIT_POISON into the flag: it has given the
 * iteration up, e.g. an exception in its turn -- word 3 of the sticky error words is raised (vln_persistent_check reports the
 * iteration as invalid) and the stream goes on.  vln_host_wait_fetch (ABI v18): the same wait and then, in the same launch, a copy
 * of nbytes (multiple of 16) from src_dev (pinned host memory, device-visible address) to dst -- what the host left for the next
 * step (agent/base.py:141-178: the observation's index vectors) crosses PCIe without a launch of its own.  ack_dev (nullable;
 * pinned host memory, device-visible address): when the wait is over and the bytes are read, *want_dev is stored there -- a host that
 * runs AHEAD of the device (the next iteration's graph is already launched) must see this iteration's value in the word of turn i
 * before it rewrites turn i's mailbox and flag for the next iteration (graphs.HandshakeIterationGraph does). */
#define VLN_HOST_WAIT_POISON 0xFFFFFFFFFFFFFFFFull
#define VLN_HOST_WAIT_DEFAULT_US 2000000
int vln_host_wait(const uint64_t* flag_dev, const uint64_t* want_dev, int64_t spin_limit, uint64_t* ack_dev, vln_stream_t s);
/* ... and what a step hands BACK (ABI v18): nbytes (multiple of 8) from device memory to pinned host memory (device-visible
 * address) by one workgroup's system-scope stores -- the chosen actions of envdrop.py:198 (`a_t.cpu()`), polled by the host. */
int vln_store_to_host(const void* src, void* dst_dev, int64_t nbytes, vln_stream_t s);
int vln_host_wait_fetch(const uint64_t* flag_dev, const uint64_t* want_dev, int64_t spin_limit, const void* src_dev, void* dst,
                        int64_t nbytes, uint64_t* ack_dev, vln_stream_t s);
int64_t vln_envdrop_ws_floats(const vln_envdrop_dims* d);
int64_t vln_attn_sync_bytes(int B);   /* bytes of vln_envdrop_step.attn_sync for B episodes */
/* 1 when the folded text attention of vln_envdrop_step.kctx covers (ctype, B episodes, S tokens, D = H) on the current device with
 * this exchange buffer (four workgroups per episode co-resident, S <= 96, D <= 512, D % 32 == 0, the path not switched off) */
int vln_attn_textk_ok(int ctype, int B, int S, int D, const void* sync, int64_t sync_bytes);
int vln_envdrop_flush(vln_stream_t s);      /* issue what a chained step left pending on this stream (no-op if nothing is) */
/* Forget the pending stages of stream s WITHOUT issuing them (their rollout was abandoned: an iteration that raised); returns how
 * many were dropped.  A pending stage reads the split-K slabs its step left in vln_envdrop_step.ws: the caller keeps that buffer
 * untouched -- and alive -- until the next step call, vln_envdrop_flush or this call (EnvDropDecoder gives chained steps a
 * workspace of their own for that reason). */
int vln_envdrop_drop_pending(vln_stream_t s);
int vln_envdrop_step_fwd(const vln_envdrop_dims* d, const vln_envdrop_weights* w, vln_envdrop_step* io, vln_stream_t s);
int vln_envdrop_step_bwd(const vln_envdrop_dims* d, const vln_envdrop_weights* w, vln_envdrop_step* io,
                         vln_envdrop_grads* g, vln_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* VLN_HIP_H */
