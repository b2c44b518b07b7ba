// Fused elementwise stages: LSTM 4-gate cell epilogue (sums the split-K gate slabs, adds both biases,
// applies the i,f,g,o nonlinearities and the state update in one pass), its backward, Philox dropout
// helpers and the in-place environmental feature dropout (policy.py:226-231).
#include "vln_internal.h"
#include "step_bodies.h"
#include "../../include/vln_hip.h"

namespace vln {

// ---------------------------------------------------------------------------
// LSTM cell pointwise (torch.nn.LSTMCell semantics, gate order i,f,g,o)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_pw_fwd_kernel(LstmPwFwd a, int iters) {
  __shared__ float sg[4][64];
  lstm_pw_fwd_body(a, (int)blockIdx.x, (int)gridDim.x, iters, (int)threadIdx.x, sg);
}
int lstm_pointwise_fwd(hipStream_t st, const LstmPwFwd& a) {
  long total = (long)a.B * a.H;
  const long groups = (total + 63) / 64;
  int blocks = (int)groups;
  if (blocks > 4096) blocks = 4096;
  const int iters = (int)((groups + blocks - 1) / blocks);
  VLN_LAUNCH(lstm_pw_fwd_kernel, dim3(blocks), dim3(256), 0, st, a, iters);
  VLN_CHECK_LAUNCH("lstm_pointwise_fwd");
  return VLN_OK;
}

__global__ __launch_bounds__(256) void lstm_pw_bwd_kernel(LstmPwBwd a) {
  lstm_pw_bwd_body(a, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}
int lstm_pointwise_bwd(hipStream_t st, const LstmPwBwd& a) {
  long total = (long)a.B * a.H;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  VLN_LAUNCH(lstm_pw_bwd_kernel, dim3(blocks), dim3(256), 0, st, a);
  VLN_CHECK_LAUNCH("lstm_pointwise_bwd");
  return VLN_OK;
}

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_dropout_kernel(const float* x, long ldx, float* y, long ldy,
                                                            int rows, int cols, DropSpec d) {
  const long total = (long)rows * cols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int r = (int)(e / cols), c = (int)(e % cols);
    y[(long)r * ldy + c] = x[(long)r * ldx + c] * dropout_scale1(d.seed, d.off(), (uint32_t)e, d.p);
  }
}
int scale_dropout(hipStream_t st, const float* x, long ldx, float* y, long ldy, int rows, int cols, DropSpec d) {
  long total = (long)rows * cols;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  VLN_LAUNCH(scale_dropout_kernel, dim3(blocks), dim3(256), 0, st, x, ldx, y, ldy, rows, cols, d);
  VLN_CHECK_LAUNCH("scale_dropout");
  return VLN_OK;
}

__global__ __launch_bounds__(256) void export_mask_kernel(float* out, long n, DropSpec d) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    out[e] = dropout_scale1(d.seed, d.off(), (uint32_t)e, d.p);
}
int export_dropout_mask(hipStream_t st, float* out, long n, DropSpec d) {
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  VLN_LAUNCH(export_mask_kernel, dim3(blocks), dim3(256), 0, st, out, n, d);
  VLN_CHECK_LAUNCH("export_dropout_mask");
  return VLN_OK;
}

__global__ __launch_bounds__(256) void fill_kernel(float* p, long n, float v) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) p[e] = v;
}
int fill_f32(hipStream_t st, float* p, long n, float v) {
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  VLN_LAUNCH(fill_kernel, dim3(blocks), dim3(256), 0, st, p, n, v);
  VLN_CHECK_LAUNCH("fill_f32");
  return VLN_OK;
}

__global__ __launch_bounds__(256) void add_inplace_kernel(float* y, long ldy, const float* x, long ldx, int rows,
                                                          int cols) {
  const long total = (long)rows * cols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int r = (int)(e / cols), c = (int)(e % cols);
    y[(long)r * ldy + c] += x[(long)r * ldx + c];
  }
}
int add_inplace(hipStream_t st, float* y, long ldy, const float* x, long ldx, int rows, int cols) {
  long total = (long)rows * cols;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  VLN_LAUNCH(add_inplace_kernel, dim3(blocks), dim3(256), 0, st, y, ldy, x, ldx, rows, cols);
  VLN_CHECK_LAUNCH("add_inplace");
  return VLN_OK;
}

// ---------------------------------------------------------------------------
// Environmental feature dropout, in place on the image part of every [img | angle] row.
// Element index for the mask = row*img + d (image part only); 4 elements per Philox call.
// Optionally emits a bf16 copy of the WHOLE row (image part dropped, angle tail verbatim) that the
// bf16 attention path streams instead of the fp32 tensor.
// ---------------------------------------------------------------------------
template <typename TX>
__global__ __launch_bounds__(256) void feat_dropout_kernel(TX* x, long rows, int img, int angle, DropSpec d,
                                                           bf16_raw* copy) {
  const int F = img + angle;
  const int q4 = img >> 2;                       // img % 4 == 0 checked on the host
  const long total4 = rows * (long)(F >> 2);     // F % 4 == 0 checked on the host
  const int f4 = F >> 2;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (long)gridDim.x * blockDim.x) {
    const long r = e / f4;
    const int c4 = (int)(e % f4);
    TX* p = x + r * F + c4 * 4;
    float v[4];
    Elt<TX>::ld4(p, v);
    if (c4 < q4) {
      if (d.p > 0.f) {
        float m[4];
        dropout_scale4(d.seed, d.off(), (uint32_t)(r * q4 + c4), d.p, m);
        v[0] *= m[0]; v[1] *= m[1]; v[2] *= m[2]; v[3] *= m[3];
        Elt<TX>::st4(p, v);
      }
    }
    if (copy) Elt<bf16_raw>::st4(copy + r * F + c4 * 4, v);
  }
}
int feat_dropout_inplace(hipStream_t st, void* x, int xtype, long rows, int img, int angle, DropSpec d,
                         void* copy_bf16) {
  if ((img & 3) || (angle & 3)) { set_error("feat_dropout: img/angle sizes must be multiples of 4"); return VLN_ERR_ARG; }
  if (rows <= 0) return VLN_OK;
  if (d.p <= 0.f && !copy_bf16) return VLN_OK;
  long total4 = rows * (long)((img + angle) >> 2);
  int blocks = (int)((total4 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (xtype == W_BF16)
    VLN_LAUNCH(feat_dropout_kernel<bf16_raw>, dim3(blocks), dim3(256), 0, st, (bf16_raw*)x, rows, img, angle, d, (bf16_raw*)copy_bf16);
  else
    VLN_LAUNCH(feat_dropout_kernel<float>, dim3(blocks), dim3(256), 0, st, (float*)x, rows, img, angle, d, (bf16_raw*)copy_bf16);
  VLN_CHECK_LAUNCH("feat_dropout");
  return VLN_OK;
}

}  // namespace vln

// ---------------------------------------------------------------------------
// A9: in-place candidate mask + CrossEntropyLoss(ignore_index) + Categorical log-prob/entropy in ONE pass
// (follower.py:123-139, envdrop.py:173-195, monitor.py:146-176).  One wave per batch row (C <= 64 candidates
// per lane-slot, looped beyond).  Saves the masked softmax so backward is a single scaled subtraction.
// ---------------------------------------------------------------------------
namespace vln {
struct CeArgs {
  float* logits; long ld; const long long* target; const unsigned char* mask; float* loss; float* probs;
  const long long* action; float* logp; float* entropy; int B, C; long ignore_index; int write_mask;
};
// one wave per row; returns the row's CE term (wave-uniform)
__device__ __forceinline__ float ce_row(const CeArgs& a, int b, int lane) {
  const int C = a.C;
  const unsigned char* mask = a.mask;
  float* lg = a.logits + (long)b * a.ld;
  float mx = -INFINITY;
  for (int c = lane; c < C; c += 64) {
    float v = lg[c];
    if (mask && mask[(long)b * C + c]) {
      v = -INFINITY;
      if (a.write_mask) lg[c] = v;               // logits.masked_fill_(candidate_mask, -inf), in place like the caller
    }
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = (mask && mask[(long)b * C + c]) ? -INFINITY : lg[c];
    sum += __expf(v - mx);
  }
  sum = wave_sum(sum);
  const float lse = mx + __logf(sum);
  const long tgt = a.target ? a.target[b] : a.ignore_index;
  const float eps = 1.1920928955078125e-07f;     // torch.distributions clamp_probs
  float ent = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = (mask && mask[(long)b * C + c]) ? -INFINITY : lg[c];
    const float p = __expf(v - lse);
    if (a.probs) a.probs[(long)b * C + c] = p;
    const float pc = fminf(fmaxf(p, eps), 1.f - eps);
    ent -= p * __logf(pc);
    if (a.action && a.logp && c == a.action[b]) a.logp[b] = __logf(pc);
  }
  ent = wave_sum(ent);
  const float l = (tgt == a.ignore_index) ? 0.f : (lse - lg[tgt]);
  if (lane == 0) {
    if (a.entropy) a.entropy[b] = ent;
    if (a.loss) a.loss[b] = l;
  }
  return l;
}
__global__ __launch_bounds__(256) void masked_ce_fwd_kernel(CeArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b < a.B) ce_row(a, b, lane);
}
// reduction="sum" in the same launch: ONE workgroup, one THREAD per row (a row is <= a few dozen candidates: three
// short serial passes beat a wave's cross-lane reductions, and 64 rows then cost one wave), block sum in a fixed order.
__device__ __forceinline__ float ce_row_serial(const CeArgs& a, int b) {
  const int C = a.C;
  const unsigned char* mk = a.mask ? a.mask + (long)b * C : nullptr;
  float* lg = a.logits + (long)b * a.ld;
  float mx = -INFINITY;
  for (int c = 0; c < C; ++c) {
    float v = lg[c];
    if (mk && mk[c]) {
      v = -INFINITY;
      if (a.write_mask) lg[c] = v;
    }
    mx = fmaxf(mx, v);
  }
  float sum = 0.f;
  for (int c = 0; c < C; ++c) sum += __expf(((mk && mk[c]) ? -INFINITY : lg[c]) - mx);
  const float lse = mx + __logf(sum);
  const long tgt = a.target ? a.target[b] : a.ignore_index;
  const float eps = 1.1920928955078125e-07f;
  float ent = 0.f;
  for (int c = 0; c < C; ++c) {
    const float p = __expf(((mk && mk[c]) ? -INFINITY : lg[c]) - lse);
    if (a.probs) a.probs[(long)b * C + c] = p;
    const float pc = fminf(fmaxf(p, eps), 1.f - eps);
    ent -= p * __logf(pc);
    if (a.action && a.logp && c == a.action[b]) a.logp[b] = __logf(pc);
  }
  const float l = (tgt == a.ignore_index) ? 0.f : (lse - lg[tgt]);
  if (a.entropy) a.entropy[b] = ent;
  if (a.loss) a.loss[b] = l;
  return l;
}
// Rows of <= 16 candidates (the navigation graphs have <= 14 + STOP): the whole row is fetched with 16 independent
// loads from clamped addresses (one memory round trip instead of three dependent passes over freshly written
// logits), the arithmetic then runs in registers.  Same operation order as ce_row_serial: identical results.
__device__ __forceinline__ float ce_row_regs(const CeArgs& a, int b) {
  const int C = a.C;
  const unsigned char* mk = a.mask ? a.mask + (long)b * C : nullptr;
  float* lg = a.logits + (long)b * a.ld;
  float v[16];
  bool m[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int cc = min(c, C - 1);
    v[c] = lg[cc];
    m[c] = mk ? (mk[cc] != 0) : false;
  }
  const long tgt = a.target ? a.target[b] : a.ignore_index;
  const long act = (a.action && a.logp) ? a.action[b] : -1;
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (m[c]) v[c] = -INFINITY;
    if (c < C) mx = fmaxf(mx, v[c]);
  }
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) if (c < C) sum += __expf(v[c] - mx);
  const float lse = mx + __logf(sum);
  const float eps = 1.1920928955078125e-07f;
  float ent = 0.f, vt = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (c < C) {
      if (m[c] && a.write_mask) lg[c] = -INFINITY;
      const float p = __expf(v[c] - lse);
      if (a.probs) a.probs[(long)b * C + c] = p;
      const float pc = fminf(fmaxf(p, eps), 1.f - eps);
      ent -= p * __logf(pc);
      if (c == act) a.logp[b] = __logf(pc);
      if (c == tgt) vt = v[c];
    }
  }
  const float l = (tgt == a.ignore_index) ? 0.f : (lse - vt);
  if (a.entropy) a.entropy[b] = ent;
  if (a.loss) a.loss[b] = l;
  return l;
}
// mean != 0 (nn.CrossEntropyLoss's default reduction, follower.py:62): loss_sum[0] = sum / #rows with a target, loss_sum[1] = 1 / that
// count (what the backward scales by) -- the count is formed in the same launch (0 rows: 0/0 = nan, as torch)
__global__ __launch_bounds__(256) void masked_ce_fwd_sum_kernel(CeArgs a, float* loss_sum, int mean) {
  __shared__ float part[4];
  __shared__ float cnt[4];
  float acc = 0.f, n = 0.f;
  for (int b = threadIdx.x; b < a.B; b += 256) {
    acc += (a.C <= 16) ? ce_row_regs(a, b) : ce_row_serial(a, b);
    if (mean && a.target[b] != a.ignore_index) n += 1.f;
  }
  acc = wave_sum(acc);
  if (mean) n = wave_sum(n);
  if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = acc; cnt[threadIdx.x >> 6] = n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float total = (part[0] + part[1]) + (part[2] + part[3]);
    if (mean) {
      const float c = (cnt[0] + cnt[1]) + (cnt[2] + cnt[3]);
      loss_sum[0] = total / c;
      loss_sum[1] = 1.f / c;
    } else {
      loss_sum[0] = total;
    }
  }
}

// the mean of LARGE problems (the speaker's 5120 x 992 word logits): masked_ce_fwd_kernel has left the per-row losses in `rows` (a
// wave per row over the whole chip); one workgroup sums them and counts the rows with a target, in a fixed order
__global__ __launch_bounds__(256) void masked_ce_mean_finish_kernel(const float* rows, const long long* target, int B, long ignore_index,
                                                                    float* loss_sum) {
  __shared__ float part[4];
  __shared__ float cnt[4];
  float acc = 0.f, n = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    acc += rows[b];
    if (target[b] != ignore_index) n += 1.f;
  }
  acc = wave_sum(acc);
  n = wave_sum(n);
  if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = acc; cnt[threadIdx.x >> 6] = n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float c = (cnt[0] + cnt[1]) + (cnt[2] + cnt[3]);
    loss_sum[0] = ((part[0] + part[1]) + (part[2] + part[3])) / c;
    loss_sum[1] = 1.f / c;
  }
}

// dlogits[b,c] = dloss[b] * (p - onehot(target))   (0 for ignored rows; p = 0 at masked slots)
// dloss_stride 0: one scalar upstream gradient for every row (the backward of the fused sum)
// scale (nullable): one device scalar multiplied into every row's gradient (the 1 / count of the mean reduction)
__global__ __launch_bounds__(256) void masked_ce_bwd_kernel(const float* probs, const long long* target, const float* dloss,
                                                            long dloss_stride, float* dlogits, int B, int C,
                                                            long ignore_index, const float* scale) {
  const long total = (long)B * C;
  const float sc = scale ? scale[0] : 1.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int b = (int)(e / C), c = (int)(e % C);
    const long t = target[b];
    const float up = scale ? dloss[b * dloss_stride] * sc : dloss[b * dloss_stride];
    dlogits[e] = (t == ignore_index) ? 0.f : up * (probs[e] - (c == t ? 1.f : 0.f));
  }
}

// ---------------------------------------------------------------------------
// Self-Monitor step loss (monitor.py:146-165) in one launch each way: the action CE, the progress target built from the
// distances (monitor.py:155-157: (start - cur) / start, 1 within 3 m of the goal, the prediction itself for ended
// episodes -- read on the device, the reference pulls cur_prog_val to the host every step for this), the MSE and the mix
//   t == 0: CE          t > 0: lam * MSE + (1 - lam) * CE
// with nn.CrossEntropyLoss(ignore_index) / nn.MSELoss() means (non-curriculum) or per-episode terms (reduction="none",
// the curriculum criteria).  One workgroup, one thread per episode, fixed-order block sums.
// ---------------------------------------------------------------------------
struct MonLossArgs {
  CeArgs ce;                                   // logits / target / mask / probs (loss, action, logp, entropy unused)
  const float* progress; long ldp;             // [B] cur_prog_val
  const float* start_dist; const float* cur_dist; const unsigned char* ended;
  float* prog_target;                          // [B] out, kept for backward
  float* out;                                  // per_sample: [B], else [1]
  float* stats;                                // [2]: mean progress MSE (the agent's progress_loss record), #rows with a target
  int t; float lam; int per_sample;
};
__global__ __launch_bounds__(256) void monitor_loss_fwd_kernel(MonLossArgs m) {
  __shared__ float p_ce[4], p_sq[4], p_n[4];
  const int B = m.ce.B;
  float ce_acc = 0.f, sq_acc = 0.f, n_acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float l = (m.ce.C <= 16) ? ce_row_regs(m.ce, b) : ce_row_serial(m.ce, b);
    const float p = m.progress[(long)b * m.ldp];
    const float sd = m.start_dist[b], cd = m.cur_dist[b];
    float pt = (sd - cd) / sd;
    if (cd <= 3.0f) pt = 1.0f;
    if (m.ended[b]) pt = p;
    m.prog_target[b] = pt;
    const float sq = (p - pt) * (p - pt);
    if (m.per_sample) m.out[b] = (m.t == 0) ? l : m.lam * sq + (1.f - m.lam) * l;
    ce_acc += l; sq_acc += sq;
    n_acc += (m.ce.target[b] != m.ce.ignore_index) ? 1.f : 0.f;
  }
  ce_acc = wave_sum(ce_acc); sq_acc = wave_sum(sq_acc); n_acc = wave_sum(n_acc);
  if ((threadIdx.x & 63) == 0) { p_ce[threadIdx.x >> 6] = ce_acc; p_sq[threadIdx.x >> 6] = sq_acc; p_n[threadIdx.x >> 6] = n_acc; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float ce = (p_ce[0] + p_ce[1]) + (p_ce[2] + p_ce[3]);
    const float sq = (p_sq[0] + p_sq[1]) + (p_sq[2] + p_sq[3]);
    const float n = (p_n[0] + p_n[1]) + (p_n[2] + p_n[3]);
    const float mse = sq / (float)B;
    m.stats[0] = mse; m.stats[1] = n;
    if (!m.per_sample) {
      const float cem = ce / n;                          // mean over the rows that have a target (0/0 = nan like torch)
      m.out[0] = (m.t == 0) ? cem : m.lam * mse + (1.f - m.lam) * cem;
    }
  }
}
struct MonLossBwdArgs {
  const float* probs; const long long* target; const float* progress; long ldp; const float* prog_target; const float* stats;
  const float* dloss; long dloss_stride;       // [1] (stride 0) or [B]
  float* dlogits; float* dprogress;            // [B,C], [B]
  int B, C, t; float lam; int per_sample; long ignore_index;
};
__global__ __launch_bounds__(256) void monitor_loss_bwd_kernel(MonLossBwdArgs m) {
  const long total = (long)m.B * m.C;
  const float wce = (m.t == 0 ? 1.f : 1.f - m.lam) * (m.per_sample ? 1.f : 1.f / m.stats[1]);
  const float wsq = (m.t == 0 ? 0.f : m.lam * 2.f) * (m.per_sample ? 1.f : 1.f / (float)m.B);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total + m.B; e += (long)gridDim.x * blockDim.x) {
    if (e < total) {
      const int b = (int)(e / m.C), c = (int)(e % m.C);
      const long tg = m.target[b];
      m.dlogits[e] = (tg == m.ignore_index) ? 0.f : m.dloss[b * m.dloss_stride] * wce * (m.probs[e] - (c == tg ? 1.f : 0.f));
    } else {
      const int b = (int)(e - total);
      m.dprogress[b] = m.dloss[b * m.dloss_stride] * wsq * (m.progress[(long)b * m.ldp] - m.prog_target[b]);
    }
  }
}

// The Self-Monitor loss of a WHOLE ROLLOUT, sum_t cur_loss_t (monitor.py:146-165,196), in one launch each way (round 6): nothing on
// the rollout's dependent chain reads a step's loss, so the T launches (and the T of the backward, and the T - 1 additions of the
// running total) only lengthened the stream.  One workgroup; thread per (step, episode) row, same per-row arithmetic as
// monitor_loss_fwd_kernel; the row terms go through LDS and thread t sums step t in episode order.
constexpr int kMonMultiMaxT = 16;
constexpr int kMonMultiRowsMax = 4096;
struct MonMultiArgs {
  float* logits[kMonMultiMaxT]; const long long* target[kMonMultiMaxT]; const unsigned char* mask[kMonMultiMaxT]; float* probs[kMonMultiMaxT];
  const float* progress[kMonMultiMaxT]; const float* start_dist[kMonMultiMaxT]; const float* cur_dist[kMonMultiMaxT];
  const unsigned char* ended[kMonMultiMaxT]; float* prog_target[kMonMultiMaxT]; float* dlogits[kMonMultiMaxT]; float* dprogress[kMonMultiMaxT];
  int C[kMonMultiMaxT], ld[kMonMultiMaxT], ldp[kMonMultiMaxT];
  int T, B, t0; long ignore_index; float lam;
  float* out; float* stats;               // [1] (+= when accumulate), [T][2] = {mean progress MSE, rows with a target} per step
  int accumulate;
  const float* dloss;                     // backward: [1]
};
__global__ __launch_bounds__(1024) void monitor_loss_multi_fwd_kernel(MonMultiArgs m) {
  __shared__ float rce[kMonMultiRowsMax], rsq[kMonMultiRowsMax];
  __shared__ unsigned char rhas[kMonMultiRowsMax];        // the row has a target (the per-step sums below read LDS only: a global load
  __shared__ float steploss[kMonMultiMaxT];               // per row in that serial loop cost 25 us at T 7 x B 128)
  const int rows = m.T * m.B;
  for (int r = threadIdx.x; r < rows; r += (int)blockDim.x) {       // (1024 threads: a row each at B 128 / T 7 -- 25 us with 256, the rows in turn)
    const int t = r / m.B, b = r - t * m.B;
    CeArgs a{m.logits[t], (long)m.ld[t], m.target[t], m.mask[t], nullptr, m.probs[t], nullptr, nullptr, nullptr, m.B, m.C[t], m.ignore_index, 0};
    rce[r] = (a.C <= 16) ? ce_row_regs(a, b) : ce_row_serial(a, b);
    const float p = m.progress[t][(long)b * m.ldp[t]];
    const float sd = m.start_dist[t][b], cd = m.cur_dist[t][b];
    float pt = (sd - cd) / sd;
    if (cd <= 3.0f) pt = 1.0f;
    if (m.ended[t][b]) pt = p;
    m.prog_target[t][b] = pt;
    rsq[r] = (p - pt) * (p - pt);
    rhas[r] = (m.target[t][b] != m.ignore_index) ? 1 : 0;
  }
  __syncthreads();
  if ((int)threadIdx.x < m.T) {
    const int t = threadIdx.x;
    float ce = 0.f, sq = 0.f, n = 0.f;
    for (int b = 0; b < m.B; ++b) {
      ce += rce[t * m.B + b]; sq += rsq[t * m.B + b];
      n += rhas[t * m.B + b] ? 1.f : 0.f;
    }
    const float mse = sq / (float)m.B, cem = ce / n;          // (no row with a target: 0 / 0 = nan, as torch)
    m.stats[2 * t] = mse; m.stats[2 * t + 1] = n;
    steploss[t] = (m.t0 + t == 0) ? cem : m.lam * mse + (1.f - m.lam) * cem;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int t = 0; t < m.T; ++t) v += steploss[t];
    m.out[0] = m.accumulate ? m.out[0] + v : v;
  }
}
__global__ __launch_bounds__(256) void monitor_loss_multi_bwd_kernel(MonMultiArgs m) {
  const int t = blockIdx.x;
  const int C = m.C[t];
  const long total = (long)m.B * C;
  const bool first = (m.t0 + t == 0);
  const float up = m.dloss[0];
  const float wce = (first ? 1.f : 1.f - m.lam) / m.stats[2 * t + 1];
  const float wsq = (first ? 0.f : m.lam * 2.f) / (float)m.B;
  for (long e = threadIdx.x; e < total + m.B; e += 256) {
    if (e < total) {
      const int b = (int)(e / C), c = (int)(e % C);
      const long tg = m.target[t][b];
      m.dlogits[t][e] = (tg == m.ignore_index) ? 0.f : up * wce * (m.probs[t][e] - (c == tg ? 1.f : 0.f));
    } else {
      const int b = (int)(e - total);
      m.dprogress[t][b] = up * wsq * (m.progress[t][(long)b * m.ldp[t]] - m.prog_target[t][b]);
    }
  }
}

// The IL loss of a whole rollout, ml_loss = sum_t CE_t (envdrop.py:178-179), in ONE launch after the last decoder step
// instead of one per step: nothing on the rollout's dependent chain needs the loss, so the T small launches (and the T
// backward ones) only lengthen it.  One workgroup; thread per (step, episode) row; rows of <= 16 candidates in registers.
// Same per-row arithmetic as masked_ce_fwd_sum_kernel; the total is summed in a fixed order (rows strided by 256 per
// thread, then the block tree).  Backward: one workgroup per step.
struct CeMultiArgs {
  float* logits[VLN_CE_MAX_STEPS]; const long long* target[VLN_CE_MAX_STEPS]; const unsigned char* mask[VLN_CE_MAX_STEPS];
  float* probs[VLN_CE_MAX_STEPS]; float* dlogits[VLN_CE_MAX_STEPS]; int C[VLN_CE_MAX_STEPS]; int ld[VLN_CE_MAX_STEPS];
  int T, B; long ignore_index;
};
// Mean PER STEP (nn.CrossEntropyLoss(ignore_index)'s default reduction applied to every step's batch, then summed over the steps:
// follower.py:62,123-139): loss = scale * sum_t (sum_b CE_tb / n_t), n_t = step t's rows with a target; inv_counts[t] = 1 / n_t is left
// for the backward.  One workgroup; the row losses go through LDS, thread t sums step t's rows in episode order (fixed order).
constexpr int kCeMeanRowsMax = 8192;
__global__ __launch_bounds__(1024) void masked_ce_multi_mean_fwd_kernel(CeMultiArgs m, float* loss_sum, int accumulate, float scale, float* inv_counts) {
  __shared__ float rowloss[kCeMeanRowsMax];
  __shared__ unsigned char rowhas[kCeMeanRowsMax];        // the row has a target (the serial sums below read LDS only)
  __shared__ float stepmean[VLN_CE_MAX_STEPS];
  const int rows = m.T * m.B;
  for (int r = threadIdx.x; r < rows; r += (int)blockDim.x) {
    const int t = r / m.B, b = r - t * m.B;
    CeArgs a{m.logits[t], (long)m.ld[t], m.target[t], m.mask[t], nullptr, m.probs[t], nullptr, nullptr, nullptr, m.B, m.C[t],
             m.ignore_index, 0};
    rowloss[r] = (a.C <= 16) ? ce_row_regs(a, b) : ce_row_serial(a, b);
    rowhas[r] = (m.target[t][b] != m.ignore_index) ? 1 : 0;
  }
  __syncthreads();
  if ((int)threadIdx.x < m.T) {
    const int t = threadIdx.x;
    float acc = 0.f, n = 0.f;
    for (int b = 0; b < m.B; ++b) {
      acc += rowloss[t * m.B + b];
      if (rowhas[t * m.B + b]) n += 1.f;
    }
    stepmean[t] = acc / n;                       // (no row with a target: 0 / 0 = nan, as torch)
    inv_counts[t] = 1.f / n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int t = 0; t < m.T; ++t) v += stepmean[t];
    v *= scale;
    loss_sum[0] = accumulate ? loss_sum[0] + v : v;
  }
}
// (256 or 512 threads -- blockDim.x: at the headline's 7 x 64 rows every thread of 512 takes ONE row, its loads are not queued
// behind another row's; the total is summed in a fixed order either way: rows strided by the block per thread, waves, wave pairs)
__global__ __launch_bounds__(512) void masked_ce_multi_fwd_kernel(CeMultiArgs m, float* loss_sum, int accumulate, float scale) {
  __shared__ float part[8];
  float acc = 0.f;
  const int rows = m.T * m.B;
  for (int r = threadIdx.x; r < rows; r += (int)blockDim.x) {
    const int t = r / m.B, b = r - t * m.B;
    CeArgs a{m.logits[t], (long)m.ld[t], m.target[t], m.mask[t], nullptr, m.probs[t], nullptr, nullptr, nullptr, m.B, m.C[t],
             m.ignore_index, 0};
    acc += (a.C <= 16) ? ce_row_regs(a, b) : ce_row_serial(a, b);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = (part[0] + part[1]) + (part[2] + part[3]);
    if (blockDim.x > 256) v += (part[4] + part[5]) + (part[6] + part[7]);
    v *= scale;
    loss_sum[0] = accumulate ? loss_sum[0] + v : v;
  }
}
// Per-episode form (SELF-PACE consumes the loss VECTOR, envdrop.py:70,178-179 with reduction="none"; curriculum.py:296):
// loss_rows[b] = scale * sum_t CE_t[b].  Block = 64 episodes x 4 step groups (group g takes steps g, g+4, ...); the four
// partials of an episode are added in a fixed order.
__global__ __launch_bounds__(256) void masked_ce_multi_rows_kernel(CeMultiArgs m, float* loss_rows, int accumulate, float scale) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int b = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (b < m.B) {
    for (int t = grp; t < m.T; t += 4) {
      CeArgs a{m.logits[t], (long)m.ld[t], m.target[t], m.mask[t], nullptr, m.probs[t], nullptr, nullptr, nullptr, m.B, m.C[t],
               m.ignore_index, 0};
      acc += (a.C <= 16) ? ce_row_regs(a, b) : ce_row_serial(a, b);
    }
  }
  part[grp][lane] = acc;
  __syncthreads();
  if (grp == 0 && b < m.B) {
    const float v = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) * scale;
    loss_rows[b] = accumulate ? loss_rows[b] + v : v;
  }
}
// dloss_stride 0: one upstream scalar (the summed form); 1: one per episode (the per-episode form)
__global__ __launch_bounds__(256) void masked_ce_multi_bwd_kernel(CeMultiArgs m, const float* dloss, int dloss_stride, float scale,
                                                                  const float* inv_counts) {
  const int t = blockIdx.x;
  const int C = m.C[t];
  const float* probs = m.probs[t];
  const long long* target = m.target[t];
  float* dl = m.dlogits[t];
  const float sc = inv_counts ? scale * inv_counts[t] : scale;          // (the mean per step: 1 / n_t, left by the forward)
  for (int e = threadIdx.x; e < m.B * C; e += 256) {
    const int b = e / C, c = e - b * C;
    const long tg = target[b];
    dl[e] = (tg == m.ignore_index) ? 0.f : dloss[b * dloss_stride] * sc * (probs[e] - (c == tg ? 1.f : 0.f));
  }
}
}  // namespace vln

// ---------------------------------------------------------------------------------------------------------------
// A2C sweep of the EnvDrop rollout (envdrop.py:235-264) in one launch: one thread per episode walks the steps backwards,
//   R_t = gamma R_{t+1} + r_t  (R_T = last_value where the episode has not ended),  A_t = R_t - V_t (a constant),
//   loss_b = sum_t m_t ( -logp_t A_t + 1/2 (R_t - V_t)^2 - c H_t ),
// and leaves the partial derivatives behind (d/dlogp = -A m, d/dV = -(R - V) m, d/dH = -c m) so that backward is one
// elementwise launch.  total = sum of the masks (the reference's normaliser) is counted in the same pass.
// ---------------------------------------------------------------------------------------------------------------
namespace vln {
__global__ __launch_bounds__(256) void a2c_fwd_kernel(const float* logp, const float* ent, const float* val, const float* reward,
                                                      const uint8_t* mask, const float* last_value, const uint8_t* ended, int T, int B,
                                                      float gamma, float ent_coef, float* loss_b, float* dlogp, float* dval,
                                                      float* dent, float* total) {
  __shared__ float part[4];
  float cnt = 0.f;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
    float R = ended[b] ? 0.f : last_value[b];
    float acc = 0.f;
    for (int t = T - 1; t >= 0; --t) {
      const long i = (long)t * B + b;
      R = R * gamma + reward[i];
      const float m = mask[i] ? 1.f : 0.f;
      const float adv = R - val[i];
      const float h = ent ? ent[i] : 0.f;
      acc += -logp[i] * adv * m;               // same order of accumulation as the reference's three += per step
      acc += 0.5f * (adv * adv) * m;
      if (ent) acc += -ent_coef * h * m;
      dlogp[i] = -adv * m;
      dval[i] = -adv * m;
      if (dent) dent[i] = -ent_coef * m;
      cnt += m;
    }
    loss_b[b] = acc;
  }
  if (total) {                                 // single workgroup when total is requested (host sizes the grid)
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) total[0] = (part[0] + part[1]) + (part[2] + part[3]);
  }
}
__global__ __launch_bounds__(256) void a2c_bwd_kernel(const float* dloss_b, long stride, const float* dlogp, const float* dval,
                                                      const float* dent, int T, int B, float* glogp, float* gval, float* gent) {
  const long n = (long)T * B;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float g = dloss_b[(i % B) * stride];
    if (glogp) glogp[i] = g * dlogp[i];
    if (gval) gval[i] = g * dval[i];
    if (gent && dent) gent[i] = g * dent[i];
  }
}
}  // namespace vln

extern "C" int vln_a2c_loss_fwd(const float* logp, const float* ent, const float* val, const float* reward, const uint8_t* mask,
                                const float* last_value, const uint8_t* ended, int T, int B, float gamma, float ent_coef,
                                float* loss_b, float* dlogp, float* dval, float* dent, float* total, void* s) {
  if (!logp || !val || !reward || !mask || !last_value || !ended || !loss_b || !dlogp || !dval || T <= 0 || B <= 0 || (ent && !dent)) {
    vln::set_error("vln_a2c_loss_fwd: bad args");
    return VLN_ERR_ARG;
  }
  VLN_LAUNCH(vln::a2c_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, logp, ent, val, reward, mask, last_value, ended, T, B,
                     gamma, ent_coef, loss_b, dlogp, dval, dent, total);
  VLN_CHECK_LAUNCH("a2c_loss_fwd");
  return VLN_OK;
}
extern "C" int vln_a2c_loss_bwd(const float* dloss_b, int64_t dloss_stride, const float* dlogp, const float* dval, const float* dent,
                                int T, int B, float* glogp, float* gval, float* gent, void* s) {
  if (!dloss_b || !dlogp || !dval || T <= 0 || B <= 0) { vln::set_error("vln_a2c_loss_bwd: bad args"); return VLN_ERR_ARG; }
  long n = (long)T * B;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 256) blocks = 256;
  VLN_LAUNCH(vln::a2c_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, dloss_b, (long)dloss_stride, dlogp, dval, dent, T,
                     B, glogp, gval, gent);
  VLN_CHECK_LAUNCH("a2c_loss_bwd");
  return VLN_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm1d (+ optional ReLU) of the Self-Monitor agent's BN-MLP (units.py:210-242, policy.py:148-149) in ONE launch
// forward and ONE backward.  A workgroup owns a strip of 16 columns and ALL rows (64 row lanes), so the batch statistics
// need no second kernel: mean, then the centred second moment (second pass over the strip, L2-resident: 64 B per row),
// reduced through LDS; the normalised rows are written and the running statistics updated (momentum, unbiased
// variance, num_batches_tracked) as torch.nn.BatchNorm1d does.
// Backward, training mode:  dx = g rstd / R * (R dy - sum dy - xhat sum(dy xhat)),  dgamma = sum dy xhat, dbeta = sum dy
// (dy masked by y > 0 when the ReLU is fused); eval mode: dx = dy g rstd with the running statistics.
// ---------------------------------------------------------------------------------------------------------------
namespace vln {
struct BnArgs {
  const float* x; long ldx; float* y; long ldy;
  const float* gamma; const float* beta; float* run_mean; float* run_var; long long* nbt;
  float* save_mean; float* save_rstd;
  int R, D; float eps, momentum; int training, relu;
  DropSpec drop;                    // optional dropout between the normalisation and the ReLU (MLPwithBN: BN, Dropout, ReLU)
  const unsigned char* row_zero;    // optional [R]: rows whose output is forced to 0 (padded candidate slots, policy.py:148-149)
  // TWO SEGMENTS (row-chunked form only): rows [0, R1) and [R1, R) are two independent batches -- the Self-Monitor's BN-MLP runs
  // on the previous action (B rows) and on the candidates (B*C rows) with the same weights, policy.py:140-149 -- normalised with
  // their OWN statistics in one launch pair.  Segment 1 uses drop2, indexes its Philox elements and row_zero from ITS first row
  // (exactly what a second call would do), keeps its saved statistics `stat2` floats after segment 0's, and the running
  // statistics take both updates in order (num_batches_tracked += 2).  R1 == 0: one segment.
  int R1; long stat2; DropSpec drop2;
  const float* x2; long ldx2;       // nullable (two segments): segment 1's rows are x2 + (r - R1) * ldx2 -- the two batches read in place
};
// chunk -> (segment, first row, rows in the chunk, rows of the segment, the segment's first chunk and chunk count)
struct BnSegChunk { int seg, r0, n, seg_rows, seg_r0, first, count; };
__device__ __forceinline__ float4 bn_strip_sum(float4 v, float4 (*part)[4], int rl, int cg) {
  part[rl][cg] = v;
  __syncthreads();
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int k = 0; k < 64; ++k) { const float4 q = part[k][cg]; t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }
  __syncthreads();
  return t;
}
// Workgroups are dealt to the 8 XCDs round-robin by block index.  A workgroup owns a 64-byte column strip, i.e. HALF of every
// 128-byte line it touches: with the identity mapping the two halves of a line are fetched by two different XCDs' L2s (twice
// the HBM traffic).  This mapping gives each XCD a CONTIGUOUS range of strips instead.
__device__ __forceinline__ int xcd_chunked_block(int b, int nblk) {
  const int x = b & 7, j = b >> 3;
  const int q = nblk >> 3, rem = nblk & 7;          // XCD y owns q + (y < rem) blocks
  return x * q + min(x, rem) + j;
}
__global__ __launch_bounds__(256) void bn_fwd_kernel(BnArgs a) {
  __shared__ float4 part[64][4];
  const int cg = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = xcd_chunked_block(blockIdx.x, gridDim.x) * 16 + cg * 4;
  const bool c_ok = c < a.D;                                   // D % 4 == 0
  const int cc = c_ok ? c : 0;
  const float* xp = a.x + cc;
  float4 mean, rstd;
  if (a.training) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c_ok)
#pragma unroll 4
      for (int r = rl; r < a.R; r += 64) {
        const float4 t = *reinterpret_cast<const float4*>(xp + (long)r * a.ldx);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
    s = bn_strip_sum(s, part, rl, cg);
    const float inv = 1.f / (float)a.R;
    mean = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c_ok)
#pragma unroll 4
      for (int r = rl; r < a.R; r += 64) {                      // second pass over the strip: L2-resident by now
        const float4 t = *reinterpret_cast<const float4*>(xp + (long)r * a.ldx);
        const float dx = t.x - mean.x, dy = t.y - mean.y, dz = t.z - mean.z, dw = t.w - mean.w;
        q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
      }
    q = bn_strip_sum(q, part, rl, cg);
    const float4 var = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
    rstd = make_float4(rsqrtf(var.x + a.eps), rsqrtf(var.y + a.eps), rsqrtf(var.z + a.eps), rsqrtf(var.w + a.eps));
    if (rl == 0 && c_ok) {
      if (a.save_mean) *reinterpret_cast<float4*>(a.save_mean + c) = mean;
      if (a.save_rstd) *reinterpret_cast<float4*>(a.save_rstd + c) = rstd;
      if (a.run_mean) {
        const float m = a.momentum, ub = (a.R > 1) ? (float)a.R / (float)(a.R - 1) : 1.f;
        float4 rm = *reinterpret_cast<float4*>(a.run_mean + c), rv = *reinterpret_cast<float4*>(a.run_var + c);
        rm.x = (1.f - m) * rm.x + m * mean.x; rm.y = (1.f - m) * rm.y + m * mean.y;
        rm.z = (1.f - m) * rm.z + m * mean.z; rm.w = (1.f - m) * rm.w + m * mean.w;
        rv.x = (1.f - m) * rv.x + m * var.x * ub; rv.y = (1.f - m) * rv.y + m * var.y * ub;
        rv.z = (1.f - m) * rv.z + m * var.z * ub; rv.w = (1.f - m) * rv.w + m * var.w * ub;
        *reinterpret_cast<float4*>(a.run_mean + c) = rm;
        *reinterpret_cast<float4*>(a.run_var + c) = rv;
      }
    }
    if (a.nbt && blockIdx.x == 0 && threadIdx.x == 0) *a.nbt += 1;
  } else {
    const float4 rm = *reinterpret_cast<const float4*>(a.run_mean + cc), rv = *reinterpret_cast<const float4*>(a.run_var + cc);
    mean = rm;
    rstd = make_float4(rsqrtf(rv.x + a.eps), rsqrtf(rv.y + a.eps), rsqrtf(rv.z + a.eps), rsqrtf(rv.w + a.eps));
  }
  if (!c_ok) return;
  const float4 g = a.gamma ? *reinterpret_cast<const float4*>(a.gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 bt = a.beta ? *reinterpret_cast<const float4*>(a.beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int r = rl; r < a.R; r += 64) {
    const float4 t = *reinterpret_cast<const float4*>(xp + (long)r * a.ldx);
    float4 o = make_float4((t.x - mean.x) * rstd.x * g.x + bt.x, (t.y - mean.y) * rstd.y * g.y + bt.y,
                           (t.z - mean.z) * rstd.z * g.z + bt.z, (t.w - mean.w) * rstd.w * g.w + bt.w);
    if (a.drop.p > 0.f) {
      float m[4];
      dropout_scale4(a.drop.seed, a.drop.off(), (uint32_t)(((long)r * a.D + c) >> 2), a.drop.p, m);
      o.x *= m[0]; o.y *= m[1]; o.z *= m[2]; o.w *= m[3];
    }
    if (a.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    if (a.row_zero && a.row_zero[r]) o = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(a.y + (long)r * a.ldy + c) = o;
  }
}

struct BnBwdArgs {
  const float* x; long ldx; const float* dy; long lddy; const float* y; long ldy;   // y only for the fused ReLU mask
  const float* gamma; const float* mean; const float* rstd;     // training: saved batch stats; eval: running mean, var
  float* dx; long lddx; float* dgamma; float* dbeta;
  int R, D; float eps; int training, relu, accumulate;
  DropSpec drop; const unsigned char* row_zero;     // as in the forward
  int R1; long stat2; DropSpec drop2;               // two segments, as in the forward (row-chunked form only)
  const float* x2; long ldx2;                       // as in the forward
};
__global__ __launch_bounds__(256) void bn_bwd_kernel(BnBwdArgs a) {
  __shared__ float4 part[64][4];
  const int cg = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = xcd_chunked_block(blockIdx.x, gridDim.x) * 16 + cg * 4;
  const bool c_ok = c < a.D;
  const int cc = c_ok ? c : 0;
  const float4 mean = *reinterpret_cast<const float4*>(a.mean + cc);
  float4 rstd = *reinterpret_cast<const float4*>(a.rstd + cc);
  if (!a.training) rstd = make_float4(rsqrtf(rstd.x + a.eps), rsqrtf(rstd.y + a.eps), rsqrtf(rstd.z + a.eps), rsqrtf(rstd.w + a.eps));
  const float4 g = a.gamma ? *reinterpret_cast<const float4*>(a.gamma + cc) : make_float4(1.f, 1.f, 1.f, 1.f);
  auto grad_at = [&](int r) {                // dy of row r with the fused ReLU's mask applied
    float4 dv = *reinterpret_cast<const float4*>(a.dy + (long)r * a.lddy + cc);
    if (a.row_zero && a.row_zero[r]) dv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.drop.p > 0.f) {
      float m[4];
      dropout_scale4(a.drop.seed, a.drop.off(), (uint32_t)(((long)r * a.D + cc) >> 2), a.drop.p, m);
      dv.x *= m[0]; dv.y *= m[1]; dv.z *= m[2]; dv.w *= m[3];
    }
    if (a.relu) {
      const float4 yv = *reinterpret_cast<const float4*>(a.y + (long)r * a.ldy + cc);
      dv.x = yv.x > 0.f ? dv.x : 0.f; dv.y = yv.y > 0.f ? dv.y : 0.f; dv.z = yv.z > 0.f ? dv.z : 0.f; dv.w = yv.w > 0.f ? dv.w : 0.f;
    }
    return dv;
  };
  auto xhat_at = [&](int r) {
    const float4 xv = *reinterpret_cast<const float4*>(a.x + (long)r * a.ldx + cc);
    return make_float4((xv.x - mean.x) * rstd.x, (xv.y - mean.y) * rstd.y, (xv.z - mean.z) * rstd.z, (xv.w - mean.w) * rstd.w);
  };
  float4 sd = make_float4(0.f, 0.f, 0.f, 0.f), sdx = sd;
  if (c_ok)
#pragma unroll 4
    for (int r = rl; r < a.R; r += 64) {
      const float4 dv = grad_at(r), xh = xhat_at(r);
      sd.x += dv.x; sd.y += dv.y; sd.z += dv.z; sd.w += dv.w;
      sdx.x += dv.x * xh.x; sdx.y += dv.y * xh.y; sdx.z += dv.z * xh.z; sdx.w += dv.w * xh.w;
    }
  sd = bn_strip_sum(sd, part, rl, cg);
  sdx = bn_strip_sum(sdx, part, rl, cg);
  if (!c_ok) return;
  if (rl == 0) {
    if (a.dgamma) {
      float4 o = sdx;
      if (a.accumulate) { const float4 p = *reinterpret_cast<float4*>(a.dgamma + c); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
      *reinterpret_cast<float4*>(a.dgamma + c) = o;
    }
    if (a.dbeta) {
      float4 o = sd;
      if (a.accumulate) { const float4 p = *reinterpret_cast<float4*>(a.dbeta + c); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
      *reinterpret_cast<float4*>(a.dbeta + c) = o;
    }
  }
  if (!a.dx) return;
  const float inv = 1.f / (float)a.R;
#pragma unroll 4
  for (int r = rl; r < a.R; r += 64) {       // second pass: the strip is L2-resident
    const float4 dv = grad_at(r);
    float4 o;
    if (a.training) {
      const float4 xh = xhat_at(r);
      o.x = g.x * rstd.x * (dv.x - inv * (sd.x + xh.x * sdx.x));
      o.y = g.y * rstd.y * (dv.y - inv * (sd.y + xh.y * sdx.y));
      o.z = g.z * rstd.z * (dv.z - inv * (sd.z + xh.z * sdx.z));
      o.w = g.w * rstd.w * (dv.w - inv * (sd.w + xh.w * sdx.w));
    } else {
      o = make_float4(dv.x * g.x * rstd.x, dv.y * g.y * rstd.y, dv.z * g.z * rstd.z, dv.w * g.w * rstd.w);
    }
    *reinterpret_cast<float4*>(a.dx + (long)r * a.lddx + c) = o;
  }
}
// ---- row-chunked forms for tall inputs (R >= 512: the candidates' BN-MLP runs on B * C = 1024 rows) ----------------------
// One workgroup per 16-column strip walks ALL rows two or three times: D / 16 workgroups (8 for the 128-wide layer) and
// R / 64 dependent loop trips each -- 20-40 us for 0.5-16 MB.  Here the rows are cut into chunks of 128 (two rows per
// thread, kept in registers between the passes) and the grid is (strips, chunks):
//   phase A  per-chunk statistics into `ws`  (forward: mean and M2 of the chunk; backward: sum dy and sum dy * xhat)
//   phase B  every workgroup merges the chunks' statistics for its strip in a fixed order (forward: Chan's parallel
//            mean / M2 merge -- no E[x^2] - E[x]^2 cancellation) and finishes its own rows; chunk 0 writes the per-feature
//            outputs (saved / running statistics; d gamma, d beta).
// Two launches of ~6 us each instead of one of 20-40.
constexpr int kBnChunk = 128;
struct BnChunkWs { float* part; int nchunk; };       // part [nchunk][2][D]
__host__ __device__ __forceinline__ int bn_nchunk(int R, int R1) {
  return R1 > 0 ? (R1 + kBnChunk - 1) / kBnChunk + (R - R1 + kBnChunk - 1) / kBnChunk : (R + kBnChunk - 1) / kBnChunk;
}
__device__ __forceinline__ BnSegChunk bn_chunk(int ch, int R, int R1) {
  if (R1 <= 0) return BnSegChunk{0, ch * kBnChunk, min(kBnChunk, R - ch * kBnChunk), R, 0, 0, (R + kBnChunk - 1) / kBnChunk};
  const int n0 = (R1 + kBnChunk - 1) / kBnChunk;
  if (ch < n0) return BnSegChunk{0, ch * kBnChunk, min(kBnChunk, R1 - ch * kBnChunk), R1, 0, 0, n0};
  const int c = ch - n0, R2 = R - R1;
  return BnSegChunk{1, R1 + c * kBnChunk, min(kBnChunk, R2 - c * kBnChunk), R2, R1, n0, (R2 + kBnChunk - 1) / kBnChunk};
}

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
// row r of the input: segment 1 may live in its own array (BnArgs.x2)
__device__ __forceinline__ const float* bn_xrow(const float* x, long ldx, const float* x2, long ldx2, const BnSegChunk& sc, int r) {
  return (sc.seg && x2) ? x2 + (long)(r - sc.seg_r0) * ldx2 : x + (long)r * ldx;
}

__global__ __launch_bounds__(256) void bn_fwd_stats_kernel(BnArgs a, BnChunkWs w) {
  __shared__ float4 part[64][4];
  const int cg = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = xcd_chunked_block(blockIdx.x, gridDim.x) * 16 + cg * 4;
  const bool c_ok = c < a.D;
  const int cc = c_ok ? c : 0;
  const int ch = blockIdx.y;
  const BnSegChunk sc = bn_chunk(ch, a.R, a.R1);
  const int r0 = sc.r0, n = sc.n;
  const int ra = r0 + rl, rb = r0 + 64 + rl;
  const bool oka = c_ok && rl < n, okb = c_ok && 64 + rl < n;
  const float4 z = f4(0.f);
  const float4 ta = oka ? *reinterpret_cast<const float4*>(bn_xrow(a.x, a.ldx, a.x2, a.ldx2, sc, ra) + cc) : z;
  const float4 tb = okb ? *reinterpret_cast<const float4*>(bn_xrow(a.x, a.ldx, a.x2, a.ldx2, sc, rb) + cc) : z;
  float4 s = make_float4(ta.x + tb.x, ta.y + tb.y, ta.z + tb.z, ta.w + tb.w);
  s = bn_strip_sum(s, part, rl, cg);
  const float inv = 1.f / (float)n;
  const float4 mean = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  float4 q = z;
  if (oka) { const float dx = ta.x - mean.x, dy = ta.y - mean.y, dz = ta.z - mean.z, dw = ta.w - mean.w; q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw; }
  if (okb) { const float dx = tb.x - mean.x, dy = tb.y - mean.y, dz = tb.z - mean.z, dw = tb.w - mean.w; q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw; }
  q = bn_strip_sum(q, part, rl, cg);
  if (rl == 0 && c_ok) {
    *reinterpret_cast<float4*>(w.part + ((long)ch * 2 + 0) * a.D + c) = mean;
    *reinterpret_cast<float4*>(w.part + ((long)ch * 2 + 1) * a.D + c) = q;
  }
}
__global__ __launch_bounds__(256) void bn_fwd_apply_kernel(BnArgs a, BnChunkWs w) {
  const int cg = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = xcd_chunked_block(blockIdx.x, gridDim.x) * 16 + cg * 4;
  if (c >= a.D) return;
  const int ch = blockIdx.y;
  const BnSegChunk sc = bn_chunk(ch, a.R, a.R1);
  const int r0 = sc.r0;
  float4 mean, rstd;
  if (a.training) {
    // a segment's batch statistics from its chunks' (mean, M2), merged in chunk order (Chan)
    auto merged = [&](const BnSegChunk& g, float4& mean_o, float4& var_o) {
      float4 ms = f4(0.f);
      for (int i = 0; i < g.count; ++i) {
        const float ni = (float)min(kBnChunk, g.seg_rows - i * kBnChunk);
        const float4 mi = *reinterpret_cast<const float4*>(w.part + ((long)(g.first + i) * 2) * a.D + c);
        ms.x += ni * mi.x; ms.y += ni * mi.y; ms.z += ni * mi.z; ms.w += ni * mi.w;
      }
      const float inv = 1.f / (float)g.seg_rows;
      mean_o = make_float4(ms.x * inv, ms.y * inv, ms.z * inv, ms.w * inv);
      float4 m2 = f4(0.f);
      for (int i = 0; i < g.count; ++i) {
        const float ni = (float)min(kBnChunk, g.seg_rows - i * kBnChunk);
        const float4 mi = *reinterpret_cast<const float4*>(w.part + ((long)(g.first + i) * 2) * a.D + c);
        const float4 qi = *reinterpret_cast<const float4*>(w.part + ((long)(g.first + i) * 2 + 1) * a.D + c);
        const float dx = mi.x - mean_o.x, dy = mi.y - mean_o.y, dz = mi.z - mean_o.z, dw = mi.w - mean_o.w;
        m2.x += qi.x + ni * dx * dx; m2.y += qi.y + ni * dy * dy; m2.z += qi.z + ni * dz * dz; m2.w += qi.w + ni * dw * dw;
      }
      var_o = make_float4(m2.x * inv, m2.y * inv, m2.z * inv, m2.w * inv);
    };
    float4 var;
    merged(sc, mean, var);
    rstd = make_float4(rsqrtf(var.x + a.eps), rsqrtf(var.y + a.eps), rsqrtf(var.z + a.eps), rsqrtf(var.w + a.eps));
    if (ch == sc.first && rl == 0) {          // the segment's first chunk saves the segment's statistics
      if (a.save_mean) *reinterpret_cast<float4*>(a.save_mean + sc.seg * a.stat2 + c) = mean;
      if (a.save_rstd) *reinterpret_cast<float4*>(a.save_rstd + sc.seg * a.stat2 + c) = rstd;
    }
    if (ch == 0 && rl == 0) {                 // ONE workgroup per strip updates the running statistics: segment 0, then segment 1
      if (a.run_mean) {
        float4 rm = *reinterpret_cast<float4*>(a.run_mean + c), rv = *reinterpret_cast<float4*>(a.run_var + c);
        const int nseg = a.R1 > 0 ? 2 : 1;
        for (int sgi = 0; sgi < nseg; ++sgi) {
          float4 mu = mean, vr = var;
          int rows = sc.seg_rows;
          if (sgi == 1) {
            const BnSegChunk g1 = bn_chunk((a.R1 + kBnChunk - 1) / kBnChunk, a.R, a.R1);
            merged(g1, mu, vr);
            rows = g1.seg_rows;
          }
          const float m = a.momentum, ub = (rows > 1) ? (float)rows / (float)(rows - 1) : 1.f;
          rm.x = (1.f - m) * rm.x + m * mu.x; rm.y = (1.f - m) * rm.y + m * mu.y;
          rm.z = (1.f - m) * rm.z + m * mu.z; rm.w = (1.f - m) * rm.w + m * mu.w;
          rv.x = (1.f - m) * rv.x + m * vr.x * ub; rv.y = (1.f - m) * rv.y + m * vr.y * ub;
          rv.z = (1.f - m) * rv.z + m * vr.z * ub; rv.w = (1.f - m) * rv.w + m * vr.w * ub;
        }
        *reinterpret_cast<float4*>(a.run_mean + c) = rm;
        *reinterpret_cast<float4*>(a.run_var + c) = rv;
      }
      if (a.nbt && blockIdx.x == 0 && cg == 0) *a.nbt += (a.R1 > 0 ? 2 : 1);
    }
  } else {
    mean = *reinterpret_cast<const float4*>(a.run_mean + c);
    const float4 rv = *reinterpret_cast<const float4*>(a.run_var + c);
    rstd = make_float4(rsqrtf(rv.x + a.eps), rsqrtf(rv.y + a.eps), rsqrtf(rv.z + a.eps), rsqrtf(rv.w + a.eps));
  }
  const float4 g = a.gamma ? *reinterpret_cast<const float4*>(a.gamma + c) : f4(1.f);
  const float4 bt = a.beta ? *reinterpret_cast<const float4*>(a.beta + c) : f4(0.f);
  const DropSpec& dsp = sc.seg ? a.drop2 : a.drop;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int r = r0 + k * 64 + rl;
    if (k * 64 + rl >= sc.n) continue;
    const int rs = r - sc.seg_r0;                     // row within its segment: what a call on the segment alone would index
    const float4 t = *reinterpret_cast<const float4*>(bn_xrow(a.x, a.ldx, a.x2, a.ldx2, sc, r) + c);
    float4 o = make_float4((t.x - mean.x) * rstd.x * g.x + bt.x, (t.y - mean.y) * rstd.y * g.y + bt.y,
                           (t.z - mean.z) * rstd.z * g.z + bt.z, (t.w - mean.w) * rstd.w * g.w + bt.w);
    if (dsp.p > 0.f) {
      float m[4];
      dropout_scale4(dsp.seed, dsp.off(), (uint32_t)(((long)rs * a.D + c) >> 2), dsp.p, m);
      o.x *= m[0]; o.y *= m[1]; o.z *= m[2]; o.w *= m[3];
    }
    if (a.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    if (a.row_zero && (a.R1 <= 0 || sc.seg == 1) && a.row_zero[rs]) o = f4(0.f);
    *reinterpret_cast<float4*>(a.y + (long)r * a.ldy + c) = o;
  }
}

// backward: `phase` 0 = per-chunk sums of dy and dy * xhat into ws; 1 = merge + d gamma / d beta (chunk 0) + dx of the chunk
template <int kPhase>
__global__ __launch_bounds__(256) void bn_bwd_chunk_kernel(BnBwdArgs a, BnChunkWs w) {
  __shared__ float4 part[64][4];
  const int cg = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = xcd_chunked_block(blockIdx.x, gridDim.x) * 16 + cg * 4;
  const bool c_ok = c < a.D;
  const int cc = c_ok ? c : 0;
  const int ch = blockIdx.y;
  const BnSegChunk sc = bn_chunk(ch, a.R, a.R1);
  const int r0 = sc.r0;
  const long so = a.training ? sc.seg * a.stat2 : 0;          // eval mode: both segments use the running statistics
  const float4 mean = *reinterpret_cast<const float4*>(a.mean + so + cc);
  float4 rstd = *reinterpret_cast<const float4*>(a.rstd + so + cc);
  const DropSpec& dsp = sc.seg ? a.drop2 : a.drop;
  if (!a.training) rstd = make_float4(rsqrtf(rstd.x + a.eps), rsqrtf(rstd.y + a.eps), rsqrtf(rstd.z + a.eps), rsqrtf(rstd.w + a.eps));
  float4 dv[2], xh[2];
  bool ok[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int r = r0 + k * 64 + rl;
    const int rs = r - sc.seg_r0;
    ok[k] = c_ok && k * 64 + rl < sc.n;
    dv[k] = f4(0.f); xh[k] = f4(0.f);
    if (ok[k]) {
      float4 d = *reinterpret_cast<const float4*>(a.dy + (long)r * a.lddy + cc);
      if (a.row_zero && (a.R1 <= 0 || sc.seg == 1) && a.row_zero[rs]) d = f4(0.f);
      if (dsp.p > 0.f) {
        float m[4];
        dropout_scale4(dsp.seed, dsp.off(), (uint32_t)(((long)rs * a.D + cc) >> 2), dsp.p, m);
        d.x *= m[0]; d.y *= m[1]; d.z *= m[2]; d.w *= m[3];
      }
      if (a.relu) {
        const float4 yv = *reinterpret_cast<const float4*>(a.y + (long)r * a.ldy + cc);
        d.x = yv.x > 0.f ? d.x : 0.f; d.y = yv.y > 0.f ? d.y : 0.f; d.z = yv.z > 0.f ? d.z : 0.f; d.w = yv.w > 0.f ? d.w : 0.f;
      }
      dv[k] = d;
      const float4 xv = *reinterpret_cast<const float4*>(bn_xrow(a.x, a.ldx, a.x2, a.ldx2, sc, r) + cc);
      xh[k] = make_float4((xv.x - mean.x) * rstd.x, (xv.y - mean.y) * rstd.y, (xv.z - mean.z) * rstd.z, (xv.w - mean.w) * rstd.w);
    }
  }
  if constexpr (kPhase == 0) {
    float4 sd = make_float4(dv[0].x + dv[1].x, dv[0].y + dv[1].y, dv[0].z + dv[1].z, dv[0].w + dv[1].w);
    float4 sdx = make_float4(dv[0].x * xh[0].x + dv[1].x * xh[1].x, dv[0].y * xh[0].y + dv[1].y * xh[1].y,
                             dv[0].z * xh[0].z + dv[1].z * xh[1].z, dv[0].w * xh[0].w + dv[1].w * xh[1].w);
    sd = bn_strip_sum(sd, part, rl, cg);
    sdx = bn_strip_sum(sdx, part, rl, cg);
    if (rl == 0 && c_ok) {
      *reinterpret_cast<float4*>(w.part + ((long)ch * 2 + 0) * a.D + c) = sd;
      *reinterpret_cast<float4*>(w.part + ((long)ch * 2 + 1) * a.D + c) = sdx;
    }
  } else {
    if (!c_ok) return;
    // this segment's sums (for its rows' dx); d gamma / d beta take every chunk of BOTH segments (chunk 0's workgroup)
    float4 sd = f4(0.f), sdx = f4(0.f), td = f4(0.f), tdx = f4(0.f);
    for (int i = 0; i < w.nchunk; ++i) {
      const float4 p0 = *reinterpret_cast<const float4*>(w.part + ((long)i * 2) * a.D + c);
      const float4 p1 = *reinterpret_cast<const float4*>(w.part + ((long)i * 2 + 1) * a.D + c);
      td.x += p0.x; td.y += p0.y; td.z += p0.z; td.w += p0.w;
      tdx.x += p1.x; tdx.y += p1.y; tdx.z += p1.z; tdx.w += p1.w;
      if (i >= sc.first && i < sc.first + sc.count) {
        sd.x += p0.x; sd.y += p0.y; sd.z += p0.z; sd.w += p0.w;
        sdx.x += p1.x; sdx.y += p1.y; sdx.z += p1.z; sdx.w += p1.w;
      }
    }
    if (ch == 0 && rl == 0) {
      const float4 sd_all = td, sdx_all = tdx;
      if (a.dgamma) {
        float4 o = sdx_all;
        if (a.accumulate) { const float4 p = *reinterpret_cast<float4*>(a.dgamma + c); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
        *reinterpret_cast<float4*>(a.dgamma + c) = o;
      }
      if (a.dbeta) {
        float4 o = sd_all;
        if (a.accumulate) { const float4 p = *reinterpret_cast<float4*>(a.dbeta + c); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
        *reinterpret_cast<float4*>(a.dbeta + c) = o;
      }
    }
    if (!a.dx) return;
    const float4 g = a.gamma ? *reinterpret_cast<const float4*>(a.gamma + c) : f4(1.f);
    const float inv = 1.f / (float)sc.seg_rows;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (!ok[k]) continue;
      const int r = r0 + k * 64 + rl;
      float4 o;
      if (a.training) {
        o.x = g.x * rstd.x * (dv[k].x - inv * (sd.x + xh[k].x * sdx.x));
        o.y = g.y * rstd.y * (dv[k].y - inv * (sd.y + xh[k].y * sdx.y));
        o.z = g.z * rstd.z * (dv[k].z - inv * (sd.z + xh[k].z * sdx.z));
        o.w = g.w * rstd.w * (dv[k].w - inv * (sd.w + xh[k].w * sdx.w));
      } else {
        o = make_float4(dv[k].x * g.x * rstd.x, dv[k].y * g.y * rstd.y, dv[k].z * g.z * rstd.z, dv[k].w * g.w * rstd.w);
      }
      *reinterpret_cast<float4*>(a.dx + (long)r * a.lddx + c) = o;
    }
  }
}
static inline bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace vln

// ---------------------------------------------------------------------------------------------------------------
// The sampled-action branch of the rollouts (envdrop.py:186-195) in ONE launch: probs = softmax(mask(logits)),
// a ~ Categorical(probs) (drawn with the kernels' Philox stream when no action is given), log pi(a) and the entropy with
// torch.distributions' clamp, one thread per episode (<= 64 candidates).  Backward (one launch):
//   d logits_j = dlogp (1[j = a] - p_j) - dent p_j (log p_j + H)        (clamped / masked slots carry no gradient)
// ---------------------------------------------------------------------------------------------------------------
namespace vln {
// One WAVE per episode, lane c = candidate c (C <= 64): the row is read once, max / sum / entropy are wave reductions, the
// inverse-CDF draw is a 6-step inclusive scan + ballot.  (One thread per episode walked the row four times: ~32 dependent
// loads, 9.6 us per launch on a sampled rollout's critical path.)
__global__ __launch_bounds__(256) void categorical_fwd_kernel(const float* logits, long ld, const unsigned char* mask,
                                                              const long long* action_in, long long* action_out, float* probs,
                                                              float* logp, float* ent, int B, int C, uint64_t seed, uint64_t offset,
                                                              const uint64_t* offset_base_dev) {
  if (offset_base_dev) offset += *offset_base_dev * 8ull;       // device clock (vln_tick): the launch arguments repeat, the draws do not
  const float eps = 1.1920928955078125e-07f;
  const int lane = threadIdx.x & 63;
  for (int b = blockIdx.x * 4 + (threadIdx.x >> 6); b < B; b += gridDim.x * 4) {
    const bool in = lane < C;
    const bool masked = in && mask && mask[(long)b * C + lane];
    const float l = (in && !masked) ? logits[(long)b * ld + lane] : -INFINITY;
    const float mx = wave_max(l);
    const float e = (in && !masked) ? __expf(l - mx) : 0.f;
    const float inv = 1.f / wave_sum(e);
    const float pc = e * inv;
    long a;
    if (action_in) a = action_in[b];
    else {                                          // inverse-CDF draw from the row's own Philox word
      const Philox4 r = philox4x32_10(seed, offset, (uint32_t)b);
      const float u = (float)(r.x >> 8) * (1.0f / 16777216.0f);
      float cum = pc;                               // inclusive prefix sum over the lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(cum, o, 64);
        if (lane >= o) cum += t;
      }
      const unsigned long long hit = __ballot(in && u < cum), live = __ballot(in && pc > 0.f);
      a = hit ? (long)(__ffsll((long long)hit) - 1) : (live ? (long)(63 - __clzll((long long)live)) : 0);
    }
    const float lc = __logf(fminf(fmaxf(pc, eps), 1.f - eps));
    const float H = -wave_sum(in ? pc * lc : 0.f);
    const float la = (a >= 0 && a < C) ? __shfl(lc, (int)a, 64) : 0.f;      // wave-uniform `a`
    if (in) probs[(long)b * C + lane] = pc;
    if (lane == 0) {
      if (action_out) action_out[b] = a;
      logp[b] = la;
      ent[b] = H;
    }
  }
}
__global__ __launch_bounds__(256) void categorical_bwd_kernel(const float* probs, const long long* action, const float* dlogp,
                                                              const float* dent, float* dlogits, int B, int C) {
  const float eps = 1.1920928955078125e-07f;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
    const float* p = probs + (long)b * C;
    const long a = action[b];
    const float gl = dlogp ? dlogp[b] : 0.f, ge = dent ? dent[b] : 0.f;
    // clamp_probs passes no gradient where it clamps: only slots with eps <= p <= 1 - eps count in the p-gradients
    float gp_dot = 0.f;                              // sum_j (dL/dp_j) p_j over the live slots
    for (int c = 0; c < C; ++c) {
      const float pc = p[c];
      const bool live = pc >= eps && pc <= 1.f - eps;
      float gp = 0.f;                                // dL/dp_c
      if (live) gp = ((c == a) ? gl / pc : 0.f) - ge * (__logf(pc) + 1.f);
      else gp = -ge * __logf(fminf(fmaxf(pc, eps), 1.f - eps));      // H = -sum p log(clamp p): the p factor still differentiates
      gp_dot += gp * pc;
    }
    for (int c = 0; c < C; ++c) {
      const float pc = p[c];
      const bool live = pc >= eps && pc <= 1.f - eps;
      float gp = 0.f;
      if (live) gp = ((c == a) ? gl / pc : 0.f) - ge * (__logf(pc) + 1.f);
      else gp = -ge * __logf(fminf(fmaxf(pc, eps), 1.f - eps));
      dlogits[(long)b * C + c] = pc * (gp - gp_dot);   // softmax backward; masked slots have p = 0 -> 0
    }
  }
}
// every step of a sampled rollout in one launch: blockIdx.y = step, upstream gradients [T,B] (row t = step t)
struct CatMultiArgs {
  const float* probs[VLN_CE_MAX_STEPS]; const long long* action[VLN_CE_MAX_STEPS]; float* dlogits[VLN_CE_MAX_STEPS];
  int C[VLN_CE_MAX_STEPS]; int T, B; const float* dlogp; const float* dent;
};
__global__ __launch_bounds__(256) void categorical_multi_bwd_kernel(CatMultiArgs m) {
  const int t = blockIdx.y;
  const int C = m.C[t];
  const float eps = 1.1920928955078125e-07f;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < m.B; b += gridDim.x * blockDim.x) {
    const float* p = m.probs[t] + (long)b * C;
    const long a = m.action[t][b];
    const float gl = m.dlogp ? m.dlogp[(long)t * m.B + b] : 0.f, ge = m.dent ? m.dent[(long)t * m.B + b] : 0.f;
    float gp_dot = 0.f;                              // same arithmetic, same order as categorical_bwd_kernel
    for (int c = 0; c < C; ++c) {
      const float pc = p[c];
      const bool live = pc >= eps && pc <= 1.f - eps;
      float gp = 0.f;
      if (live) gp = ((c == a) ? gl / pc : 0.f) - ge * (__logf(pc) + 1.f);
      else gp = -ge * __logf(fminf(fmaxf(pc, eps), 1.f - eps));
      gp_dot += gp * pc;
    }
    float* dl = m.dlogits[t] + (long)b * C;
    for (int c = 0; c < C; ++c) {
      const float pc = p[c];
      const bool live = pc >= eps && pc <= 1.f - eps;
      float gp = 0.f;
      if (live) gp = ((c == a) ? gl / pc : 0.f) - ge * (__logf(pc) + 1.f);
      else gp = -ge * __logf(fminf(fmaxf(pc, eps), 1.f - eps));
      dl[c] = pc * (gp - gp_dot);
    }
  }
}
}  // namespace vln

extern "C" int vln_categorical_multi_bwd(const vln_cat_step* steps, int T, int B, const float* dlogp, const float* dent, void* s) {
  if (!steps || T <= 0 || T > VLN_CE_MAX_STEPS || B <= 0 || (!dlogp && !dent)) { vln::set_error("vln_categorical_multi_bwd: bad args"); return VLN_ERR_ARG; }
  vln::CatMultiArgs m{};
  m.T = T; m.B = B; m.dlogp = dlogp; m.dent = dent;
  for (int t = 0; t < T; ++t) {
    if (!steps[t].probs || !steps[t].action || !steps[t].dlogits || steps[t].C <= 0) { vln::set_error("vln_categorical_multi_bwd: null step pointer"); return VLN_ERR_ARG; }
    m.probs[t] = steps[t].probs; m.action[t] = (const long long*)steps[t].action; m.dlogits[t] = steps[t].dlogits; m.C[t] = steps[t].C;
  }
  VLN_LAUNCH(vln::categorical_multi_bwd_kernel, dim3((B + 255) / 256, T), dim3(256), 0, (hipStream_t)s, m);
  VLN_CHECK_LAUNCH("categorical_multi_bwd");
  return VLN_OK;
}

extern "C" int vln_categorical_fwd(const float* logits, int64_t ld, const uint8_t* cand_mask, const int64_t* action_in,
                                   int64_t* action_out, float* probs, float* logp, float* entropy, int B, int C, uint64_t seed,
                                   uint64_t offset, const uint64_t* offset_base_dev, void* s) {
  if (!logits || !probs || !logp || !entropy || B <= 0 || C <= 0 || (!action_in && !action_out)) {
    vln::set_error("vln_categorical_fwd: bad args");
    return VLN_ERR_ARG;
  }
  if (C > 64) { vln::set_error("vln_categorical_fwd: at most 64 candidates"); return VLN_ERR_ARG; }
  VLN_LAUNCH(vln::categorical_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)s, logits, (long)ld, cand_mask,
                     (const long long*)action_in, (long long*)action_out, probs, logp, entropy, B, C, seed, offset, offset_base_dev);
  VLN_CHECK_LAUNCH("categorical_fwd");
  return VLN_OK;
}
extern "C" int vln_categorical_bwd(const float* probs, const int64_t* action, const float* dlogp, const float* dent, float* dlogits,
                                   int B, int C, void* s) {
  if (!probs || !action || !dlogits || B <= 0 || C <= 0) { vln::set_error("vln_categorical_bwd: bad args"); return VLN_ERR_ARG; }
  VLN_LAUNCH(vln::categorical_bwd_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)s, probs,
                     (const long long*)action, dlogp, dent, dlogits, B, C);
  VLN_CHECK_LAUNCH("categorical_bwd");
  return VLN_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Pieces of the Self-Monitor decoder step (policy.py:119-166) that are not GEMMs or attention rows:
//   pe_dropout      : positioned context = dropout(ctx + pe[:L])                       (units.py:188-207)
//   monitor_head    : h_pm = dropout(sigmoid(W_m[h_0; w_cands]) * tanh(c_1)); progress = tanh(w_c . [ctx_attn; h_pm] + b_c)
//                     forward and backward, one workgroup per episode                  (policy.py:119-130)
//   add_n           : out (+)= sum of up to four strided matrices (gradient contributions of one tensor)
// ---------------------------------------------------------------------------------------------------------------
namespace vln {
__global__ __launch_bounds__(256) void pe_dropout_kernel(const float* ctx, const float* pe, float* out, int B, int L, int H, DropSpec dr) {
  const long total4 = (long)B * L * H / 4;                // H % 4 == 0
  const long lh4 = (long)L * H / 4;
  for (long e4 = (long)blockIdx.x * blockDim.x + threadIdx.x; e4 < total4; e4 += (long)gridDim.x * blockDim.x) {
    const float4 x = *reinterpret_cast<const float4*>(ctx + e4 * 4);
    const float4 q = *reinterpret_cast<const float4*>(pe + (e4 % lh4) * 4);
    float m[4] = {1.f, 1.f, 1.f, 1.f};
    if (dr.p > 0.f) dropout_scale4(dr.seed, dr.off(), (uint32_t)e4, dr.p, m);
    *reinterpret_cast<float4*>(out + e4 * 4) = make_float4((x.x + q.x) * m[0], (x.y + q.y) * m[1], (x.z + q.z) * m[2], (x.w + q.w) * m[3]);
  }
}

struct MonHeadArgs {
  const float* mg; const float* c1; const float* word_w; const float* wc; const float* bc;   // wc [L+H], bc [1]
  float* mem; float* prog;                        // fwd out: mem [B,H] (dropped gate product), prog [B]
  int B, L, H; DropSpec dr;
  // backward
  const float* dprog; const float* dc1_ext; const float* dww_ext;      // [B], [B,H] nullable, [B,L] nullable
  float* dmg; float* dc1; float* dww; float* Z; float* dpre;           // [B,H], [B,H], [B,L], [B,L+H] (rows dpre*[word_w|mem]), [B]
  SlabVec mgs; float* mg_out;      // forward, mgs.p != null: the gate product still in split-K slabs (+ bias); summed here and written to mg_out
};
__global__ __launch_bounds__(256) void monitor_head_fwd_kernel(MonHeadArgs a) {
  __shared__ float part[4];
  const int b = blockIdx.x, H = a.H, L = a.L;
  float acc = 0.f;
  for (int j = threadIdx.x; j < H; j += 256) {
    const long i = (long)b * H + j;
    float gate;
    if (a.mgs.p) { gate = a.mgs.at(b, j); a.mg_out[i] = gate; }
    else gate = a.mg[i];
    const float m = sigmoidf_(gate) * tanhf(a.c1[i]) * dropout_scale1(a.dr.seed, a.dr.off(), (uint32_t)i, a.dr.p);
    a.mem[i] = m;
    acc += a.wc[L + j] * m;
  }
  for (int l = threadIdx.x; l < L; l += 256) acc += a.wc[l] * a.word_w[(long)b * L + l];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) a.prog[b] = tanhf((part[0] + part[1]) + (part[2] + part[3]) + a.bc[0]);
}
__global__ __launch_bounds__(256) void monitor_head_bwd_kernel(MonHeadArgs a) {
  const int b = blockIdx.x, H = a.H, L = a.L;
  const float pv = a.prog[b];
  const float dpre = (a.dprog ? a.dprog[b] : 0.f) * (1.f - pv * pv);
  if (threadIdx.x == 0) a.dpre[b] = dpre;
  for (int j = threadIdx.x; j < H; j += 256) {
    const long i = (long)b * H + j;
    const float sg = sigmoidf_(a.mg[i]), tc = tanhf(a.c1[i]);
    const float dm = dpre * a.wc[L + j] * dropout_scale1(a.dr.seed, a.dr.off(), (uint32_t)i, a.dr.p);
    a.dmg[i] = dm * tc * sg * (1.f - sg);
    a.dc1[i] = dm * sg * (1.f - tc * tc) + (a.dc1_ext ? a.dc1_ext[i] : 0.f);
    a.Z[(long)b * (L + H) + L + j] = dpre * a.mem[i];
  }
  for (int l = threadIdx.x; l < L; l += 256) {
    const long i = (long)b * L + l;
    a.dww[i] = dpre * a.wc[l] + (a.dww_ext ? a.dww_ext[i] : 0.f);
    a.Z[(long)b * (L + H) + l] = dpre * a.word_w[i];
  }
}

struct AddNArgs { const float* src[4]; long ld[4]; int n; float* out; long ldo; int rows, cols, accumulate; };
__device__ __forceinline__ void add_n_body(const AddNArgs& a) {
  const long total = (long)a.rows * a.cols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / a.cols, c = e % a.cols;
    float v = a.accumulate ? a.out[r * a.ldo + c] : 0.f;
    for (int i = 0; i < a.n; ++i) v += a.src[i][r * a.ld[i] + c];
    a.out[r * a.ldo + c] = v;
  }
}
__global__ __launch_bounds__(256) void add_n_kernel(AddNArgs a) { add_n_body(a); }
// up to 4 independent sums in ONE launch (blockIdx.y = the job; a one-source job is a strided copy): same bits as 4 launches
struct AddNMulti { AddNArgs j[4]; };
__global__ __launch_bounds__(256) void add_n_multi_kernel(AddNMulti m) { add_n_body(m.j[blockIdx.y]); }
// The same sums with sources that still lie in split-K slabs (SlabVec): each source's slabs are added in slab order first, then the
// sources in their order -- the bits of "reduce each product, then add_n" without the reduce launches.
// drop (p > 0): the sum is multiplied by a dropout mask whose element index is r * drop_cols + drop_col0 + c -- a column block of
// a wider dropped row (the Follower's drop([a_prev | pano]), policy.py:49-51)
struct AddNSvArgs { SlabVec src[4]; int n; float* out; long ldo; int rows, cols, vec; DropSpec drop; int drop_cols, drop_col0; };
struct AddNSvMulti { AddNSvArgs j[4]; };
__global__ __launch_bounds__(256) void add_n_sv_multi_kernel(AddNSvMulti m) {
  const AddNSvArgs& a = m.j[blockIdx.y];
  if (a.vec) {
    const int c4 = a.cols >> 2;
    const long total = (long)a.rows * c4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
      const long r = e / c4, c = (e % c4) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int i = 0; i < a.n; ++i) { const float4 t = a.src[i].at4(r, c); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
      if (a.drop.p > 0.f) {
        float m[4];
        dropout_scale4(a.drop.seed, a.drop.off(), (uint32_t)((r * a.drop_cols + a.drop_col0 + c) >> 2), a.drop.p, m);
        v.x *= m[0]; v.y *= m[1]; v.z *= m[2]; v.w *= m[3];
      }
      *reinterpret_cast<float4*>(a.out + r * a.ldo + c) = v;
    }
    return;
  }
  const long total = (long)a.rows * a.cols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / a.cols, c = e % a.cols;
    float v = 0.f;
    for (int i = 0; i < a.n; ++i) v += a.src[i].at(r, c);
    if (a.drop.p > 0.f) v *= dropout_scale1(a.drop.seed, a.drop.off(), (uint32_t)(r * a.drop_cols + a.drop_col0 + c), a.drop.p);
    a.out[r * a.ldo + c] = v;
  }
}
}  // namespace vln
int vln::add_n_multi(hipStream_t st, const AddNJob* jobs, int n) {
  if (!jobs || n < 1 || n > 4) { set_error("add_n_multi: 1..4 jobs"); return VLN_ERR_ARG; }
  AddNMulti m{};
  long most = 0;
  for (int i = 0; i < n; ++i) {
    const AddNJob& q = jobs[i];
    if (!q.out || q.rows <= 0 || q.cols <= 0 || q.n < 1 || q.n > 4) { set_error("add_n_multi: bad job %d", i); return VLN_ERR_ARG; }
    AddNArgs& a = m.j[i];
    for (int k = 0; k < 4; ++k) { a.src[k] = k < q.n ? q.src[k] : nullptr; a.ld[k] = k < q.n ? q.ld[k] : 0; if (k < q.n && !q.src[k]) { set_error("add_n_multi: null source"); return VLN_ERR_ARG; } }
    a.n = q.n; a.out = q.out; a.ldo = q.ldo; a.rows = q.rows; a.cols = q.cols; a.accumulate = 0;
    const long t = (long)q.rows * q.cols;
    if (t > most) most = t;
  }
  int blocks = (int)((most + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  VLN_LAUNCH(add_n_multi_kernel, dim3(blocks, n), dim3(256), 0, st, m);
  VLN_CHECK_LAUNCH("add_n_multi");
  return VLN_OK;
}

int vln::add_n_sv_multi(hipStream_t st, const AddNSvJob* jobs, int n) {
  if (!jobs || n < 1 || n > 4) { set_error("add_n_sv_multi: 1..4 jobs"); return VLN_ERR_ARG; }
  AddNSvMulti m{};
  long most = 0;
  for (int i = 0; i < n; ++i) {
    const AddNSvJob& q = jobs[i];
    if (!q.out || q.rows <= 0 || q.cols <= 0 || q.n < 1 || q.n > 4) { set_error("add_n_sv_multi: bad job %d", i); return VLN_ERR_ARG; }
    AddNSvArgs& a = m.j[i];
    bool vec = (q.cols & 3) == 0 && (q.ldo & 3) == 0 && al16p(q.out);
    for (int k = 0; k < q.n; ++k) {
      const SlabVec& v = q.src[k];
      if (!v.p || v.n < 1) { set_error("add_n_sv_multi: null source"); return VLN_ERR_ARG; }
      vec = vec && al16p(v.p) && (v.ld & 3) == 0 && (v.stride & 3) == 0 && al16p(v.bias);
      a.src[k] = v;
    }
    if (q.drop.p > 0.f && ((q.drop_cols & 3) || (q.drop_col0 & 3))) vec = false;
    a.n = q.n; a.out = q.out; a.ldo = q.ldo; a.rows = q.rows; a.cols = q.cols; a.vec = vec ? 1 : 0;
    a.drop = q.drop; a.drop_cols = q.drop_cols; a.drop_col0 = q.drop_col0;
    const long t = (long)q.rows * (vec ? q.cols / 4 : q.cols);
    if (t > most) most = t;
  }
  int blocks = (int)((most + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  VLN_LAUNCH(add_n_sv_multi_kernel, dim3(blocks, n), dim3(256), 0, st, m);
  VLN_CHECK_LAUNCH("add_n_sv_multi");
  return VLN_OK;
}

extern "C" int vln_pe_dropout(const float* ctx, const float* pe, float* out, int B, int L, int H, uint64_t seed, uint64_t offset,
                              float p, void* s) {
  if (!ctx || !pe || !out || B <= 0 || L <= 0 || H <= 0 || (H & 3)) { vln::set_error("vln_pe_dropout: bad args (H %% 4 == 0)"); return VLN_ERR_ARG; }
  long t4 = (long)B * L * H / 4;
  int blocks = (int)((t4 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  VLN_LAUNCH(vln::pe_dropout_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, ctx, pe, out, B, L, H, vln::tls_drop(seed, offset, p));
  VLN_CHECK_LAUNCH("pe_dropout");
  return VLN_OK;
}
extern "C" int vln_monitor_head_fwd(const float* mg, const float* c1, const float* word_w, const float* wc, const float* bc, float* mem,
                                    float* prog, int B, int L, int H, uint64_t seed, uint64_t offset, float p, void* s) {
  if (!mg || !c1 || !word_w || !wc || !bc || !mem || !prog || B <= 0) { vln::set_error("vln_monitor_head_fwd: bad args"); return VLN_ERR_ARG; }
  vln::MonHeadArgs a{};
  a.mg = mg; a.c1 = c1; a.word_w = word_w; a.wc = wc; a.bc = bc; a.mem = mem; a.prog = prog; a.B = B; a.L = L; a.H = H;
  a.dr = vln::tls_drop(seed, offset, p);
  VLN_LAUNCH(vln::monitor_head_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)s, a);
  VLN_CHECK_LAUNCH("monitor_head_fwd");
  return VLN_OK;
}
int vln::monitor_head_fwd_sv(hipStream_t st, SlabVec mg, float* mg_out, const float* c1, const float* word_w, const float* wc, const float* bc,
                             float* mem, float* prog, int B, int L, int H, uint64_t seed, uint64_t offset, float p) {
  if (!mg.p || !mg_out || !c1 || !word_w || !wc || !bc || !mem || !prog || B <= 0) { set_error("monitor_head_fwd_sv: bad args"); return VLN_ERR_ARG; }
  MonHeadArgs a{};
  a.mgs = mg; a.mg_out = mg_out; a.c1 = c1; a.word_w = word_w; a.wc = wc; a.bc = bc; a.mem = mem; a.prog = prog; a.B = B; a.L = L; a.H = H;
  a.dr = tls_drop(seed, offset, p);
  VLN_LAUNCH(monitor_head_fwd_kernel, dim3(B), dim3(256), 0, st, a);
  VLN_CHECK_LAUNCH("monitor_head_fwd");
  return VLN_OK;
}
extern "C" int vln_monitor_head_bwd(const float* mg, const float* c1, const float* word_w, const float* wc, const float* mem,
                                    const float* prog, const float* dprog, const float* dc1_ext, const float* dww_ext, float* dmg,
                                    float* dc1, float* dww, float* Z, float* dpre, int B, int L, int H, uint64_t seed, uint64_t offset,
                                    float p, void* s) {
  if (!mg || !c1 || !word_w || !wc || !mem || !prog || !dmg || !dc1 || !dww || !Z || !dpre || B <= 0) {
    vln::set_error("vln_monitor_head_bwd: bad args");
    return VLN_ERR_ARG;
  }
  vln::MonHeadArgs a{};
  a.mg = mg; a.c1 = c1; a.word_w = word_w; a.wc = wc; a.mem = const_cast<float*>(mem); a.prog = const_cast<float*>(prog);
  a.B = B; a.L = L; a.H = H; a.dr = vln::tls_drop(seed, offset, p);
  a.dprog = dprog; a.dc1_ext = dc1_ext; a.dww_ext = dww_ext; a.dmg = dmg; a.dc1 = dc1; a.dww = dww; a.Z = Z; a.dpre = dpre;
  VLN_LAUNCH(vln::monitor_head_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)s, a);
  VLN_CHECK_LAUNCH("monitor_head_bwd");
  return VLN_OK;
}
extern "C" int vln_add_n(float* out, int64_t ldo, int rows, int cols, const float* s0, int64_t ld0, const float* s1, int64_t ld1,
                         const float* s2, int64_t ld2, const float* s3, int64_t ld3, int accumulate, void* s) {
  if (!out || rows <= 0 || cols <= 0 || !s0) { vln::set_error("vln_add_n: bad args"); return VLN_ERR_ARG; }
  vln::AddNArgs a{};
  const float* src[4] = {s0, s1, s2, s3}; const int64_t ld[4] = {ld0, ld1, ld2, ld3};
  a.n = 0;
  for (int i = 0; i < 4; ++i) if (src[i]) { a.src[a.n] = src[i]; a.ld[a.n] = (long)ld[i]; a.n++; }
  a.out = out; a.ldo = (long)ldo; a.rows = rows; a.cols = cols; a.accumulate = accumulate;
  long total = (long)rows * cols;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  VLN_LAUNCH(vln::add_n_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, a);
  VLN_CHECK_LAUNCH("add_n");
  return VLN_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Small row-wise elementwise forms of the Speaker-Follower step (ActionScoring, units.py:163-185; tanh backward):
//   VLN_EW_MUL         y[r,c] = a[r,c] * b[r*ldb + c]            (ldb = 0: b is one row vector, e.g. linear_out.weight)
//   VLN_EW_ADD_SCALAR  y[r,c] = a[r,c] + b[0]                    (linear_out.bias)
//   VLN_EW_TANH_GRAD   y[r,c] = a[r,c] * (1 - b[r,c]^2)          (b = tanh output)
//   VLN_EW_MUL_ROWSUM  y[r,c] = a[r,c] * sum_{j<nb} b[r*ldb + j] (a = NULL: 1, i.e. the row sums themselves)
// One workgroup per row (rows <= a few hundred, cols <= a few thousand): the row sum is a block reduction in a fixed order.
// ---------------------------------------------------------------------------------------------------------------
namespace vln {
struct EwArgs { const float* a; long lda; const float* b; long ldb; int nb; float* y; long ldy; int rows, cols, op; };
__device__ __forceinline__ void ew_row(const EwArgs& e, int r);
__global__ __launch_bounds__(256) void ew_kernel(EwArgs e) { ew_row(e, (int)blockIdx.x); }
// up to 4 independent row-wise forms in ONE launch (blockIdx.y = the job): same bits as 4 launches
struct EwMulti { EwArgs j[4]; };
__global__ __launch_bounds__(256) void ew_multi_kernel(EwMulti m) {
  const EwArgs& e = m.j[blockIdx.y];
  if ((int)blockIdx.x < e.rows) ew_row(e, (int)blockIdx.x);
}
__device__ __forceinline__ void ew_row(const EwArgs& e, int r) {
  __shared__ float part[4];
  float rs = 0.f;
  if (e.op == 3) {
    float acc = 0.f;
    for (int j = threadIdx.x; j < e.nb; j += 256) acc += e.b[(long)r * e.ldb + j];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    rs = (part[0] + part[1]) + (part[2] + part[3]);
  }
  const float b0 = (e.op == 1) ? e.b[0] : 0.f;
  for (int c = threadIdx.x; c < e.cols; c += 256) {
    const float av = e.a ? e.a[(long)r * e.lda + c] : 1.f;
    float v;
    if (e.op == 0) v = av * e.b[(long)r * e.ldb + c];
    else if (e.op == 1) v = av + b0;
    else if (e.op == 2) { const float t = e.b[(long)r * e.ldb + c]; v = av * (1.f - t * t); }
    else v = av * rs;
    e.y[(long)r * e.ldy + c] = v;
  }
}
}  // namespace vln
int vln::ew_multi(hipStream_t st, const EwJob* jobs, int n) {
  if (!jobs || n < 1 || n > 4) { set_error("ew_multi: 1..4 jobs"); return VLN_ERR_ARG; }
  EwMulti m{};
  int rows = 0;
  for (int i = 0; i < n; ++i) {
    const EwJob& q = jobs[i];
    if (!q.y || !q.b || q.rows <= 0 || q.cols <= 0 || q.op < 0 || q.op > 3 || (!q.a && q.op != 3) || (q.op == 3 && q.nb <= 0)) {
      set_error("ew_multi: bad job %d", i);
      return VLN_ERR_ARG;
    }
    m.j[i] = EwArgs{q.a, q.lda, q.b, q.ldb, q.nb, q.y, q.ldy, q.rows, q.cols, q.op};
    if (q.rows > rows) rows = q.rows;
  }
  VLN_LAUNCH(ew_multi_kernel, dim3(rows, n), dim3(256), 0, st, m);
  VLN_CHECK_LAUNCH("ew_multi");
  return VLN_OK;
}
extern "C" int vln_ew(int op, const float* a, int64_t lda, const float* b, int64_t ldb, int nb, float* y, int64_t ldy, int rows,
                      int cols, void* s) {
  if (!y || !b || rows <= 0 || cols <= 0 || op < 0 || op > 3 || (!a && op != 3) || (op == 3 && nb <= 0)) {
    vln::set_error("vln_ew: bad args");
    return VLN_ERR_ARG;
  }
  vln::EwArgs e{a, (long)lda, b, (long)ldb, nb, y, (long)ldy, rows, cols, op};
  VLN_LAUNCH(vln::ew_kernel, dim3(rows), dim3(256), 0, (hipStream_t)s, e);
  VLN_CHECK_LAUNCH("ew");
  return VLN_OK;
}

extern "C" int vln_bn_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_rstd,
                          int R, int D, float eps, float momentum, int training, int relu, uint64_t seed, uint64_t offset,
                          float p_drop, const uint8_t* row_zero, float* ws, int64_t ws_floats, void* s) {
  using namespace vln;
  if (!x || !y || R <= 0 || D <= 0 || (D & 3) || (ldx & 3) || (ldy & 3) || !al16p(x) || !al16p(y) ||
      (!training && (!running_mean || !running_var)) || (training && (!save_mean || !save_rstd))) {
    set_error("vln_bn_fwd: bad args (D %% 4 == 0, 16-byte aligned rows)");
    return VLN_ERR_ARG;
  }
  return bn_fwd_seg(x, ldx, y, ldy, gamma, beta, running_mean, running_var, num_batches_tracked, save_mean, save_rstd, R, 0, 0, D, eps,
                    momentum, training, relu, seed, offset, 0, p_drop, row_zero, ws, ws_floats, s);
}
int vln::bn_fwd_seg(const float* x, int64_t ldx, float* y, int64_t ldy, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_rstd, int R, int R1, int64_t stat2, int D,
                    float eps, float momentum, int training, int relu, uint64_t seed, uint64_t offset, uint64_t offset2, float p_drop,
                    const uint8_t* row_zero, float* ws, int64_t ws_floats, void* s, const float* x2, int64_t ldx2) {
  if (R1 < 0 || R1 >= R) { set_error("bn_fwd: bad segment split"); return VLN_ERR_ARG; }
  if (x2 && (R1 <= 0 || (ldx2 & 3) || !al16p(x2))) { set_error("bn_fwd: a second input array needs two segments and 16-byte aligned rows"); return VLN_ERR_ARG; }
  BnArgs a{x, (long)ldx, y, (long)ldy, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, save_mean, save_rstd,
           R, D, eps, momentum, training, relu, tls_drop(seed, offset, p_drop), row_zero, R1, (long)stat2, tls_drop(seed, offset2, p_drop),
           x2, (long)ldx2};
  const int nchunk = bn_nchunk(R, R1);
  if (R1 > 0 && !(ws && al16p(ws) && ws_floats >= (int64_t)nchunk * 2 * D)) { set_error("bn_fwd: the two-segment form needs its workspace"); return VLN_ERR_ARG; }
  if ((R1 > 0 || R >= 512) && ((!training) || (ws && al16p(ws) && ws_floats >= (int64_t)nchunk * 2 * D))) {     // tall input: row-chunked form
    BnChunkWs w{ws, nchunk};
    dim3 grid((D + 15) / 16, nchunk);
    if (training) VLN_LAUNCH(bn_fwd_stats_kernel, grid, dim3(256), 0, (hipStream_t)s, a, w);
    VLN_LAUNCH(bn_fwd_apply_kernel, grid, dim3(256), 0, (hipStream_t)s, a, w);
    VLN_CHECK_LAUNCH("bn_fwd (chunked)");
    return VLN_OK;
  }
  VLN_LAUNCH(bn_fwd_kernel, dim3((D + 15) / 16), dim3(256), 0, (hipStream_t)s, a);
  VLN_CHECK_LAUNCH("bn_fwd");
  return VLN_OK;
}
extern "C" int vln_bn_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* gamma,
                          const float* mean, const float* rstd_or_var, float* dx, int64_t lddx, float* dgamma, float* dbeta, int R,
                          int D, float eps, int training, int relu, int accumulate, uint64_t seed, uint64_t offset, float p_drop,
                          const uint8_t* row_zero, float* ws, int64_t ws_floats, void* s) {
  using namespace vln;
  if (!x || !dy || !mean || !rstd_or_var || (relu && !y) || R <= 0 || D <= 0 || (D & 3) || (ldx & 3) || (lddy & 3) || (lddx & 3) ||
      (relu && (ldy & 3)) || !al16p(x) || !al16p(dy) || (dx && !al16p(dx)) ) {
    set_error("vln_bn_bwd: bad args");
    return VLN_ERR_ARG;
  }
  return bn_bwd_seg(x, ldx, dy, lddy, y, ldy, gamma, mean, rstd_or_var, dx, lddx, dgamma, dbeta, R, 0, 0, D, eps, training, relu, accumulate,
                    seed, offset, 0, p_drop, row_zero, ws, ws_floats, s);
}
int vln::bn_bwd_seg(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* gamma,
                    const float* mean, const float* rstd_or_var, float* dx, int64_t lddx, float* dgamma, float* dbeta, int R, int R1,
                    int64_t stat2, int D, float eps, int training, int relu, int accumulate, uint64_t seed, uint64_t offset, uint64_t offset2,
                    float p_drop, const uint8_t* row_zero, float* ws, int64_t ws_floats, void* s, const float* x2, int64_t ldx2) {
  if (R1 < 0 || R1 >= R) { set_error("bn_bwd: bad segment split"); return VLN_ERR_ARG; }
  if (x2 && (R1 <= 0 || (ldx2 & 3) || !al16p(x2))) { set_error("bn_bwd: a second input array needs two segments and 16-byte aligned rows"); return VLN_ERR_ARG; }
  BnBwdArgs a{x, (long)ldx, dy, (long)lddy, y, (long)ldy, gamma, mean, rstd_or_var, dx, (long)lddx, dgamma, dbeta, R, D, eps,
              training, relu, accumulate, tls_drop(seed, offset, p_drop), row_zero, R1, (long)stat2, tls_drop(seed, offset2, p_drop),
              x2, (long)ldx2};
  const int nchunk = bn_nchunk(R, R1);
  if (R1 > 0 && !(ws && al16p(ws) && ws_floats >= (int64_t)nchunk * 2 * D)) { set_error("bn_bwd: the two-segment form needs its workspace"); return VLN_ERR_ARG; }
  if ((R1 > 0 || R >= 512) && ws && al16p(ws) && ws_floats >= (int64_t)nchunk * 2 * D) {       // tall input: row-chunked form
    BnChunkWs w{ws, nchunk};
    VLN_LAUNCH(bn_bwd_chunk_kernel<0>, dim3((D + 15) / 16, nchunk), dim3(256), 0, (hipStream_t)s, a, w);
    VLN_LAUNCH(bn_bwd_chunk_kernel<1>, dim3((D + 15) / 16, dx ? nchunk : 1), dim3(256), 0, (hipStream_t)s, a, w);
    VLN_CHECK_LAUNCH("bn_bwd (chunked)");
    return VLN_OK;
  }
  VLN_LAUNCH(bn_bwd_kernel, dim3((D + 15) / 16), dim3(256), 0, (hipStream_t)s, a);
  VLN_CHECK_LAUNCH("bn_bwd");
  return VLN_OK;
}

extern "C" int vln_masked_ce_fwd(float* logits, int64_t ld, const int64_t* target, const uint8_t* cand_mask, float* loss,
                                 float* loss_sum, float* probs, const int64_t* action, float* logp, float* entropy, int B,
                                 int C, int64_t ignore_index, int write_mask, void* s) {
  if (!logits || B <= 0 || C <= 0) { vln::set_error("vln_masked_ce_fwd: bad args"); return VLN_ERR_ARG; }
  vln::CeArgs a{logits, (long)ld, (const long long*)target, cand_mask, loss, probs, (const long long*)action, logp, entropy,
                B, C, (long)ignore_index, write_mask};
  if (loss_sum) VLN_LAUNCH(vln::masked_ce_fwd_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, a, loss_sum, 0);
  else VLN_LAUNCH(vln::masked_ce_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)s, a);
  VLN_CHECK_LAUNCH("masked_ce_fwd");
  return VLN_OK;
}
extern "C" int vln_masked_ce_bwd(const float* probs, const int64_t* target, const float* dloss, int64_t dloss_stride,
                                 float* dlogits, int B, int C, int64_t ignore_index, void* s) {
  if (!probs || !target || !dloss || !dlogits || B <= 0 || C <= 0) { vln::set_error("vln_masked_ce_bwd: bad args"); return VLN_ERR_ARG; }
  long total = (long)B * C;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  VLN_LAUNCH(vln::masked_ce_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, probs, (const long long*)target,
                     dloss, (long)dloss_stride, dlogits, B, C, (long)ignore_index, (const float*)nullptr);
  VLN_CHECK_LAUNCH("masked_ce_bwd");
  return VLN_OK;
}
// reduction = "mean" in the same launch: mean_out[0] = mean over the rows with a target, mean_out[1] = 1 / their count
extern "C" int vln_masked_ce_mean_fwd(float* logits, int64_t ld, const int64_t* target, const uint8_t* cand_mask, float* mean_out,
                                      float* probs, int B, int C, int64_t ignore_index, float* rows_scratch, void* s) {
  if (!logits || !target || !mean_out || B <= 0 || C <= 0) { vln::set_error("vln_masked_ce_mean_fwd: bad args"); return VLN_ERR_ARG; }
  vln::CeArgs a{logits, (long)ld, (const long long*)target, cand_mask, nullptr, probs, nullptr, nullptr, nullptr, B, C, (long)ignore_index, 0};
  if (rows_scratch && (long)B * C > VLN_CE_MEAN_ONE_LAUNCH_MAX) {     // one workgroup would walk B rows of C logits serially
    a.loss = rows_scratch;
    VLN_LAUNCH(vln::masked_ce_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)s, a);
    VLN_LAUNCH(vln::masked_ce_mean_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, (const float*)rows_scratch, (const long long*)target, B,
               (long)ignore_index, mean_out);
  } else
  VLN_LAUNCH(vln::masked_ce_fwd_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, a, mean_out, 1);
  VLN_CHECK_LAUNCH("masked_ce_mean_fwd");
  return VLN_OK;
}
// d logits of the mean: dloss[0] * inv_count[0] * (p - onehot)
extern "C" int vln_masked_ce_mean_bwd(const float* probs, const int64_t* target, const float* dloss, const float* inv_count, float* dlogits,
                                      int B, int C, int64_t ignore_index, void* s) {
  if (!probs || !target || !dloss || !inv_count || !dlogits || B <= 0 || C <= 0) { vln::set_error("vln_masked_ce_mean_bwd: bad args"); return VLN_ERR_ARG; }
  long total = (long)B * C;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  VLN_LAUNCH(vln::masked_ce_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, probs, (const long long*)target, dloss, 0L, dlogits, B, C,
             (long)ignore_index, inv_count);
  VLN_CHECK_LAUNCH("masked_ce_mean_bwd");
  return VLN_OK;
}

extern "C" int vln_monitor_loss_fwd(float* logits, int64_t ld, const int64_t* target, const uint8_t* cand_mask, const float* progress,
                                    int64_t ldp, const float* start_dist, const float* cur_dist, const uint8_t* ended, int t,
                                    float lam, int per_sample, float* probs, float* prog_target, float* out, float* stats, int B,
                                    int C, int64_t ignore_index, void* s) {
  if (!logits || !target || !progress || !start_dist || !cur_dist || !ended || !probs || !prog_target || !out || !stats || B <= 0 ||
      C <= 0 || ld < C || t < 0) {
    vln::set_error("vln_monitor_loss_fwd: bad args");
    return VLN_ERR_ARG;
  }
  vln::MonLossArgs m{};
  m.ce = vln::CeArgs{logits, (long)ld, (const long long*)target, cand_mask, nullptr, probs, nullptr, nullptr, nullptr, B, C,
                     (long)ignore_index, 0};
  m.progress = progress; m.ldp = (long)ldp; m.start_dist = start_dist; m.cur_dist = cur_dist; m.ended = ended;
  m.prog_target = prog_target; m.out = out; m.stats = stats; m.t = t; m.lam = lam; m.per_sample = per_sample;
  VLN_LAUNCH(vln::monitor_loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, m);
  VLN_CHECK_LAUNCH("monitor_loss_fwd");
  return VLN_OK;
}
extern "C" int vln_monitor_loss_bwd(const float* probs, const int64_t* target, const float* progress, int64_t ldp,
                                    const float* prog_target, const float* stats, const float* dloss, int64_t dloss_stride, int t,
                                    float lam, int per_sample, float* dlogits, float* dprogress, int B, int C,
                                    int64_t ignore_index, void* s) {
  if (!probs || !target || !progress || !prog_target || !stats || !dloss || !dlogits || !dprogress || B <= 0 || C <= 0) {
    vln::set_error("vln_monitor_loss_bwd: bad args");
    return VLN_ERR_ARG;
  }
  vln::MonLossBwdArgs m{probs, (const long long*)target, progress, (long)ldp, prog_target, stats, dloss, (long)dloss_stride,
                        dlogits, dprogress, B, C, t, lam, per_sample, (long)ignore_index};
  int blocks = (int)(((long)B * (C + 1) + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  VLN_LAUNCH(vln::monitor_loss_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, m);
  VLN_CHECK_LAUNCH("monitor_loss_bwd");
  return VLN_OK;
}

static int mon_multi_fill(vln::MonMultiArgs& m, const vln_monitor_loss_step* steps, int T, int B, int t0, float lam, int64_t ignore_index,
                          float* stats, bool bwd, const char* who) {
  if (!steps || T <= 0 || T > VLN_MONITOR_LOSS_MAX_STEPS || B <= 0 || (long)T * B > vln::kMonMultiRowsMax || t0 < 0 || !stats) { vln::set_error(who); return VLN_ERR_ARG; }
  m.T = T; m.B = B; m.t0 = t0; m.lam = lam; m.ignore_index = (long)ignore_index; m.stats = stats;
  for (int t = 0; t < T; ++t) {
    const vln_monitor_loss_step& q = steps[t];
    if (!q.logits || !q.target || !q.probs || !q.progress || !q.start_dist || !q.cur_dist || !q.ended || !q.prog_target || q.C <= 0 || q.ld < q.C ||
        (bwd && (!q.dlogits || !q.dprogress))) { vln::set_error(who); return VLN_ERR_ARG; }
    m.logits[t] = q.logits; m.target[t] = (const long long*)q.target; m.mask[t] = q.cand_mask; m.probs[t] = q.probs;
    m.progress[t] = q.progress; m.start_dist[t] = q.start_dist; m.cur_dist[t] = q.cur_dist; m.ended[t] = q.ended;
    m.prog_target[t] = q.prog_target; m.dlogits[t] = q.dlogits; m.dprogress[t] = q.dprogress;
    m.C[t] = q.C; m.ld[t] = (int)q.ld; m.ldp[t] = (int)q.ldp;
  }
  return VLN_OK;
}
extern "C" int vln_monitor_loss_multi_fwd(const vln_monitor_loss_step* steps, int T, int B, int t0, float lam, int64_t ignore_index, float* out,
                                          float* stats, int accumulate, void* s) {
  static_assert(VLN_MONITOR_LOSS_MAX_STEPS == vln::kMonMultiMaxT, "header / kernel step capacity");
  vln::MonMultiArgs m{};
  if (!out) { vln::set_error("vln_monitor_loss_multi_fwd: bad args"); return VLN_ERR_ARG; }
  const int rc = mon_multi_fill(m, steps, T, B, t0, lam, ignore_index, stats, false, "vln_monitor_loss_multi_fwd: bad args");
  if (rc) return rc;
  m.out = out; m.accumulate = accumulate;
  VLN_LAUNCH(vln::monitor_loss_multi_fwd_kernel, dim3(1), dim3((long)T * B > 512 ? 1024 : ((long)T * B > 256 ? 512 : 256)), 0, (hipStream_t)s, m);
  VLN_CHECK_LAUNCH("monitor_loss_multi_fwd");
  return VLN_OK;
}
extern "C" int vln_monitor_loss_multi_bwd(const vln_monitor_loss_step* steps, int T, int B, int t0, float lam, int64_t ignore_index,
                                          const float* stats, const float* dloss, void* s) {
  vln::MonMultiArgs m{};
  if (!dloss) { vln::set_error("vln_monitor_loss_multi_bwd: bad args"); return VLN_ERR_ARG; }
  const int rc = mon_multi_fill(m, steps, T, B, t0, lam, ignore_index, const_cast<float*>(stats), true, "vln_monitor_loss_multi_bwd: bad args");
  if (rc) return rc;
  m.dloss = dloss;
  VLN_LAUNCH(vln::monitor_loss_multi_bwd_kernel, dim3(T), dim3(256), 0, (hipStream_t)s, m);
  VLN_CHECK_LAUNCH("monitor_loss_multi_bwd");
  return VLN_OK;
}

static int ce_multi_fill(vln::CeMultiArgs& m, const vln_ce_step* steps, int T, int B, int64_t ignore_index, bool bwd, const char* who) {
  if (!steps || T <= 0 || T > VLN_CE_MAX_STEPS || B <= 0) { vln::set_error(who); return VLN_ERR_ARG; }
  m.T = T; m.B = B; m.ignore_index = (long)ignore_index;
  for (int t = 0; t < T; ++t) {
    const vln_ce_step& q = steps[t];
    if (!q.logits || !q.target || !q.probs || q.C <= 0 || q.ld < q.C || (bwd && !q.dlogits)) { vln::set_error(who); return VLN_ERR_ARG; }
    m.logits[t] = q.logits; m.target[t] = (const long long*)q.target; m.mask[t] = q.cand_mask; m.probs[t] = q.probs;
    m.dlogits[t] = q.dlogits; m.C[t] = q.C; m.ld[t] = (int)q.ld;
  }
  return VLN_OK;
}
extern "C" int vln_masked_ce_multi_fwd(const vln_ce_step* steps, int T, int B, int64_t ignore_index, float scale, float* loss_sum,
                                       float* loss_rows, int accumulate, float* inv_counts, void* s) {
  vln::CeMultiArgs m{};
  if (!loss_sum == !loss_rows) { vln::set_error("vln_masked_ce_multi_fwd: exactly one of loss_sum / loss_rows"); return VLN_ERR_ARG; }
  const int rc = ce_multi_fill(m, steps, T, B, ignore_index, false, "vln_masked_ce_multi_fwd: bad args");
  if (rc) return rc;
  if (inv_counts) {
    if (!loss_sum || (long)T * B > vln::kCeMeanRowsMax) { vln::set_error("vln_masked_ce_multi_fwd: the mean per step needs loss_sum and T * B <= %d", vln::kCeMeanRowsMax); return VLN_ERR_ARG; }
    VLN_LAUNCH(vln::masked_ce_multi_mean_fwd_kernel, dim3(1), dim3((long)T * B > 512 ? 1024 : ((long)T * B > 256 ? 512 : 256)), 0, (hipStream_t)s, m,
               loss_sum, accumulate, scale, inv_counts);
  } else
  if (loss_sum)
    VLN_LAUNCH(vln::masked_ce_multi_fwd_kernel, dim3(1), dim3((long)T * B > 256 ? 512 : 256), 0, (hipStream_t)s, m, loss_sum, accumulate, scale);
  else
    VLN_LAUNCH(vln::masked_ce_multi_rows_kernel, dim3((B + 63) / 64), dim3(256), 0, (hipStream_t)s, m, loss_rows, accumulate,
                       scale);
  VLN_CHECK_LAUNCH("masked_ce_multi_fwd");
  return VLN_OK;
}
extern "C" int vln_masked_ce_multi_bwd(const vln_ce_step* steps, int T, int B, int64_t ignore_index, float scale, const float* dloss,
                                       int64_t dloss_stride, const float* inv_counts, void* s) {
  vln::CeMultiArgs m{};
  if (!dloss || (dloss_stride != 0 && dloss_stride != 1)) { vln::set_error("vln_masked_ce_multi_bwd: bad args"); return VLN_ERR_ARG; }
  const int rc = ce_multi_fill(m, steps, T, B, ignore_index, true, "vln_masked_ce_multi_bwd: bad args");
  if (rc) return rc;
  VLN_LAUNCH(vln::masked_ce_multi_bwd_kernel, dim3(T), dim3(256), 0, (hipStream_t)s, m, dloss, (int)dloss_stride, scale, inv_counts);
  VLN_CHECK_LAUNCH("masked_ce_multi_bwd");
  return VLN_OK;
}
