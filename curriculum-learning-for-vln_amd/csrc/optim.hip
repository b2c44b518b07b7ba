// Fused clip-grad-norm + optimizer step over FLAT fp32 buffers.
// Reference: engine/trainer.py:17-21 (optim_switcher adam / rms / sgd, torch defaults), :423-427 (EnvDrop):
// clip_grad_norm(encoder, 40); clip_grad_norm(decoder, 40); RMSprop(lr=1e-4) (alpha 0.99, eps 1e-8, no momentum, not
// centered); Follower :65-67 two Adam instances, Monitor :219-222 one Adam (betas 0.9/0.999, eps 1e-8).
// torch issues ~10 multi-tensor launches for this; here: one partial-sum launch + one update launch over the
// flat parameter / gradient / square-average buffers (the gradient buffer is dp.GradBucket's all-reduce bucket).
// Clip groups are contiguous element ranges; group norms are reduced deterministically (per-block partials).
#include <math.h>

#include "vln_internal.h"
#include "../../include/vln_hip.h"

namespace vln {

constexpr int kOptBlock = 256;
constexpr int kOptPerThread = 16;                    // 4 x float4
constexpr int kOptChunk = kOptBlock * kOptPerThread; // elements per block

struct OptGroups {   // up to 8 clip groups; group g covers elements [begin[g], begin[g+1])
  long begin[9];
  int blk0[9];       // first block of each group (blocks never straddle groups)
  int ngroups;
};

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float t = (threadIdx.x < 4) ? sh[threadIdx.x] : 0.f;
  t = wave_sum(t);
  __syncthreads();
  return t;   // valid in wave 0 (all lanes)
}

__global__ __launch_bounds__(256) void opt_sumsq_kernel(const float* g, OptGroups gr, float* partial) {
  __shared__ float sh[4];
  int grp = 0;
  while (grp + 1 < gr.ngroups && (int)blockIdx.x >= gr.blk0[grp + 1]) ++grp;
  const long base = gr.begin[grp] + (long)((int)blockIdx.x - gr.blk0[grp]) * kOptChunk;
  const long end = gr.begin[grp + 1];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kOptPerThread / 4; ++i) {
    const long e = base + ((long)i * kOptBlock + threadIdx.x) * 4;
    if (e + 3 < end) {
      const float4 v = *reinterpret_cast<const float4*>(g + e);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (long k = e; k < end && k < e + 4; ++k) s += g[k] * g[k];
    }
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

enum OptMode { OPT_RMSPROP = 0, OPT_ADAM = 1, OPT_SGD = 2 };
struct OptHyper {
  float lr, a, b, eps;          // rmsprop: a = alpha; adam: a = beta1, b = beta2
  float bc1, bc2_rsqrt;         // adam bias corrections: 1 - beta1^t, 1 / sqrt(1 - beta2^t)
  float grad_scale;
  int zero_grads;               // 1: the update also clears the gradient buffer it has just consumed (the next zero_grad() is free)
  float max_norm[8];              // per clip group; 0 = that group is not clipped (trainer.py:425-426 clips encoder and decoder, not the critic)
  // adam, optional: the step count t = *step_dev + step_add lives on the device (whole-iteration graphs: the launch arguments
  // must repeat); the bias corrections are then formed by the kernel instead of the host
  const long long* step_dev; long long step_add;
};

template <int MODE>
__device__ __forceinline__ void opt_update(float& p, float g, float& s1, float& s2, const OptHyper& h) {
  if (MODE == OPT_RMSPROP) {            // torch.optim.RMSprop defaults: no momentum, not centered
    s1 = h.a * s1 + (1.f - h.a) * g * g;
    p -= h.lr * g / (sqrtf(s1) + h.eps);
  } else if (MODE == OPT_ADAM) {        // torch.optim.Adam defaults: no weight decay, no amsgrad
    s1 = h.a * s1 + (1.f - h.a) * g;
    s2 = h.b * s2 + (1.f - h.b) * g * g;
    const float denom = sqrtf(s2) * h.bc2_rsqrt + h.eps;
    p -= (h.lr / h.bc1) * (s1 / denom);
  } else {                              // torch.optim.SGD defaults: plain
    p -= h.lr * g;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void opt_step_kernel(float* p, float* g, float* s1, float* s2, OptGroups gr,
                                                       const float* partial, float* norms_out, OptHyper h) {
  __shared__ float sh[4];
  __shared__ float s_coef, s_bc1, s_bc2r;
  int grp = 0;
  while (grp + 1 < gr.ngroups && (int)blockIdx.x >= gr.blk0[grp + 1]) ++grp;
  // total norm of this block's group (every block re-reduces its group's partials: a few hundred floats)
  float s = 0.f;
  for (int b = gr.blk0[grp] + threadIdx.x; b < gr.blk0[grp + 1]; b += kOptBlock) s += partial[b];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) {
    const float norm = sqrtf(s) * h.grad_scale;
    float c = (h.max_norm[grp] > 0.f) ? h.max_norm[grp] / (norm + 1e-6f) : 1.f;   // torch.nn.utils.clip_grad_norm_
    s_coef = h.grad_scale * (c < 1.f ? c : 1.f);
    if (MODE == OPT_ADAM && h.step_dev) {
      const double t = (double)(*h.step_dev + h.step_add);
      s_bc1 = (float)(1.0 - pow((double)h.a, t));
      s_bc2r = (float)(1.0 / sqrt(1.0 - pow((double)h.b, t)));
    }
    if (norms_out && (int)blockIdx.x == gr.blk0[grp]) norms_out[grp] = norm;
  }
  __syncthreads();
  const float coef = s_coef;
  if (MODE == OPT_ADAM && h.step_dev) { h.bc1 = s_bc1; h.bc2_rsqrt = s_bc2r; }
  const long base = gr.begin[grp] + (long)((int)blockIdx.x - gr.blk0[grp]) * kOptChunk;
  const long end = gr.begin[grp + 1];
#pragma unroll
  for (int i = 0; i < kOptPerThread / 4; ++i) {
    const long e = base + ((long)i * kOptBlock + threadIdx.x) * 4;
    if (e + 3 < end) {
      const float4 gv = *reinterpret_cast<const float4*>(g + e);
      float4 pv = *reinterpret_cast<const float4*>(p + e);
      float4 av = (MODE != OPT_SGD) ? *reinterpret_cast<const float4*>(s1 + e) : make_float4(0, 0, 0, 0);
      float4 bv = (MODE == OPT_ADAM) ? *reinterpret_cast<const float4*>(s2 + e) : make_float4(0, 0, 0, 0);
      opt_update<MODE>(pv.x, gv.x * coef, av.x, bv.x, h);
      opt_update<MODE>(pv.y, gv.y * coef, av.y, bv.y, h);
      opt_update<MODE>(pv.z, gv.z * coef, av.z, bv.z, h);
      opt_update<MODE>(pv.w, gv.w * coef, av.w, bv.w, h);
      if (MODE != OPT_SGD) *reinterpret_cast<float4*>(s1 + e) = av;
      if (MODE == OPT_ADAM) *reinterpret_cast<float4*>(s2 + e) = bv;
      *reinterpret_cast<float4*>(p + e) = pv;
      if (h.zero_grads) *reinterpret_cast<float4*>(g + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (long k = e; k < end && k < e + 4; ++k) {
        float pk = p[k], ak = (MODE != OPT_SGD) ? s1[k] : 0.f, bk = (MODE == OPT_ADAM) ? s2[k] : 0.f;
        opt_update<MODE>(pk, g[k] * coef, ak, bk, h);
        if (MODE != OPT_SGD) s1[k] = ak;
        if (MODE == OPT_ADAM) s2[k] = bk;
        p[k] = pk;
        if (h.zero_grads) g[k] = 0.f;
      }
    }
  }
}

static int opt_launch(int mode, float* params, float* grads, float* s1, float* s2, const int64_t* group_begin,
                      int ngroups, float* partial, float* norms_out, OptHyper h, const float* max_norms, hipStream_t st,
                      const char* what) {
  if (!params || !grads || !group_begin || !partial || ngroups < 1 || ngroups > 8 || (mode != OPT_SGD && !s1) ||
      (mode == OPT_ADAM && !s2)) {
    set_error("%s: bad args", what);
    return VLN_ERR_ARG;
  }
  for (int g = 0; g < 8; ++g) h.max_norm[g] = (max_norms && g < ngroups) ? max_norms[g] : 0.f;
  h.zero_grads = h.grad_scale < 0.f ? 1 : 0;       // grad_scale < 0: scale by |grad_scale| and clear the gradients afterwards
  h.grad_scale = fabsf(h.grad_scale);
  OptGroups gr;
  gr.ngroups = ngroups;
  int blk = 0;
  for (int g = 0; g <= ngroups; ++g) {
    gr.begin[g] = group_begin[g];
    if (group_begin[g] % 4) { set_error("%s: group offsets must be multiples of 4", what); return VLN_ERR_ARG; }
    gr.blk0[g] = blk;
    if (g < ngroups) blk += (int)((group_begin[g + 1] - group_begin[g] + kOptChunk - 1) / kOptChunk);
  }
  if (blk <= 0) return VLN_OK;
  VLN_LAUNCH(opt_sumsq_kernel, dim3(blk), dim3(kOptBlock), 0, st, grads, gr, partial);
  if (mode == OPT_RMSPROP)
    VLN_LAUNCH(opt_step_kernel<OPT_RMSPROP>, dim3(blk), dim3(kOptBlock), 0, st, params, grads, s1, s2, gr, partial, norms_out, h);
  else if (mode == OPT_ADAM)
    VLN_LAUNCH(opt_step_kernel<OPT_ADAM>, dim3(blk), dim3(kOptBlock), 0, st, params, grads, s1, s2, gr, partial, norms_out, h);
  else
    VLN_LAUNCH(opt_step_kernel<OPT_SGD>, dim3(blk), dim3(kOptBlock), 0, st, params, grads, s1, s2, gr, partial, norms_out, h);
  VLN_CHECK_LAUNCH(what);
  return VLN_OK;
}

}  // namespace vln

using namespace vln;

// group_begin: ngroups+1 element offsets (multiples of 4) into the flat buffers; partial: >= total blocks floats;
// max_norms: HOST array of ngroups clip norms (0 = group not clipped) or NULL (no clipping)
extern "C" int64_t vln_rmsprop_partial_floats(const int64_t* group_begin, int ngroups) {
  long blocks = 0;
  for (int g = 0; g < ngroups; ++g) blocks += (group_begin[g + 1] - group_begin[g] + kOptChunk - 1) / kOptChunk;
  return blocks;
}
extern "C" int vln_rmsprop_clip_step(float* params, float* grads, float* square_avg, const int64_t* group_begin,
                                     int ngroups, float* partial, float* norms_out, float lr, float alpha, float eps,
                                     const float* max_norms, float grad_scale, vln_stream_t s) {
  OptHyper h{lr, alpha, 0.f, eps, 1.f, 1.f, grad_scale, 0, {}, nullptr, 0};
  return opt_launch(OPT_RMSPROP, params, grads, square_avg, nullptr, group_begin, ngroups, partial, norms_out, h,
                    max_norms, (hipStream_t)s, "vln_rmsprop_clip_step");
}
extern "C" int vln_adam_clip_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                                  const int64_t* group_begin, int ngroups, float* partial, float* norms_out, float lr,
                                  float beta1, float beta2, float eps, int64_t step, const int64_t* step_dev, const float* max_norms,
                                  float grad_scale, vln_stream_t s) {
  if (!step_dev && step < 1) { set_error("vln_adam_clip_step: step counts from 1"); return VLN_ERR_ARG; }
  const double st = step_dev ? 1.0 : (double)step;
  const double bc1 = 1.0 - pow((double)beta1, st), bc2 = 1.0 - pow((double)beta2, st);
  OptHyper h{lr, beta1, beta2, eps, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale, 0, {}, reinterpret_cast<const long long*>(step_dev), (long long)step};
  return opt_launch(OPT_ADAM, params, grads, exp_avg, exp_avg_sq, group_begin, ngroups, partial, norms_out, h,
                    max_norms, (hipStream_t)s, "vln_adam_clip_step");
}
extern "C" int vln_sgd_clip_step(float* params, float* grads, const int64_t* group_begin, int ngroups, float* partial,
                                 float* norms_out, float lr, const float* max_norms, float grad_scale, vln_stream_t s) {
  OptHyper h{lr, 0.f, 0.f, 0.f, 1.f, 1.f, grad_scale, 0, {}, nullptr, 0};
  return opt_launch(OPT_SGD, params, grads, nullptr, nullptr, group_begin, ngroups, partial, norms_out, h, max_norms,
                    (hipStream_t)s, "vln_sgd_clip_step");
}
