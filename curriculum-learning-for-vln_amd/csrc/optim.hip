// Fused clip-grad-norm + RMSprop step over FLAT fp32 buffers.
// Reference: engine/trainer.py:423-427 (EnvDrop): clip_grad_norm(encoder, 40); clip_grad_norm(decoder, 40);
// torch.optim.RMSprop(lr=1e-4) with torch defaults (alpha 0.99, eps 1e-8, no momentum, not centered).
// torch issues ~10 multi-tensor launches for this; here: one partial-sum launch + one update launch over the
// flat parameter / gradient / square-average buffers (the gradient buffer is dp.GradBucket's all-reduce bucket).
// Clip groups are contiguous element ranges; group norms are reduced deterministically (per-block partials).
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

__global__ __launch_bounds__(256) void opt_rmsprop_kernel(float* p, const float* g, float* sq, OptGroups gr,
                                                          const float* partial, float* norms_out, float lr, float alpha,
                                                          float eps, float max_norm, float grad_scale) {
  __shared__ float sh[4];
  __shared__ float s_coef;
  int grp = 0;
  while (grp + 1 < gr.ngroups && (int)blockIdx.x >= gr.blk0[grp + 1]) ++grp;
  // total norm of this block's group (every block re-reduces its group's partials: a few hundred floats)
  float s = 0.f;
  for (int b = gr.blk0[grp] + threadIdx.x; b < gr.blk0[grp + 1]; b += kOptBlock) s += partial[b];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) {
    const float norm = sqrtf(s) * grad_scale;
    float c = (max_norm > 0.f) ? max_norm / (norm + 1e-6f) : 1.f;       // torch.nn.utils.clip_grad_norm_
    s_coef = grad_scale * (c < 1.f ? c : 1.f);
    if (norms_out && (int)blockIdx.x == gr.blk0[grp]) norms_out[grp] = norm;
  }
  __syncthreads();
  const float coef = s_coef;
  const long base = gr.begin[grp] + (long)((int)blockIdx.x - gr.blk0[grp]) * kOptChunk;
  const long end = gr.begin[grp + 1];
#pragma unroll
  for (int i = 0; i < kOptPerThread / 4; ++i) {
    const long e = base + ((long)i * kOptBlock + threadIdx.x) * 4;
    if (e + 3 < end) {
      float4 gv = *reinterpret_cast<const float4*>(g + e);
      float4 sv = *reinterpret_cast<const float4*>(sq + e);
      float4 pv = *reinterpret_cast<const float4*>(p + e);
      float gg[4] = {gv.x * coef, gv.y * coef, gv.z * coef, gv.w * coef};
      float ss[4] = {sv.x, sv.y, sv.z, sv.w};
      float pp[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ss[k] = alpha * ss[k] + (1.f - alpha) * gg[k] * gg[k];
        pp[k] -= lr * gg[k] / (sqrtf(ss[k]) + eps);
      }
      *reinterpret_cast<float4*>(sq + e) = make_float4(ss[0], ss[1], ss[2], ss[3]);
      *reinterpret_cast<float4*>(p + e) = make_float4(pp[0], pp[1], pp[2], pp[3]);
    } else {
      for (long k = e; k < end && k < e + 4; ++k) {
        const float gk = g[k] * coef;
        const float sk = alpha * sq[k] + (1.f - alpha) * gk * gk;
        sq[k] = sk;
        p[k] -= lr * gk / (sqrtf(sk) + eps);
      }
    }
  }
}

}  // namespace vln

using namespace vln;

// group_begin: ngroups+1 element offsets (multiples of 4) into the flat buffers; partial: >= total blocks floats
extern "C" int64_t vln_rmsprop_partial_floats(const int64_t* group_begin, int ngroups) {
  long blocks = 0;
  for (int g = 0; g < ngroups; ++g) blocks += (group_begin[g + 1] - group_begin[g] + kOptChunk - 1) / kOptChunk;
  return blocks;
}
extern "C" int vln_rmsprop_clip_step(float* params, const float* grads, float* square_avg, const int64_t* group_begin,
                                     int ngroups, float* partial, float* norms_out, float lr, float alpha, float eps,
                                     float max_norm, float grad_scale, vln_stream_t s) {
  if (!params || !grads || !square_avg || !group_begin || !partial || ngroups < 1 || ngroups > 8) {
    set_error("vln_rmsprop_clip_step: bad args");
    return VLN_ERR_ARG;
  }
  OptGroups gr;
  gr.ngroups = ngroups;
  int blk = 0;
  for (int g = 0; g <= ngroups; ++g) {
    gr.begin[g] = group_begin[g];
    if (group_begin[g] % 4) { set_error("vln_rmsprop_clip_step: group offsets must be multiples of 4"); return VLN_ERR_ARG; }
    gr.blk0[g] = blk;
    if (g < ngroups) blk += (int)((group_begin[g + 1] - group_begin[g] + kOptChunk - 1) / kOptChunk);
  }
  if (blk <= 0) return VLN_OK;
  hipStream_t st = (hipStream_t)s;
  hipLaunchKernelGGL(opt_sumsq_kernel, dim3(blk), dim3(kOptBlock), 0, st, grads, gr, partial);
  hipLaunchKernelGGL(opt_rmsprop_kernel, dim3(blk), dim3(kOptBlock), 0, st, params, grads, square_avg, gr, partial, norms_out,
                     lr, alpha, eps, max_norm, grad_scale);
  VLN_CHECK_LAUNCH("rmsprop_clip_step");
  return VLN_OK;
}
