// Skinny-M MFMA GEMMs for the decoder/encoder linears (gfx950).
//
//   gemm_nt : Y[M,N] = X[M,K] * W[N,K]^T       forward linears and, with the transposed weight shadow,
//                                               the dX products.  M is the episode batch (64..128) or
//                                               batch*candidates; the WEIGHT stream is the HBM-bound operand.
//   gemm_tn : D[N,K] (+)= A[Mt,N]^T * X[Mt,K]  deferred weight gradients, contraction over (steps x batch).
//
// Tiling (gemm_nt): one 256-thread workgroup = 4 waves computes a 64(M) x 64(N) tile over one K-chunk
// (cross-workgroup split-K fills the 256 CUs; partial slabs are summed by the consumer / reduce pass, which
// keeps results deterministic -- no float atomics).  Each wave owns 16 output columns: its W fragments go
// global->VGPR directly (each lane reads 32 contiguous bytes of one weight row: 16 rows x 128 B per wave
// instruction pair, whole cache lines), the X tile is shared by the 4 waves through LDS (144-B padded rows,
// ds_read_b128).  fp32 uses v_mfma_f32_16x16x4_f32 (exact fp32), bf16 uses v_mfma_f32_16x16x32_bf16 with
// fp32 accumulate; lane group q = lane>>4 owns k = k0 + q*VK .. +VK so both operands read 32 B per lane.
#include <type_traits>

#include "vln_internal.h"
#include "prologue_bodies.h"
#include "shadow_bodies.h"
#include "layout_bodies.h"
#include "step_bodies.h"
#include "../../include/vln_hip.h"

namespace vln {

#include "gemm_nt_body.h"
#include "gemm_rows.h"

template <typename TW, int PD, bool kFast, int NT = 1>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNTArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[gemm_nt_smem_bytes(GemmCfg<TW>::kPlanes)];
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd) {
    // Products with MANY row tiles (the encoder's two M = B * L products): the column tiles of one row tile read the same
    // [64, K] block of X.  Workgroups are dealt to the 8 XCDs round-robin in launch order, so in grid order those column tiles
    // sit on different XCDs and each L2 fetches the block for itself (d-embedding product: 175 MB of fabric traffic against
    // 48 MB algorithmic, profiles/round4_gemm_nt_by_shape.txt).  XCD x takes the contiguous range of tiles [x, x + 1) * T / 8
    // instead: the sharers of a block run together behind ONE L2.  (The host sets the flag only when T % 8 == 0.)
    const int L = bx + (int)gridDim.x * (by + (int)gridDim.y * bz), per = (int)(gridDim.x * gridDim.y * gridDim.z) >> 3;
    const int t = (L & 7) * per + (L >> 3);
    bx = t % (int)gridDim.x;
    const int r = t / (int)gridDim.x;
    by = r % (int)gridDim.y; bz = r / (int)gridDim.y;
  }
  const VBlock vb{bx, by, bz, (int)threadIdx.x, smem};
  gemm_nt_body<TW, PD, kFast, NT>(a, vb, true, gemm_nt_nsteps(a, vb.by, GemmCfg<TW>::BK), [] {});
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#include "gemm_nt_n16.h"

// narrow outputs: 16-column workgroups with the K split inside the workgroup (no slabs, no reduce launch)
// tunable[3] = largest K it takes (every workgroup streams the WHOLE X once: long contractions belong to the split-K path)
static bool n16_applies(int M, int N, int K, int wtype) {     // (W_F32S / W_F32X operands take the 64-column kernel)
  return wtype != W_F32S && wtype != W_F32X && g_tunable[2] && N <= 1024 && M <= 256 && K >= 64 && (g_tunable[3] <= 0 || K <= g_tunable[3]);
}

struct PostedLayout { bool on = false; LayoutArgs a; };
static thread_local PostedLayout g_posted_layout;
static int launch_n16(hipStream_t st, const float* X, long ldx, const void* W, int wtype, long ldw, float* Y, long ldy,
                      int M, int N, int K, const float* bias, int act, float* Y2, long ldy2, DropSpec drop) {
  const int BK = (wtype == W_BF16) ? 64 : 32;
  GemmN16Args a;
  a.X = X; a.ldx = ldx; a.W = W; a.ldw = ldw; a.Y = Y; a.ldy = ldy; a.bias = bias; a.act = act;
  a.Y2 = Y2; a.ldy2 = ldy2; a.drop = drop; a.M = M; a.N = N; a.K = K;
  a.kq = ((K + 3) / 4 + BK - 1) / BK * BK;
  a.xvec = aligned16(X) && (ldx % 4 == 0);
  a.wvec = aligned16(W) && (ldw % (wtype == W_BF16 ? 8 : 4) == 0);
  dim3 grid((N + 15) / 16, 1, (M + 63) / 64), block(256);
  const double bytes = (double)N * K * (wtype == W_BF16 ? 2 : 4) + 4.0 * M * K + 4.0 * M * N;
  if (g_posted_layout.on) {                          // a posted layout change rides behind the product's tiles
    const LayoutArgs lay = g_posted_layout.a;
    g_posted_layout.on = false;
    const int gx = (int)grid.x, gz = (int)grid.z;
    if (wtype == W_BF16) launch_timed(K_GEMM_NT, bytes, gemm_nt_n16_layout_kernel<bf16_raw>, dim3(gx * gz + lay.blocks), block, 0, st, a, gx, gz, lay);
    else launch_timed(K_GEMM_NT, bytes, gemm_nt_n16_layout_kernel<float>, dim3(gx * gz + lay.blocks), block, 0, st, a, gx, gz, lay);
    VLN_CHECK_LAUNCH("gemm_nt_n16 + layout");
    return VLN_OK;
  }
  if (wtype == W_BF16) launch_timed(K_GEMM_NT, bytes, gemm_nt_n16_kernel<bf16_raw>, grid, block, 0, st, a);
  else launch_timed(K_GEMM_NT, bytes, gemm_nt_n16_kernel<float>, grid, block, 0, st, a);
  VLN_CHECK_LAUNCH("gemm_nt_n16");
  return VLN_OK;
}
__global__ __launch_bounds__(256) void layout_posted_kernel(LayoutArgs lay) {
  if (lay.kind == 0) tm_to_bm_body(lay.src, lay.dst, lay.dst_lp, lay.B, lay.L, lay.W, lay.dr, lay.vec, (long)blockIdx.x, (long)gridDim.x);
  else bm_to_tm_body(lay.src, lay.dst, lay.B, lay.L, lay.W, lay.dr, lay.vec, (long)blockIdx.x, (long)gridDim.x);
}
int layout_post(int kind, const float* src, float* dst, void* dst_lp, int B, int L, int W, DropSpec dr) {
  if ((kind != 0 && kind != 1) || !src || !dst || B <= 0 || L <= 0 || W <= 0 || (kind == 1 && dst_lp)) { set_error("vln_layout_post: bad args"); return VLN_ERR_ARG; }
  if (g_posted_layout.on) { set_error("vln_layout_post: a posted layout change is still pending (vln_layout_post_flush)"); return VLN_ERR_ARG; }
  int vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(dst_lp)) & 15) == 0;
  if (vec && W % 8 == 0) vec = 2;
  const long n = (long)B * L * W / (vec == 2 ? 8 : vec ? 4 : 1);
  long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
  g_posted_layout.on = true;
  g_posted_layout.a = LayoutArgs{kind, src, dst, static_cast<bf16_raw*>(dst_lp), B, L, W, dr, vec, (int)blocks};
  return VLN_OK;
}
int layout_post_flush(hipStream_t st) {
  if (!g_posted_layout.on) return VLN_OK;
  g_posted_layout.on = false;
  VLN_LAUNCH(layout_posted_kernel, dim3((unsigned)g_posted_layout.a.blocks), dim3(256), 0, st, g_posted_layout.a);
  VLN_CHECK_LAUNCH("layout (posted)");
  return VLN_OK;
}

// How many K-chunks (= partial slabs) a product is split into: a pure function of its shape, the workspace and the tunables, so a
// caller that DEFERS the consumption of the slabs (envdrop.hip: the chained steps) can name them without issuing the product.
static int gemm_nt_split(int nb, int mb, int ksteps, int M, int N, bool have_ws, long ws_floats, bool to_caller, int* steps_per_out) {
  // split K until ~2 workgroups per CU are in flight; keep >= 2 K-steps per chunk.  (Measured on MI355X:
  // many co-resident workgroups with one K-step of prefetch each beat one fat workgroup per CU with all of
  // its loads in flight: 8.5 vs 12.5 us average per launch in the EnvDrop step, profiles/round1_notes.md.)
  int nsplit = 1;
  if (have_ws) {
    const int target = g_tunable[0];                 // workgroups wanted in flight (default 384)
    nsplit = target / (nb * mb);
    if (nsplit > ksteps / 2) nsplit = ksteps / 2;
    // wide-and-shallow products (N >= 2048 columns, K <= 8 steps: the H->F query projections) already fill 32+
    // workgroups; splitting them only buys a reduce launch
    if (g_tunable[1] && nb * mb >= 32 && ksteps <= 8 && !to_caller) nsplit = 1;
    if (nsplit < 1) nsplit = 1;
    long per = (long)M * N;
    if ((long)nsplit * per > ws_floats) nsplit = (int)(ws_floats / per);
    if (nsplit < 1) nsplit = 1;
  }
  int steps_per = (ksteps + nsplit - 1) / nsplit;
  nsplit = (ksteps + steps_per - 1) / steps_per;
  if (steps_per_out) *steps_per_out = steps_per;
  return nsplit;
}
// slabs gemm_nt(..., nsplit_out != nullptr) will leave for a [M,K] x [N,K]^T product of weight type `wtype` (64-column tiles)
int gemm_nt_slabs(int M, int N, int K, int wtype, long ws_floats) {
  const int BK = (wtype == W_F32) ? 32 : 64;
  return gemm_nt_split((N + 63) / 64, (M + 63) / 64, (K + BK - 1) / BK, M, N, true, ws_floats, true, nullptr);
}

// slabs the PLAIN call gemm_nt(..., nsplit_out = nullptr) would reduce with its own launch (1: it finishes in the product's launch)
int gemm_nt_plain_slabs(int M, int N, int K, int wtype, long ws_floats) {
  if (n16_applies(M, N, K, wtype)) return 1;
  const int BK = (wtype == W_F32) ? 32 : 64;
  return gemm_nt_split((N + 63) / 64, (M + 63) / 64, (K + BK - 1) / BK, M, N, true, ws_floats, false, nullptr);
}
// A product whose CONSUMER sums the split-K slabs (SlabVec, bias included): the slabs are left in the caller's bump-allocated area
// (several products' slabs live at once).  The K split is the one of products handed to a caller (gemm_nt_slabs): where the plain
// call splits too it is the same split -- the bits of the plain call + reduce_epilogue -- and the wide-and-shallow products the
// plain call leaves un-split (a reduce launch would cost more than the split buys) ARE split here, the summing being free.  A
// product the narrow-output kernel takes, or that does not fit the area, goes through the plain call into Y (*out = that matrix).
int gemm_nt_to_consumer(hipStream_t st, SlabArea& ar, const float* X, long ldx, const void* W, int wtype, long ldw, float* Y, long ldy,
                        int M, int N, int K, const float* bias, SlabVec* out) {
  if (!out || !ar.base) { set_error("gemm_nt_to_consumer: null pointer"); return VLN_ERR_ARG; }
  const long per = (long)M * N;
  if (g_tunable[9] || ar.left < per || n16_applies(M, N, K, wtype) || gemm_nt_slabs(M, N, K, wtype, ar.left) <= 1) {      // tunable[9] = 1: always the plain call (A/B)
    if (!Y) { set_error("gemm_nt_to_consumer: the product is not split and no output matrix was given"); return VLN_ERR_ARG; }
    const int r = gemm_nt(st, X, ldx, W, wtype, ldw, Y, ldy, M, N, K, bias, ACT_NONE, ar.base, ar.left, nullptr);
    *out = plain_vec(Y, ldy);
    return r;
  }
  int n = 1;
  const int r = gemm_nt(st, X, ldx, W, wtype, ldw, nullptr, 0, M, N, K, nullptr, ACT_NONE, ar.base, ar.left, &n);
  if (r != VLN_OK) return r;
  *out = SlabVec{ar.base, (long)N, n, per, bias};
  const long used = ((long)n * per + 63) & ~63L;
  ar.base += used; ar.left -= used;
  return VLN_OK;
}

// Y = act(X W^T + bias), Y2 = Y * dropout (optional): one launch for narrow outputs, else split-K slabs + reduce
int gemm_nt_fused(hipStream_t st, const float* X, long ldx, const void* W, int wtype, long ldw, float* Y, long ldy, int M,
                  int N, int K, const float* bias, int act, float* Y2, long ldy2, DropSpec drop, float* ws, long ws_floats) {
  if (M <= 0 || N <= 0 || K <= 0) { set_error("gemm_nt_fused: bad dims"); return VLN_ERR_ARG; }
  if (!(act & ACT_ACCUM) && n16_applies(M, N, K, wtype)) return launch_n16(st, X, ldx, W, wtype, ldw, Y, ldy, M, N, K, bias, act, Y2, ldy2, drop);
  int nsplit = 1;
  int r = gemm_nt(st, X, ldx, W, wtype, ldw, nullptr, 0, M, N, K, nullptr, ACT_NONE, ws, ws_floats, &nsplit);
  if (r != VLN_OK) return r;
  return reduce_epilogue(st, ws, nsplit, (long)M * N, N, Y, ldy, M, N, bias, act, Y2, ldy2, drop);
}

// one instantiation of gemm_rows_kernel (LDS above 64 KB needs the attribute, set once)
template <typename TW, int NRB>
static int launch_rows_t(hipStream_t st, int tiles, const GemmNTArgs& a, RowTiling rt, double bytes) {
  constexpr int smem = gemm_rows_smem_bytes<TW, NRB>();
  static bool attr_set = false;
  if (!attr_set) {
    if (smem > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rows_kernel<TW, NRB>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
      (void)hipGetLastError();
      return -1;
    }
    attr_set = true;
  }
  launch_timed(K_GEMM_NT, bytes, gemm_rows_kernel<TW, NRB>, dim3(tiles), dim3(256), (unsigned)smem, st, a, rt);
  VLN_CHECK_LAUNCH("gemm_rows");
  return VLN_OK;
}
// -1: no instantiation for this (weight type, tile height) -- the caller falls back to gemm_nt's tiles
static int launch_rows(hipStream_t st, int wtype, int rb_max, int tiles, const GemmNTArgs& a, RowTiling rt, double bytes) {
#define VLN_ROWS(TW)                                                   \
  do {                                                                 \
    switch (rb_max) {                                                  \
      case 3: return launch_rows_t<TW, 3>(st, tiles, a, rt, bytes);    \
      case 4: return launch_rows_t<TW, 4>(st, tiles, a, rt, bytes);    \
      case 5: return launch_rows_t<TW, 5>(st, tiles, a, rt, bytes);    \
      case 6: return launch_rows_t<TW, 6>(st, tiles, a, rt, bytes);    \
      case 7: return launch_rows_t<TW, 7>(st, tiles, a, rt, bytes);    \
      case 8: return launch_rows_t<TW, 8>(st, tiles, a, rt, bytes);    \
      default: return -1;                                              \
    }                                                                  \
  } while (0)
  if (wtype == W_F32) VLN_ROWS(float);
  if (wtype == W_F32S) VLN_ROWS(f32s_raw);
  if (wtype == W_F32X) VLN_ROWS(f32x_raw);
#undef VLN_ROWS
  return -1;
}

// host-only view of gemm_rows_plan for the tests: 1 and the tiling when a tall [M, K] x [N, K]^T product would take the row-block
// tiling on a device of `cus` compute units, 0 when it keeps gemm_nt's 64-row tiles
int gemm_rows_tiling(int M, int N, int cus, int* n_big, int* rb_big, int* tiles) {
  RowTiling rt{0, 0}; int t = 0, rbm = 0;
  if (M < 256 || !gemm_rows_plan(M, N, cus, &rt, &t, &rbm)) return 0;
  if (n_big) *n_big = rt.n_big;
  if (rb_big) *rb_big = rt.rb_big;
  if (tiles) *tiles = t;
  return 1;
}

int gemm_nt(hipStream_t st, const float* X, long ldx, const void* W, int wtype, long ldw, float* Y, long ldy,
            int M, int N, int K, const float* bias, int act, float* ws, long ws_floats, int* nsplit_out) {
  if (M <= 0 || N <= 0 || K <= 0) { set_error("gemm_nt: bad dims %d %d %d", M, N, K); return VLN_ERR_ARG; }
  if (nsplit_out == nullptr && !(act & ACT_ACCUM) && n16_applies(M, N, K, wtype))
    return launch_n16(st, X, ldx, W, wtype, ldw, Y, ldy, M, N, K, bias, act, nullptr, 0, DropSpec{0, 0, 0.f});
  const int BK = (wtype == W_F32) ? 32 : 64;
  // 128-column tiles for the wide AND deep products consumed from slabs (LSTM gates, d xcat: N, K >= 2048): measured
  // NO faster than 64-column tiles (11.3 vs 10.3 us and 11.2 vs 11.1 us; the pointwise consumer then reads 15 slabs
  // instead of 8) -- the X re-reads hit L2 and are not what bounds these launches.  Opt-in only: tunable[1] bit 1.
  const bool fast_ok = aligned16(X) && (ldx % 4 == 0) && aligned16(W) && (ldw % (wtype == W_BF16 ? 8 : 4) == 0) && (K % BK == 0) &&
                       g_tunable[5] != 2;
  const bool wide = nsplit_out != nullptr && N >= 2048 && K >= 2048 && M <= 64 && fast_ok && (g_tunable[1] & 2) &&
                    (wtype == W_F32 || wtype == W_BF16);       // (the split-fp32 forms have 64-column tiles only)
  const int TNW = wide ? 128 : 64;
  const int nb = (N + TNW - 1) / TNW, mb = (M + 63) / 64;
  const int ksteps = (K + BK - 1) / BK;
  // split K until ~2 workgroups per CU are in flight; keep >= 2 K-steps per chunk.  (Measured on MI355X:
  // many co-resident workgroups with one K-step of prefetch each beat one fat workgroup per CU with all of
  // its loads in flight: 8.5 vs 12.5 us average per launch in the EnvDrop step, profiles/round1_notes.md.)
  int steps_per = 1;
  int nsplit = gemm_nt_split(nb, mb, ksteps, M, N, ws != nullptr, ws_floats, nsplit_out != nullptr, &steps_per);
  GemmNTArgs a;
  a.X = X; a.ldx = ldx; a.W = W; a.ldw = ldw;
  a.M = M; a.N = N; a.K = K; a.kchunk = steps_per * BK;
  a.bias = bias; a.act = act;
  a.xvec = aligned16(X) && (ldx % 4 == 0);
  a.wvec = aligned16(W) && (ldw % (wtype == W_BF16 ? 8 : 4) == 0);
  const bool to_slabs = (nsplit > 1) || (nsplit_out != nullptr);
  if (to_slabs && ws == nullptr) { set_error("gemm_nt: slab output needs a workspace"); return VLN_ERR_ARG; }
  if (to_slabs) { a.Y = ws; a.ldy = N; a.slab_stride = (long)M * N; }
  else { a.Y = Y; a.ldy = ldy; a.slab_stride = 0; }
  dim3 grid(nb, nsplit, mb), block(256);
  a.xcd = (mb >= 8 && nb > 1 && ((long)nb * nsplit * mb) % 8 == 0 && g_tunable[8] != 0) ? 1 : 0;
  {
    // algorithmic bytes: the weight stream once + activations in + result out
    const double bytes = (double)N * K * (wtype == W_BF16 ? 2 : 4) + 4.0 * M * K + 4.0 * M * N;
    // fast form: aligned operands and every K-chunk a whole number of BK steps
    const bool fast = a.xvec && a.wvec && (K % BK == 0) && g_tunable[5] != 2;      // tunable[5] = 2: bounds-checked kernel (A/B)
    // Depth 4 was measured slower than depth 2 on every decoder shape (scripts/gemm_probe.hip: LSTM gates 13.0 vs 11.7 us;
    // the M = 5120 encoder projection 52 vs 37 us: 224 VGPRs halve the workgroups per CU): opt-in only, tunable[5] = 4.
    const bool deep = g_tunable[5] == 4 && steps_per > 2;
    // tall activations whose 64-row tiles do not fill whole rounds of the CUs: 16-row-block tiling (gemm_rows.h), same bits
    // (not for bf16-streamed weights: two bf16 MFMAs per fragment pair leave nothing to pipeline -- measured 43 vs 42 us and 40 vs 38 us)
    if (!to_slabs && fast && M >= 256 && wtype != W_BF16 && g_tunable[12] == 0) {
      RowTiling rt; int tiles = 0, rbm = 0;
      if (gemm_rows_plan(M, N, device_cus(), &rt, &tiles, &rbm)) {
        a.xcd = (tiles % 8 == 0 && nb > 1 && g_tunable[8] != 0) ? 1 : 0;
        const int r = launch_rows(st, wtype, rbm, tiles, a, rt, bytes);
        if (r != -1) return r;
        a.xcd = (mb >= 8 && nb > 1 && ((long)nb * nsplit * mb) % 8 == 0 && g_tunable[8] != 0) ? 1 : 0;
      }
    }
#define VLN_NT_LAUNCH(TW, PDv, FASTv) launch_timed(K_GEMM_NT, bytes, gemm_nt_kernel<TW, PDv, FASTv>, grid, block, 0, st, a)
    if (wtype != W_F32 && wtype != W_BF16 && wtype != W_F32S && wtype != W_F32X) { set_error("gemm_nt: unknown weight type %d", wtype); return VLN_ERR_ARG; }
    if (wide && fast && wtype != W_F32S && wtype != W_F32X) {
      if (wtype == W_BF16) launch_timed(K_GEMM_NT, bytes, gemm_nt_kernel<bf16_raw, 2, true, 2>, grid, block, 0, st, a);
      else launch_timed(K_GEMM_NT, bytes, gemm_nt_kernel<float, 2, true, 2>, grid, block, 0, st, a);
    } else if (wtype == W_F32S) {
      if (fast) VLN_NT_LAUNCH(f32s_raw, 2, true);
      else VLN_NT_LAUNCH(f32s_raw, 1, false);
    } else if (wtype == W_F32X) {
      if (fast) VLN_NT_LAUNCH(f32x_raw, 2, true);
      else VLN_NT_LAUNCH(f32x_raw, 1, false);
    } else if (wtype == W_BF16) {
      if (fast) { if (deep) VLN_NT_LAUNCH(bf16_raw, 4, true); else VLN_NT_LAUNCH(bf16_raw, 2, true); }
      else VLN_NT_LAUNCH(bf16_raw, 1, false);
    } else {
      if (fast) { if (deep) VLN_NT_LAUNCH(float, 4, true); else VLN_NT_LAUNCH(float, 2, true); }
      else VLN_NT_LAUNCH(float, 1, false);
    }
#undef VLN_NT_LAUNCH
  }
  VLN_CHECK_LAUNCH("gemm_nt");
  if (nsplit_out) { *nsplit_out = nsplit; return VLN_OK; }   // caller consumes the slabs itself
  if (to_slabs)
    return reduce_epilogue(st, ws, nsplit, (long)M * N, N, Y, ldy, M, N, bias, act, nullptr, 0, DropSpec{0, 0, 0.f});
  return VLN_OK;
}

// ---------------------------------------------------------------------------
// reduce + epilogue
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_epilogue_kernel(ReduceEpiArgs a) {
  reduce_epilogue_body(a, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

int reduce_epilogue(hipStream_t st, const float* slabs, int nsplit, long slab_stride, long lds, float* out,
                    long ldo, int M, int N, const float* bias, int act, float* out2, long ldo2, DropSpec drop) {
  long total = (long)M * N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  const ReduceEpiArgs a{slabs, nsplit, slab_stride, lds, out, ldo, M, N, bias, act, out2, ldo2, drop};
  VLN_LAUNCH(reduce_epilogue_kernel, dim3(blocks), dim3(256), 0, st, a);
  VLN_CHECK_LAUNCH("reduce_epilogue");
  return VLN_OK;
}

// ---------------------------------------------------------------------------
// gemm_tn: D[N,K] (+)= A[Mt,N]^T X[Mt,K]   (fp32 in, exact fp32 MFMA)
// 64x64 output tile per workgroup; both operands are staged row-major through LDS (coalesced along their
// contiguous dim) and read back as 4-byte fragments: the transposition costs no extra pass.
// ---------------------------------------------------------------------------
constexpr int kTnStride = 80;  // floats per staged row: 64 + 16 keeps the two k-slots of a half-wave on
                               // disjoint banks for ds_read_b32
struct GemmTNArgs {
  const float* A; long lda; const float* X; long ldx; float* D; long ldd;
  int Mt, N, K, accumulate, avec, xvec;
  int mchunk; long slab_stride;   // split over the contraction rows: block z handles rows [z*mchunk, ...)
};

__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTNArgs a) {
  __shared__ __attribute__((aligned(16))) float sA[2][32 * kTnStride];
  __shared__ __attribute__((aligned(16))) float sX[2][32 * kTnStride];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int mbeg = blockIdx.z * a.mchunk;
  const int mend = min(a.Mt, mbeg + a.mchunk);
  const int fi = lane & 15, fq = lane >> 4;
  const int sm = tid >> 3, sc = (tid & 7) * 8;   // staging: row in the 32-row step, 8-float column segment
  float ra[8], rx[8];

  auto load_tile = [&](const float* P, long ld, int c0, int C, int vec, int mbase, float (&r)[8]) {
    const int m = mbase + sm;
    const bool ok = m < mend;
    const float* p = P + (long)(ok ? m : 0) * ld + c0 + sc;
    if (ok && vec && (c0 + sc + 8) <= C) {
      float4 t0 = *reinterpret_cast<const float4*>(p);
      float4 t1 = *reinterpret_cast<const float4*>(p + 4);
      r[0] = t0.x; r[1] = t0.y; r[2] = t0.z; r[3] = t0.w; r[4] = t1.x; r[5] = t1.y; r[6] = t1.z; r[7] = t1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = (ok && (c0 + sc + j) < C) ? p[j] : 0.0f;
    }
  };
  auto store_tile = [&](float* S, const float (&r)[8]) {
    float* d = S + sm * kTnStride + sc;
    *reinterpret_cast<float4*>(d) = make_float4(r[0], r[1], r[2], r[3]);
    *reinterpret_cast<float4*>(d + 4) = make_float4(r[4], r[5], r[6], r[7]);
  };

  f32x4 acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nsteps = (mend - mbeg + 31) / 32;
  if (nsteps > 0) {
    load_tile(a.A, a.lda, n0, a.N, a.avec, mbeg, ra);
    load_tile(a.X, a.ldx, k0, a.K, a.xvec, mbeg, rx);
  }
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    store_tile(sA[buf], ra);
    store_tile(sX[buf], rx);
    __syncthreads();
    if (s + 1 < nsteps) {
      load_tile(a.A, a.lda, n0, a.N, a.avec, mbeg + (s + 1) * 32, ra);
      load_tile(a.X, a.ldx, k0, a.K, a.xvec, mbeg + (s + 1) * 32, rx);
    }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int m = kk * 4 + fq;
      const float av = sA[buf][m * kTnStride + wave * 16 + fi];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const float xv = sX[buf][m * kTnStride + kb * 16 + fi];
        acc[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xv, acc[kb], 0, 0, 0);
      }
    }
  }
  float* D = a.D + (long)blockIdx.z * a.slab_stride;
  float old[4][4];                 // all 16 old values in flight at once, from clamped addresses (see gemm_tn_x3_kernel)
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = min(k0 + kb * 16 + fi, a.K - 1), row = min(n0 + wave * 16 + fq * 4 + r, a.N - 1);
      old[kb][r] = a.accumulate ? D[(long)row * a.ldd + col] : 0.f;
    }
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const int col = k0 + kb * 16 + fi;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = n0 + wave * 16 + fq * 4 + r;
      if (row < a.N && col < a.K) D[(long)row * a.ldd + col] = acc[kb][r] + old[kb][r];
    }
  }
}

// ---------------------------------------------------------------------------
// gemm_tn, split-bf16 form (the bf16 compute mode's weight gradients): both fp32 operands are split x = hi + lo into
// bf16 planes and D += Ah^T Xh + Ah^T Xl + Al^T Xh on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (the dropped
// lo*lo term is 2^-16 relative): 3 x 16 cycles per 32 contraction rows instead of 8 x 32 on the f32 MFMA.
// 128(N) x 128(K) output tile per workgroup (operand re-reads halve against 64 x 64), 32 contraction rows per step.
// Staging: threads 0-127 own the dY operand, 128-255 the X operand; a thread loads a 4-column x 8-row block with eight
// coalesced float4 loads (32 lanes x 16 B = 512 contiguous bytes per row), so its registers already hold, per column,
// the 8 consecutive contraction rows one MFMA lane supplies -- the transposition costs no shuffle.  LDS image per
// operand and plane: [column][row group of 8] x 16 B, 80-byte rows (conflict-free ds_read_b128 / ds_write_b128).
// ---------------------------------------------------------------------------
// LDS image per operand and plane: [row group of 8 (4)][column slot (4 sections x 36)] x 16 B.  Column c of the tile
// lives at slot (c % 4) * 36 + c / 4: the 32 lanes of a staging store (columns 4*cg + c, cg = lane) hit consecutive
// 16-byte slots, and the 16 lanes of an MFMA fragment read (columns 16*t + fi) hit 16 distinct 4-bank groups.
constexpr int kX3Plane = 4 * 144 * 16;      // bytes
struct TnTile {           // one 128 x 128 output tile of D[N,K] (+)= A[Mt,N]^T X[Mt,K] over rows [mbeg, mend)
  const float* A; long lda; const float* X; long ldx; float* D; long ldd;
  int N, K, accumulate, n0, k0, mbeg, mend;
};
__device__ __forceinline__ int x3_slot(int c) { return (c & 3) * 36 + (c >> 2); }

__device__ __forceinline__ void tn_x3_tile(const TnTile& a, unsigned char (*sm)[2][kX3Plane]) {
  constexpr int PD = 2;                              // row steps in flight besides the one being multiplied
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, fq = lane >> 4;
  // staging role
  const int op = tid >> 7, t = tid & 127, cg = t & 31, mq = t >> 5;
  const float* P = op ? a.X : a.A;
  const long ld = op ? a.ldx : a.lda;
  const int C = op ? a.K : a.N;
  const int col = (op ? a.k0 : a.n0) + cg * 4;
  const bool col_ok = col < C;                       // C % 4 == 0 (host check): a float4 never straddles the edge
  const float* pc = P + (col_ok ? col : 0);
  float4 r[PD][8];
  auto load = [&](float4 (&q)[8], int mbase) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = mbase + mq * 8 + i;
      const float4 v = *reinterpret_cast<const float4*>(pc + (long)min(m, a.mend - 1) * ld);
      const bool ok = col_ok && m < a.mend;          // select, not a branch: the loads stay unconditional
      q[i] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
  };
  auto store = [&](const float4 (&q)[8]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      bf16x8 h, l;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = (c == 0) ? q[i].x : (c == 1) ? q[i].y : (c == 2) ? q[i].z : q[i].w;
        h[i] = (__bf16)v;
        l[i] = (__bf16)(v - (float)h[i]);
      }
      const int off = (mq * 144 + c * 36 + cg) * 16;
      *reinterpret_cast<bf16x8*>(&sm[op][0][off]) = h;
      *reinterpret_cast<bf16x8*>(&sm[op][1][off]) = l;
    }
  };
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nsteps = (a.mend - a.mbeg + 31) / 32;
#pragma unroll
  for (int p = 0; p < PD; ++p)
    if (p < nsteps) load(r[p], a.mbeg + p * 32);
  for (int s0 = 0; s0 < nsteps; s0 += PD) {
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const int s = s0 + p;
      if (s < nsteps) {
        store(r[p]);
        __syncthreads();
        if (s + PD < nsteps) load(r[p], a.mbeg + (s + PD) * 32);
        bf16x8 ah[4], al[4], xh[4], xl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int oa = (fq * 144 + x3_slot(wn + i * 16 + fi)) * 16, ox = (fq * 144 + x3_slot(wk + i * 16 + fi)) * 16;
          ah[i] = *reinterpret_cast<const bf16x8*>(&sm[0][0][oa]);
          al[i] = *reinterpret_cast<const bf16x8*>(&sm[0][1][oa]);
          xh[i] = *reinterpret_cast<const bf16x8*>(&sm[1][0][ox]);
          xl[i] = *reinterpret_cast<const bf16x8*>(&sm[1][1][ox]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], xh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], xl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], xh[j], acc[i][j], 0, 0, 0);
          }
        __syncthreads();
      }
    }
  }
  // Epilogue.  With `accumulate` the old values are fetched 16 at a time from clamped addresses (no branch around a
  // load: a bounds-checked read-modify-write per element makes every one of the 64 a separate round trip to memory).
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float old[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kcol = min(a.k0 + wk + j * 16 + fi, a.K - 1), nrow = min(a.n0 + wn + i * 16 + fq * 4 + q, a.N - 1);
        old[j][q] = a.accumulate ? a.D[(long)nrow * a.ldd + kcol] : 0.f;
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kcol = a.k0 + wk + j * 16 + fi;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nrow = a.n0 + wn + i * 16 + fq * 4 + q;
        if (nrow < a.N && kcol < a.K) a.D[(long)nrow * a.ldd + kcol] = acc[i][j][q] + old[j][q];
      }
    }
  }
}

__global__ __launch_bounds__(256) void gemm_tn_x3_kernel(GemmTNArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[2][2][kX3Plane];
  const int mbeg = blockIdx.z * a.mchunk;
  TnTile t{a.A, a.lda, a.X, a.ldx, a.D + (long)blockIdx.z * a.slab_stride, a.ldd, a.N, a.K, a.accumulate,
           (int)blockIdx.y * 128, (int)blockIdx.x * 128, mbeg, min(a.Mt, mbeg + a.mchunk)};
  tn_x3_tile(t, sm);
}

// Every weight gradient of a module in ONE launch: a workgroup finds its job from the prefix sums of the jobs' tile
// counts (all jobs contract over the same Mt rows: the decoder's (steps x batch) stash rows).  The small products no
// longer pay a launch + a row-split + a slab reduce each (7 launches of 14-21 us and 13 reduce launches per iteration).
struct WgradJobs {
  vln_wgrad_job j[VLN_WGRAD_MAX_JOBS];
  int tile0[VLN_WGRAD_MAX_JOBS + 1];
  long slab0[VLN_WGRAD_MAX_JOBS];     // msplit > 1: offset of the job's first slab in ws (floats)
  float* ws;
  int n, Mt, msplit, mchunk, ntiles, per_xcd;
};
// XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own L2.  Tiles
// that share an operand block (a row of tiles shares its dY columns) are numbered consecutively, so giving XCD x the
// contiguous range [x*per_xcd, (x+1)*per_xcd) lets those re-reads hit that XCD's L2 instead of the fabric.
__global__ __launch_bounds__(256) void wgrad_grouped_x3_kernel(WgradJobs a) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[2][2][kX3Plane];
  const int lt = ((int)blockIdx.x & 7) * a.per_xcd + ((int)blockIdx.x >> 3);
  if (((int)blockIdx.x >> 3) >= a.per_xcd || lt >= a.ntiles) return;
  int ji = 0;
  while (ji + 1 < a.n && lt >= a.tile0[ji + 1]) ++ji;
  const vln_wgrad_job& q = a.j[ji];
  const int tile = lt - a.tile0[ji];
  const int nbk = (q.K + 127) / 128;
  const int mbeg = (int)blockIdx.y * a.mchunk;
  TnTile t{q.dy, q.ld_dy, q.x, q.ld_x, q.dw, q.ld_dw, q.N, q.K, q.accumulate, (tile / nbk) * 128, (tile % nbk) * 128, mbeg,
           min(a.Mt, mbeg + a.mchunk)};
  if (a.msplit > 1) { t.D = a.ws + a.slab0[ji] + (long)blockIdx.y * q.N * q.K; t.ldd = q.K; t.accumulate = 0; }
  tn_x3_tile(t, sm);
}
// dw (+)= sum of the job's msplit slabs, every job in one launch (float4 columns: K % 4 == 0)
__global__ __launch_bounds__(256) void wgrad_grouped_reduce_kernel(WgradJobs a) {
  for (int ji = 0; ji < a.n; ++ji) {
    const vln_wgrad_job& q = a.j[ji];
    const long total4 = (long)q.N * q.K / 4, per = (long)q.N * q.K;
    const float* sl = a.ws + a.slab0[ji];
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (long)gridDim.x * blockDim.x) {
      const long r = (e * 4) / q.K, c = (e * 4) % q.K;
      float4 v = *reinterpret_cast<const float4*>(sl + e * 4);
      for (int s = 1; s < a.msplit; ++s) {
        const float4 t = *reinterpret_cast<const float4*>(sl + (long)s * per + e * 4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      float* d = q.dw + r * q.ld_dw + c;
      if (q.accumulate) { v.x += d[0]; v.y += d[1]; v.z += d[2]; v.w += d[3]; }
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  }
}

// D[r, c] (+)= sum_s slabs[s][r*cols + c]
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* slabs, int nsplit, long slab_stride, float* D,
                                                           long ldd, int rows, int cols, int accumulate) {
  const long total = (long)rows * cols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int r = (int)(e / cols), c = (int)(e % cols);
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += slabs[(long)s * slab_stride + e];
    float* d = D + (long)r * ldd + c;
    *d = accumulate ? (*d + v) : v;
  }
}
static int reduce_slabs(hipStream_t st, const float* slabs, int nsplit, long slab_stride, float* D, long ldd, int rows,
                        int cols, int accumulate) {
  long total = (long)rows * cols;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  VLN_LAUNCH(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, slabs, nsplit, slab_stride, D, ldd, rows, cols, accumulate);
  VLN_CHECK_LAUNCH("reduce_slabs");
  return VLN_OK;
}

int gemm_tn(hipStream_t st, const float* A, long lda, const float* X, long ldx, float* D, long ldd, int Mt,
            int N, int K, int accumulate, float* ws, long ws_floats, int precision) {
  if (N <= 0 || K <= 0 || Mt < 0) { set_error("gemm_tn: bad dims"); return VLN_ERR_ARG; }
  GemmTNArgs a{A, lda, X, ldx, D, ldd, Mt, N, K, accumulate, 0, 0, 0, 0};
  a.avec = aligned16(A) && (lda % 4 == 0);
  a.xvec = aligned16(X) && (ldx % 4 == 0);
  // precision 1 (bf16 compute mode): split-bf16 MFMA form on 128 x 128 tiles, when the operands allow float4 columns
  const bool x3 = precision == 1 && a.avec && a.xvec && (N % 4 == 0) && (K % 4 == 0) && Mt > 0 && g_tunable[6] != 1;
  const int T = x3 ? 128 : 64, MS = x3 ? 32 : 32;
  const int nbk = (K + T - 1) / T, nbn = (N + T - 1) / T;
  // outputs with few tiles and a long contraction (encoder: Mt = L*B): split the rows over workgroups, slabs + reduce
  int msplit = 1;
  if (ws) {
    msplit = (x3 ? 256 : 512) / (nbk * nbn);
    const int max_by_rows = Mt / 128;
    if (msplit > max_by_rows) msplit = max_by_rows;
    const long per = (long)N * K;
    if ((long)msplit * per > ws_floats) msplit = (int)(ws_floats / per);
    if (msplit < 1) msplit = 1;
  }
  int mchunk = ((Mt + msplit - 1) / msplit + MS - 1) / MS * MS;
  if (mchunk < MS) mchunk = MS;
  msplit = (Mt + mchunk - 1) / mchunk;
  if (msplit < 1) msplit = 1;
  a.mchunk = mchunk;
  if (msplit > 1) { a.D = ws; a.ldd = K; a.slab_stride = (long)N * K; a.accumulate = 0; }
  dim3 grid(nbk, nbn, msplit), block(256);
  {
    const double bytes = 4.0 * ((double)Mt * N + (double)Mt * K + (double)N * K * (accumulate ? 2 : 1));
    if (x3) launch_timed(K_GEMM_TN, bytes, gemm_tn_x3_kernel, grid, block, 0, st, a);
    else launch_timed(K_GEMM_TN, bytes, gemm_tn_kernel, grid, block, 0, st, a);
  }
  VLN_CHECK_LAUNCH("gemm_tn");
  if (msplit > 1) return reduce_slabs(st, ws, msplit, (long)N * K, D, ldd, N, K, accumulate);
  return VLN_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Packed form of the grouped weight gradients (the default for precision 1).
//   pass 1  wgrad_pack_kernel: every operand matrix [Mt, C] fp32 -> two bf16 planes (hi, lo) in MFMA-FRAGMENT order:
//           block (column tile ct of 16, row step ms of 32) = 64 lanes x 16 B, lane (fi, fq) = column 16 ct + fi, rows
//           32 ms + 8 fq .. +8.  Blocks of one column tile are consecutive over ms.  Each element is converted ONCE
//           (the LDS-staged kernel converts it again in every tile that re-reads it: 18x for the LSTM's dgates).
//   pass 2  wgrad_packed_kernel: a wave's fragment for (tile, step) is ONE coalesced 1 KiB load straight into the
//           registers the MFMA reads -- no LDS, no barrier, no conversion; waves run independently with the next
//           step's 16 loads in flight behind the current step's 48 MFMAs.
// ---------------------------------------------------------------------------------------------------------------
#include "wgrad_ride.h"
using PackJobs = PackJobsT<VLN_WGRAD_MAX_JOBS>;
using PackedJobs = PackedJobsT<VLN_WGRAD_MAX_JOBS>;
__global__ __launch_bounds__(256) void wgrad_pack_kernel(PackJobs a) { wgrad_pack_block(a, (int)blockIdx.x); }
template <int TERMS>
__global__ __launch_bounds__(256) void wgrad_packed_kernel(PackedJobs a) { wgrad_packed_tile<TERMS>(a, (int)blockIdx.x, (int)blockIdx.y); }

// bytes of workspace the packed path wants for these jobs (pack area + row-split slabs)
static long wgrad_packed_ws_floats(const vln_wgrad_job* jobs, int n, int Mt, int* msplit_out, long* area_floats_out) {
  const int MS = (Mt + 31) / 32;
  long area = 0, elems = 0;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    area += 2L * (((jobs[i].N + 15) / 16) + ((jobs[i].K + 15) / 16)) * MS * 1024;
    elems += (long)jobs[i].N * jobs[i].K;
    tiles += ((jobs[i].N + 127) / 128) * ((jobs[i].K + 127) / 128);
  }
  int msplit = 1;
  if (tiles < 256) { msplit = 256 / tiles; if (msplit > MS / 4) msplit = MS / 4; if (msplit < 1) msplit = 1; }
  if (msplit_out) *msplit_out = msplit;
  if (area_floats_out) *area_floats_out = area / 4;
  return area / 4 + (msplit > 1 ? (long)msplit * elems : 0);
}

// Job tables of the packed form (pack launch + contraction launch) for `n` products over Mt rows; returns false when the
// workspace does not hold them.  NJ = capacity of the tables (the library-wide one, or a ride's).
template <int NJ>
static bool wgrad_packed_tables(const vln_wgrad_job* jobs, int n, int Mt, float* ws, long ws_floats, int terms, int seg_rows,
                                const int64_t* dy_seg, const int64_t* x_seg, PackJobsT<NJ>& pk, PackedJobsT<NJ>& g, int* pack_blocks,
                                double* bytes_out) {
  const int MS = (Mt + 31) / 32;
  int msplit = 1; long area_floats = 0;
  const long need = wgrad_packed_ws_floats(jobs, n, Mt, &msplit, &area_floats);
  if (n > NJ || need > ws_floats || !aligned16(ws)) return false;
  pk.area = reinterpret_cast<unsigned char*>(ws); pk.Mt = Mt; pk.MS = MS; pk.n = 0; pk.lo = terms == 3 ? 1 : 0; pk.seg_rows = seg_rows;
  g.area = pk.area; g.ws = ws + area_floats; g.Mt = Mt; g.MS = MS; g.n = n;
  long off = 0; int blk = 0, t = 0; long slab = 0;
  double bytes = 0.0;
  const int rblocks = (MS + 1) / 2;
  auto add_pack = [&](const float* src, long ld, int C, long seg, int rows) -> long {
    for (int i = 0; i < pk.n; ++i)                                    // an operand shared by two products is packed once
      if (pk.j[i].src == src && pk.j[i].ld == ld && pk.j[i].C == C && pk.j[i].seg_stride == seg && pk.j[i].rows == rows) return pk.j[i].dst;
    PackJob& q = pk.j[pk.n];
    q.src = src; q.ld = ld; q.C = C; q.dst = off; q.seg_stride = seg; q.rows = rows;
    pk.blk0[pk.n] = blk;
    blk += ((C + 127) / 128) * rblocks;
    off += 2L * ((C + 15) / 16) * MS * 1024;
    pk.n++;
    return q.dst;
  };
  for (int i = 0; i < n; ++i) {
    g.j[i] = jobs[i];
    const int jr = (jobs[i].rows > 0 && jobs[i].rows < Mt) ? jobs[i].rows : Mt;     // a product over fewer rows: the rest packs as zeros
    g.pa[i] = add_pack(jobs[i].dy, jobs[i].ld_dy, jobs[i].N, dy_seg ? (long)dy_seg[i] : 0, jr);
    g.px[i] = add_pack(jobs[i].x, jobs[i].ld_x, jobs[i].K, x_seg ? (long)x_seg[i] : 0, jr);
    g.tile0[i] = t;
    t += ((jobs[i].N + 127) / 128) * ((jobs[i].K + 127) / 128);
    g.slab0[i] = slab; slab += (long)msplit * jobs[i].N * jobs[i].K;
    bytes += 4.0 * ((double)Mt * jobs[i].N + (double)Mt * jobs[i].K + (double)jobs[i].N * jobs[i].K * (jobs[i].accumulate ? 2 : 1));
  }
  pk.blk0[pk.n] = blk;
  g.tile0[n] = t; g.ntiles = t; g.per_xcd = (t + 7) / 8;
  g.schunk = (MS + msplit - 1) / msplit;
  g.msplit = (MS + g.schunk - 1) / g.schunk;
  *pack_blocks = blk;
  if (bytes_out) *bytes_out = bytes;
  return true;
}

// ---- a POSTED product (vln_linear_fwd_post): Y = X W^T that the NEXT packed weight-gradient call issues as extra workgroups of its
// pack launch.  The encoder backward's d x = dgates W_ih (M = L * B rows, 320 tiles of 32 K-steps: 36 us, bound by its first-touch
// latencies and by 1.25 tiles per CU) and the pack of the same dgates for the layer's weight gradients (HBM-bound, 21 us) do not depend
// on each other: in one launch the pack's blocks fill the slots the product's tiles leave.  Same tiles, same kernel body as the
// stand-alone call: bit-identical.  A post that no packed call takes is issued on its own by vln_linear_fwd_post_flush.
struct PostedProduct { bool on = false; const float* X; long ldx; const void* W; int wtype; long ldw; float* Y; long ldy; int M, N, K; };
static thread_local PostedProduct g_posted;
// ... and POSTED column sums (vln_colsum_post: the layer's bias gradients, sums over the BPTT's per-row-block partials): they depend on
// nothing the weight gradients produce either and ride in the same launch, behind the product's tiles
using ColsumJobs = ColsumJobsT<VLN_COLSUM_MAX_JOBS>;                  // (defined with the column-sum launches below)
template <int NJ>
static int colsum_tables(const vln_colsum_job* jobs, int n, int rows, float* ws, long ws_floats, int seg_rows, const int64_t* seg_stride,
                         ColsumJobsT<NJ>& a, int* blocks);
struct PostedColsums { bool on = false; vln_colsum_job j[VLN_COLSUM_MAX_JOBS]; int n, rows; };
static thread_local PostedColsums g_posted_cs;

template <typename TW>
__global__ __launch_bounds__(256) void wgrad_pack_gemm_kernel(PackJobs pk, GemmNTArgs a, int gx, int gz, ColsumJobs cs, int cs_blocks) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[gemm_nt_smem_bytes(GemmCfg<TW>::kPlanes)];
  const int ng = gx * gz;
  if ((int)blockIdx.x >= ng + cs_blocks) { wgrad_pack_block(pk, (int)blockIdx.x - ng - cs_blocks); return; }
  if ((int)blockIdx.x >= ng) { colsum_grouped_block(cs, (int)blockIdx.x - ng, 0, reinterpret_cast<float4 (*)[4]>(smem)); return; }
  int t = (int)blockIdx.x;
  if (a.xcd) t = (t & 7) * (ng >> 3) + (t >> 3);          // gemm_nt_kernel's XCD-aware tile order (ng % 8 == 0)
  const VBlock vb{t % gx, 0, t / gx, (int)threadIdx.x, smem};
  gemm_nt_body<TW, 2, true, 1>(a, vb, true, gemm_nt_nsteps(a, 0, GemmCfg<TW>::BK), [] {});
}
// the posted product as gemm_nt's un-split fast launch would run it: its argument block and tile grid; false: not that form
static bool posted_args(const PostedProduct& p, GemmNTArgs* a, int* gx, int* gz) {
  if (p.wtype != W_BF16 && p.wtype != W_F32) return false;
  const int BK = (p.wtype == W_F32) ? 32 : 64;
  const bool fast = aligned16(p.X) && (p.ldx % 4 == 0) && aligned16(p.W) && (p.ldw % (p.wtype == W_BF16 ? 8 : 4) == 0) && (p.K % BK == 0) &&
                    g_tunable[5] != 2 && g_tunable[5] != 4;
  if (!fast || n16_applies(p.M, p.N, p.K, p.wtype) || gemm_nt_plain_slabs(p.M, p.N, p.K, p.wtype, 1L << 40) != 1) return false;
  const int nb = (p.N + 63) / 64, mb = (p.M + 63) / 64;
  a->X = p.X; a->ldx = p.ldx; a->W = p.W; a->ldw = p.ldw; a->Y = p.Y; a->ldy = p.ldy; a->slab_stride = 0;
  a->bias = nullptr; a->act = ACT_NONE; a->M = p.M; a->N = p.N; a->K = p.K; a->kchunk = ((p.K + BK - 1) / BK) * BK;
  a->xvec = 1; a->wvec = 1;
  a->xcd = (mb >= 8 && nb > 1 && ((long)nb * mb) % 8 == 0 && g_tunable[8] != 0) ? 1 : 0;
  *gx = nb; *gz = mb;
  return true;
}
int linear_fwd_post(const float* X, long ldx, const void* W, int wtype, long ldw, float* Y, long ldy, int M, int N, int K) {
  if (!X || !W || !Y || M <= 0 || N <= 0 || K <= 0) { set_error("vln_linear_fwd_post: bad args"); return VLN_ERR_ARG; }
  if (g_posted.on) { set_error("vln_linear_fwd_post: a posted product is still pending (vln_linear_fwd_post_flush)"); return VLN_ERR_ARG; }
  g_posted = PostedProduct{true, X, ldx, W, wtype, ldw, Y, ldy, M, N, K};
  return VLN_OK;
}
int linear_fwd_post_flush(hipStream_t st, float* ws, long ws_floats) {
  if (!g_posted.on) return VLN_OK;
  const PostedProduct p = g_posted;
  g_posted.on = false;
  return gemm_nt(st, p.X, p.ldx, p.W, p.wtype, p.ldw, p.Y, p.ldy, p.M, p.N, p.K, nullptr, ACT_NONE, ws, ws_floats, nullptr);
}
int colsum_grouped(hipStream_t st, const vln_colsum_job* jobs, int n, int rows, float* ws, long ws_floats, int seg_rows, const int64_t* seg_stride);
int colsum_post(const vln_colsum_job* jobs, int n, int rows) {
  if (!jobs || n <= 0 || n > VLN_COLSUM_MAX_JOBS || rows <= 0) { set_error("vln_colsum_post: 1..%d jobs over > 0 rows", VLN_COLSUM_MAX_JOBS); return VLN_ERR_ARG; }
  if (g_posted_cs.on) { set_error("vln_colsum_post: posted column sums are still pending (vln_colsum_post_flush)"); return VLN_ERR_ARG; }
  g_posted_cs.on = true; g_posted_cs.n = n; g_posted_cs.rows = rows;
  for (int i = 0; i < n; ++i) g_posted_cs.j[i] = jobs[i];
  return VLN_OK;
}
// what an aborted caller left posted (it raised between its post and its flush): forgotten, its buffers may be gone
int posted_drop() {
  const int n = (g_posted.on ? 1 : 0) + (g_posted_cs.on ? 1 : 0) + (g_posted_layout.on ? 1 : 0);
  g_posted.on = false; g_posted_cs.on = false; g_posted_layout.on = false;
  return n;
}
int colsum_post_flush(hipStream_t st, float* ws, long ws_floats) {
  if (!g_posted_cs.on) return VLN_OK;
  g_posted_cs.on = false;
  return colsum_grouped(st, g_posted_cs.j, g_posted_cs.n, g_posted_cs.rows, ws, ws_floats, 0, nullptr);
}

static int wgrad_grouped_packed(hipStream_t st, const vln_wgrad_job* jobs, int n, int Mt, float* ws, long ws_floats, int terms,
                                int seg_rows = 0, const int64_t* dy_seg = nullptr, const int64_t* x_seg = nullptr) {
  PackJobs pk; PackedJobs g;
  int blk = 0; double bytes = 0.0;
  if (!wgrad_packed_tables<VLN_WGRAD_MAX_JOBS>(jobs, n, Mt, ws, ws_floats, terms, seg_rows, dy_seg, x_seg, pk, g, &blk, &bytes))
    return -1;                                                        // caller falls back to the LDS-staged kernel
  GemmNTArgs pa; int gx = 0, gz = 0;
  if (g_posted.on && posted_args(g_posted, &pa, &gx, &gz)) {          // the posted product's tiles first, the pack blocks behind them
    ColsumJobs cs{}; int csb = 0;                                     // ... and the posted column sums between them (one pass each)
    if (g_posted_cs.on && colsum_tables<VLN_COLSUM_MAX_JOBS>(g_posted_cs.j, g_posted_cs.n, g_posted_cs.rows, nullptr, 0, 0, nullptr, cs, &csb) == VLN_OK &&
        cs.rsplit == 1) g_posted_cs.on = false;
    else csb = 0;
    if (g_posted.wtype == W_BF16) VLN_LAUNCH(wgrad_pack_gemm_kernel<bf16_raw>, dim3(gx * gz + csb + blk), dim3(256), 0, st, pk, pa, gx, gz, cs, csb);
    else VLN_LAUNCH(wgrad_pack_gemm_kernel<float>, dim3(gx * gz + csb + blk), dim3(256), 0, st, pk, pa, gx, gz, cs, csb);
    g_posted.on = false;
  } else
  VLN_LAUNCH(wgrad_pack_kernel, dim3(blk), dim3(256), 0, st, pk);
  if (terms == 3) launch_timed(K_GEMM_TN, bytes, wgrad_packed_kernel<3>, dim3(g.per_xcd * 8, g.msplit), dim3(256), 0, st, g);
  else launch_timed(K_GEMM_TN, bytes, wgrad_packed_kernel<1>, dim3(g.per_xcd * 8, g.msplit), dim3(256), 0, st, g);
  if (g.msplit > 1) {
    WgradJobs r;
    r.n = n; r.Mt = Mt; r.ws = g.ws; r.msplit = g.msplit;
    long elems = 0;
    for (int i = 0; i < n; ++i) { r.j[i] = jobs[i]; r.slab0[i] = g.slab0[i]; elems += (long)jobs[i].N * jobs[i].K; }
    long b = (elems / 4 / n + 255) / 256;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    VLN_LAUNCH(wgrad_grouped_reduce_kernel, dim3((unsigned)b), dim3(256), 0, st, r);
  }
  return (int)check_hip(hipGetLastError(), "wgrad_grouped_packed");
}

// Rollout-level form for per-step C calls: every operand is n_seg blocks of [seg_rows, C], block t at base + t * stride.  One
// pack + one contraction over Mt = n_seg * seg_rows rows when the packed kernel takes the operands; otherwise one grouped call
// per segment, accumulating.
int wgrad_grouped_seg(hipStream_t st, const vln_wgrad_job* jobs, int n, int seg_rows, int n_seg, const int64_t* dy_seg,
                      const int64_t* x_seg, int precision, float* ws, long ws_floats) {
  if (n <= 0 || n > VLN_WGRAD_MAX_JOBS || seg_rows <= 0 || n_seg <= 0 || !dy_seg || !x_seg) { set_error("wgrad_grouped_seg: bad args"); return VLN_ERR_ARG; }
  if (n_seg == 1) return wgrad_grouped(st, jobs, n, seg_rows, precision, ws, ws_floats);
  bool ok = precision != 0 && g_tunable[6] != 1 && g_tunable[6] != 2;
  for (int i = 0; i < n && ok; ++i) {
    const vln_wgrad_job& q = jobs[i];
    ok = q.dy && q.x && q.dw && aligned16(q.dy) && aligned16(q.x) && (q.ld_dy % 4 == 0) && (q.ld_x % 4 == 0) && (q.N % 4 == 0) &&
         (q.K % 4 == 0) && (dy_seg[i] % 4 == 0) && (x_seg[i] % 4 == 0);
  }
  if (ok) {
    const int r = wgrad_grouped_packed(st, jobs, n, seg_rows * n_seg, ws, ws_floats, precision == 2 ? 1 : 3, seg_rows, dy_seg, x_seg);
    if (r >= 0) return r;
  }
  vln_wgrad_job seg[VLN_WGRAD_MAX_JOBS];
  for (int t = 0; t < n_seg; ++t) {
    for (int i = 0; i < n; ++i) {
      seg[i] = jobs[i];
      seg[i].dy = jobs[i].dy + (long)t * dy_seg[i];
      seg[i].x = jobs[i].x + (long)t * x_seg[i];
      if (t) seg[i].accumulate = 1;
    }
    const int r = wgrad_grouped(st, seg, n, seg_rows, precision, ws, ws_floats);
    if (r != VLN_OK) return r;
  }
  return VLN_OK;
}
int64_t wgrad_grouped_ws_floats(const vln_wgrad_job* jobs, int n, int Mt) { return wgrad_packed_ws_floats(jobs, n, Mt, nullptr, nullptr); }

int wgrad_grouped(hipStream_t st, const vln_wgrad_job* jobs, int n, int Mt, int precision, float* ws, long ws_floats) {
  if (n <= 0 || Mt <= 0) { set_error("wgrad_grouped: bad args"); return VLN_ERR_ARG; }
  {
    // products over their OWN row counts (a ride nobody carried): the jobs over the call's rows stay one grouped call -- the same
    // launches as without the odd ones -- and each odd one is a call of its own
    int odd = 0;
    for (int i = 0; i < n; ++i) odd += (jobs[i].rows > 0 && jobs[i].rows != Mt) ? 1 : 0;
    if (odd) {
      if (n > VLN_WGRAD_MAX_JOBS) { set_error("wgrad_grouped: too many jobs"); return VLN_ERR_ARG; }
      vln_wgrad_job same[VLN_WGRAD_MAX_JOBS];
      int ns = 0;
      for (int i = 0; i < n; ++i) {
        vln_wgrad_job q = jobs[i];
        const int r = q.rows;
        q.rows = 0;
        if (r > 0 && r != Mt) {
          if (r > Mt) { set_error("wgrad_grouped: job %d has more rows than the call", i); return VLN_ERR_ARG; }
          const int rc = wgrad_grouped(st, &q, 1, r, precision, ws, ws_floats);
          if (rc != VLN_OK) return rc;
        } else {
          same[ns++] = q;
        }
      }
      return ns ? wgrad_grouped(st, same, ns, Mt, precision, ws, ws_floats) : VLN_OK;
    }
  }
  // precision 2 = plain bf16 operands on the packed grouped kernel; where that kernel cannot be used it degrades to the
  // (more accurate) split form
  const int terms = precision == 2 ? 1 : 3;
  if (precision == 2) precision = 1;
  bool ok = precision == 1 && g_tunable[6] != 1;
  for (int i = 0; i < n && ok; ++i) {
    const vln_wgrad_job& q = jobs[i];
    ok = aligned16(q.dy) && aligned16(q.x) && (q.ld_dy % 4 == 0) && (q.ld_x % 4 == 0) && (q.N % 4 == 0) && (q.K % 4 == 0);
  }
  if (!ok) {       // exact fp32 form (or operands the grouped kernel does not take): one launch per product
    for (int i = 0; i < n; ++i) {
      const vln_wgrad_job& q = jobs[i];
      int r = gemm_tn(st, q.dy, q.ld_dy, q.x, q.ld_x, q.dw, q.ld_dw, Mt, q.N, q.K, q.accumulate, ws, ws_floats, precision);
      if (r != VLN_OK) return r;
    }
    return VLN_OK;
  }
  if (n <= VLN_WGRAD_MAX_JOBS && g_tunable[6] != 2) {      // tunable[6] = 2: LDS-staged grouped kernel (A/B)
    const int r = wgrad_grouped_packed(st, jobs, n, Mt, ws, ws_floats, terms);
    if (r >= 0) return r;
  }
  for (int base = 0; base < n; base += VLN_WGRAD_MAX_JOBS) {
    WgradJobs a;
    a.n = (n - base < VLN_WGRAD_MAX_JOBS) ? n - base : VLN_WGRAD_MAX_JOBS;
    a.Mt = Mt; a.ws = ws;
    int t = 0;
    long elems = 0;
    double bytes = 0.0;
    for (int i = 0; i < a.n; ++i) {
      const vln_wgrad_job& q = jobs[base + i];
      if (!q.dy || !q.x || !q.dw || q.N <= 0 || q.K <= 0) { set_error("wgrad_grouped: bad job %d", base + i); return VLN_ERR_ARG; }
      a.j[i] = q;
      a.tile0[i] = t;
      t += ((q.N + 127) / 128) * ((q.K + 127) / 128);
      elems += (long)q.N * q.K;
      bytes += 4.0 * ((double)Mt * q.N + (double)Mt * q.K + (double)q.N * q.K * (q.accumulate ? 2 : 1));
    }
    a.tile0[a.n] = t;
    a.ntiles = t;
    a.per_xcd = (t + 7) / 8;
    // few tiles, long contraction (the encoder: Mt = L*B rows): split the rows too, slabs in ws + one grouped reduce
    int msplit = 1;
    if (ws && t < 256) {
      msplit = 256 / t;
      if (msplit > Mt / 128) msplit = Mt / 128;
      if ((long)msplit * elems > ws_floats) msplit = (int)(ws_floats / elems);
      if (msplit < 1) msplit = 1;
    }
    int mchunk = ((Mt + msplit - 1) / msplit + 31) / 32 * 32;
    msplit = (Mt + mchunk - 1) / mchunk;
    a.msplit = msplit; a.mchunk = mchunk;
    long off = 0;
    for (int i = 0; i < a.n; ++i) { a.slab0[i] = off; off += (long)msplit * a.j[i].N * a.j[i].K; }
    launch_timed(K_GEMM_TN, bytes, wgrad_grouped_x3_kernel, dim3(a.per_xcd * 8, msplit), dim3(256), 0, st, a);
    if (msplit > 1) {
      long b = (elems / 4 / a.n + 255) / 256;
      if (b > 1024) b = 1024;
      if (b < 1) b = 1;
      VLN_LAUNCH(wgrad_grouped_reduce_kernel, dim3((unsigned)b), dim3(256), 0, st, a);
    }
  }
  VLN_CHECK_LAUNCH("wgrad_grouped");
  return VLN_OK;
}

// ---------------------------------------------------------------------------
// colsum: bias gradients.  grid (cols/64, row chunks); chunks > 1 go through the workspace + reduce.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* A, long lda, float* out, long out_stride, int rows,
                                                     int cols, int rchunk, int accumulate) {
  // block = 64 columns x 4 row-lanes; rows strided by 4
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int rbeg = blockIdx.y * rchunk, rend = min(rows, rbeg + rchunk);
  float s = 0.f;
  if (c < cols)
    for (int r = rbeg + rl; r < rend; r += 4) s += A[(long)r * lda + c];
  part[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < cols) {
    float t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    float* o = out + (long)blockIdx.y * out_stride + c;
    *o = accumulate ? *o + t : t;
  }
}
// aligned fast path: 16 float4 column groups x 16 row lanes, 4 loads in flight per thread
__global__ __launch_bounds__(256) void colsum4_kernel(const float* A, long lda, float* out, long out_stride, int rows,
                                                      int cols, int rchunk, int accumulate) {
  __shared__ float4 part[16][16];
  const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cg * 4;
  const int rbeg = blockIdx.y * rchunk, rend = min(rows, rbeg + rchunk);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  if (c < cols) {                       // cols % 4 == 0: a float4 never straddles the edge
    const float* p = A + c;
    int r = rbeg + rl;
    for (; r + 48 < rend; r += 64) {
      const float4 a = *reinterpret_cast<const float4*>(p + (long)r * lda);
      const float4 b = *reinterpret_cast<const float4*>(p + (long)(r + 16) * lda);
      const float4 d = *reinterpret_cast<const float4*>(p + (long)(r + 32) * lda);
      const float4 e = *reinterpret_cast<const float4*>(p + (long)(r + 48) * lda);
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
      s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
      s2.x += d.x; s2.y += d.y; s2.z += d.z; s2.w += d.w;
      s3.x += e.x; s3.y += e.y; s3.z += e.z; s3.w += e.w;
    }
    for (; r < rend; r += 16) {
      const float4 a = *reinterpret_cast<const float4*>(p + (long)r * lda);
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    }
  }
  part[rl][cg] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                             (s0.w + s1.w) + (s2.w + s3.w));
  __syncthreads();
  if (threadIdx.x < 64) {
    const int g = threadIdx.x >> 2, e = threadIdx.x & 3, cc = blockIdx.x * 64 + threadIdx.x;
    if (cc < cols) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += reinterpret_cast<const float*>(&part[k][g])[e];
      float* o = out + (long)blockIdx.y * out_stride + cc;
      *o = accumulate ? *o + t : t;
    }
  }
}

int colsum(hipStream_t st, const float* A, long lda, float* out, int rows, int cols, int accumulate, float* ws,
           long ws_floats) {
  const int nbc = (cols + 63) / 64;
  if (aligned16(A) && (lda % 4 == 0) && (cols % 4 == 0)) {
    int rsplit = 1;
    if (ws) {
      rsplit = 512 / nbc;
      if (rsplit > rows / 64) rsplit = rows / 64;
      if ((long)rsplit * cols > ws_floats) rsplit = (int)(ws_floats / cols);
      if (rsplit < 1) rsplit = 1;
    }
    int rchunk = ((rows + rsplit - 1) / rsplit + 15) / 16 * 16;
    rsplit = (rows + rchunk - 1) / rchunk;
    if (rsplit > 1) {
      VLN_LAUNCH(colsum4_kernel, dim3(nbc, rsplit), dim3(256), 0, st, A, lda, ws, (long)cols, rows, cols, rchunk, 0);
      VLN_CHECK_LAUNCH("colsum4");
      return reduce_slabs(st, ws, rsplit, cols, out, cols, 1, cols, accumulate);
    }
    VLN_LAUNCH(colsum4_kernel, dim3(nbc, 1), dim3(256), 0, st, A, lda, out, 0L, rows, cols, rchunk, accumulate);
    VLN_CHECK_LAUNCH("colsum4");
    return VLN_OK;
  }
  int rsplit = 1;
  if (ws) {
    rsplit = 256 / nbc;
    if (rsplit > rows / 64) rsplit = rows / 64;
    if ((long)rsplit * cols > ws_floats) rsplit = (int)(ws_floats / cols);
    if (rsplit < 1) rsplit = 1;
  }
  int rchunk = (rows + rsplit - 1) / rsplit;
  if (rchunk < 1) rchunk = 1;
  rsplit = (rows + rchunk - 1) / rchunk;
  if (rsplit < 1) rsplit = 1;
  if (rsplit > 1) {
    VLN_LAUNCH(colsum_kernel, dim3(nbc, rsplit), dim3(256), 0, st, A, lda, ws, (long)cols, rows, cols, rchunk, 0);
    VLN_CHECK_LAUNCH("colsum");
    return reduce_slabs(st, ws, rsplit, cols, out, cols, 1, cols, accumulate);
  }
  VLN_LAUNCH(colsum_kernel, dim3(nbc, 1), dim3(256), 0, st, A, lda, out, 0L, rows, cols, rchunk, accumulate);
  VLN_CHECK_LAUNCH("colsum");
  return VLN_OK;
}

// ---- every bias gradient of a module in one launch ---------------------------------------------------------------
// job: out1[c] (and out2[c]: the LSTM's b_ih and b_hh receive the same sum) (+)= sum_r A[r*lda + c] over the SAME
// `rows` for all jobs.  A workgroup owns 16 columns (4 float4 groups) x 64 row lanes and a row chunk; with one chunk
// the sums go straight to the outputs, else partials go to ws and one second launch finishes every job.
using ColsumJobs = ColsumJobsT<VLN_COLSUM_MAX_JOBS>;
__global__ __launch_bounds__(256) void colsum_grouped_kernel(ColsumJobs a) {
  __shared__ float4 part[64][4];
  colsum_grouped_block(a, (int)blockIdx.x, (int)blockIdx.y, part);
}
__global__ __launch_bounds__(256) void colsum_grouped_finish_kernel(ColsumJobs a) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.total_cols; i += gridDim.x * blockDim.x) {
    int ji = 0;
    while (ji + 1 < a.n && i >= a.col0[ji + 1]) ++ji;
    const vln_colsum_job& q = a.j[ji];
    const int cc = i - a.col0[ji];
    float t = 0.f;
    for (int s = 0; s < a.rsplit; ++s) t += a.ws[(long)s * a.total_cols + i];
    if (q.out1) q.out1[cc] = q.accumulate ? q.out1[cc] + t : t;
    if (q.out2) q.out2[cc] = q.accumulate ? q.out2[cc] + t : t;
  }
}
template <int NJ>
static int colsum_tables(const vln_colsum_job* jobs, int n, int rows, float* ws, long ws_floats, int seg_rows, const int64_t* seg_stride,
                         ColsumJobsT<NJ>& a, int* blocks) {
  if (n <= 0 || n > NJ || rows <= 0) { set_error("colsum_grouped: bad args (n = %d)", n); return VLN_ERR_ARG; }
  a.n = n; a.rows = rows; a.ws = ws; a.seg_rows = seg_stride ? seg_rows : 0;
  for (int i = 0; i < n; ++i) a.seg_stride[i] = seg_stride ? (long)seg_stride[i] : 0;
  int blk = 0, col = 0;
  for (int i = 0; i < n; ++i) {
    const vln_colsum_job& q = jobs[i];
    if (!q.A || q.cols <= 0 || (q.cols & 3) || (q.lda & 3) || !aligned16(q.A) || (!q.out1 && !q.out2)) {
      set_error("colsum_grouped: job %d needs 16-byte aligned rows and cols %% 4 == 0", i);
      return VLN_ERR_ARG;
    }
    a.j[i] = q; a.blk0[i] = blk; a.col0[i] = col;
    blk += (q.cols + 15) / 16; col += q.cols;
  }
  a.blk0[n] = blk; a.col0[n] = col; a.total_cols = col;
  int rsplit = 1;
  if (ws && rows > 1024) {
    rsplit = 1024 / blk;
    if (rsplit > rows / 512) rsplit = rows / 512;
    if ((long)rsplit * col > ws_floats) rsplit = (int)(ws_floats / col);
    if (rsplit < 1) rsplit = 1;
  }
  a.rchunk = ((rows + rsplit - 1) / rsplit + 63) / 64 * 64;
  a.rsplit = (rows + a.rchunk - 1) / a.rchunk;
  *blocks = blk;
  return VLN_OK;
}
int colsum_grouped(hipStream_t st, const vln_colsum_job* jobs, int n, int rows, float* ws, long ws_floats, int seg_rows,
                   const int64_t* seg_stride) {
  ColsumJobs a;
  int blk = 0;
  const int r = colsum_tables<VLN_COLSUM_MAX_JOBS>(jobs, n, rows, ws, ws_floats, seg_rows, seg_stride, a, &blk);
  if (r) return r;
  VLN_LAUNCH(colsum_grouped_kernel, dim3(blk, a.rsplit), dim3(256), 0, st, a);
  if (a.rsplit > 1) VLN_LAUNCH(colsum_grouped_finish_kernel, dim3((a.total_cols + 255) / 256), dim3(256), 0, st, a);
  VLN_CHECK_LAUNCH("colsum_grouped");
  return VLN_OK;
}

// A ride (wgrad_ride.h): the same job tables, for the passenger workgroups of the backward recurrence launch.  false = these
// jobs do not ride (exact-fp32 / unaligned operands, a row split that needs a reduce launch, too many jobs, a small workspace):
// the caller issues them as their own launches.
bool wgrad_ride_prepare(const vln_wgrad_job* jobs, int n, int Mt, int precision, const vln_colsum_job* cjobs, int nc, float* ws,
                        long ws_floats, WgradRideArgs* out) {
  if (n <= 0 || n > kRideWgradJobs || nc < 0 || nc > kRideColsumJobs || Mt <= 0 || Mt > 1024 || !ws) return false;
  if (precision != 1 && precision != 2) return false;
  if (g_tunable[6] == 1 || g_tunable[6] == 2) return false;
  for (int i = 0; i < n; ++i) {
    const vln_wgrad_job& q = jobs[i];
    if (!(q.dy && q.x && q.dw && aligned16(q.dy) && aligned16(q.x) && (q.ld_dy % 4 == 0) && (q.ld_x % 4 == 0) && (q.N % 4 == 0) && (q.K % 4 == 0) &&
          q.N > 0 && q.K > 0)) return false;
  }
  WgradRideArgs& r = *out;
  r.terms = precision == 2 ? 1 : 3;
  if (!wgrad_packed_tables<kRideWgradJobs>(jobs, n, Mt, ws, ws_floats, r.terms, 0, nullptr, nullptr, r.pk, r.g, &r.pack_blocks, nullptr)) return false;
  if (r.g.msplit != 1) return false;
  r.tiles = r.g.per_xcd * 8;
  r.cs_blocks = 0; r.cs.n = 0;
  if (nc > 0) {
    if (colsum_tables<kRideColsumJobs>(cjobs, nc, Mt, nullptr, 0, 0, nullptr, r.cs, &r.cs_blocks) != VLN_OK) return false;
    if (r.cs.rsplit != 1) return false;
  }
  r.on = 1;
  return true;
}

// ---------------------------------------------------------------------------
// weight shadows: transposed / cast copies refreshed once per optimizer step
// ---------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* W, long ldw, TO* Wt, long ldt, int N,
                                                             int K) {
  __shared__ float tile[64][65];
  const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int n = n0 + r, k = k0 + tx;
    tile[r][tx] = (n < N && k < K) ? W[(long)n * ldw + k] : 0.0f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int k = k0 + r, n = n0 + tx;
    if (k < K && n < N) Elt<TO>::st(Wt + (long)k * ldt + n, tile[tx][r]);
  }
}
int transpose_cast(hipStream_t st, const float* W, long ldw, void* Wt, int out_type, long ldt, int N, int K) {
  dim3 grid((K + 63) / 64, (N + 63) / 64), block(256);
  if (out_type == W_BF16)
    VLN_LAUNCH(transpose_cast_kernel<bf16_raw>, grid, block, 0, st, W, ldw, (bf16_raw*)Wt, ldt, N, K);
  else
    VLN_LAUNCH(transpose_cast_kernel<float>, grid, block, 0, st, W, ldw, (float*)Wt, ldt, N, K);
  VLN_CHECK_LAUNCH("transpose_cast");
  return VLN_OK;
}

template <typename TO>
__global__ __launch_bounds__(256) void cast_copy_kernel(const float* W, long ldw, TO* out, long ldo, int rows,
                                                        int cols) {
  const long total = (long)rows * cols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int r = (int)(e / cols), c = (int)(e % cols);
    Elt<TO>::st(out + (long)r * ldo + c, W[(long)r * ldw + c]);
  }
}
int cast_copy(hipStream_t st, const float* W, long ldw, void* out, int out_type, long ldo, int rows, int cols) {
  long total = (long)rows * cols;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (out_type == W_BF16)
    VLN_LAUNCH(cast_copy_kernel<bf16_raw>, dim3(blocks), dim3(256), 0, st, W, ldw, (bf16_raw*)out, ldo, rows, cols);
  else
    VLN_LAUNCH(cast_copy_kernel<float>, dim3(blocks), dim3(256), 0, st, W, ldw, (float*)out, ldo, rows, cols);
  VLN_CHECK_LAUNCH("cast_copy");
  return VLN_OK;
}

// ---- all shadows of a module in ONE launch (shadow_bodies.h) --------------------------------------------------------
using ShadowJobs = ShadowJobsT<VLN_SHADOW_MAX_JOBS>;
__global__ __launch_bounds__(256) void shadow_refresh_kernel(ShadowJobs a) {
  __shared__ float lds[64][65];
  shadow_block<false>(a, (int)blockIdx.x, lds);
}
int shadow_refresh(hipStream_t st, const vln_shadow_job* jobs, int n) {
  for (int base = 0; base < n; base += VLN_SHADOW_MAX_JOBS) {
    ShadowJobs a; int t = 0;
    int r = shadow_jobs(jobs + base, (n - base < VLN_SHADOW_MAX_JOBS) ? n - base : VLN_SHADOW_MAX_JOBS, &a, &t); if (r) return r;
    VLN_LAUNCH(shadow_refresh_kernel, dim3(t), dim3(256), 0, st, a);
  }
  VLN_CHECK_LAUNCH("shadow_refresh");
  return VLN_OK;
}

// ---- the iteration's prologue as ONE launch ------------------------------------------------------------------------------------------
// Block ranges: [0, nf) pull the batch out of pinned host memory (PCIe-bound, ~39 us alone), [nf, nf + tiles) refresh the weight
// shadows of the modules whose parameters the previous optimizer step changed (HBM-bound, 21 us as two launches), the last block
// ticks the device clock.  None of the three depends on another; as separate launches they were 70 us of the iteration's dependent
// chain (pull 39, tick 4.5, refreshes 5.8 + 15.6 and three boundaries).
__global__ __launch_bounds__(256) void prologue_kernel(FetchArgs f, int nf, ShadowJobs sj, int tiles, TickArgs tk) {
  __shared__ float lds[64][65];
  const int b = (int)blockIdx.x;
  if (b < nf) { host_fetch_body(f, b, nf, (int)threadIdx.x); return; }
  if (b < nf + tiles) { shadow_block<false>(sj, b - nf, lds); return; }
  tick_body(tk, (int)threadIdx.x);
}

}  // namespace vln

extern "C" int vln_prologue(const uint64_t* slots_dev, int ring, uint64_t* seq, uint32_t* done, void* dst, int64_t nbytes,
                            const vln_tick_item* ticks, int n_ticks, const vln_shadow_job* jobs, int n_jobs, vln_stream_t s) {
  using namespace vln;
  if (n_jobs < 0 || n_jobs > VLN_SHADOW_MAX_JOBS || n_ticks < 0) { set_error("vln_prologue: 0..%d shadow jobs, >= 0 tick items", VLN_SHADOW_MAX_JOBS); return VLN_ERR_ARG; }
  FetchArgs f{}; int nf = 0;
  if (slots_dev) { int r = fetch_args(slots_dev, ring, seq, done, dst, nbytes, &f, &nf); if (r) return r; }
  ShadowJobs sj{}; int tiles = 0;
  if (n_jobs) { int r = shadow_jobs(jobs, n_jobs, &sj, &tiles); if (r) return r; }
  TickArgs tk{};
  if (n_ticks) { int r = tick_args(ticks, n_ticks, &tk); if (r) return r; }
  const int blocks = nf + tiles + (n_ticks ? 1 : 0);
  if (blocks <= 0) return VLN_OK;
  VLN_LAUNCH(prologue_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, f, nf, sj, tiles, tk);
  VLN_CHECK_LAUNCH("prologue");
  return VLN_OK;
}
