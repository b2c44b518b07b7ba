// Stage-by-stage timing of the persistent (bi)LSTM recurrence kernels: one workgroup records s_memrealtime (100 MHz)
// at the VLN_STAMP points of encoder_persist.h for every time step.  Standalone program:
//   make -C curriculum-learning-for-vln_amd/csrc && scripts/build_lstm_probe.sh && gpurun -- ./scripts/lstm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

__device__ unsigned long long g_stamps[128 * 8];
#ifndef PROBE_X
#define PROBE_X 3
#endif
#define VLN_STAMP(k)                                                                          \
  do {                                                                                        \
    if (threadIdx.x == 0 && blockIdx.x == PROBE_X && blockIdx.y == 0 && blockIdx.z == 0)      \
      g_stamps[step * 8 + (k)] = __builtin_amdgcn_s_memrealtime();                            \
  } while (0)

#include "../curriculum-learning-for-vln_amd/csrc/encoder.hip"

static void fill(std::vector<float>& v, float scale) { for (auto& x : v) x = scale * ((rand() & 0xffff) / 32768.f - 1.f); }
static unsigned short bf16_of(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x8000u) >> 16); }

static void report(const char* name, int L, int nst, const char* const* lab) {
  std::vector<unsigned long long> h(128 * 8);
  hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamps), h.size() * 8);
  double acc[8] = {0}, tot = 0; int n = 0;
  for (int s = 5; s < L - 5; ++s) {
    for (int k = 1; k < nst; ++k) acc[k] += (double)(h[s * 8 + k] - h[s * 8 + k - 1]) * 0.01;
    tot += (double)(h[(s + 1) * 8] - h[s * 8]) * 0.01;
    ++n;
  }
  printf("%s: per-step %.2f us |", name, tot / n);
  for (int k = 1; k < nst; ++k) printf(" %s %.2f", lab[k], acc[k] / n);
  printf(" | loop-around %.2f\n", (tot - (acc[1] + acc[2] + acc[3] + acc[4] + acc[5])) / n);
}

int main(int argc, char** argv) {
  const int B = 64, L = 80, Hd = 256, dirs = 2, G = dirs * 4 * Hd, Y = dirs * Hd;
  const int wtype = (argc > 1 && !strcmp(argv[1], "fp32")) ? VLN_F32 : VLN_BF16;
  srand(1);
  std::vector<float> xproj((size_t)L * B * G), whh((size_t)dirs * 4 * Hd * Hd), dy((size_t)L * B * Y);
  fill(xproj, 1.f); fill(whh, 0.06f); fill(dy, 0.1f);
  std::vector<int> len(B);
  for (int b = 0; b < B; ++b) len[b] = L - (b * (L - 8)) / B;       // sorted descending, 80 .. ~9
  float *d_x, *d_hp, *d_cp, *d_y, *d_act, *d_tc, *d_hc, *d_cc, *d_dy, *d_dg, *d_dh, *d_dc; void *d_w, *d_wt, *d_sync; int* d_len;
  const size_t st = (size_t)dirs * L * B * Hd * 4;
  hipMalloc(&d_x, xproj.size() * 4); hipMalloc(&d_hp, st); hipMalloc(&d_cp, st); hipMalloc(&d_y, (size_t)L * B * Y * 4);
  hipMalloc(&d_act, (size_t)L * B * G * 4); hipMalloc(&d_tc, (size_t)L * B * Y * 4); hipMalloc(&d_hc, B * Y * 4); hipMalloc(&d_cc, B * Y * 4);
  hipMalloc(&d_dy, dy.size() * 4); hipMalloc(&d_dg, (size_t)L * B * G * 4); hipMalloc(&d_dh, dirs * B * Hd * 4); hipMalloc(&d_dc, dirs * B * Hd * 4);
  hipMalloc(&d_len, B * 4);
  const long sync_bytes = vln_lstm_sync_ws_bytes(B, Hd, dirs);
  hipMalloc(&d_sync, sync_bytes);
  hipMemset(d_sync, 0, sync_bytes);
  const size_t wel = whh.size();
  hipMalloc(&d_w, wel * 4); hipMalloc(&d_wt, wel * 4);
  std::vector<float> wt(wel);                                         // [dirs][Hd][4Hd] transpose for the backward
  for (int d = 0; d < dirs; ++d) for (int r = 0; r < 4 * Hd; ++r) for (int c = 0; c < Hd; ++c)
    wt[((size_t)d * Hd + c) * 4 * Hd + r] = whh[((size_t)d * 4 * Hd + r) * Hd + c];
  if (wtype == VLN_BF16) {
    std::vector<unsigned short> a(wel), b(wel);
    for (size_t i = 0; i < wel; ++i) { a[i] = bf16_of(whh[i]); b[i] = bf16_of(wt[i]); }
    hipMemcpy(d_w, a.data(), wel * 2, hipMemcpyHostToDevice); hipMemcpy(d_wt, b.data(), wel * 2, hipMemcpyHostToDevice);
  } else {
    hipMemcpy(d_w, whh.data(), wel * 4, hipMemcpyHostToDevice); hipMemcpy(d_wt, wt.data(), wel * 4, hipMemcpyHostToDevice);
  }
  hipMemcpy(d_x, xproj.data(), xproj.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_dy, dy.data(), dy.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_len, len.data(), B * 4, hipMemcpyHostToDevice);
  hipMemset(d_dh, 0, dirs * B * Hd * 4); hipMemset(d_dc, 0, dirs * B * Hd * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // protocol 2 = counter + payload hand-off (round 1), 1 = data-tagged granules; interleaved on the same box
  for (int round = 0; round < 6; ++round) {
  const int proto = (round % 3 == 0) ? 2 : (round % 3 == 1 ? 1 : 3);
  vln_set_persistent(proto);
  printf("---- protocol %s\n", proto == 2 ? "counter (fwd + bwd)" : proto == 1 ? "default: granules fwd, counter bwd" : "granules (fwd + bwd)");
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    int r = vln_lstm_seq_fwd(d_x, d_w, wtype, d_len, d_hp, d_cp, d_y, d_act, d_tc, d_hc, d_cc, B, L, Hd, dirs, nullptr, nullptr, d_sync, sync_bytes, -1, nullptr, nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r) { printf("fwd failed: %s\n", vln_last_error_string()); return 1; }
    if (rep == 3) { printf("fwd launch %.1f us  ", ms * 1e3); { const char* lab[] = {"", "wait/sweep", "load->LDS", "mfma", "pointwise+handoff store", "drain+arrive / bookkeeping"}; report("fwd", L, 6, lab); } }
  }
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(d_dh, 0, dirs * B * Hd * 4); hipMemset(d_dc, 0, dirs * B * Hd * 4);
    hipEventRecord(e0, 0);
    int r = vln_lstm_seq_bwd(d_dy, d_wt, wtype, d_len, d_act, d_tc, d_cp, d_dg, d_dh, d_dc, nullptr, nullptr, B, L, Hd, dirs, d_sync, sync_bytes, -1, nullptr, nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r) { printf("bwd failed: %s\n", vln_last_error_string()); return 1; }
    if (rep == 3) {
      printf("bwd launch %.1f us  ", ms * 1e3);
      // the backward walks step = L-1 .. 0: flip the stamp rows so report() sees increasing time
      std::vector<unsigned long long> h(128 * 8), f(128 * 8);
      hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamps), h.size() * 8);
      for (int s = 0; s < L; ++s) for (int k = 0; k < 8; ++k) f[s * 8 + k] = h[(L - 1 - s) * 8 + k];
      hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), f.data(), f.size() * 8);
      const char* lab[] = {"", "wait", "load+reduce", "pointwise+tile", "mfma+partial stores", "drain+arrive"};
      report("bwd", L, 6, lab);
    }
  }
  {
    std::vector<float> dg((size_t)L * B * G);
    hipMemcpy(dg.data(), d_dg, dg.size() * 4, hipMemcpyDeviceToHost);
    double s1 = 0, s2 = 0; for (float v : dg) { s1 += v; s2 += (double)v * v; }
    unsigned stw[64]; hipMemcpy(stw, d_sync, 256, hipMemcpyDeviceToHost);
    printf("dgates sum %.6g sumsq %.6g  status %u\n", s1, s2, stw[32]);
  }
  float chk[4]; hipMemcpy(chk, d_hc, 16, hipMemcpyDeviceToHost);
  printf("hcat[0..3] = %g %g %g %g\n", chk[0], chk[1], chk[2], chk[3]);
  }
  return 0;
}
