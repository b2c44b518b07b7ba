// Which CUs / XCDs does a CU-masked stream (hipExtStreamCreateWithCUMask) dispatch to on MI355X?
//   hipcc -O2 --offload-arch=gfx950 scripts/cumask_probe.hip -o scripts/cumask_probe && gpurun -- ./scripts/cumask_probe
// Every workgroup records XCC_ID and HW_ID (SE / CU fields); the host prints, per mask, how many distinct (xcc, se, cu) were
// used and the per-XCC workgroup counts.  Used to lay out the "recurrence CUs" / "streaming CUs" split.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <set>
#include <vector>

__global__ void whoami(unsigned* out, int spin) {
  unsigned xcc = 0, hw = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = hw; }
  // stay resident a little so that the grid spreads over every CU the mask allows
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) {}
}

static void run(const char* name, const std::vector<uint32_t>& mask, bool masked) {
  hipStream_t st;
  hipError_t e = masked ? hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) : hipStreamCreate(&st);
  if (e != hipSuccess) { printf("%s: stream create failed: %s\n", name, hipGetErrorString(e)); return; }
  const int n = 2048;
  unsigned* d; hipMalloc(&d, n * 8);
  hipLaunchKernelGGL(whoami, dim3(n), dim3(256), 0, st, d, 2000);     // 20 us resident
  hipStreamSynchronize(st);
  std::vector<unsigned> h(n * 2);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per_xcc;
  std::set<unsigned long long> cus;
  std::map<unsigned, std::set<unsigned>> cu_of_xcc;
  for (int i = 0; i < n; ++i) {
    const unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_xcc[xcc]++;
    cus.insert(((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu);
    cu_of_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  printf("%-28s distinct CUs %3zu | per XCC wgs:", name, cus.size());
  for (auto& kv : per_xcc) printf(" x%u:%d(%zu cu)", kv.first, kv.second, cu_of_xcc[kv.first].size());
  printf("\n");
  hipFree(d);
  hipStreamDestroy(st);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device: %s, %d CUs\n", p.name, p.multiProcessorCount);
  run("no mask", {}, false);
  const int words = 8;                                   // 256 bits
  auto mk = [&](auto pred) { std::vector<uint32_t> m(words, 0u); for (int i = 0; i < 256; ++i) if (pred(i)) m[i / 32] |= 1u << (i % 32); return m; };
  run("all 256 bits", mk([](int) { return true; }), true);
  run("bits 0..127", mk([](int i) { return i < 128; }), true);
  run("bits 128..255", mk([](int i) { return i >= 128; }), true);
  run("bits 0..31", mk([](int i) { return i < 32; }), true);
  run("bits 32..63", mk([](int i) { return i >= 32 && i < 64; }), true);
  run("even bits", mk([](int i) { return (i & 1) == 0; }), true);
  run("odd bits", mk([](int i) { return (i & 1) == 1; }), true);
  run("bits i%8 == 0", mk([](int i) { return i % 8 == 0; }), true);
  run("bits i%8 == 3", mk([](int i) { return i % 8 == 3; }), true);
  run("bits i%16 < 8", mk([](int i) { return i % 16 < 8; }), true);
  run("bits i%16 >= 8", mk([](int i) { return i % 16 >= 8; }), true);
  run("bits (i/8)%2 == 0", mk([](int i) { return (i / 8) % 2 == 0; }), true);
  run("bits i%64 < 32", mk([](int i) { return i % 64 < 32; }), true);
  run("bit 0 only", mk([](int i) { return i == 0; }), true);
  run("bit 1 only", mk([](int i) { return i == 1; }), true);
  run("bit 8 only", mk([](int i) { return i == 8; }), true);
  run("bit 32 only", mk([](int i) { return i == 32; }), true);
  return 0;
}
