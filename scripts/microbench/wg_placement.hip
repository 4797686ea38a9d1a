// Where do the workgroups of a k_step-shaped launch land?  4096 single-wave workgroups with k_step's footprint (128 VGPRs = 4 waves per SIMD,
// ~10 KB of LDS each: 16 per CU), all resident at once; each records the hardware ids of the wave slot it got.  Prints, per SIMD, which
// blockIdx values share it -- what an arena -> workgroup order must know if it wants to keep two expensive arenas off the same SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o build_variants/wg_placement scripts/microbench/wg_placement.hip && build_variants/wg_placement
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>
#include <algorithm>
extern __shared__ unsigned char lds[];
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) probe(unsigned *out, long long spin) {
  // HW_REG_HW_ID = 4: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx9 layout); XCC_ID = 20
  unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
  unsigned xcc = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20);
  long long t0 = wall_clock64();
  volatile unsigned char *l = lds; l[threadIdx.x] = 1;
  while (wall_clock64() - t0 < spin) { }
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main(int argc, char **argv) {
  int n = argc > 1 ? atoi(argv[1]) : 4096; int ldsb = argc > 2 ? atoi(argv[2]) : 9904;
  unsigned *d; hipMalloc(&d, n * 8); hipMemset(d, 0xff, n * 8);
  hipLaunchKernelGGL(probe, dim3(n), dim3(64), ldsb, 0, d, 30000LL);   // 100 MHz wall clock: 300 us
  hipDeviceSynchronize();
  std::vector<unsigned> h(2 * n); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> by_simd;
  for (int b = 0; b < n; b++) {
    unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    by_simd[(xcc << 12) | (se << 8) | (sh << 7) | (cu << 2) | simd].push_back(b);
  }
  printf("%d workgroups on %zu SIMDs\n", n, by_simd.size());
  std::map<size_t, int> hist; for (auto &kv : by_simd) hist[kv.second.size()]++;
  for (auto &kv : hist) printf("  %d SIMDs hold %zu workgroups\n", kv.second, kv.first);
  int shown = 0;
  for (auto &kv : by_simd) { if (shown++ >= 12) break; printf("  xcc %u se %u sh %u cu %2u simd %u:", kv.first >> 12, (kv.first >> 8) & 7, (kv.first >> 7) & 1, (kv.first >> 2) & 0x1f, kv.first & 3); for (int b : kv.second) printf(" %d", b); printf("\n"); }
  // the stride structure: for workgroup b, which other b' share its SIMD?  print differences for a few
  for (int b : {0, 1, 2, 8, 100, 1000}) { if (b >= n) continue; for (auto &kv : by_simd) if (std::find(kv.second.begin(), kv.second.end(), b) != kv.second.end()) { printf("  blockIdx %d shares its SIMD with:", b); for (int o : kv.second) if (o != b) printf(" %d", o); printf("\n"); } }
  return 0;
}
