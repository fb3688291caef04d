// Where do the workgroups of a launch land?  N workgroups of T threads with L bytes of dynamic LDS each spin for ~30 us and record
// (XCC, SE, SH, CU) from the hardware-id registers plus their start / end time: how many distinct CUs a launch of 16 ... 256 big
// workgroups really occupies, and whether two of them were put on one CU (back to back) while other CUs stayed empty.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/wg_placement.hip -o /tmp/wg_placement && /tmp/wg_placement
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <set>
#include <map>
#include <algorithm>
__global__ void k(unsigned long long *out, int spin_ticks) {
  extern __shared__ int sm[];
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
    const unsigned long long t1 = wall_clock64();
    sm[0] = (int)hw;
    out[blockIdx.x * 4 + 0] = hw;
    out[blockIdx.x * 4 + 1] = xcc;
    out[blockIdx.x * 4 + 2] = t0;
    out[blockIdx.x * 4 + 3] = t1;
  }
  __syncthreads();
}
int main() {
  unsigned long long *d;
  hipMalloc(&d, 4096 * 4 * 8);
  const int configs[][2] = {{512, 144 * 1024}, {256, 128 * 1024}, {512, 64 * 1024}, {256, 32 * 1024}};
  for (auto &c : configs) {
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("== %d threads, %d KB LDS per workgroup\n", c[0], c[1] / 1024);
    for (int n : {16, 64, 96, 104, 112, 120, 128, 160, 192, 224, 256, 320, 512}) {
      hipMemset(d, 0, 4096 * 4 * 8);
      hipLaunchKernelGGL(k, dim3(n), dim3(c[0]), c[1], 0, d, 3000);     // 3000 ticks of 100 MHz = 30 us
      std::vector<unsigned long long> h(n * 4);
      if (hipMemcpy(h.data(), d, n * 32, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 2; }
      std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned long long>>> cus;
      std::map<unsigned, int> per_xcc;
      unsigned long long tmin = ~0ull, tmax = 0;
      for (int i = 0; i < n; i++) {
        const unsigned hw = (unsigned)h[i * 4], xcc = (unsigned)h[i * 4 + 1] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cus[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back({h[i * 4 + 2], h[i * 4 + 3]});
        per_xcc[xcc]++;
        tmin = std::min(tmin, h[i * 4 + 2]);
        tmax = std::max(tmax, h[i * 4 + 3]);
      }
      int shared = 0, most = 0;
      for (auto &e : cus) {
        if (e.second.size() > 1) shared++;
        most = std::max(most, (int)e.second.size());
      }
      printf("  %4d workgroups: %3d distinct CUs, %3d CUs ran more than one (max %d on one CU), launch %.1f us;  per XCC:", n, (int)cus.size(),
             shared, most, (tmax - tmin) / 100.0);
      for (auto &e : per_xcc) printf(" %d", e.second);
      printf("\n");
    }
  }
  return 0;
}
