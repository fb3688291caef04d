// Micro-check for the fp32 conv on one wave per SIMD (f32_conv4_sk_kernel): does an LDS-DMA through a buffer resource
// (buffer_load_dwordx4 ... offen lds) write ZEROS for lanes whose offset is out of range, as the register form does?
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/buffer_lds_oob.hip -o gpurun_out/buffer_lds_oob && gpurun_out/buffer_lds_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void k(const float *p, int n, float *out) {
  extern __shared__ float sm[];
  for (int i = threadIdx.x; i < 512; i += 64) sm[i] = -7.f;         // garbage the DMA has to overwrite
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, n * 4, 0x00020000);
  unsigned voff = threadIdx.x * 16;
  if (threadIdx.x % 3 == 1) voff = 0x80000000u;                      // out of range -> zeros expected
  const unsigned soff = 1024;                                        // uniform offset: floats 256 ..
  const unsigned m0v = (unsigned)(uintptr_t)sm;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(voff), "s"(r), "s"(soff) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = sm[i];
}
int main() {
  const int n = 4096;
  std::vector<float> h(n);
  for (int i = 0; i < n; i++) h[i] = 1.f + i;
  float *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, n, o);
  std::vector<float> r(256);
  if (hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 2; }
  int bad = 0;
  for (int l = 0; l < 64; l++)
    for (int e = 0; e < 4; e++) {
      const float want = (l % 3 == 1) ? 0.f : 1.f + 256 + l * 4 + e;
      if (r[l * 4 + e] != want) { if (bad < 8) printf("lane %d elem %d: got %g want %g\n", l, e, r[l * 4 + e], want); bad++; }
    }
  printf("buffer_load_dwordx4 ... offen lds with out-of-range lanes: %s (%d mismatches)\n", bad ? "UNEXPECTED" : "zeros written, in-range lanes copied", bad);
  return bad ? 1 : 0;
}
