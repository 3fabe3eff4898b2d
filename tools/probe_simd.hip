// which SIMD does wave w of a 512-thread workgroup land on?  (HW_ID register, gfx9: simd_id bits 5:4, cu_id 11:8, wave_id 3:0)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void k(unsigned* out) {
  unsigned id = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d; hipMalloc(&d, 4 * 8 * 4 * 4);
  hipLaunchKernelGGL(k, dim3(4), dim3(512), 100 * 1024, 0, d);
  unsigned h[32]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 4; ++b) { for (int w = 0; w < 8; ++w) printf("b%d w%d: wave_id %u simd %u cu %u | ", b, w, h[b*8+w] & 15, (h[b*8+w] >> 4) & 3, (h[b*8+w] >> 8) & 15); printf("\n"); }
  return 0;
}
