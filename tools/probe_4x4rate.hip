// Issue rate of v_mfma_f32_4x4x1_16b_f32 by the number of accumulators it rotates over (gfx950, one wave per SIMD):
// the narrow phases of the fused kernels (output layer, gW3, gW1) are priced at 8 cycles per instruction in DESIGN.md.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_4x4rate tools/probe_4x4rate.hip && tools/probe_4x4rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
#define SB __builtin_amdgcn_sched_barrier(0)

template <int NACC, int MIX>     // MIX: one 32x32x2 after every 8 narrow ones (how the kernels' drains look)
__global__ __launch_bounds__(256, 1) void k_rate(float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[8];
  for (int a = 0; a < 8; ++a) acc[a] = (f32x4)(0.f);
  f32x16 big = (f32x16)(0.f);
  float a0 = 1.0f + lane * 1e-3f, b0 = 0.5f;
  long long t0 = 0;
  for (int it = 0; it < iters + 1; ++it) {
    if (it == 1) { SB; t0 = __builtin_readcyclecounter(); SB; }
#pragma unroll
    for (int m = 0; m < 256; ++m) {
      acc[m % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, acc[m % NACC], 0, 0, 0);
      SB;
      if (MIX && (m & 7) == 7) { big = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, big, 0, 0, 0); SB; }
    }
  }
  SB; long long t1 = __builtin_readcyclecounter(); SB;
  float s = 0.f;
  for (int a = 0; a < 8; ++a) s += acc[a].x + acc[a].y;
  for (int r = 0; r < 16; ++r) s += big[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int MIX>
void run(float* out, long long* cyc) {
  const int iters = 200;
  hipLaunchKernelGGL((k_rate<NACC, MIX>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  CK(hipDeviceSynchronize());
  long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
  const double per = (double)h / iters / 256.0;
  if (MIX) printf("%d accumulators + a 32x32x2 after every 8: %6.2f cycles per 4x4x1 slot (8 narrow + 1 wide = %6.1f; priced 8 x 8 + 64 = 128)\n", NACC, per, per * 8);
  else printf("%d accumulator(s): %6.2f cycles per v_mfma_f32_4x4x1\n", NACC, per);
}

int main() {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 64));
  run<1, 0>(out, cyc); run<2, 0>(out, cyc); run<4, 0>(out, cyc); run<8, 0>(out, cyc);
  run<4, 1>(out, cyc); run<8, 1>(out, cyc);
  return 0;
}
