// Two waves per SIMD: does the vector-ALU work of one wave run under the fp32 MFMAs of the other?  (gfx950)
// probe_fill.hip showed that ONE wave cannot hide its own VALU instructions behind its MFMAs.  Here a 512-thread workgroup puts
// two waves on every SIMD: waves 0-3 issue a stream of v_mfma_f32_32x32x2_f32, waves 4-7 a stream of independent v_fma_f32 /
// v_exp_f32 / packed fma.  Each side is timed alone and together (s_memtime per wave): if the pipes are independent both keep their
// stand-alone time.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_pair tools/probe_pair.hip && tools/probe_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
#define SB __builtin_amdgcn_sched_barrier(0)
enum { V_FMA = 0, V_EXP, V_PK, V_MFMA4, V_MFMA_TOO };

template <int KIND>
__global__ __launch_bounds__(512, 1) void k_pair(float* out, long long* cyc, int iters, int run_mfma, int run_valu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2] = {(f32x16)(0.f), (f32x16)(0.f)};
  float a0 = 1.0f + lane * 1e-3f, b0 = 0.5f;
  float x[16];
  f32x2 y[8];
  for (int i = 0; i < 16; ++i) x[i] = 0.1f * i + lane;
  for (int i = 0; i < 8; ++i) y[i] = f32x2{0.1f * i, 1.0f + lane};
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 small = (f32x4)(0.f);
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (run_mfma)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m) { acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[m & 1], 0, 0, 0); SB; }
      }
  } else if (run_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 1024; ++k) {            // 1024 x 4 cycles = the 4096 cycles of the other wave's 64 MFMAs
        const int e = k & 15;
        if (KIND == V_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[e]) : "v"(a0), "v"(b0));
        if (KIND == V_EXP) { if ((k & 3) == 0) asm volatile("v_exp_f32 %0, %1" : "=v"(x[e]) : "v"(x[(e + 7) & 15])); }      // quarter rate: 256 x 16 cycles
        if (KIND == V_PK) { if ((k & 1) == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y[e & 7]) : "v"(y[(e + 3) & 7]), "v"(y[(e + 5) & 7])); }   // 512 x 8 cycles
        if (KIND == V_MFMA4) { if ((k & 1) == 0) small = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, small, 0, 0, 0); }          // 512 x 8 cycles of the matrix pipe
        if (KIND == V_MFMA_TOO) { if ((k & 15) == 0) acc[(k >> 4) & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[(k >> 4) & 1], 0, 0, 0); }
        SB;
      }
    }
  }
  SB;
  long long t1 = __builtin_readcyclecounter();
  float s = small.x;
  for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int i = 0; i < 16; ++i) s += x[i];
  for (int i = 0; i < 8; ++i) s += y[i].x + y[i].y;
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 200;
  long long h[3][8];
  for (int mode = 0; mode < 3; ++mode) {
    const int rm = mode != 1, rv = mode != 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k_pair<KIND>, dim3(256), dim3(512), 0, 0, out, cyc, iters, rm, rv);
      CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h[mode], cyc, sizeof(long long) * 8, hipMemcpyDeviceToHost));
  }
  printf("%-34s per 64 MFMAs / per 1024 VALU slots:  MFMA alone %6.0f   VALU alone %6.0f   together: MFMA %6.0f  VALU %6.0f\n", name,
         (double)h[0][0] / iters, (double)h[1][4] / iters, (double)h[2][0] / iters, (double)h[2][4] / iters);
}

int main() {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 64));
  run<V_FMA>("v_fma_f32 x 1024", out, cyc);
  run<V_EXP>("v_exp_f32 x 256", out, cyc);
  run<V_PK>("v_pk_fma_f32 x 512", out, cyc);
  run<V_MFMA4>("v_mfma_f32_4x4x1 x 512", out, cyc);
  run<V_MFMA_TOO>("v_mfma_f32_32x32x2 x 64 (both)", out, cyc);
  return 0;
}
