// What does an instruction cost when it sits between two fp32 MFMAs of ONE wave per SIMD?  (gfx950)
// The Fisher-vector-product kernel runs one 64-lane wave per SIMD; everything that is not an MFMA is issued in the 64-cycle
// shadow of a v_mfma_f32_32x32x2_f32.  This probe times a loop of 64 such MFMAs (two alternating accumulators, like the
// kernel's layer products) with K "filler" instructions of one kind after every MFMA -- independent v_fma_f32, a dependent
// fma -> mul pair, v_accvgpr moves, ds_read_b128 / ds_write_b32 / ds_write2_b32 with their s_waitcnt, s_nop -- and prints
// cycles per MFMA (s_memtime): 64.0 = free, anything above is what the filler costs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_fill tools/probe_fill.hip && tools/probe_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
#define SB __builtin_amdgcn_sched_barrier(0)

enum { F_NONE = 0, F_FMA, F_FMAMUL_DEP, F_ACCREAD, F_DSREAD128, F_DSWRITE32, F_DSWRITE2, F_NOP, F_MUL_ON_ACC, F_WAITCNT, F_SALU, F_M4x4 };

template <int KIND, int K, int NACC>
__global__ __launch_bounds__(256, 1) void k_fill(float* out, long long* cyc, int iters) {
  __shared__ float lds[4 * 64 * 40];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* my = lds + wave * 64 * 40;
  for (int i = lane; i < 64 * 40; i += 64) my[i] = 0.001f * i;
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) acc[a] = (f32x16)(0.f);
  f32x16 side = (f32x16)(1.0f);
  f32x4 small = (f32x4)(0.f);
  float a0 = 1.0f + lane * 1e-3f, b0 = 0.5f;
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = 0.1f * i + lane;
  f32x4 ld = (f32x4)(0.f);
  long long t0 = 0, t1 = 0;
  for (int it = 0; it < iters + 1; ++it) {
    if (it == 1) { SB; t0 = __builtin_readcyclecounter(); SB; }
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[m % NACC], 0, 0, 0);
      SB;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int e = (m * K + k) & 15;
        if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[e]) : "v"(a0), "v"(b0));
        if (KIND == F_FMAMUL_DEP) { if (k & 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[e]) : "v"(x[(e + 15) & 15])); else asm volatile("v_fma_f32 %0, %1, %1, 1.0" : "=v"(x[e]) : "v"(a0)); }
        if (KIND == F_ACCREAD) asm volatile("v_mov_b32 %0, %1" : "=v"(x[e]) : "v"(x[(e + 5) & 15]));
        if (KIND == F_DSREAD128) { asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"((int)(((lane * 36 + 4 * e) & 2047) * 4 + wave * 64 * 40 * 4))); }
        if (KIND == F_DSWRITE32) asm volatile("ds_write_b32 %0, %1" :: "v"((int)((lane + 36 * e) * 4 + wave * 64 * 40 * 4)), "v"(x[e]));
        if (KIND == F_DSWRITE2) asm volatile("ds_write2_b32 %0, %1, %2 offset1:36" :: "v"((int)((lane + 72 * e) * 4 + wave * 64 * 40 * 4)), "v"(x[e]), "v"(x[(e + 1) & 15]));
        if (KIND == F_NOP) asm volatile("s_nop 0");
        if (KIND == F_MUL_ON_ACC) side[e] *= x[e];                       // VALU on registers of an accumulator-sized tuple
        if (KIND == F_WAITCNT) asm volatile("s_waitcnt lgkmcnt(0)");
        if (KIND == F_SALU) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
        if (KIND == F_M4x4) small = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, small, 0, 0, 0);
        SB;
      }
      if (KIND == F_DSREAD128 && K > 0) { asm volatile("s_waitcnt lgkmcnt(0)"); x[0] += ld.x; SB; }
    }
  }
  SB; t1 = __builtin_readcyclecounter(); SB;
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int i = 0; i < 16; ++i) s += x[i] + side[i];
  s += small.x + ld.y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// bursts: 8 MFMAs, then B vector-ALU instructions in a row (what the Fisher-vector-product kernel does since r03)
template <int KIND, int B>
__global__ __launch_bounds__(256, 1) void k_burst(float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[2];
  for (int a = 0; a < 2; ++a) acc[a] = (f32x16)(0.f);
  float a0 = 1.0f + lane * 1e-3f, b0 = 0.5f;
  float x[32];
  for (int i = 0; i < 32; ++i) x[i] = 0.1f * i + lane;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 y[16];
  for (int i = 0; i < 16; ++i) y[i] = f32x2{0.1f * i, 1.0f + lane};
  long long t0 = 0, t1 = 0;
  for (int it = 0; it < iters + 1; ++it) {
    if (it == 1) { SB; t0 = __builtin_readcyclecounter(); SB; }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
#pragma unroll
      for (int m = 0; m < 8; ++m) { acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[m & 1], 0, 0, 0); SB; }
#pragma unroll
      for (int k = 0; k < B; ++k) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[k & 31]) : "v"(a0), "v"(b0));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(y[k & 15]) : "v"(y[(k + 1) & 15]));
        if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(y[k & 15]) : "v"(y[(k + 1) & 15]));
        if (KIND == 3) asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0" : "+v"(x[k & 31]) :: "a0");
        SB;
      }
    }
  }
  SB; t1 = __builtin_readcyclecounter(); SB;
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int i = 0; i < 32; ++i) s += x[i];
  for (int i = 0; i < 16; ++i) s += y[i].x + y[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int B>
void runb(const char* name, float* out, long long* cyc) {
  const int iters = 100;
  hipLaunchKernelGGL((k_burst<KIND, B>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  CK(hipDeviceSynchronize());
  long long c;
  CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double per_group = (double)c / (iters * 8.0);
  printf("burst of %3d %-28s after 8 MFMAs: %8.1f cycles / group = 512 + %.1f  (%.2f per instruction)\n", B, name, per_group, per_group - 512.0, B ? (per_group - 512.0) / B : 0.0);
}

template <int KIND, int K, int NACC>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 200;
  hipLaunchKernelGGL((k_fill<KIND, K, NACC>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  CK(hipDeviceSynchronize());
  long long c;
  CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  printf("%-34s K=%2d acc=%d : %7.2f cycles / MFMA  (+%.2f per filler)\n", name, K, NACC, (double)c / (iters * 64.0), K ? ((double)c / (iters * 64.0) - 64.0) / K : 0.0);
}

int main() {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 64));
  run<F_NONE, 0, 2>("bare", out, cyc);
  run<F_NONE, 0, 1>("bare, ONE accumulator", out, cyc);
  run<F_NONE, 0, 4>("bare, 4 accumulators", out, cyc);
  run<F_FMA, 1, 2>("v_fma_f32 independent", out, cyc);
  run<F_FMA, 2, 2>("v_fma_f32 independent", out, cyc);
  run<F_FMA, 4, 2>("v_fma_f32 independent", out, cyc);
  run<F_FMA, 8, 2>("v_fma_f32 independent", out, cyc);
  run<F_FMA, 12, 2>("v_fma_f32 independent", out, cyc);
  run<F_FMA, 2, 1>("v_fma_f32, ONE accumulator", out, cyc);
  run<F_FMA, 2, 4>("v_fma_f32, 4 accumulators", out, cyc);
  run<F_FMAMUL_DEP, 2, 2>("fma -> dependent mul", out, cyc);
  run<F_FMAMUL_DEP, 4, 2>("fma -> dependent mul", out, cyc);
  run<F_ACCREAD, 2, 2>("v_mov_b32", out, cyc);
  run<F_ACCREAD, 4, 2>("v_mov_b32", out, cyc);
  run<F_MUL_ON_ACC, 2, 2>("v_mul on a 16-register tuple", out, cyc);
  run<F_DSREAD128, 1, 2>("ds_read_b128 + waitcnt per MFMA", out, cyc);
  run<F_DSREAD128, 2, 2>("ds_read_b128 + waitcnt per MFMA", out, cyc);
  run<F_DSWRITE32, 1, 2>("ds_write_b32", out, cyc);
  run<F_DSWRITE32, 2, 2>("ds_write_b32", out, cyc);
  run<F_DSWRITE2, 1, 2>("ds_write2_b32", out, cyc);
  run<F_DSWRITE2, 2, 2>("ds_write2_b32", out, cyc);
  run<F_NOP, 1, 2>("s_nop 0", out, cyc);
  run<F_NOP, 4, 2>("s_nop 0", out, cyc);
  run<F_WAITCNT, 1, 2>("s_waitcnt lgkmcnt(0) (nothing pending)", out, cyc);
  run<F_WAITCNT, 2, 2>("s_waitcnt lgkmcnt(0) (nothing pending)", out, cyc);
  run<F_SALU, 2, 2>("s_add_u32", out, cyc);
  run<F_M4x4, 1, 2>("v_mfma_f32_4x4x1 (8 cycles each)", out, cyc);
  run<F_M4x4, 4, 2>("v_mfma_f32_4x4x1 (8 cycles each)", out, cyc);
  runb<0, 0>("(none)", out, cyc);
  runb<0, 8>("v_fma_f32", out, cyc);
  runb<0, 32>("v_fma_f32", out, cyc);
  runb<0, 64>("v_fma_f32", out, cyc);
  runb<1, 16>("v_pk_fma_f32 (2 per lane)", out, cyc);
  runb<1, 32>("v_pk_fma_f32 (2 per lane)", out, cyc);
  runb<2, 32>("v_pk_mul_f32 (2 per lane)", out, cyc);
  runb<3, 32>("accvgpr write + read pair", out, cyc);
  return 0;
}
