// Isolated check of the LDS-transposed 16-byte store epilogue tried for k_gemm_p (r04; profiles/r04_lw/experiment_epilogue_stores.log):
// each wave writes a 32 x 32 block in MFMA accumulator layout to an LDS scratch, reads it back as rows and stores it with
// buffer_store_dwordx4 (register soffset).  Correct here -- in the real kernel a v_fma_f32 scheduled right behind such a store
// clobbered its first data register (gfx950 store-data hazard the compiler does not cover for MUBUF stores with a register soffset).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_dbg/tr_test tools/probe_store_hazard.hip && tools/_dbg/tr_test   -> "bad 0"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f1;
typedef __attribute__((address_space(3))) f32x4 lds_f4;
constexpr int LD = 36;
__device__ __forceinline__ int unit_of(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__global__ void k(float* C, int ldc) {
  __shared__ __attribute__((aligned(16))) float lds[8 * 32 * LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hi = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wv / 4, wn = wv % 4;
  float* scr = lds + wv * 32 * LD;
  auto pin = [](const float* p) { uint32_t a = (uint32_t)(uintptr_t)(const lds_f1*)p; asm volatile("" : "+v"(a)); return a; };
  const uint32_t scw = pin(scr + (4 * hi) * LD + j), scr4 = pin(scr + (lane >> 3) * LD + 4 * (lane & 7));
  const uint32_t laneR4 = ((uint32_t)(lane >> 3) * (uint32_t)ldc + 4u * (uint32_t)(lane & 7)) * 4u;
  const __amdgpu_buffer_rsrc_t r_c = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, -1, 0x00020000);
  uint32_t ldc4 = ldc * 4u;
  asm volatile("" : "+s"(ldc4));
  auto rowof = [](int r) { return (r & 3) + 8 * (r >> 2); };
  for (int mt = 0; mt < 2; ++mt)
    for (int nt = 0; nt < 2; ++nt) {
      const uint32_t so = (uint32_t)(wm * 64 + mt * 32) * ldc4 + (uint32_t)(wn * 64 + nt * 32) * 4u;
      if (mt + nt) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + mt * 32 + unit_of(r, hi), col = wn * 64 + nt * 32 + j;
        *(lds_f1*)(uintptr_t)(scw + (uint32_t)(rowof(r) * LD * 4)) = (float)(row * 1000 + col);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 row = *(const lds_f4*)(uintptr_t)(scr4 + (uint32_t)(8 * q * LD * 4));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, row), r_c, laneR4, so + (uint32_t)(8 * q) * ldc4, 0);
      }
    }
}
int main() {
  const int ldc = 256; float* C; hipMalloc(&C, 128 * ldc * 4); hipMemset(C, 0, 128 * ldc * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, C, ldc); hipDeviceSynchronize();
  std::vector<float> h(128 * ldc); hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int r = 0; r < 128; ++r) for (int c = 0; c < 256; ++c) if (h[r * ldc + c] != (float)(r * 1000 + c)) { if (bad < 10) printf("bad at %d %d: %f\n", r, c, h[r * ldc + c]); ++bad; }
  printf("bad %d\n", bad); return 0;
}
