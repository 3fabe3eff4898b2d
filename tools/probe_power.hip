// Does memory traffic cost an fp32-MFMA-bound kernel CLOCK (not cycles)?  r03's probes (probe_fill.hip) showed that global / LDS
// traffic beside fp32 MFMAs is free in CYCLES.  The r04 elimination builds of the layer-wise output-layer pass showed a
// kernel whose time grows with every traffic component put back, whatever the schedule.  This probe separates the two: every
// SIMD of the chip issues back-to-back 32x32x2 fp32 MFMAs (four independent accumulators); per 16 MFMAs every lane additionally
// requests L x 16 bytes of a stream that comes from HBM (each workgroup walks its own 64 MB region with buffer loads -- lane offset
// fixed, position on the scalar unit, results "used" eight requests later without a vector-ALU instruction: nothing but the memory
// instructions themselves is added to the MFMA stream).  Reported per variant: shader cycles and wall time per 16-MFMA group, the
// clock the chip sustained (cycle counter / 100 MHz real-time counter), the HBM rate of the stream.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_power tools/probe_power.hip && tools/probe_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int L, bool STORE, int NT, bool B4, bool LOAD = true, int GLDS = 0>
__global__ __launch_bounds__(NT) void k_probe(const float* __restrict__ src, float* __restrict__ dst, size_t region_f4, long long* out, int groups) {
  const int lane = threadIdx.x;
  const f32x4* p = (const f32x4*)src + (size_t)blockIdx.x * region_f4;
  f32x4* q = (f32x4*)dst + (size_t)blockIdx.x * region_f4;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) acc[a] = (f32x16)(0.f);
  float a0 = 1.0f + lane * 1e-3f, b0 = 0.5f;
  f32x4 ring[8];
  for (int i = 0; i < 8; ++i) ring[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 sink = {0.f, 0.f, 0.f, 0.f};
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)q, 0, -1, 0x00020000);
  const int voff = lane * (B4 ? 4 : 16);
  const uint32_t region_bytes = (uint32_t)(region_f4 * 16);
  uint32_t so = 0u;
  long long t0 = 0, c0 = 0;
  for (int g = 0; g < groups + 8; ++g) {
    if (g == 8) { __builtin_amdgcn_sched_barrier(0); t0 = (long long)__builtin_amdgcn_s_memrealtime(); c0 = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int l = 0; l < L; ++l) {
      // (the request made eight slots ago is "used" without a vector-ALU instruction -- those are NOT free beside fp32 MFMAs -- and
      //  the stream position advances on the scalar unit: buffer addressing, lane offset fixed, scalar offset bumped)
      asm volatile("" : : "v"(ring[l & 7]));
      if (GLDS) {
        // the same bytes straight into LDS (global_load_lds_dwordx4: wave-uniform LDS base + lane x 16, no VGPR write-back);
        // GLDS == 2 additionally reads them back from LDS into registers a ring turn later (ds_read_b128)
        __shared__ __attribute__((aligned(16))) char stage[NT / 64][8][1024];
        if (GLDS == 2) ring[l & 7] = *(const f32x4*)&stage[threadIdx.x >> 6][l & 7][(threadIdx.x & 63) * 16];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)p + so + (uint32_t)voff),
                                         (__attribute__((address_space(3))) void*)&stage[threadIdx.x >> 6][l & 7][0], 16, 0, 0);
      }
      else if (!LOAD) {}
      else if (B4) ring[l & 7].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, (int)so, 0));
      else ring[l & 7] = __builtin_bit_cast(f32x4, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rs, voff, (int)so, 0));
      if (STORE) __builtin_amdgcn_raw_buffer_store_b128((u32x4_t)__builtin_bit_cast(u32x4_t, ring[(l + 4) & 7]), rd, voff, (int)so, 0);
      so += (uint32_t)(NT * (B4 ? 4 : 16));
      if (so >= region_bytes) so = 0u;
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) { acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[m & 3], 0, 0, 0); }
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_sched_barrier(0);
  const long long t1 = (long long)__builtin_amdgcn_s_memrealtime(), c1 = (long long)__builtin_readcyclecounter();
  float s = sink.x + sink.y + sink.z + sink.w;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int i = 0; i < 8; ++i) s += ring[i].x;
  if (s == 123.456f) dst[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = c1 - c0; }
}

template <int L, bool STORE, int NT = 256, bool B4 = false, bool LOAD = true, int GLDS = 0>
void run(const char* name, const float* src, float* dst, size_t region_f4, long long* out, int nwg) {
  const int groups = 20000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_probe<L, STORE, NT, B4, LOAD, GLDS>), dim3(nwg), dim3(NT), 0, 0, src, dst, region_f4, out, groups);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double tf = (double)(groups + 8) * 16.0 * 4096.0 * (NT / 64) * nwg / (ms * 1e-3) * 1e-12;    // by the HOST's clock: the whole chip's MFMA rate
  long long h[2];
  CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
  const double ns = h[0] * 10.0 / groups, cyc = (double)h[1] / groups, ghz = (double)h[1] / (h[0] * 10.0);
  const double bytes = (double)L * (B4 ? 4.0 : 16.0) * NT * nwg * ((STORE ? 1.0 : 0.0) + (LOAD ? 1.0 : 0.0));          // per group, whole chip
  printf("%-46s %7.1f cycles %7.1f ns per 16 MFMAs   %.3f GHz   stream %.2f TB/s   %.1f TFLOP/s (events)\n", name, cyc, ns, ghz, bytes / ns * 1e-3, tf);
}

int main() {
  int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  const int nwg = prop.multiProcessorCount;
  const size_t region_f4 = (size_t)(64 << 20) / 16;             // 64 MB per workgroup: 16 GB in all, far beyond every cache
  float *src, *dst; long long* out;
  CK(hipMalloc(&src, region_f4 * 16 * nwg)); CK(hipMalloc(&dst, region_f4 * 16 * nwg)); CK(hipMalloc(&out, 64));
  CK(hipMemset(src, 0, region_f4 * 16 * nwg));
  printf("%d workgroups of 4 waves (one per SIMD), 16 fp32 32x32x2 MFMAs = 1024 pipe cycles per group\n", nwg);
  run<0, false>("MFMA only", src, dst, region_f4, out, nwg);
  run<1, false>("+ 1 x 16 B load per lane and group", src, dst, region_f4, out, nwg);
  run<2, false>("+ 2 loads", src, dst, region_f4, out, nwg);
  run<4, false>("+ 4 loads", src, dst, region_f4, out, nwg);
  run<8, false>("+ 8 loads", src, dst, region_f4, out, nwg);
  run<2, true>("+ 2 loads + 2 stores", src, dst, region_f4, out, nwg);
  run<4, true>("+ 4 loads + 4 stores", src, dst, region_f4, out, nwg);
  printf("the same loads straight into LDS (global_load_lds_dwordx4), without / with a ds_read_b128 of the data afterwards:\n");
  run<1, false, 256, false, true, 1>("+ 1 x 16 B global -> LDS", src, dst, region_f4, out, nwg);
  run<2, false, 256, false, true, 1>("+ 2 global -> LDS", src, dst, region_f4, out, nwg);
  run<4, false, 256, false, true, 1>("+ 4 global -> LDS", src, dst, region_f4, out, nwg);
  run<1, false, 256, false, true, 2>("+ 1 x (global -> LDS, LDS -> registers)", src, dst, region_f4, out, nwg);
  run<2, false, 256, false, true, 2>("+ 2 x (global -> LDS, LDS -> registers)", src, dst, region_f4, out, nwg);
  run<4, false, 256, false, true, 2>("+ 4 x (global -> LDS, LDS -> registers)", src, dst, region_f4, out, nwg);
  printf("stores only (16 B per lane; the HBM WRITE rate):\n");
  run<1, true, 256, false, false>("+ 1 store per lane and group", src, dst, region_f4, out, nwg);
  run<2, true, 256, false, false>("+ 2 stores", src, dst, region_f4, out, nwg);
  run<4, true, 256, false, false>("+ 4 stores", src, dst, region_f4, out, nwg);
  run<8, true, 256, false, false>("+ 8 stores", src, dst, region_f4, out, nwg);
  run<2, true, 512, false, false>("+ 2 stores, two waves per SIMD", src, dst, region_f4, out, nwg);
  printf("4-byte loads (256 B per wave and instruction instead of 1 KB):\n");
  run<1, false, 256, true>("+ 1 x 4 B load per lane and group", src, dst, region_f4, out, nwg);
  run<4, false, 256, true>("+ 4 x 4 B loads", src, dst, region_f4, out, nwg);
  run<16, false, 256, true>("+ 16 x 4 B loads", src, dst, region_f4, out, nwg);
  printf("two waves per SIMD (512 threads; cycles and ns per 16 MFMAs of ONE wave: 2048 pipe cycles when the pipe is full):\n");
  run<0, false, 64>("MFMA only, ONE wave per CU", src, dst, region_f4, out, nwg);
  run<0, false, 128>("MFMA only, two waves per CU", src, dst, region_f4, out, nwg);
  run<0, false, 512>("MFMA only", src, dst, region_f4, out, nwg);
  run<0, false, 1024>("MFMA only, four waves per SIMD", src, dst, region_f4, out, nwg);
  run<1, false, 512>("+ 1 x 16 B load per lane and group", src, dst, region_f4, out, nwg);
  run<2, false, 512>("+ 2 loads", src, dst, region_f4, out, nwg);
  run<4, false, 512>("+ 4 loads", src, dst, region_f4, out, nwg);
  return 0;
}
