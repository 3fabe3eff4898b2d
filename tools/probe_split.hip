// VERDICT r03 item 7, part (i): what would the cached Fisher-vector product cost per 32-sample tile if its dense products ran on
// the bf16 matrix pipe with every fp32 operand split three ways (hi / mid / lo, the six piece products with i + j <= 2, fp32
// accumulation)?  The error of that arithmetic is measured separately, exactly, on the CPU (tools/probe_split_error.py: at
// native-fp32 level).  This probe measures the INSTRUCTION STREAM a tile would issue, one wave per SIMD, 256 workgroups (the
// chip's clocks under load), no LDS / memory traffic (which the r03 probes showed to be free beside MFMAs):
//
//   f32     the stream of today's kernel: 276 v_mfma_f32_32x32x2_f32 + 480 v_mfma_f32_4x4x1_16b_f32 + 260 vector-ALU
//           instructions in six bursts (tools/isa_hist.sh; 23.2 k cycles per tile measured in the kernel itself)
//   bf16x3  the same K-extent on v_mfma_f32_32x32x16_bf16: 276 x 2 / 16 = 34.5 instructions per piece product, x 6 = 207; the
//           narrow output-layer products stay on the fp32 4x4x1 form (M = #actions: a 32 x 32 x 16 tile would be 80 % padding);
//           PLUS the vector-ALU work of splitting the six activation tiles a tile's products consume as B operands (x~ and h1, h2
//           could come pre-split from K1's cache; t1, t2, delta2, delta1 are formed on the fly): 32 elements per lane and tile
//           matrix, 9 instructions per element pair (3 v_cvt_pk_bf16_f32, 4 unpack shifts / masks, 2 v_pk_add_f32 residuals)
//           -- either as bursts in front of the products ("burst") or at most 5 per bf16-MFMA gap ("interleaved": what
//           MI355X_MICROARCH.md measures as hidden beside a 32-cycle bf16 MFMA; an fp32 MFMA hides nothing)
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_split tools/probe_split.hip && tools/probe_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
#define SB __builtin_amdgcn_sched_barrier(0)

// one element pair (x0, x1) -> packed bf16 hi / mid / lo; 9 instructions
#define SPLIT_PAIR(XX_, H, M, L)                                                                          \
  do {                                                                                                  \
    unsigned h_, m_;                                                                                    \
    f32x2 hf_, r_, mf_;                                                                                 \
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h_) : "v"((XX_).x), "v"((XX_).y));                   \
    asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(hf_.x) : "v"(h_));                                   \
    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hf_.y) : "v"(h_));                               \
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r_) : "v"(XX_), "v"(hf_));    \
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m_) : "v"(r_.x), "v"(r_.y));                     \
    asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(mf_.x) : "v"(m_));                                   \
    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(mf_.y) : "v"(m_));                               \
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r_) : "v"(r_), "v"(mf_));   \
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(L) : "v"(r_.x), "v"(r_.y));                      \
    H = h_; M = m_;                                                                                     \
  } while (0)

template <int B>
__device__ __forceinline__ void valu_burst(f32x2 (&y)[16]) {
#pragma unroll
  for (int k = 0; k < B; ++k) { asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(y[k & 15]) : "v"(y[(k + 1) & 15])); SB; }
}

// MODE 0: f32 stream.  MODE 1: bf16x3, split work as bursts.  MODE 2: bf16x3, split work <= 5 instructions per bf16-MFMA gap.
// MODE 3: bf16x3 MFMAs only (no split work): the matrix-pipe floor of the split formulation.
template <int MODE>
__global__ __launch_bounds__(256, 1) void k_tile(float* out, long long* cyc, int tiles) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) acc[a] = (f32x16)(0.f);
  f32x4 small[4];
  for (int a = 0; a < 4; ++a) small[a] = (f32x4)(0.f);
  float a0 = 1.0f + lane * 1e-3f, b0 = 0.5f;
  f32x2 y[16];
  for (int i = 0; i < 16; ++i) y[i] = f32x2{0.1f * i, 1.0f + lane};
  f32x2 xs[16];                                     // one activation tile of this lane: 32 fp32 values
  for (int i = 0; i < 16; ++i) xs[i] = f32x2{0.37f * i + lane, 1.0f / (1 + i + lane)};
  i32x4 av = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80}, bv = av;
  unsigned sink = 0;
  long long t0 = 0, t1 = 0;
  for (int it = 0; it < tiles + 1; ++it) {
    if (it == 1) { SB; t0 = __builtin_readcyclecounter(); SB; }
    if (MODE == 0) {
      // six phases: 46 dense MFMAs + 80 narrow ones + a burst of ~43 vector-ALU instructions each
#pragma unroll
      for (int ph = 0; ph < 6; ++ph) {
#pragma unroll
        for (int m = 0; m < 46; ++m) { acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[m & 3], 0, 0, 0); SB; }
#pragma unroll
        for (int m = 0; m < 80; ++m) { small[m & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, small[m & 3], 0, 0, 0); SB; }
        valu_burst<43>(y);
      }
    } else {
#pragma unroll
      for (int ph = 0; ph < 6; ++ph) {
        unsigned H[16], M[16], L[16];
        if (MODE == 1) {
#pragma unroll
          for (int p = 0; p < 16; ++p) { SPLIT_PAIR(xs[p], H[p], M[p], L[p]); SB; }
#pragma unroll
          for (int p = 0; p < 16; ++p) sink ^= H[p] ^ M[p] ^ L[p];
        }
        // 207 / 6 = 34.5 bf16 MFMAs per phase: 35, 34 alternating
#pragma unroll
        for (int m = 0; m < 34 + (ph & 1); ++m) {
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(av), "v"(bv));
          SB;
          if (MODE == 2 && m < 32 && (m & 1) == 0) {     // 16 pairs x 9 instructions over the phase's first 32 gaps: 4.5 per gap
            const int p = m >> 1;
            SPLIT_PAIR(xs[p], H[p], M[p], L[p]);
            sink ^= H[p] ^ M[p] ^ L[p];
            SB;
          }
        }
#pragma unroll
        for (int m = 0; m < 80; ++m) { small[m & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, small[m & 3], 0, 0, 0); SB; }
        valu_burst<43>(y);
      }
    }
  }
  SB; t1 = __builtin_readcyclecounter(); SB;
  float s = 0.f;
  for (int a = 0; a < 4; ++a) { for (int r = 0; r < 16; ++r) s += acc[a][r]; s += small[a].x; }
  for (int i = 0; i < 16; ++i) s += y[i].x + y[i].y + xs[i].x;
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(sink & 1);
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
double run(const char* name, float* out, long long* cyc) {
  const int tiles = 200;
  hipLaunchKernelGGL((k_tile<MODE>), dim3(256), dim3(256), 0, 0, out, cyc, tiles);
  CK(hipDeviceSynchronize());
  long long c;
  CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double per = (double)c / tiles;
  printf("%-72s %9.0f cycles / tile\n", name, per);
  return per;
}

int main() {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 64));
  const double f = run<0>("f32: 276 x 32x32x2_f32 + 480 x 4x4x1 + 258 VALU in bursts", out, cyc);
  const double p = run<3>("bf16x3 matrix work only: 207 x 32x32x16_bf16 + 480 x 4x4x1 + 258 VALU", out, cyc);
  const double b = run<1>("bf16x3 + splitting 6 activation tiles (864 VALU) as bursts", out, cyc);
  const double i = run<2>("bf16x3 + the same split work, <= 5 instructions per bf16-MFMA gap", out, cyc);
  printf("ratio f32 / bf16x3: matrix work only %.2f, split in bursts %.2f, split interleaved %.2f\n", f / p, f / b, f / i);
  printf("(today's kernel: 23.2 k cycles per tile, 0.317-0.321 ms per product; the stream above leaves out LDS operand traffic and waits,\n"
         " which the fp32 kernel hides completely and a 3x shorter matrix phase may not)\n");
  return 0;
}
