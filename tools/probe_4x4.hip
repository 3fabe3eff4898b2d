// layout probe for v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4x1)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D, float* cyc) {
  int l = threadIdx.x;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
  f32x4 c0 = {0,0,0,0}, c1 = {0,0,0,0};
  long long t0 = clock64();
  for (int i = 0; i < 1000; ++i) { c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c0, 0, 0, 0); }
  long long t1 = clock64();
  for (int i = 0; i < 1000; ++i) { c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c1, 0, 0, 0); }
  long long t2 = clock64();
  if (l == 0) { cyc[0] = (t1 - t0) / 2000.f; cyc[1] = (t2 - t1) / 2000.f; }
  D[256 + l] = c0[0] + c1[1];
}
int main() {
  float hA[64], hB[64], hD[256], hc[2];
  for (int i = 0; i < 64; ++i) { hA[i] = 1 + i; hB[i] = 100 + 3 * i; }
  float *A, *B, *D, *c; hipMalloc(&A, 256); hipMalloc(&B, 256); hipMalloc(&D, 4096); hipMalloc(&c, 8);
  hipMemcpy(A, hA, 256, hipMemcpyHostToDevice); hipMemcpy(B, hB, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(A, B, D, c); hipDeviceSynchronize();
  hipMemcpy(hD, D, 1024, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 8, hipMemcpyDeviceToHost);
  // hypothesis: block = l/4, D[l][r] = A[4*block + r] * B[l]
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { float ref = hA[4 * (l / 4) + r] * hB[l]; if (hD[l * 4 + r] != ref) ++bad; }
  printf("4x4x1 hypothesis D[lane][r] = A[4*(lane/4)+r]*B[lane]: %s (bad=%d)\n", bad ? "FAIL" : "PASS", bad);
  if (bad) for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, hD[l*4], hD[l*4+1], hD[l*4+2], hD[l*4+3]);
  printf("cycles/instr dependent %.1f two-acc %.1f\n", hc[0], hc[1]);
  return 0;
}
