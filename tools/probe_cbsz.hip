// semantics of the A-broadcast controls (cbsz / abid) of v_mfma_f32_4x4x1_16b_f32 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CBSZ, int ABID>
__global__ void k(const float* A, const float* B, float* D) {
  int l = threadIdx.x;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c, CBSZ, ABID, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}
template <int CBSZ, int ABID>
void run(float* A, float* B, float* D, const float* hA, const float* hB) {
  float hD[256];
  k<CBSZ, ABID><<<1, 64>>>(A, B, D); hipDeviceSynchronize();
  hipMemcpy(hD, D, 1024, hipMemcpyDeviceToHost);
  // hypothesis: blocks are grouped in 2^CBSZ; every block of a group uses the A values of block (group_base + ABID)
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int blk = l / 4, grp = blk >> CBSZ, src = (grp << CBSZ) + ABID;
    float ref = hA[4 * src + r] * hB[l];
    if (hD[l * 4 + r] != ref) ++bad;
  }
  printf("cbsz=%d abid=%d: %s (bad=%d)  lane0: %g %g, lane 36: %g %g\n", CBSZ, ABID, bad ? "FAIL" : "PASS", bad, hD[0], hD[1], hD[36*4], hD[36*4+1]);
}
int main() {
  float hA[64], hB[64];
  for (int i = 0; i < 64; ++i) { hA[i] = 1 + i; hB[i] = 100 + 3 * i; }
  float *A, *B, *D; hipMalloc(&A, 256); hipMalloc(&B, 256); hipMalloc(&D, 4096);
  hipMemcpy(A, hA, 256, hipMemcpyHostToDevice); hipMemcpy(B, hB, 256, hipMemcpyHostToDevice);
  run<0, 0>(A, B, D, hA, hB);
  run<1, 1>(A, B, D, hA, hB);
  run<2, 3>(A, B, D, hA, hB);
  run<3, 0>(A, B, D, hA, hB);
  run<3, 5>(A, B, D, hA, hB);
  run<3, 7>(A, B, D, hA, hB);
  run<4, 9>(A, B, D, hA, hB);
  return 0;
}
