// Layout / feature probe for gfx950 used while designing the fused NPG kernels.
// Checks: (1) mfma_f32_32x32x2f32 operand + accumulator lane maps, (2) the
// "accumulator feeds next B operand" chaining trick with a k-permuted A operand,
// (3) cross-half exchange (shfl_xor 32), (4) fast tanh accuracy, (5) MFMA issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// D[32x32] = A[32x2] * B[2x32]; lane l supplies A[l&31][l>>5], B[l>>5][l&31]
__global__ void k_layout(const float* A, const float* B, float* D) {
  int l = threadIdx.x;
  float a = A[(l & 31) * 2 + (l >> 5)];
  float b = B[(l >> 5) * 32 + (l & 31)];
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[l * 16 + r] = c[r];
}

// chain: H = W1[32xK1] * X[K1x32] ; Y = W2[32x32] * H  (all "transposed" form, samples = columns)
__global__ void k_chain(const float* W1, const float* X, const float* W2, float* Y, int K1) {
  int l = threadIdx.x, j = l & 31, hi = l >> 5;
  f32x16 h = {0};
  for (int s = 0; s < K1 / 2; ++s) {
    float a = W1[j * K1 + 2 * s + hi];
    float b = X[(2 * s + hi) * 32 + j];
    h = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, h, 0, 0, 0);
  }
  // lane (j,hi) reg r holds H[row = (r&3) + 8*(r>>2) + 4*hi][col j]
  f32x16 y = {0};
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    int k = (s & 3) + 8 * (s >> 2) + 4 * hi;
    float a = W2[j * 32 + k];
    y = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h[s], y, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) Y[l * 16 + r] = y[r];
}

__global__ void k_misc(float* out, const float* xs, int n) {
  int l = threadIdx.x;
  float v = (float)l;
  out[l] = __shfl_xor(v, 32);
  for (int i = l; i < n; i += 64) {
    float x = xs[i];
    float e = __expf(2.0f * x);
    float t1 = 1.0f - 2.0f / (e + 1.0f);            // plain
    float t2 = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f); // hw rcp
    out[64 + i] = t1; out[64 + n + i] = t2; out[64 + 2 * n + i] = tanhf(x);
  }
}

__global__ void k_rate(float* out, int iters) {
  f32x16 c0 = {0}, c1 = {0};
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
  }
  long long t1 = clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
  }
  long long t2 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (float)(t1 - t0) / (4.f * iters); out[1] = (float)(t2 - t1) / (4.f * iters); }
  out[2 + threadIdx.x] = c0[0] + c1[3];
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s arch %s CUs %d clock %d kHz memclk %d kHz L2 %d lds/block %zu regs/block %d\n", p.name, p.gcnArchName,
         p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.l2CacheSize, p.sharedMemPerBlock, p.regsPerBlock);
  // ---- layout
  {
    std::vector<float> A(64), B(64), D(64 * 16);
    for (int i = 0; i < 64; ++i) { A[i] = 1.0f + i * 0.37f; B[i] = -2.0f + i * i * 0.011f; }
    float *dA, *dB, *dD; CK(hipMalloc(&dA, 256)); CK(hipMalloc(&dB, 256)); CK(hipMalloc(&dD, 64 * 16 * 4));
    CK(hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice));
    k_layout<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, 64 * 16 * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
      int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      float ref = fmaf(A[row * 2 + 1], B[32 + col], A[row * 2] * B[col]);
      if (fabsf(ref - D[l * 16 + r]) > 1e-4f * fabsf(ref) + 1e-5f) ++bad;
    }
    printf("LAYOUT32x32x2 %s (bad=%d)\n", bad ? "FAIL" : "PASS", bad);
  }
  // ---- chain
  {
    const int K1 = 8;
    std::vector<float> W1(32 * K1), X(K1 * 32), W2(32 * 32), Y(64 * 16), H(32 * 32), Yr(32 * 32);
    srand(1);
    for (auto& v : W1) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto& v : X) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto& v : W2) v = rand() / (float)RAND_MAX - 0.5f;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < K1; ++k) s += (double)W1[i * K1 + k] * X[k * 32 + j]; H[i * 32 + j] = (float)s; }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < 32; ++k) s += (double)W2[i * 32 + k] * H[k * 32 + j]; Yr[i * 32 + j] = (float)s; }
    float *dW1, *dX, *dW2, *dY; CK(hipMalloc(&dW1, W1.size() * 4)); CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW2, W2.size() * 4)); CK(hipMalloc(&dY, Y.size() * 4));
    CK(hipMemcpy(dW1, W1.data(), W1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW2, W2.data(), W2.size() * 4, hipMemcpyHostToDevice));
    k_chain<<<1, 64>>>(dW1, dX, dW2, dY, K1); CK(hipDeviceSynchronize());
    CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; double maxerr = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
      int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      double e = fabs(Yr[row * 32 + col] - Y[l * 16 + r]); if (e > maxerr) maxerr = e; if (e > 1e-4) ++bad;
    }
    printf("CHAIN %s (bad=%d maxerr=%.3g)\n", bad ? "FAIL" : "PASS", bad, maxerr);
  }
  // ---- misc
  {
    const int n = 4096; std::vector<float> xs(n), out(64 + 3 * n);
    for (int i = 0; i < n; ++i) xs[i] = -9.0f + 18.0f * i / (n - 1);
    float *dx, *dout; CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dout, out.size() * 4));
    CK(hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice));
    k_misc<<<1, 64>>>(dout, dx, n); CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int l = 0; l < 64; ++l) if (out[l] != (float)(l ^ 32)) ++bad;
    printf("SHFL_XOR32 %s\n", bad ? "FAIL" : "PASS");
    double e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < n; ++i) { double t = tanh((double)xs[i]); e1 = fmax(e1, fabs(out[64 + i] - t)); e2 = fmax(e2, fabs(out[64 + n + i] - t)); e3 = fmax(e3, fabs(out[64 + 2 * n + i] - t)); }
    printf("TANH maxabs err: expf/div %.3g  expf/rcp %.3g  tanhf %.3g\n", e1, e2, e3);
  }
  // ---- rate
  {
    float* dout; CK(hipMalloc(&dout, 70 * 4)); float out[2];
    k_rate<<<1, 64>>>(dout, 1000); CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, dout, 8, hipMemcpyDeviceToHost));
    printf("MFMA32x32x2 cycles/instr: dependent %.1f  two-acc %.1f\n", out[0], out[1]);
  }
  return 0;
}
