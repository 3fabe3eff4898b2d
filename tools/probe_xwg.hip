// What does a per-step exchange between two persistent workgroups cost?  (gfx950)
// The MLP-baseline trainer (mlp_fit.h) is one workgroup on one CU: 27 us per minibatch step, a strictly sequential chain.  Splitting
// a step's 64 samples over TWO workgroups needs every step: each writes its 19 457-float gradient (78 KB), signals, waits for the
// other's, reads it.  This probe times exactly that between two co-resident 256-thread workgroups, 2 000 rounds:
//   placement   workgroups 0 and 8 of the launch (same XCD: the dispatcher deals workgroups round-robin over the 8 XCDs) or 0 and 1
//   memory      ordinary device memory with agent-scope release / acquire fences (the portable way; on gfx950 the release writes
//               back the XCD's L2), ordinary memory with "the stores have been acknowledged" only (valid for one XCD: both CUs
//               share the L2), uncached memory (hipDeviceMallocUncached; valid anywhere)
// Waits are bounded (a lost partner ends the probe with an error, it cannot hang the GPU).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_xwg tools/probe_xwg.hip && tools/probe_xwg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NV = 19 * 256;            // float4 per gradient: 19 456 floats

enum { M_FENCE = 0, M_LIGHT = 1 };

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_xwg(f32x4* slots, unsigned* flags, long long* cyc, float* out, int partner_block, int rounds, int* err) {
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == partner_block ? 1 : -1);
  if (me < 0) return;
  const int tid = threadIdx.x;
  f32x4* mine = slots + (size_t)me * 2 * NV;          // [parity][NV]
  const f32x4* theirs = slots + (size_t)(1 - me) * 2 * NV;
  unsigned* myflag = flags + 64 * me;
  unsigned* theirflag = flags + 64 * (1 - me);
  f32x4 acc = (f32x4)(0.f);
  f32x4 g[19];
  for (int i = 0; i < 19; ++i) g[i] = (f32x4)(1.0f + 0.001f * tid + i);
  __shared__ int bad;
  if (tid == 0) bad = 0;
  __syncthreads();
  long long t0 = 0;
  for (int r = 0; r < rounds + 10; ++r) {
    if (r == 10) t0 = __builtin_readcyclecounter();
    const int par = r & 1;
    for (int i = 0; i < 19; ++i) mine[par * NV + i * 256 + tid] = g[i] + acc;
    if (MODE == M_FENCE) __threadfence();                                        // agent-scope release
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(myflag, (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long w0 = wall_clock64();
      while ((int)(__hip_atomic_load(theirflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)(r + 1)) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - w0 > 100000000ull) { bad = 1; break; }             // 1 s
      }
    }
    __syncthreads();
    if (bad) { if (tid == 0) *err = 1; return; }
    if (MODE == M_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                      // (buffer_inv: the CU's vector L1 must not serve stale lines)
    for (int i = 0; i < 19; ++i) acc += theirs[par * NV + i * 256 + tid] * 1e-6f;
  }
  const long long t1 = __builtin_readcyclecounter();
  out[me * 256 + tid] = acc.x + acc.y + acc.z + acc.w;
  if (tid == 0) cyc[me] = t1 - t0;
}

template <int MODE>
void run(const char* name, bool uncached, int partner, int rounds) {
  f32x4* slots; unsigned* flags; long long* cyc; float* out; int* err;
  const size_t sb = (size_t)2 * 2 * NV * sizeof(f32x4);
  if (uncached) { CK(hipExtMallocWithFlags((void**)&slots, sb, hipDeviceMallocUncached)); CK(hipExtMallocWithFlags((void**)&flags, 1024, hipDeviceMallocUncached)); }
  else { CK(hipMalloc(&slots, sb)); CK(hipMalloc(&flags, 1024)); }
  CK(hipMalloc(&cyc, 64)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&err, 4));
  CK(hipMemset(slots, 0, sb)); CK(hipMemset(flags, 0, 1024)); CK(hipMemset(err, 0, 4)); CK(hipMemset(cyc, 0, 64));
  hipLaunchKernelGGL(k_xwg<MODE>, dim3(partner + 1), dim3(256), 0, 0, slots, flags, cyc, out, partner, rounds, err);
  CK(hipDeviceSynchronize());
  long long h[2]; int timed_out;
  CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&timed_out, err, 4, hipMemcpyDeviceToHost));
  if (timed_out) printf("%-64s partner did not answer (timeout)\n", name);
  else printf("%-64s %7.0f cycles per exchange (78 KB each way + flag), %5.2f us at 2.4 GHz\n", name, (double)h[0] / rounds, (double)h[0] / rounds / 2400.0);
  CK(hipFree(slots)); CK(hipFree(flags)); CK(hipFree(cyc)); CK(hipFree(out)); CK(hipFree(err));
}

int main() {
  const int rounds = 2000;
  run<M_FENCE>("same XCD (blocks 0, 8), device memory, agent-scope fences", false, 8, rounds);
  run<M_LIGHT>("same XCD (blocks 0, 8), device memory, store acknowledgement only", false, 8, rounds);
  run<M_LIGHT>("same XCD (blocks 0, 8), uncached memory", true, 8, rounds);
  run<M_FENCE>("other XCD (blocks 0, 1), device memory, agent-scope fences", false, 1, rounds);
  run<M_LIGHT>("other XCD (blocks 0, 1), uncached memory", true, 1, rounds);
  return 0;
}
