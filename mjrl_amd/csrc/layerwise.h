// layerwise.h -- layer-by-layer path for policies the fused kernel cannot hold on chip
// (hidden sizes > 64, obs dim > 31, any number of hidden layers incl. none).
//
// Same math as fused_policy.h, organised as fp32-MFMA GEMMs with fused epilogues over
// activations that stay resident in HBM for the whole update:
//   forward   H_{l+1} = tanh(H_l W_l^T + b_l)                        (NT GEMM + epilogue)
//   tangent   T_{l+1} = (H_l V_l^T + T_l W_l^T + c_l) (1 - H_{l+1}^2) (two NT GEMMs, one acc)
//   backward  D_l     = (D_{l+1} W_l) (1 - H_l^2)                     (NN GEMM + epilogue)
//   wgrad     gW_l    = D_{l+1}^T H_l                                  (TN GEMM, split over samples)
// Forward activations are cached per (batch, policy) binding, so the 10-25 Fisher-vector
// products of one CG solve reuse them (theta is fixed during CG).
//
// Reference semantics: mjrl/utils/fc_network.py:39-52, mjrl/policies/gaussian_mlp.py:99-145,
// mjrl/algos/batch_reinforce.py:40-58, mjrl/algos/npg_cg.py:62-81.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

#include "fused_policy.h"
#include "lw_head.h"
#include "vecops.h"

namespace mjx {

enum { EPI_STORE = 0, EPI_BIAS_TANH, EPI_BIAS_AFFINE, EPI_TANGENT, EPI_BACK, EPI_BIAS, EPI_BIAS_RELU, EPI_BACK_RELU, EPI_RBACK, EPI_FVP_HEAD };

struct GemmArgs {
  int M, N, npairs;
  int K[2];
  const float* A[2]; int64_t a_rs[2], a_ks[2];   // A(i,k) = A[i*a_rs + k*a_ks]
  const float* B[2]; int64_t b_cs[2], b_ks[2];   // B(k,j) = B[j*b_cs + k*b_ks]
  float* C; int64_t ldc;                          // C(i,j) = C[i*ldc + j*c_cs] (+ z*c_zs for split-K); c_cs == 0 means 1
  int64_t c_zs, c_cs;
  float* colsum;                                  // optional [gridDim.y][cs_ld]: per-block column sums of the stored values (bias gradients)
  int64_t cs_ld;                                  // row stride of colsum (0: N)
  const float* bias;                              // per column
  const float* aux; int64_t ld_aux;               // activation for the (1 - y^2) factor
  const float* aux2; const float* aux3;           // EPI_RBACK: tangent activation t and the pre-activation cotangent
  const float* osc; const float* osh;             // per-column affine (EPI_BIAS_AFFINE)
  const float* ls; float inv_N;                   // EPI_FVP_HEAD: log_std per column, 1 / N_global
  int fast;                                       // set by launch_tile: the interior fast path of the 256-column tiles may be used
  int rows_padded;                                // A / aux / C are workspace blocks with rows allocated up to a multiple of 128
  int epi;
#ifdef MJX_PHASE_CLOCK
  long long* clk;                                 // timing build: 8 int64 per workgroup (tools/lw_clock.py)
#endif
};

// Per-DEVICE one-time state of the layer-wise launchers (function-static flags configured only the device that
// happened to be current at first use -- a second context on another GPU of the same process launched > 64 KB-LDS kernels
// without the attribute and read a foreign device's ticket counter).
inline int lw_cur_device() { int dev = 0; (void)hipGetDevice(&dev); return dev; }
inline void lw_set_dyn_lds(const void* kern, int bytes) {                   // hipFuncAttributeMaxDynamicSharedMemorySize, once per (device, kernel)
  static std::mutex mu;
  static std::vector<std::pair<int, const void*>> done;
  const int dev = lw_cur_device();
  std::lock_guard<std::mutex> lk(mu);
  for (auto& e : done) if (e.first == dev && e.second == kern) return;
  (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done.push_back({dev, kern});
}
inline int lw_ncu() {                                                        // compute units of the current device
  static std::mutex mu;
  static std::vector<std::pair<int, int>> cache;
  const int dev = lw_cur_device();
  std::lock_guard<std::mutex> lk(mu);
  for (auto& e : cache) if (e.first == dev) return e.second;
  int c = 256;
  (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
  cache.push_back({dev, c});
  return c;
}
inline int* lw_ticket_ring() {                                               // 256 ticket counters (+ leaver counts) on the current device; nullptr if the allocation fails
  static std::mutex mu;
  static std::vector<std::pair<int, int*>> rings;
  const int dev = lw_cur_device();
  std::lock_guard<std::mutex> lk(mu);
  for (auto& e : rings) if (e.first == dev) return e.second;
  int* p = nullptr;                                                          // [256 tickets][256 leaver counts], zero between launches
  if (hipMalloc(&p, 512 * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
  else if (hipMemset(p, 0, 512 * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); p = nullptr; }
  rings.push_back({dev, p});
  return p;
}

#ifdef MJX_PHASE_CLOCK
// timing build (-DMJX_PHASE_CLOCK): every workgroup of a k_gemm launch leaves its entry / first-MFMA / loop-end / exit
// times (s_memrealtime, 100 MHz) and the CU it ran on; mjx_set_debug_buffer hands the buffer over and resets the slot
// counter, each launch_tile call takes the next slot of LW_CLK_SLOT int64
constexpr int LW_CLK_SLOT = 8 + 8 * 16384, LW_CLK_SLOTS = 24;
inline long long*& lw_clk_buf() { static long long* p = nullptr; return p; }
inline int& lw_clk_slot() { static int s = 0; return s; }
#define LW_STAMP(k) do { if (g.clk && tid == 0 && blockIdx.z == 0) { __builtin_amdgcn_sched_barrier(0); \
  g.clk[8 + 8 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); \
  if ((k) == 1 || (k) == 2) g.clk[8 + 8 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) + 5 + (k)] = (long long)__builtin_readcyclecounter(); \
  __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define LW_STAMP(k) do {} while (0)

#endif

constexpr int GBN = 128, GBK = 32, GLD = GBK + 4;

// C = sum_p A_p * B_p with v_mfma_f32_32x32x2_f32; 4 waves in a 2x2 grid of 64x64 sub-tiles, K tiles of 32.
// Operand tiles go global -> registers -> LDS in the 16-byte granules they are contiguous in:
//   * a K-contiguous operand (activations, W as "NT") is kept as [row][k] (row stride 36) and its MFMA fragments
//     are read with ONE ds_read_b128 per 32-row block and 4 k-steps;
//   * a row-contiguous operand (W as "NN", the sample-major operands of the weight gradient) is kept as [k][row]
//     (stride 132) and read with ds_read_b32;
//   k-slot `hi` of step (q, t) carries k = 8q + 4hi + t for A and B alike (the permutation trick of the fused kernels).
// Operand tiles are double-buffered in LDS (one barrier per k-tile); tile k + 1 is stored and tile k + 2 requested from
// global memory (register-staged) before the MFMAs of tile k.  blockIdx.z splits the K range of pair 0 (wgrad: K = samples).
// BN = 128: 2x2 waves of 64x64; BN = 32 (narrow outputs: action heads, value heads): 4x1 waves of 32x32, so a
// 17-column product is padded to 32 instead of 128 columns.
// threads per workgroup: 8 waves at BN >= 128 (one workgroup per CU, two waves per SIMD), 4 waves of 32 x 32 at BN = 32.
// Tile shapes (BM x BN, per-wave sub-tile, accumulator registers):
//   128 x 128   32 x 64    32     the round-1 shape: every operand is re-read once per 128 output columns / rows
//   128 x 256   64 x 64    64     sample-major products with >= 256 output columns (hidden layers of 256 / 512 units):
//                                 the activation operand is read ONCE per 256 columns -- these GEMMs sit at the ridge of
//                                 the roofline when their operands stream from HBM, so halving the re-reads
//                                 is worth more than occupancy
//   128 x 128   64 x 64    64     in 256-thread workgroups (4 waves), two workgroups per CU: one workgroup's epilogue
//                                 (activation loads, stores) and pipeline fill run under the other's matrix-core loop
template <int BN> constexpr int gemm_threads() { return BN >= 128 ? 512 : 256; }

template <int BM, int BN, int NTH = gemm_threads<BN>()>
__global__ __launch_bounds__(NTH, 2) void k_gemm(GemmArgs g) {      // (second argument: waves per SIMD; 4 would need <= 128 VGPRs: spills, measured slower)
  constexpr int WN = BN / 64 > 0 ? BN / 64 : 1;   // waves along N
  constexpr int WMc = (NTH / 64) / WN;            // waves along M
  constexpr int TM = BM / WMc, TN = BN / WN;     // per-wave tile
  constexpr int MT = TM / 32, NT = TN / 32;
  // two buffers per operand (dynamic LDS: 72 KB at 128 x 128, 108 KB at 128 x 256, 144 KB at 256 x 256): tile k + 1 is
  // stored while tile k is being multiplied, ONE barrier per k-tile
  constexpr int ASZ = (BM * GLD > GBK * (BM + 4)) ? BM * GLD : GBK * (BM + 4), BSZ = (BN * GLD > GBK * (BN + 4)) ? BN * GLD : GBK * (BN + 4);
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* As = gsm;                       // [2][ASZ]
  float* Bs = gsm + 2 * ASZ;             // [2][BSZ]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hi = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  LW_STAMP(0);
#ifdef MJX_PHASE_CLOCK
  if (g.clk && tid == 0 && blockIdx.z == 0) {
    long long* q = g.clk + 8 + 8 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x);
    q[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
    q[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
    if (blockIdx.x == 0 && blockIdx.y == 0) {
      g.clk[0] = g.M; g.clk[1] = g.N; g.clk[2] = g.K[0]; g.clk[3] = g.npairs > 1 ? g.K[1] : 0; g.clk[4] = g.epi; g.clk[5] = BN;
      g.clk[6] = gridDim.x * gridDim.y; g.clk[7] = gridDim.z;
    }
  }
#endif
  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f32x16)(0.f);

  for (int p = 0; p < g.npairs; ++p) {
    int kbeg = 0, kend = g.K[p];
    if (gridDim.z > 1) {
      int chunk = (g.K[p] + gridDim.z - 1) / gridDim.z;
      chunk = ((chunk + GBK - 1) / GBK) * GBK;
      kbeg = blockIdx.z * chunk;
      kend = min(g.K[p], kbeg + chunk);
    }
    if (kbeg >= kend) continue;
    // one operand tile = 128 rows x 32 k = 1024 float4; 2 (4) per thread.  mode 0: K-contiguous, mode 1: row-contiguous,
    // mode 2: anything else (scalar gather)
    auto mode_of = [](const float* ptr, int64_t rs, int64_t ks) {
      if (ks == 1 && (rs & 3) == 0 && (((uintptr_t)ptr) & 15) == 0) return 0;
      if (rs == 1 && (ks & 3) == 0 && (((uintptr_t)ptr) & 15) == 0) return 1;
      return 2;
    };
    const float* __restrict__ Ap = g.A[p];
    const float* __restrict__ Bp = g.B[p];
    const int64_t ars = g.a_rs[p], aks = g.a_ks[p], bcs = g.b_cs[p], bks = g.b_ks[p];
    const int amode = mode_of(Ap, ars, aks), bmode = mode_of(Bp, bcs, bks);
    constexpr int CA = BM * 8 / NTH, CB = BN * 8 / NTH;        // float4 per thread and operand tile
    f32x4 ra[CA], rb[CB];
    // RT = rows of the tile (128 or 32): 8 float4 per row in the [row][k] image, RT/4 per k-row in the [k][row] image
    auto gload1 = [&](auto rt, auto& r, const float* __restrict__ P, int mode, int64_t rs, int64_t ks, int r0, int R, int k0) {
      constexpr int RT = decltype(rt)::value, NC = RT * 8 / NTH, Q = RT / 4;
      // interior tiles (block-uniform test): unconditional 16-byte loads, nothing else in the way
      if (mode != 2 && r0 + RT <= R && k0 + GBK <= kend) {
        if (mode == 0) {
#pragma unroll
          for (int c = 0; c < NC; ++c) { const int idx = tid + NTH * c; r[c] = *(const f32x4*)(P + (int64_t)(r0 + (idx >> 3)) * rs + k0 + 4 * (idx & 7)); }
        } else {
#pragma unroll
          for (int c = 0; c < NC; ++c) { const int idx = tid + NTH * c; r[c] = *(const f32x4*)(P + (int64_t)(k0 + idx / Q) * ks + r0 + 4 * (idx % Q)); }
        }
        return;
      }
      // edge tiles / unaligned operands: element-wise, clamped address + select (no branches)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int idx = tid + NTH * c;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = (mode == 1) ? r0 + 4 * (idx % Q) + e : r0 + (idx >> 3);
          const int k = (mode == 1) ? k0 + idx / Q : k0 + 4 * (idx & 7) + e;
          const bool ok = (row < R) && (k < kend);
          const float v = P[ok ? (int64_t)row * rs + (int64_t)k * ks : 0];
          r[c][e] = ok ? v : 0.f;
        }
      }
    };
    // LDS image: K-contiguous operands as [row][k] (stride GLD), row-contiguous ones as [k][row] (stride RT + 4) --
    // either way the 16-byte granule that was loaded is stored with one conflict-free ds_write_b128
    auto lstore1 = [&](auto rt, float* S, const auto& r, int mode) {
      constexpr int RT = decltype(rt)::value, NC = RT * 8 / NTH, Q = RT / 4;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int idx = tid + NTH * c;
        if (mode == 1) *(f32x4*)&S[(idx / Q) * (RT + 4) + 4 * (idx % Q)] = r[c];
        else *(f32x4*)&S[(idx >> 3) * GLD + 4 * (idx & 7)] = r[c];
      }
    };
    // operand fragment of the 32-row block at row0 for the 4 k-steps of group q: k-slot `hi` of step t carries
    // k = 8q + 4hi + t (one ds_read_b128 from the [row][k] image, four ds_read_b32 from the [k][row] image)
    auto frag = [&](auto rt, const float* S, int mode, int row0, int q) {
      constexpr int RT = decltype(rt)::value;
      if (mode == 1) {
        // (volatile: four single ds_read_b32 with 16-bit byte offsets.  Paired into ds_read2_b32 -- 8-bit offsets, 1 020 bytes of reach
        //  against rows (RT + 4) * 4 = 528 / 1 040 bytes apart -- every k-group re-based with v_add_u32: 16-25 vector-ALU instructions
        //  per k-tile and wave between the MFMAs, which fp32 MFMAs do not hide; LDS instructions they do.)
        const volatile lds_float* p = (const volatile lds_float*)(const lds_float*)&S[(8 * q + 4 * hi) * (RT + 4) + row0 + j];
        return f32x4{p[0], p[RT + 4], p[2 * (RT + 4)], p[3 * (RT + 4)]};
      }
      return *(const f32x4*)&S[(row0 + j) * GLD + 8 * q + 4 * hi];
    };
    constexpr std::integral_constant<int, BM> RA{};
    constexpr std::integral_constant<int, BN> RB{};
    // The K loop is instantiated per LDS image pair (chosen once per operand pair, outside the loop), so that the body
    // is one straight-line block: fragment reads of group q + 1 and the MFMAs of group q schedule together.
    auto kloop = [&](auto ta, auto tb) {
      constexpr int LA = decltype(ta)::value, LB = decltype(tb)::value;     // 1: [k][row] image, 0: [row][k] image
      const int ntile = (kend - kbeg + GBK - 1) / GBK;
      gload1(RA, ra, Ap, amode, ars, aks, m0, g.M, kbeg);
      gload1(RB, rb, Bp, bmode, bcs, bks, n0, g.N, kbeg);
      __syncthreads();                                 // (a previous operand pair is fully consumed)
      lstore1(RA, As, ra, LA);
      lstore1(RB, Bs, rb, LB);
      if (ntile > 1) {
        gload1(RA, ra, Ap, amode, ars, aks, m0, g.M, kbeg + GBK);
        gload1(RB, rb, Bp, bmode, bcs, bks, n0, g.N, kbeg + GBK);
      }
      __syncthreads();
      if (p == 0) LW_STAMP(1);
      for (int kt = 0; kt < ntile; ++kt) {
        const float* Ac = As + (kt & 1) * ASZ;
        const float* Bc = Bs + (kt & 1) * BSZ;
        if (kt + 1 < ntile) {
          // tile kt + 1 goes into the other buffer (its last readers passed the barrier that ended iteration kt - 1), the
          // global loads of tile kt + 2 then fly under this tile's MFMAs
          lstore1(RA, As + ((kt + 1) & 1) * ASZ, ra, LA);
          lstore1(RB, Bs + ((kt + 1) & 1) * BSZ, rb, LB);
          if (kt + 2 < ntile) {
            gload1(RA, ra, Ap, amode, ars, aks, m0, g.M, kbeg + (kt + 2) * GBK);
            gload1(RB, rb, Bp, bmode, bcs, bks, n0, g.N, kbeg + (kt + 2) * GBK);
          }
        }
        f32x4 a4[MT], b4[NT], an[MT], bn[NT];
#pragma unroll
        for (int a = 0; a < MT; ++a) a4[a] = frag(RA, Ac, LA, wm * TM + 32 * a, 0);
#pragma unroll
        for (int b = 0; b < NT; ++b) b4[b] = frag(RB, Bc, LB, wn * TN + 32 * b, 0);
#pragma unroll
        for (int q = 0; q < GBK / 8; ++q) {
          if (q + 1 < GBK / 8) {
#pragma unroll
            for (int a = 0; a < MT; ++a) an[a] = frag(RA, Ac, LA, wm * TM + 32 * a, q + 1);
#pragma unroll
            for (int b = 0; b < NT; ++b) bn[b] = frag(RB, Bc, LB, wn * TN + 32 * b, q + 1);
          }
#ifdef MJX_GEMM_SETPRIO
          __builtin_amdgcn_s_setprio(1);       // (experiment) the wave that is issuing matrix-core work wins arbitration over its SIMD partner's loads / LDS traffic
#endif
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
              for (int b = 0; b < NT; ++b) acc[a][b] = MJX_MFMA(a4[a][t], b4[b][t], acc[a][b]);
#ifdef MJX_GEMM_SETPRIO
          __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
          for (int a = 0; a < MT; ++a) a4[a] = an[a];
#pragma unroll
          for (int b = 0; b < NT; ++b) b4[b] = bn[b];
        }
        __syncthreads();
      }
    };
    // Fast path of the 256-column tiles: every tile of this operand pair is interior (whole 128 x 256 block inside the
    // matrices, K range a multiple of 32, 16-byte granules) -- the steady state of the big products.  Per-thread operand
    // pointers are formed once and bumped by a constant per k-tile, and the LDS stores of tile k + 1 / the global loads of
    // tile k + 2 sit BETWEEN the MFMA groups of tile k instead of in front of them: in the general loop above all eight
    // waves leave the barrier together, do their stores / address arithmetic / load issue together (~700 cycles per
    // k-tile during which no matrix-core instruction runs on the CU) and only then start multiplying.  (Measured before
    // building this: with the activation operand served from cache -- MJX_LW_DEBUG_ALIAS_A -- the big products run 1-3 %
    // faster, so the loop is not waiting for memory; it loses its ~12 % in that phase-locked overhead.)
    auto kloop_fast = [&](auto ta, auto tb) {
      constexpr int LA = decltype(ta)::value, LB = decltype(tb)::value;
      constexpr int QA = BM / 4, QB = BN / 4;
      const int ntile = (kend - kbeg) / GBK;
      const float* pa[CA];
      const float* pb[CB];
#pragma unroll
      for (int c = 0; c < CA; ++c) {
        const int idx = tid + NTH * c;
        pa[c] = LA ? Ap + (int64_t)(kbeg + idx / QA) * aks + m0 + 4 * (idx % QA) : Ap + (int64_t)(m0 + (idx >> 3)) * ars + kbeg + 4 * (idx & 7);
      }
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const int idx = tid + NTH * c;
        pb[c] = LB ? Bp + (int64_t)(kbeg + idx / QB) * bks + n0 + 4 * (idx % QB) : Bp + (int64_t)(n0 + (idx >> 3)) * bcs + kbeg + 4 * (idx & 7);
      }
      const int64_t sa = LA ? (int64_t)GBK * aks : GBK, sb = LB ? (int64_t)GBK * bks : GBK;
      auto gloadf = [&]() {
#pragma unroll
        for (int c = 0; c < CA; ++c) { ra[c] = *(const f32x4*)pa[c]; pa[c] += sa; }
#pragma unroll
        for (int c = 0; c < CB; ++c) { rb[c] = *(const f32x4*)pb[c]; pb[c] += sb; }
      };
      gloadf();
      __syncthreads();                                 // (a previous operand pair is fully consumed)
      lstore1(RA, As, ra, LA);
      lstore1(RB, Bs, rb, LB);
      if (ntile > 1) gloadf();
      __syncthreads();
      if (p == 0) LW_STAMP(1);
      auto body = [&](int kt, auto st_tag, auto ld_tag) {
        constexpr bool ST = decltype(st_tag)::value, LD = decltype(ld_tag)::value;
        const float* Ac = As + (kt & 1) * ASZ;
        const float* Bc = Bs + (kt & 1) * BSZ;
        f32x4 a4[MT], b4[NT], an[MT], bn[NT];
#pragma unroll
        for (int a = 0; a < MT; ++a) a4[a] = frag(RA, Ac, LA, wm * TM + 32 * a, 0);
#pragma unroll
        for (int b = 0; b < NT; ++b) b4[b] = frag(RB, Bc, LB, wn * TN + 32 * b, 0);
#pragma unroll
        for (int q = 0; q < GBK / 8; ++q) {
          if (q + 1 < GBK / 8) {
#pragma unroll
            for (int a = 0; a < MT; ++a) an[a] = frag(RA, Ac, LA, wm * TM + 32 * a, q + 1);
#pragma unroll
            for (int b = 0; b < NT; ++b) bn[b] = frag(RB, Bc, LB, wn * TN + 32 * b, q + 1);
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
              for (int b = 0; b < NT; ++b) acc[a][b] = MJX_MFMA(a4[a][t], b4[b][t], acc[a][b]);
          if (q == 0 && ST) {                            // tile kt + 1 -> the other buffer, under this tile's MFMAs
            lstore1(RA, As + ((kt + 1) & 1) * ASZ, ra, LA);
            lstore1(RB, Bs + ((kt + 1) & 1) * BSZ, rb, LB);
          }
          if (q == 1 && LD) gloadf();                    // tile kt + 2 -> registers
#pragma unroll
          for (int a = 0; a < MT; ++a) a4[a] = an[a];
#pragma unroll
          for (int b = 0; b < NT; ++b) b4[b] = bn[b];
        }
        __syncthreads();
      };
      using T = std::true_type;
      using F = std::false_type;
      int kt = 0;
      for (; kt + 2 < ntile; ++kt) body(kt, T{}, T{});
      if (kt + 1 < ntile) { body(kt, T{}, F{}); ++kt; }
      body(kt, F{}, F{});
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    bool fast = false;
    if (BN == 256 && g.fast && amode != 2 && bmode != 2 && m0 + BM <= g.M && n0 + BN <= g.N && ((kend - kbeg) % GBK) == 0) fast = true;
    if (fast) {
      if constexpr (BN == 256) {
        if (amode == 1) { if (bmode == 1) kloop_fast(I1{}, I1{}); else kloop_fast(I1{}, I0{}); }
        else { if (bmode == 1) kloop_fast(I0{}, I1{}); else kloop_fast(I0{}, I0{}); }
      }
    } else {
      if (amode == 1) { if (bmode == 1) kloop(I1{}, I1{}); else kloop(I1{}, I0{}); }
      else { if (bmode == 1) kloop(I0{}, I1{}); else kloop(I0{}, I0{}); }
    }
  }
  // Epilogue, specialised once per launch (not per element): per-column constants are fetched once per 32-column
  // block, the activation operands of a 32x32 block as one batch of independent loads.
  LW_STAMP(2);
  float* Cz = g.C + (int64_t)blockIdx.z * g.c_zs;
  auto epilogue = [&](auto tag) {
    constexpr int EPI = decltype(tag)::value;
    constexpr bool USE_BIAS = EPI == EPI_BIAS_TANH || EPI == EPI_BIAS_AFFINE || EPI == EPI_TANGENT || EPI == EPI_BIAS || EPI == EPI_BIAS_RELU ||
                              EPI == EPI_FVP_HEAD;
    constexpr bool USE_AUX = EPI == EPI_TANGENT || EPI == EPI_BACK || EPI == EPI_RBACK || EPI == EPI_BACK_RELU;
    constexpr bool CSUM = EPI == EPI_BACK || EPI == EPI_BACK_RELU;      // launches that may carry g.colsum
    const int64_t ccs = g.c_cs ? g.c_cs : 1;
    float csum[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) csum[nt] = 0.f;
    // Interior blocks (the whole BM x BN block inside C, unit column stride): addresses are a wave-uniform base per 32 x 32
    // block (SGPRs) + sixteen 32-bit lane offsets formed once; the element-wise path below pays a 64-bit multiply, bounds
    // tests and selects PER ELEMENT -- measured with the per-workgroup clocks (tools/lw_clock.py) at ~10 us per workgroup,
    // 13-23 % of a 256-wide product and more than half of the K <= 39 ones.  The activation operand of all the wave's blocks
    // is requested as one batch.
    if (m0 + BM <= g.M && n0 + BN <= g.N && ccs == 1 && g.ldc < (1 << 24) && g.ld_aux < (1 << 24)) {
      const int wv = __builtin_amdgcn_readfirstlane(wave);
      const int um = m0 + (wv / WN) * TM, un = n0 + (wv % WN) * TN;        // wave-uniform origin of this wave's sub-tile
      uint32_t offC[16], offA[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        offC[r] = (uint32_t)(4 * hi) * (uint32_t)g.ldc + (uint32_t)j + (uint32_t)((r & 3) + 8 * (r >> 2)) * (uint32_t)g.ldc;
        offA[r] = (uint32_t)(4 * hi) * (uint32_t)g.ld_aux + (uint32_t)j + (uint32_t)((r & 3) + 8 * (r >> 2)) * (uint32_t)g.ld_aux;
      }
      constexpr bool AUX1 = EPI == EPI_TANGENT || EPI == EPI_BACK || EPI == EPI_BACK_RELU;
      float yall[AUX1 ? MT * NT * 16 : 1];
      if (AUX1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float* __restrict__ ab = g.aux + (int64_t)(um + mt * 32) * g.ld_aux + (un + nt * 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) yall[(mt * NT + nt) * 16 + r] = ab[offA[r]];
          }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int col = un + nt * 32 + j;
          const float bias = USE_BIAS ? g.bias[col] : 0.f;
          const float osc = (EPI == EPI_BIAS_AFFINE || EPI == EPI_FVP_HEAD) ? g.osc[col] : 1.f;
          const float osh = (EPI == EPI_BIAS_AFFINE && g.osh) ? g.osh[col] : 0.f;
          float dk = 0.f;
          if (EPI == EPI_FVP_HEAD) { const float sg = expf(g.ls[col]); dk = 2.0f / (2.0f * sg * sg + 1e-8f); }
          float* __restrict__ cb = Cz + (int64_t)(um + mt * 32) * g.ldc + (un + nt * 32);
          float y[16], t2[16], pre[16];
          if (AUX1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = yall[(mt * NT + nt) * 16 + r];
          } else if (USE_AUX) {                       // EPI_RBACK: three operands, per block
            const int64_t ao = (int64_t)(um + mt * 32) * g.ld_aux + (un + nt * 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) { y[r] = (g.aux + ao)[offA[r]]; t2[r] = (g.aux2 + ao)[offA[r]]; pre[r] = (g.aux3 + ao)[offA[r]]; }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[mt][nt][r];
            if (EPI == EPI_BIAS_TANH) v = tanhf(v + bias);
            else if (EPI == EPI_BIAS_AFFINE) v = (v + bias) * osc + osh;
            else if (EPI == EPI_FVP_HEAD) { v = (v + bias) * osc; v = osc * (dk * v * g.inv_N); }
            else if (EPI == EPI_TANGENT) v = (v + bias) * fmaf(-y[r], y[r], 1.0f);
            else if (EPI == EPI_BACK) v = v * fmaf(-y[r], y[r], 1.0f);
            else if (EPI == EPI_RBACK) v = v * fmaf(-y[r], y[r], 1.0f) - 2.0f * y[r] * t2[r] * pre[r];
            else if (EPI == EPI_BIAS) v = v + bias;
            else if (EPI == EPI_BIAS_RELU) v = fmaxf(v + bias, 0.f);
            else if (EPI == EPI_BACK_RELU) v = (y[r] > 0.f) ? v : 0.f;
            cb[offC[r]] = v;
            if (CSUM) csum[nt] += v;
          }
        }
    } else
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + wn * TN + nt * 32 + j;
        const bool cok = col < g.N;
        const int colc = cok ? col : 0;
        const float bias = USE_BIAS ? g.bias[colc] : 0.f;
        const float osc = (EPI == EPI_BIAS_AFFINE || EPI == EPI_FVP_HEAD) ? g.osc[colc] : 1.f;
        const float osh = (EPI == EPI_BIAS_AFFINE && g.osh) ? g.osh[colc] : 0.f;
        float dk = 0.f;                           // EPI_FVP_HEAD: D = 2 / (2 sigma^2 + 1e-8) of the column's action
        if (EPI == EPI_FVP_HEAD) { const float sg = expf(g.ls[colc]); dk = 2.0f / (2.0f * sg * sg + 1e-8f); }
        const int rbase = m0 + wm * TM + mt * 32;
        float y[16], t2[16], pre[16];
        if (USE_AUX) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rbase + unit_of(r, hi);
            const int64_t o = (row < g.M) ? (int64_t)row * g.ld_aux + colc : 0;
            y[r] = g.aux[o];
            if (EPI == EPI_RBACK) { t2[r] = g.aux2[o]; pre[r] = g.aux3[o]; }
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + unit_of(r, hi);
          float v = acc[mt][nt][r];
          if (EPI == EPI_BIAS_TANH) v = tanhf(v + bias);
          else if (EPI == EPI_BIAS_AFFINE) v = (v + bias) * osc + osh;
          else if (EPI == EPI_FVP_HEAD) { v = (v + bias) * osc; v = osc * (dk * v * g.inv_N); }     // the tangent of mu, then d3 = out_scale D mudot / N
          else if (EPI == EPI_TANGENT) v = (v + bias) * fmaf(-y[r], y[r], 1.0f);
          else if (EPI == EPI_BACK) v = v * fmaf(-y[r], y[r], 1.0f);
          else if (EPI == EPI_RBACK) v = v * fmaf(-y[r], y[r], 1.0f) - 2.0f * y[r] * t2[r] * pre[r];   // Pearlmutter R-backward through tanh
          else if (EPI == EPI_BIAS) v = v + bias;
          else if (EPI == EPI_BIAS_RELU) v = fmaxf(v + bias, 0.f);
          else if (EPI == EPI_BACK_RELU) v = (y[r] > 0.f) ? v : 0.f;
          const bool ok = cok && row < g.M;
          if (ok) Cz[(int64_t)row * g.ldc + (int64_t)col * ccs] = v;
          if (CSUM) csum[nt] += ok ? v : 0.f;
        }
      }
    if (CSUM && g.colsum) {
      // column sums of this block's 128 rows: lane halves (permlane swap), then the waves along M through LDS
      __syncthreads();                              // operand tiles are dead
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float t = half_sum(csum[nt]);
        if (hi == 0) As[wm * BN + wn * TN + nt * 32 + j] = t;
      }
      __syncthreads();
      if (tid < BN && n0 + tid < g.N) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WMc; ++w) t += As[w * BN + tid];
        g.colsum[(int64_t)blockIdx.y * (g.cs_ld ? g.cs_ld : g.N) + n0 + tid] = t;
      }
    }
  };
  switch (g.epi) {
    case EPI_BIAS_TANH: epilogue(std::integral_constant<int, EPI_BIAS_TANH>{}); break;
    case EPI_BIAS_AFFINE: epilogue(std::integral_constant<int, EPI_BIAS_AFFINE>{}); break;
    case EPI_TANGENT: epilogue(std::integral_constant<int, EPI_TANGENT>{}); break;
    case EPI_BACK: epilogue(std::integral_constant<int, EPI_BACK>{}); break;
    case EPI_RBACK: epilogue(std::integral_constant<int, EPI_RBACK>{}); break;
    case EPI_BIAS: epilogue(std::integral_constant<int, EPI_BIAS>{}); break;
    case EPI_BIAS_RELU: epilogue(std::integral_constant<int, EPI_BIAS_RELU>{}); break;
    case EPI_BACK_RELU: epilogue(std::integral_constant<int, EPI_BACK_RELU>{}); break;
    case EPI_FVP_HEAD: epilogue(std::integral_constant<int, EPI_FVP_HEAD>{}); break;
    default: epilogue(std::integral_constant<int, EPI_STORE>{}); break;
  }
  LW_STAMP(3);
}

// x~ = (x - in_shift) / (in_scale + 1e-8)    fc_network.py:46
}  // namespace mjx
#include "lw_gemm_p.h"
namespace mjx {

__global__ void k_normalize(const float* __restrict__ x, int64_t N, int n, int ldo, const float* __restrict__ tr, float* __restrict__ o) {
  // rows of o are padded to ldo = n rounded up to 4 floats (zeros): 16-byte operand granules for any observation width
  const int64_t tot = N * ldo;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % ldo);
    const int64_t r = i / ldo;
    o[i] = (f < n) ? (x[r * n + f] - tr[f]) / (tr[n + f] + 1e-8f) : 0.f;
  }
}

// rows of `cols` floats <-> rows padded to `ld` floats (zeros): the first layer's weight block for observation widths
// that are not a multiple of 4 (the 39 / 45 / 46-wide Adroit observations)
__global__ void k_pad_rows(const float* __restrict__ src, int rows, int cols, int ld, float* __restrict__ dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * ld; i += gridDim.x * blockDim.x) {
    const int r = i / ld, c = i - r * ld;
    dst[i] = c < cols ? src[r * cols + c] : 0.f;
  }
}
// dst[c][r] = src[r][c] (32 x 32 tiles through LDS): the hidden layers' weight blocks in the K-contiguous ("NT") layout the
// persistent GEMM's delta products run fastest with (r06)
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < cols && r < rows) dst[(int64_t)c * rows + r] = tile[tx][ty + 8 * i];
  }
}
__global__ void k_unpad_rows(const float* __restrict__ src, int rows, int cols, int ld, float* __restrict__ dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += gridDim.x * blockDim.x) {
    const int r = i / cols, c = i - r * cols;
    dst[i] = src[r * ld + c];
  }
}

// per-sample likelihood head (mean_LL / likelihood_ratio / mean_kl / CPI surrogate).
// mode 0 (VPG): writes d3 = out_scale * adv*LR/N * (a-mu)/sigma^2, partial sums [surr, count, gls(m)...]
// mode 2 (EVAL): partial sums [surr, kl]
// part: [gridDim.x][2 + MPH] doubles
constexpr int MPH = 64;   // max action dim of the layer-wise head
// r06: ONE pass with the actions on the lanes.  The r01-r05 kernel gave every thread a sample and walked its m actions in a loop
// (rows of m floats per thread: uncoalesced) -- and, to keep the log_std gradients out of a dynamically indexed register array,
// it repeated the WHOLE pass over the batch once per four actions: 7 passes at configs[4]'s 28 actions, 2.27 ms per launch for
// 0.45 GB (K1 and K3 of every update: 4.5 of 226 ms).  Now a wave takes one sample (m > 32) or two (a lane half each) per step: lane a
// holds action a's element of the row (coalesced loads / stores), the sums over the actions are wave reductions (DPP row sums +
// v_permlane16_swap / v_permlane32_swap: no LDS), and lane a keeps d log_std[a] in ONE fp64 register for the whole launch.
// Per element the arithmetic is the old kernel's (the same divisions); the sums over the actions are trees instead of chains, and the
// likelihood ratio is formed from the per-action DIFFERENCE of the two log-likelihoods (below).
__device__ __forceinline__ float head_sum32(float v) {       // every lane: the sum over its 32-lane half
  v = row16_sum(v);
  auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}
constexpr int HEAD_NT = 1024;                            // 16 waves per workgroup: the pass is latency-bound (a row or two per wave step)
__global__ __launch_bounds__(HEAD_NT) void k_head(int mode, const float* __restrict__ mu, const float* __restrict__ mu_old,
                                              const float* __restrict__ act, const float* __restrict__ adv, int64_t N, int m,
                                              const float* __restrict__ ls_new, const float* __restrict__ ls_old,
                                              const float* __restrict__ osc, float inv_N, float* __restrict__ d3,
                                              double* __restrict__ part) {
  __shared__ double sh[17];
  __shared__ double shg[HEAD_NT / 64][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  const bool wide = m > 32;                      // one sample per wave step; else two (lane half = sample)
  const int a = wide ? lane : (lane & 31), sub = wide ? 0 : (lane >> 5), SP = wide ? 1 : 2;
  const bool on = a < m;
  const float lsn_a = on ? ls_new[a] : 0.f, lso_a = on ? ls_old[a] : 0.f;
  const float sgn_a = on ? expf(lsn_a) : 1.f, sgo_a = on ? expf(lso_a) : 1.f;
  const float osc_a = (on && mode == 0) ? osc[a] : 0.f;
  const float dkl = 2.0f * sgn_a * sgn_a + 1e-8f;
  float sumn = 0.f, sumo = 0.f;
  for (int k = 0; k < m; ++k) { sumn += ls_new[k]; sumo += ls_old[k]; }
  auto asum = [&](float v) { v = head_sum32(v); return wide ? half_sum(v) : v; };
  double s_surr = 0.0, s_b = 0.0, gl = 0.0;
  const int64_t nw = (int64_t)gridDim.x * nwv;
  for (int64_t i0 = ((int64_t)blockIdx.x * nwv + wave) * SP; i0 < N; i0 += nw * SP) {
    const int64_t i = i0 + sub;
    const bool row = i < N, ok = on && row;
    const int64_t e = ok ? i * m + a : 0;
    const float x = act[e], mn = mu[e];
    const float zn = ok ? (x - mn) / sgn_a : 0.f;
    // LL_new - LL_old = sum_a -0.5 (zn - zo)(zn + zo) - (sum log_std_new - sum log_std_old): the DIFFERENCE per action, summed --
    // not two sums of magnitude ~m/2 + |sum log_std| subtracted afterwards (their fp32 rounding, ~|LL| x 6e-8, went straight into
    // the likelihood ratio: the old != new gradients sat 5e-6..1.3e-5 from the fp64 oracle because of it)
    float dll = 0.f, kl = 0.f;
    if (mu_old) {
      const float mo = mu_old[e];
      const float zo = ok ? (x - mo) / sgo_a : 0.f;
      dll = asum(-0.5f * (zn - zo) * (zn + zo)) - (sumn - sumo);
      const float Nr = (mo - mn) * (mo - mn) + sgo_a * sgo_a - sgn_a * sgn_a;
      kl = asum(ok ? Nr / dkl + lsn_a - lso_a : 0.f);
    }
    const float LR = expf(dll), ad = adv[row ? i : 0];
    if (a == 0 && row) { s_surr += (double)(LR * ad); s_b += (mode == 0) ? 1.0 : (double)kl; }
    if (mode == 0 && ok) {
      const float w = ad * LR * inv_N;
      d3[e] = osc_a * (w * zn / sgn_a);
      gl += (double)(w * (zn * zn - 1.0f));
    }
  }
  shg[wave][lane] = gl;
  __syncthreads();
  if (mode == 0 && (int)threadIdx.x < m) {       // fixed order: waves, then the two lane halves
    double t = 0.0;
    for (int w = 0; w < nwv; ++w)
      for (int h = 0; h < SP; ++h) t += shg[w][h * 32 + threadIdx.x];
    part[(size_t)blockIdx.x * (2 + MPH) + 2 + threadIdx.x] = t;
  }
  s_surr = block_sum(s_surr, sh);
  s_b = block_sum(s_b, sh);
  if (threadIdx.x == 0) { part[(size_t)blockIdx.x * (2 + MPH)] = s_surr; part[(size_t)blockIdx.x * (2 + MPH) + 1] = s_b; }
}

// ---- minibatch losses of the torch-optimizer paths, one block per minibatch ----
// loss 0: MSE(mu, a)            torch.nn.MSELoss, mean over B*m elements        behavior_cloning.py:96-105
// loss 1: -mean LL(a | mu, s)   mean_LL of gaussian_mlp.py:99-115               behavior_cloning.py:83-94
// loss 2: -mean min(LR adv, clamp(LR, 1-c, 1+c) adv), LR = exp(LL - LL_old)     ppo_clip.py:49-56
// writes d3 = dLoss/d(pre-affine output) (B x m), gls = dLoss/dlog_std (m) and the loss value.
// rows: gathered minibatch (mu, mu_old, act, adv are B-row blocks)
__global__ __launch_bounds__(256) void k_minibatch_head(int loss, const float* __restrict__ mu, const float* __restrict__ mu_old,
                                                        const float* __restrict__ act, const float* __restrict__ adv, int B, int m,
                                                        const float* __restrict__ ls, const float* __restrict__ ls_old,
                                                        const float* __restrict__ osc, float clip, float* __restrict__ d3,
                                                        float* __restrict__ gls, double* __restrict__ loss_out) {
  __shared__ double sh[17];
  __shared__ float sg[MPH], sgo[MPH], lsn[MPH], lso[MPH], oscs[MPH];
  __shared__ double glacc[MPH];
  if (threadIdx.x < m) {
    lsn[threadIdx.x] = ls[threadIdx.x]; sg[threadIdx.x] = expf(ls[threadIdx.x]);
    lso[threadIdx.x] = ls_old ? ls_old[threadIdx.x] : 0.f; sgo[threadIdx.x] = ls_old ? expf(ls_old[threadIdx.x]) : 1.f;
    oscs[threadIdx.x] = osc[threadIdx.x];
    glacc[threadIdx.x] = 0.0;
  }
  __syncthreads();
  const float invB = 1.0f / (float)B;
  double lacc = 0.0;
  float sumn = 0.f, sumo = 0.f;
  for (int a = 0; a < m; ++a) { sumn += lsn[a]; sumo += lso[a]; }
  const float c = 0.5f * (float)m * 1.8378770664093453f;
  // per-row weight w: dLoss/dLL of the row (MLE: -1/B; PPO: -adv LR mask / B)
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    if (loss == 0) {
      for (int a = 0; a < m; ++a) {
        const float e = mu[i * m + a] - act[i * m + a];
        d3[i * m + a] = oscs[a] * (2.0f * e * invB / (float)m);
        lacc += (double)e * (double)e;
      }
    } else {
      float lln = 0.f, llo = 0.f;
      for (int a = 0; a < m; ++a) {
        const float x = act[i * m + a];
        const float zn = (x - mu[i * m + a]) / sg[a];
        lln = fmaf(-0.5f * zn, zn, lln);
        if (loss == 2) { const float zo = (x - mu_old[i * m + a]) / sgo[a]; llo = fmaf(-0.5f * zo, zo, llo); }
      }
      lln = lln - sumn - c;
      float w;
      if (loss == 1) { w = -invB; lacc += (double)lln; }
      else {
        llo = llo - sumo - c;
        const float LR = expf(lln - llo), ad = adv[i];
        const float s1 = LR * ad, s2 = fminf(fmaxf(LR, 1.0f - clip), 1.0f + clip) * ad;
        lacc += (double)fminf(s1, s2);
        const bool inside = (LR >= 1.0f - clip) && (LR <= 1.0f + clip);
        // d min(s1, s2)/dLR: adv where the unclipped branch is (co-)active; ties inside the clip range add up to adv
        w = (inside || s1 < s2) ? -ad * LR * invB : 0.f;
      }
      for (int a = 0; a < m; ++a) {
        const float zn = (act[i * m + a] - mu[i * m + a]) / sg[a];
        d3[i * m + a] = oscs[a] * (w * zn / sg[a]);                  // dLL/dmu = z / sigma
      }
    }
  }
  // dLoss/dlog_std[a] = sum_rows w (z^2 - 1): one fixed-order block reduction per action
  if (loss != 0) {
    for (int a = 0; a < m; ++a) {
      double g = 0.0;
      for (int i = threadIdx.x; i < B; i += blockDim.x) {
        float lln = 0.f, llo = 0.f, w;
        if (loss == 1) w = -invB;
        else {
          for (int b = 0; b < m; ++b) {
            const float x = act[i * m + b];
            const float zn = (x - mu[i * m + b]) / sg[b];
            lln = fmaf(-0.5f * zn, zn, lln);
            const float zo = (x - mu_old[i * m + b]) / sgo[b]; llo = fmaf(-0.5f * zo, zo, llo);
          }
          const float LR = expf((lln - sumn - c) - (llo - sumo - c)), ad = adv[i];
          const float s1 = LR * ad, s2 = fminf(fmaxf(LR, 1.0f - clip), 1.0f + clip) * ad;
          const bool inside = (LR >= 1.0f - clip) && (LR <= 1.0f + clip);
          w = (inside || s1 < s2) ? -ad * LR * invB : 0.f;
        }
        const float z = (act[i * m + a] - mu[i * m + a]) / sg[a];
        g += (double)(w * (z * z - 1.0f));
      }
      g = block_sum(g, sh);
      if (threadIdx.x == 0) gls[a] = (float)g;
    }
  }
  lacc = block_sum(lacc, sh);
  if (threadIdx.x == 0 && loss_out) {
    if (loss == 0) loss_out[0] = lacc / ((double)B * (double)m);
    else loss_out[0] = -lacc / (double)B;
  }
}

// rows idx[0..B) of the batch -> contiguous minibatch blocks
__global__ void k_gather_minibatch(const float* __restrict__ obs, const float* __restrict__ act, const float* __restrict__ adv,
                                   const int32_t* __restrict__ idx, int B, int n, int m, float* __restrict__ Xb,
                                   float* __restrict__ Ab, float* __restrict__ advb) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, T = gridDim.x * blockDim.x;
  for (int i = t; i < B * n; i += T) { const int r = i / n, c = i - r * n; Xb[i] = obs[(int64_t)idx[r] * n + c]; }
  for (int i = t; i < B * m; i += T) { const int r = i / m, c = i - r * m; Ab[i] = act[(int64_t)idx[r] * m + c]; }
  if (adv) for (int i = t; i < B; i += T) advb[i] = adv[idx[i]];
}

// FVP head: d3 = out_scale * D * mudot / N,  D = 2/(2 sigma^2 + 1e-8)   (in place on mudot)
__global__ void k_fvp_head(float* __restrict__ mudot, int64_t N, int m, const float* __restrict__ ls,
                           const float* __restrict__ osc, float inv_N) {
  const int64_t tot = N * m;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
    int a = (int)(i % m);
    float s = expf(ls[a]);
    float Dk = 2.0f / (2.0f * s * s + 1e-8f);
    mudot[i] = osc[a] * (Dk * mudot[i] * inv_N);
  }
}

// General (theta_new != theta_old) KL Hessian head, per sample and action j  (gaussian_mlp.py:135-145):
//   e = mu_n - mu_o, u = sigma_n^2, D = 2u + 1e-8
//   d3  = out_scale * (2 e / D) / N                                  first-order cotangent on the pre-affine output
//   Rd3 = out_scale * ((2/D) mudot + c_ms v_s) / N,  c_ms = -8 e u / D^2
//   hs_j += (c_ms mudot + g2 v_s) / N,  g2 = -(1e-8 + 2A) 4u (1e-8 - 2u) / D^3,  A = e^2 + sigma_o^2
// mudot (in: post-affine tangent of mu) is overwritten with Rd3; part: [gridDim.x][MPH] doubles.
__global__ __launch_bounds__(256) void k_hvp_head(const float* __restrict__ mu, const float* __restrict__ mu_old,
                                                  float* __restrict__ mudot, float* __restrict__ d3, int64_t N, int m,
                                                  const float* __restrict__ ls_new, const float* __restrict__ ls_old,
                                                  const float* __restrict__ vs, const float* __restrict__ osc, float inv_N,
                                                  double* __restrict__ part) {
  __shared__ double sh[17];
  for (int a = 0; a < m; ++a) {
    const float sn = expf(ls_new[a]), so = expf(ls_old[a]);
    const float u = sn * sn, D = 2.0f * u + 1e-8f, va = vs[a], oa = osc[a];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
      const int64_t o = i * m + a;
      const float e = mu[o] - mu_old[o], md = mudot[o];
      const float cms = -8.0f * e * u / (D * D);
      const float A2 = e * e + so * so;
      const float g2 = -(1e-8f + 2.0f * A2) * 4.0f * u * (1e-8f - 2.0f * u) / (D * D * D);
      d3[o] = oa * (2.0f * e / D) * inv_N;
      mudot[o] = oa * ((2.0f / D) * md + cms * va) * inv_N;
      acc += (double)((cms * md + g2 * va) * inv_N);
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * MPH + a] = acc;
  }
}
__global__ void k_reduce_hs(const double* __restrict__ part, int G, int m, float* __restrict__ out) {
  __shared__ double sh[17];
  for (int a = 0; a < m; ++a) {
    double t = 0.0;
    for (int g = threadIdx.x; g < G; g += blockDim.x) t += part[(size_t)g * MPH + a];
    t = block_sum(t, sh);
    if (threadIdx.x == 0) out[a] = (float)t;
  }
}
// p <- p * (1 - y^2)
__global__ void k_scale_dtanh(float* __restrict__ p, const float* __restrict__ y, int64_t cnt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x) {
    float yy = y[i];
    p[i] *= fmaf(-yy, yy, 1.0f);
  }
}

// column sums of a (N x h) matrix, split over row ranges: part[z][h]
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ D, int64_t N, int h, int64_t ld, float* __restrict__ part) {
  __shared__ float sh[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  int64_t chunk = (N + gridDim.y - 1) / gridDim.y;
  int64_t lo = blockIdx.y * chunk, hi = min(N, lo + chunk);
  float a = 0.f;
  if (c < h)
    for (int64_t r = lo + rg; r < hi; r += 4) a += D[r * ld + c];
  sh[rg][threadIdx.x & 63] = a;
  __syncthreads();
  if (rg == 0 && c < h) part[(size_t)blockIdx.y * h + c] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// column sums of a narrow (N x h, h <= 32) matrix: every thread owns whole rows (contiguous h floats), so a wave
// reads one contiguous block of memory per step; part[blockIdx.x][h], fixed order
template <int HMAX>
__global__ __launch_bounds__(256) void k_colsum_narrow(const float* __restrict__ D, int64_t N, int h, float* __restrict__ part) {
  __shared__ float sh[4][HMAX];
  float acc[HMAX];
#pragma unroll
  for (int c = 0; c < HMAX; ++c) acc[c] = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < N; r += (int64_t)gridDim.x * 256) {
    const float* row = D + r * h;
#pragma unroll
    for (int c = 0; c < HMAX; ++c) if (c < h) acc[c] += row[c];
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < HMAX; ++c) {
    float v = acc[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) sh[w][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < h) part[(size_t)blockIdx.x * h + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// out[i] = sum_z part[z][i]   (fp64 accumulate, fixed order)
__global__ void k_reduce_split(const float* __restrict__ part, int Z, int64_t cnt, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x) {
    double a = 0.0;
    for (int z = 0; z < Z; ++z) a += (double)part[(size_t)z * cnt + i];
    out[i] = (float)a;
  }
}

// the same with 16-byte loads and the splits spread over the 4 waves of a workgroup (cnt % 4 == 0): thread (w, c) sums
// slabs w, w + 4, ... of column group c in fp64, the four partial sums are added in a fixed order.  One coalesced 1 KB
// read per wave and slab; Z / 4 independent loads per thread (the scalar kernel above walks all Z slabs with one thread
// and ran at ~130 GB/s: 27 % of a cfg4 Fisher-vector product in round 1).
__global__ __launch_bounds__(256) void k_reduce_split4(const float* __restrict__ part, int Z, int64_t cnt4, float* __restrict__ out) {
  __shared__ double sh[3][64][4];
  const int c = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + c;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (i < cnt4) {
    const f32x4* __restrict__ p = (const f32x4*)part + i;
#pragma unroll 4
    for (int z = w; z < Z; z += 4) {
      const f32x4 v = p[(int64_t)z * cnt4];
      a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
    }
  }
  if (w > 0) { sh[w - 1][c][0] = a0; sh[w - 1][c][1] = a1; sh[w - 1][c][2] = a2; sh[w - 1][c][3] = a3; }
  __syncthreads();
  if (w == 0 && i < cnt4) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { a0 += sh[k][c][0]; a1 += sh[k][c][1]; a2 += sh[k][c][2]; a3 += sh[k][c][3]; }
    ((f32x4*)out)[i] = f32x4{(float)a0, (float)a1, (float)a2, (float)a3};
  }
}

__global__ void k_reduce_head(const double* __restrict__ part, int G, int m, int mode, double* __restrict__ scal, float* __restrict__ gls) {
  __shared__ double sh[17];
  for (int k = 0; k < 2 + ((mode == 0) ? m : 0); ++k) {
    double a = 0.0;
    for (int g = threadIdx.x; g < G; g += blockDim.x) a += part[(size_t)g * (2 + MPH) + k];
    a = block_sum(a, sh);
    if (threadIdx.x == 0) {
      if (k < 2) scal[k] = a;
      else gls[k - 2] = (float)a;
    }
  }
  if (threadIdx.x == 0) { scal[2] = (mode == 0) ? scal[1] : 0.0; scal[3] = 0.0; }
}

// log_std block of the FVP: frac * c(sigma) * v_s   
__global__ void k_fvp_logstd(const float* __restrict__ ls, const float* __restrict__ vs, int m, float frac, float* __restrict__ out) {
  int a = threadIdx.x;
  if (a >= m) return;
  float s = expf(ls[a]);
  float u = s * s, den = 2.0f * u + 1e-8f;
  out[a] = frac * (16.0f * u * u / (den * den) - 4.0f * u / den) * vs[a];
}

struct LayerwiseWS {
  int n = 0, m = 0;
  std::vector<int> sizes;            // n, h..., m
  std::vector<int64_t> oW, ob;       // offsets in the flat vector
  int64_t oS = 0, d = 0;
  int64_t cap = 0;                   // rows allocated
  float* Xn = nullptr;               // normalised obs (N x ldx(): rows padded with zeros to a multiple of 4 floats)
  float* V1p = nullptr; float* G1p = nullptr; float* W1p = nullptr;   // first-layer direction / gradient / weight blocks with rows padded the same way (only when ldx() != n)
  // r06: W_l^T of the hidden layers l >= 1 (made by the forward pass that fills the activation cache, from the same theta): the
  // delta product delta_l W_l then has its B operand K-contiguous like the tangent products, i.e. runs as k_gemm_p<0, EPI_BACK>
  // (0.75-0.76 of the fp32-MFMA peak) instead of <1, EPI_BACK> (0.67: one ds_read_b32 per k-step and column instead of a
  // ds_read_b128 per four).  MJX_LW_WT=0: the weights as they lie.
  std::vector<float*> Wt;
  const float* wt_theta = nullptr;
  static bool wt_on() { static const bool on = [] { const char* e = getenv("MJX_LW_WT"); return !(e && e[0] == '0'); }(); return on; }
  bool wt_eligible(int l) const { return l >= 1 && l < nL() && sizes[l + 1] >= 64 && (sizes[l + 1] % 32) == 0 && (sizes[l] % 4) == 0 && sizes[l] > 32; }
  // (up to 32 features: whole 16-byte granules; beyond: whole 128-byte k-tiles, so that every k-tile segment of a row is one
  //  cache line and the first layer needs no K tail -- 376 -> 384, 39 -> 64)
  int ldx() const { return n <= 32 ? ((n + 3) & ~3) : ((n + 31) & ~31); }
  int ld_in(int l) const { return l == 0 ? ldx() : sizes[l]; }   // row stride of layer l's input activations
  std::vector<float*> H;             // hidden activations of the NEW net (cached)
  std::vector<float*> T;             // tangent / delta buffers per hidden layer
  float *mu = nullptr, *mu2 = nullptr, *d3 = nullptr;   // (N x m)
  std::vector<float*> P, RD;         // general-HVP extras: pre-activation cotangents / R-deltas (lazy)
  float* rd3 = nullptr; int64_t gen_cap = 0;
  float* part = nullptr; int64_t part_cap = 0;          // split-K partials
  // MJX_LW_OVERLAP=1 (experiment, r05): the weight gradient of layer l on a side stream beside the delta product towards layer
  // l - 1 (both only read delta_l): its own column-sum workspace, one event fork / join per layer
  float* cpart2 = nullptr; int64_t cpart2_cap = 0;
  hipStream_t side = nullptr; hipEvent_t ev_a = nullptr, ev_b = nullptr;
  double* hpart = nullptr;           // head partials
  bool fwd_valid = false;
  int64_t fwd_rows = 0;              // rows the cached activations cover (a binding narrowed to a prefix keeps them: mjx_bind_rows)
  // r06: `mu` as left by surr_vpg with old == new IS the old policy's output for rows [0, mu_rows) of `mu_obs` until something
  // overwrites it: the evaluations that close an update (K3, TRPO's trials, DAPG's surr_before) reuse it instead of running the
  // old network again -- the fused path's "ocache", for the layer-wise path.  Only the one-call updates ask for it (eval(old_cached)).
  bool mu_valid = false; int64_t mu_rows = 0; const float* mu_obs = nullptr;
  int nL() const { return (int)sizes.size() - 1; }   // number of affine layers

  void init(int n_, int m_, const std::vector<int>& hid) {
    n = n_; m = m_;
    sizes.clear(); sizes.push_back(n); for (int h : hid) sizes.push_back(h); sizes.push_back(m);
    oW.clear(); ob.clear();
    int64_t k = 0;
    for (int l = 0; l < nL(); ++l) { oW.push_back(k); k += (int64_t)sizes[l] * sizes[l + 1]; ob.push_back(k); k += sizes[l + 1]; }
    oS = k; d = k + m;
    H.assign(hid.size(), nullptr); T.assign(hid.size(), nullptr);
    P.assign(hid.size(), nullptr); RD.assign(hid.size(), nullptr);
    Wt.assign(sizes.size(), nullptr);
  }
  void invalidate() { fwd_valid = false; }
  void narrow(int64_t N) { if (N > fwd_rows) fwd_valid = false; }      // fewer rows of the same batch: the prefix of the cache stays valid
  void release() {
    hipFree(Xn); Xn = nullptr;
    hipFree(V1p); V1p = nullptr; hipFree(G1p); G1p = nullptr; hipFree(W1p); W1p = nullptr;
    for (auto& p : H) { hipFree(p); p = nullptr; }
    for (auto& p : T) { hipFree(p); p = nullptr; }
    for (auto& p : Wt) { hipFree(p); p = nullptr; }
    wt_theta = nullptr;
    hipFree(mu); hipFree(mu2); hipFree(d3); hipFree(part); hipFree(hpart); hipFree(rd3); rd3 = nullptr; gen_cap = 0;
    hipFree(cpart2); cpart2 = nullptr; cpart2_cap = 0;
    if (side) { (void)hipStreamDestroy(side); side = nullptr; }
    if (ev_a) { (void)hipEventDestroy(ev_a); ev_a = nullptr; }
    if (ev_b) { (void)hipEventDestroy(ev_b); ev_b = nullptr; }
    for (auto& p : P) { hipFree(p); p = nullptr; }
    for (auto& p : RD) { hipFree(p); p = nullptr; }
    mu = mu2 = d3 = part = nullptr; hpart = nullptr; cap = 0; part_cap = 0;
  }
  static constexpr int HEAD_G = 512;
  int reserve(int64_t N) {
    fwd_valid = false; mu_valid = false;
    if (m > MPH) return -3;
    if (N <= cap) return 0;
    int64_t newcap = (N + 127) / 128 * 128;       // whole 128-row tiles: the persistent GEMM (lw_gemm_p.h) reads / writes the padding rows
    std::vector<int> hid(sizes.begin() + 1, sizes.end() - 1);
    release();
    if (hipMalloc(&Xn, (size_t)newcap * ldx() * 4) != hipSuccess) return 2;
    (void)hipMemset(Xn, 0, (size_t)newcap * ldx() * 4);
    if (ldx() != n && nL() >= 1) {                              // padded first-layer blocks (direction / gradient)
      if (hipMalloc(&V1p, (size_t)sizes[1] * ldx() * 4) != hipSuccess) return 2;
      if (hipMalloc(&G1p, (size_t)sizes[1] * ldx() * 4) != hipSuccess) return 2;
      if (hipMalloc(&W1p, (size_t)sizes[1] * ldx() * 4) != hipSuccess) return 2;
    }
    for (size_t l = 0; l < hid.size(); ++l) {
      if (hipMalloc(&H[l], (size_t)newcap * hid[l] * 4) != hipSuccess) return 2;
      if (hipMalloc(&T[l], (size_t)newcap * hid[l] * 4) != hipSuccess) return 2;
    }
    for (int l = 1; l < nL(); ++l)
      if (wt_eligible(l) && hipMalloc(&Wt[l], (size_t)sizes[l] * sizes[l + 1] * 4) != hipSuccess) return 2;
    if (hipMalloc(&mu, (size_t)newcap * m * 4) != hipSuccess) return 2;
    if (hipMalloc(&mu2, (size_t)newcap * m * 4) != hipSuccess) return 2;
    if (hipMalloc(&d3, (size_t)newcap * m * 4) != hipSuccess) return 2;
    if (hipMalloc(&hpart, (size_t)HEAD_G * (2 + MPH) * sizeof(double)) != hipSuccess) return 2;
    cap = newcap;
    return 0;
  }

  template <int BM, int BN>
  static constexpr size_t gemm_lds_bytes() {
    return 2 * sizeof(float) * (size_t)(((BM * GLD > GBK * (BM + 4)) ? BM * GLD : GBK * (BM + 4)) + ((BN * GLD > GBK * (BN + 4)) ? BN * GLD : GBK * (BN + 4)));
  }
  // MJX_LW_TILES (A/B measurements): 0 = the round-1 shapes only (128 x 128 in 512-thread workgroups / 128 x 32),
  // 1 = 128 x 256 tiles (one workgroup per CU), 2 = 128 x 128 tiles in 256-thread workgroups, two per CU
  static int tile_mode() {
    const int m = [] { const char* e = getenv("MJX_LW_TILES"); return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1; }();
    return m;
  }
  static bool wide_tiles() { return tile_mode() == 1; }
  // rows per workgroup tile (256-row tiles were tried for the weight gradients: 256 x 256 needs 128 accumulator + 80 operand
  // registers per lane and spills 874 VGPRs at two waves per SIMD; 256 x 128 buys nothing over 128 x 256)
  static int bm_of(int, bool) { return 128; }
  // column blocks a launch_gemm call covers N columns with
  static int col_blocks(int N) {
    if (N <= 32) return 1;
    if (!wide_tiles()) return (N + 127) / 128;
    const int rem = N % 256;
    return N / 256 + (rem ? 1 : 0);
  }
  template <int BM, int BN, int NTH = gemm_threads<BN>()>
  static void launch_tile(const GemmArgs& g, int splits, hipStream_t st) {
    void (*const kern)(GemmArgs) = k_gemm<BM, BN, NTH>;
    lw_set_dyn_lds((const void*)kern, (int)gemm_lds_bytes<BM, BN>());      // double-buffered operand tiles: dynamic LDS beyond the 64 KB default
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, splits);
    constexpr size_t lds = gemm_lds_bytes<BM, BN>();
    const bool fast_on = [] { const char* e = getenv("MJX_LW_FAST"); return !(e && e[0] == '0'); }();   // MJX_LW_FAST=0: A/B
    if (g.fast != (fast_on ? 1 : 0)) { GemmArgs h = g; h.fast = fast_on ? 1 : 0; launch_tile<BM, BN, NTH>(h, splits, st); return; }
#ifdef MJX_PHASE_CLOCK
    if (lw_clk_buf() && !g.clk && lw_clk_slot() < LW_CLK_SLOTS && (int64_t)grid.x * grid.y <= 16384) {
      GemmArgs h = g; h.clk = lw_clk_buf() + (int64_t)(lw_clk_slot()++) * LW_CLK_SLOT;
      hipLaunchKernelGGL(kern, grid, dim3(NTH), lds, st, h);
      return;
    }
#endif
#ifdef MJX_GEMM_EXPERIMENTS
    // timing experiment (results are WRONG): every row of a K-contiguous A operand reads the same 128-byte line, so the
    // k-loop's activation stream comes from cache -- separates memory stalls from issue / LDS limits
    const bool alias_a = [] { const char* e = getenv("MJX_LW_DEBUG_ALIAS_A"); return e && e[0] == '1'; }();
    if (alias_a) {
      GemmArgs h = g;
      for (int p = 0; p < h.npairs; ++p) if (h.a_ks[p] == 1) h.a_rs[p] = 0;
      hipLaunchKernelGGL(kern, grid, dim3(NTH), lds, st, h);
      return;
    }
#endif
    hipLaunchKernelGGL(kern, grid, dim3(NTH), lds, st, g);
  }
  // the launch restricted to columns [j0, j0 + ncols)
  static GemmArgs col_slice(const GemmArgs& g, int j0, int ncols) {
    GemmArgs h = g;
    if (!h.cs_ld) h.cs_ld = g.N;
    h.N = ncols;
    for (int p = 0; p < g.npairs; ++p) h.B[p] = g.B[p] + (int64_t)j0 * g.b_cs[p];
    h.C = g.C + (int64_t)j0 * (g.c_cs ? g.c_cs : 1);
    if (g.colsum) h.colsum = g.colsum + j0;
    if (g.bias) h.bias = g.bias + j0;
    if (g.aux) h.aux = g.aux + j0;
    if (g.aux2) h.aux2 = g.aux2 + j0;
    if (g.aux3) h.aux3 = g.aux3 + j0;
    if (g.osc) h.osc = g.osc + j0;
    if (g.osh) h.osh = g.osh + j0;
    if (g.ls) h.ls = g.ls + j0;
    return h;
  }
  // Sample splits of a weight-gradient launch: whole rounds of workgroups over the CUs.  One workgroup per CU runs at a time
  // (LDS), so tiles x splits = 490 workgroups on 256 CUs (configs[3]: 2 row tiles x 245 splits) is two rounds, the second
  // 91 % full, each paying the per-workgroup prologue / epilogue; 2 x 128 is ONE full round of twice-as-long workgroups.
  // cost(s) = rounds(s) x (samples per workgroup + ~9 us of fixed cost in sample units); the search keeps s within [cap/4, cap].
  // Weight gradients with 33..128 columns (the first layer of a narrow observation: 512 x 40 at configs[4]) are bound by the
  // bytes they keep in flight, not by the matrix cores (3.3 us per k-tile against 1 us of MFMAs): 128 x 128 tiles in
  // 256-thread workgroups, two per CU (MJX_LW_THIN=0: one 512-thread workgroup per CU as for the wide ones)
  // samples one workgroup of a weight-gradient launch accumulates in ONE fp32 MFMA chain before its partial goes to the fp64
  // split reduction (MJX_LW_CHAIN), and the workgroup budget that caps the number of splits (MJX_LW_WG_CAP).  r05: at 1M rows x
  // 512^2 the r04 values (2 048 samples, 1 024 workgroups -> chains of 7 800 samples) left the DAPG step 1.5e-5 from the
  // reference (which itself sits 3.2e-6 from fp64 truth): round-off of a sequential fp32 chain grows with its length.  Chains of
  // 1 024 samples (up to 8 192 workgroups, 1 GB of partial slabs at that size): 7.2e-6, and the Fisher-vector product is 1 %
  // FASTER (19.27 -> 19.01 ms, tools/probe_chain_error.py: more, shorter workgroups fill the last round better).
  static int wgrad_chain() {
    const int v = [] { const char* e = getenv("MJX_LW_CHAIN"); const int x = e ? atoi(e) : 0; return x >= 256 ? x : 1024; }();
    return v;
  }
  static int wgrad_wg_cap() {
    const int v = [] { const char* e = getenv("MJX_LW_WG_CAP"); const int x = e ? atoi(e) : 0; return x >= 64 ? x : 8192; }();
    return v;
  }
  static bool thin_wgrad(int ncols) {
    const bool on = [] { const char* e = getenv("MJX_LW_THIN"); return !(e && e[0] == '0'); }();
    return on && ncols > 32 && ncols <= 128 && tile_mode() == 1;
  }
  static int pick_splits(int64_t N, int row_tiles, int ncols, int cap) {
    const int ncu = lw_ncu();
    const bool on = [] { const char* e = getenv("MJX_LW_SPLITS"); return !(e && e[0] == '0'); }();
    if (cap < 4 || !on) return cap;
    int cb = 1;                                        // column blocks of the main launch (launch_gemm)
    if (ncols > 32) {
      if (!wide_tiles()) cb = (ncols + 127) / 128;
      else { cb = ncols / 256 + ((ncols % 256) > 128 ? 1 : 0); if (cb < 1) cb = 1; }
    }
    const int64_t tl = (int64_t)row_tiles * cb;
    const int slots = ncu * (thin_wgrad(ncols) ? 2 : 1);     // (thin products run two 256-thread workgroups per CU, launch_gemm)
    int best = cap;
    double bestc = 1e300;
    for (int s = cap; s >= cap / 4 && s >= 1; --s) {
      const int64_t rounds = (tl * s + slots - 1) / slots;
      const double c = (double)rounds * ((double)N / s + 74.0);
      if (c < bestc * 0.999) { bestc = c; best = s; }
    }
    return best;
  }
  // Sample-major products in their steady-state shape go to the persistent kernel (lw_gemm_p.h).
  // MJX_LW_PERSIST=0 keeps everything on the general kernel.
  static int persistent_lb(const GemmArgs& g, int splits) {        // -> 0 / 1: the B layout it can run with, -1: not eligible
    const bool on = [] { const char* e = getenv("MJX_LW_PERSIST"); return !(e && e[0] == '0'); }();
    if (on && lw_ticket_ring() == nullptr) return -1;               // no ticket counters on this device: the general kernel serves the product
    if (!on || splits != 1 || tile_mode() != 1 || g.M < 2 * GP_BM || g.N < GP_BN || (g.N % GP_BN) != 0) return -1;
    if ((g.M % GP_BM) != 0 && !g.rows_padded) return -1;
    if ((g.epi != EPI_TANGENT && g.epi != EPI_BACK && g.epi != EPI_BIAS_TANH) || (g.c_cs != 0 && g.c_cs != 1) || g.c_zs != 0) return -1;
    if (g.epi != EPI_BIAS_TANH && !g.aux) return -1;
    if (g.epi != EPI_BACK && !g.bias) return -1;
    if (g.ldc >= (1 << 24) || g.ld_aux >= (1 << 24) || (((uintptr_t)g.C | (uintptr_t)g.aux) & 3)) return -1;
    int lb = -1;
    for (int p = 0; p < g.npairs; ++p) {
      if (g.K[p] < 2 * GP_BK || (g.K[p] % GP_BK) != 0) return -1;       // whole k-tiles, at least two
      if (g.a_ks[p] != 1 || (g.a_rs[p] & 3) != 0 || (((uintptr_t)g.A[p]) & 15) != 0) return -1;
      int l;
      if (g.b_ks[p] == 1 && (g.b_cs[p] & 3) == 0 && (((uintptr_t)g.B[p]) & 15) == 0) l = 0;
      else if (g.b_cs[p] == 1 && (g.b_ks[p] & 3) == 0 && (((uintptr_t)g.B[p]) & 15) == 0) l = 1;
      else return -1;
      if (lb >= 0 && l != lb) return -1;
      lb = l;
    }
    return lb;
  }
  template <int LB, int EPI>
  static void launch_p(const GemmArgs& g, int row_tiles, hipStream_t st) {
    void (*const kern)(GemmArgs, int, int, int*) = k_gemm_p<LB, EPI, GemmArgs>;
    lw_set_dyn_lds((const void*)kern, (int)gp_lds_bytes<LB>());
    const int ncu = lw_ncu();
    const int cbs = g.N / GP_BN, ntiles = row_tiles * cbs;
    // (a ring of counters per device, one per launch in turn: launches on different streams do not share one;
    //  persistent_lb() has checked that the ring exists)
    int* ring = lw_ticket_ring();
    static std::atomic<unsigned> turn{0};
    int* ticket = ring + (turn.fetch_add(1) & 255u);                      // (zero: the kernel's last workgroup leaves it so)
#ifdef MJX_PHASE_CLOCK
    if (lw_clk_buf() && lw_clk_slot() < LW_CLK_SLOTS) {
      GemmArgs h = g; h.clk = lw_clk_buf() + (int64_t)(lw_clk_slot()++) * LW_CLK_SLOT;
      hipLaunchKernelGGL(kern, dim3(ntiles < ncu ? ntiles : ncu), dim3(GP_NTH), gp_lds_bytes<LB>(), st, h, row_tiles, cbs, ticket);
      return;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(ntiles < ncu ? ntiles : ncu), dim3(GP_NTH), gp_lds_bytes<LB>(), st, g, row_tiles, cbs, ticket);
  }
  // (r04, measured and dropped: handing the row blocks of the last, partial ticket round -- 67 of 3 907 tiles at the configs[3]
  //  shard -- to the general kernel as 128 x 128 tiles in a second launch, to halve the tail: 4.39 -> 4.45 ms; the ticket scheme
  //  already spreads the remainder, a kernel boundary costs more than the half tile it saves.  profiles/r04_lw/ab_tail_launch.log)
  static void launch_persistent(const GemmArgs& g0, int lb, hipStream_t st) {
    GemmArgs g = g0;
    if (!g.cs_ld) g.cs_ld = g.N;
    const int row_tiles = (g.M + GP_BM - 1) / GP_BM;
    if (g.epi == EPI_BIAS_TANH) { if (lb) launch_p<1, EPI_BIAS_TANH>(g, row_tiles, st); else launch_p<0, EPI_BIAS_TANH>(g, row_tiles, st); return; }
    if (g.epi == EPI_TANGENT) { if (lb) launch_p<1, EPI_TANGENT>(g, row_tiles, st); else launch_p<0, EPI_TANGENT>(g, row_tiles, st); }
    else { if (lb) launch_p<1, EPI_BACK>(g, row_tiles, st); else launch_p<0, EPI_BACK>(g, row_tiles, st); }
  }
  // wgrad: the contraction runs over samples (split over blockIdx.z) and the M x N output is small
  static void launch_gemm(const GemmArgs& g, int splits, hipStream_t st, bool wgrad = false) {
    (void)wgrad;
    if (g.N <= 32) { launch_tile<128, 32>(g, splits, st); return; }
    { const int lb = persistent_lb(g, splits); if (lb >= 0) { launch_persistent(g, lb, st); return; } }
    if (tile_mode() == 2) { launch_tile<128, 128, 256>(g, splits, st); return; }
    if (!wide_tiles()) { launch_tile<128, 128>(g, splits, st); return; }
    // 256-column blocks; a remainder of up to 128 columns gets its own 128-column launch (a half-empty 256-column
    // block would spend matrix-core time on padding), a larger one rides in one more 256-column block
    if (splits > 1 && thin_wgrad(g.N)) {
      // (r04) up to 64 columns -- the 512 x 39 first-layer gradient of configs[4], stored 64 wide -- a 128 x 64 tile: the 128-column
      // tile spent half its matrix work and half its B-operand LDS traffic on padding (MJX_LW_THIN64=0: the r03 shape)
      const bool t64 = [] { const char* e = getenv("MJX_LW_THIN64"); return !(e && e[0] == '0'); }();
      if (t64 && g.N <= 64) { launch_tile<128, 64, 256>(g, splits, st); return; }
      launch_tile<128, 128, 256>(g, splits, st);
      return;
    }
    int n256 = (g.N / 256) * 256, rem = g.N - n256;
    if (rem > 128) { n256 = g.N; rem = 0; }
    if (n256 == g.N) { launch_tile<128, 256>(g, splits, st); return; }
    if (n256) launch_tile<128, 256>(col_slice(g, 0, n256), splits, st);
    launch_tile<128, 128>(col_slice(g, n256, rem), splits, st);
  }
  // sum of `splits` partial slabs of cnt floats (fixed order, fp64 accumulation)
  static void reduce_split(const float* part, int splits, int64_t cnt, float* out, hipStream_t st) {
    if ((cnt & 3) == 0 && (((uintptr_t)part | (uintptr_t)out) & 15) == 0 && tile_mode() != 0)
      hipLaunchKernelGGL(k_reduce_split4, dim3((unsigned)((cnt / 4 + 63) / 64)), dim3(256), 0, st, part, splits, cnt / 4, out);
    else
      hipLaunchKernelGGL(k_reduce_split, dim3(ew_grid(cnt)), dim3(256), 0, st, part, splits, cnt, out);
  }
  static int ew_grid(int64_t cnt) { int64_t g = (cnt + 255) / 256; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

  // forward pass of one parameter set; acts[l] receives hidden layer l, out receives mu
  void forward(const float* theta, const float* tr, const float* obs, int64_t N, std::vector<float*>& acts, float* out,
               hipStream_t st) {
    hipLaunchKernelGGL(k_normalize, dim3(ew_grid(N * ldx())), dim3(256), 0, st, obs, N, n, ldx(), tr, Xn);
    if (&acts == &H) {                               // the pass that fills the activation cache also transposes this theta's hidden weight blocks
      wt_theta = nullptr;
      if (wt_on()) {
        for (int l = 1; l < nL(); ++l)
          if (Wt[l]) hipLaunchKernelGGL(k_transpose, dim3((sizes[l] + 31) / 32, (sizes[l + 1] + 31) / 32), dim3(256), 0, st, theta + oW[l], sizes[l + 1], sizes[l], Wt[l]);
        wt_theta = theta;
      }
    }
    const float* in = Xn;
    for (int l = 0; l < nL(); ++l) {
      const bool last = (l == nL() - 1);
      GemmArgs g{};
      g.M = (int)N; g.N = sizes[l + 1]; g.npairs = 1; g.K[0] = sizes[l];
      g.A[0] = in; g.a_rs[0] = (l == 0) ? ldx() : sizes[l]; g.a_ks[0] = 1;
      g.B[0] = theta + oW[l]; g.b_cs[0] = sizes[l]; g.b_ks[0] = 1;
      if (l == 0 && !last && ldx() != n && W1p != nullptr) {       // padded rows: K = ldx() (the pad columns of Xn and of W1p are zero)
        hipLaunchKernelGGL(k_pad_rows, dim3(ew_grid((int64_t)sizes[1] * ldx())), dim3(256), 0, st, theta + oW[0], sizes[1], n, ldx(), W1p);
        g.K[0] = ldx(); g.B[0] = W1p; g.b_cs[0] = ldx();
      }
      g.C = last ? out : acts[l]; g.ldc = sizes[l + 1]; g.c_zs = 0;
      g.bias = theta + ob[l];
      g.epi = last ? EPI_BIAS_AFFINE : EPI_BIAS_TANH;
      g.osc = tr + 2 * n + m; g.osh = tr + 2 * n;
      g.rows_padded = last ? 0 : 1;                  // hidden activations live in workspace blocks (H / T: whole 128-row tiles)
      launch_gemm(g, 1, st);
      in = g.C;
    }
  }

  int ensure_part(int64_t floats) {
    if (floats <= part_cap) return 0;
    hipFree(part); part = nullptr; part_cap = 0;
    if (hipMalloc(&part, (size_t)floats * 4) != hipSuccess) return 2;
    part_cap = floats;
    return 0;
  }

  // backward from d3 (N x m cotangent on the pre-affine output) into grad (flat, W and b blocks).
  // Bias gradients (column sums of delta_l) come out of the GEMM that produces delta_l (per-block column sums in its
  // epilogue, reduced in a fixed order); only the top delta (d3, written by the head kernel) needs its own pass.
  // (l_start / delta0: continue below the output layer when fvp_head() already produced that layer's gradients, the
  // delta of the last hidden layer and its column sums)
  int backward(const float* theta, int64_t N, float* grad, hipStream_t st, int l_start = -1, const float* delta0 = nullptr) {
    const float* delta = delta0 ? delta0 : d3;
    bool bias_done = delta0 != nullptr;
    const bool overlap_on = [] { const char* e = getenv("MJX_LW_OVERLAP"); return e && e[0] == '1'; }();
    const bool overlap = overlap_on && N >= 65536;
    if (overlap && !side) {
      if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev_a, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&ev_b, hipEventDisableTiming) != hipSuccess) return 2;
    }
    hipStream_t main_st = st;
    if (overlap) { (void)hipEventRecord(ev_a, main_st); (void)hipStreamWaitEvent(side, ev_a, 0); }     // delta of the top layer is ready on the caller's stream
    for (int l = (l_start >= 0 ? l_start : nL() - 1); l >= 0; --l) {
      if (overlap) st = side;                        // this layer's weight gradient (and its reductions) go to the side stream
      const int ho = sizes[l + 1], hi_ = sizes[l];
      const float* in = (l == 0) ? Xn : H[l - 1];
      // weight gradient: gW[ho x hi] = delta^T (ho x N) * in (N x hi), split over samples
      const bool narrow = ho <= 32;                 // action head: compute gW^T (hi x ho) so the padding goes to 32 columns, not 128 rows
      const int rowblocks = (int)((N + 127) / 128);      // of the delta GEMM below (its column sums; sample-major launches use 128-row tiles)
      const int tbm = (narrow || hi_ <= 32) ? 128 : bm_of(ho, true);
      int tiles = narrow ? (hi_ + tbm - 1) / tbm : ((ho + tbm - 1) / tbm) * col_blocks(hi_);
      int splits = (int)((N + wgrad_chain() - 1) / wgrad_chain());
      int maxs = (wgrad_wg_cap() + tiles - 1) / tiles;
      if (splits > maxs) splits = maxs;
      if (splits < 1) splits = 1;
      splits = pick_splits(N, narrow ? (hi_ + tbm - 1) / tbm : (ho + tbm - 1) / tbm, narrow ? ho : hi_, splits);
      const int csplits = 256;
      const int rsplits = 64;                      // second-stage split of the per-row-block column sums
      // first layer of an observation width that is not a multiple of 4: the gradient block is formed with padded rows
      // (ldx() columns: 16-byte operand granules of Xn, the pad column is zero) and compacted afterwards
      const bool padw = (l == 0) && !narrow && ldx() != n && G1p != nullptr;
      const int wN = padw ? ldx() : hi_;
      if (ensure_part((int64_t)splits * ho * wN + (int64_t)csplits * ho + (int64_t)rowblocks * hi_ + (int64_t)rsplits * hi_)) return 2;
      GemmArgs g{};
      g.npairs = 1; g.K[0] = (int)N;
      if (narrow) {
        g.M = hi_; g.N = ho;
        g.A[0] = in; g.a_rs[0] = 1; g.a_ks[0] = ld_in(l);
        g.B[0] = delta; g.b_cs[0] = 1; g.b_ks[0] = ho;
        g.ldc = 1; g.c_cs = hi_;                     // C^T(i = input unit, j = output unit) -> gW[j][i]
      } else {
        g.M = ho; g.N = hi_;
        g.A[0] = delta; g.a_rs[0] = 1; g.a_ks[0] = ho;
        g.B[0] = in; g.b_cs[0] = 1; g.b_ks[0] = ld_in(l);
        g.ldc = hi_;
      }
      float* wdst = padw ? G1p : grad + oW[l];
      if (padw) { g.N = wN; g.ldc = wN; }
      // (a single split / row block -- minibatches -- writes the gradient blocks directly: no reduction launches)
      g.C = (splits == 1) ? wdst : part; g.c_zs = (int64_t)ho * wN;
      g.epi = EPI_STORE;
      launch_gemm(g, splits, st, true);
      if (splits > 1) reduce_split(part, splits, (int64_t)ho * wN, wdst, st);
      if (padw) hipLaunchKernelGGL(k_unpad_rows, dim3(ew_grid((int64_t)ho * hi_)), dim3(256), 0, st, G1p, ho, hi_, wN, grad + oW[l]);
      float* bpart = part + (int64_t)splits * ho * wN;
      float* cpart = bpart + (int64_t)csplits * ho;
      if (overlap) {
        // the delta product below runs on the caller's stream beside this layer's weight gradient: its column-sum partials get a
        // block of their own (the next layer's gradient slabs on the side stream may grow over `cpart`)
        const int64_t need = (int64_t)rowblocks * hi_ + (int64_t)rsplits * hi_;
        if (need > cpart2_cap) { hipFree(cpart2); cpart2 = nullptr; cpart2_cap = 0; if (hipMalloc(&cpart2, (size_t)need * 4) != hipSuccess) return 2; cpart2_cap = need; }
        cpart = cpart2;
      }
      if (!bias_done) {
        const int cs = (N <= 4096) ? 1 : csplits;
        float* dst = (cs == 1) ? grad + ob[l] : bpart;
        if (ho <= 32) hipLaunchKernelGGL(k_colsum_narrow<32>, dim3(cs), dim3(256), 0, st, delta, N, ho, dst);
        else hipLaunchKernelGGL(k_colsum, dim3((ho + 63) / 64, cs), dim3(256), 0, st, delta, N, ho, (int64_t)ho, dst);
        if (cs > 1)
          hipLaunchKernelGGL(k_reduce_partials, dim3((ho + 15) / 16), dim3(256), 0, st, bpart, cs, ho, grad + ob[l],
                             (const float*)nullptr, (const float*)nullptr, 0, 0.f);
      }
      if (overlap) st = main_st;                     // ... the delta product towards the layer below stays on the caller's stream
      if (l > 0) {
        // delta_{l} = (delta_{l+1} W_l) (1 - H_{l-1}^2)   -> T[l-1]   (+ its column sums = grad b_{l-1})
        GemmArgs b{};
        b.M = (int)N; b.N = hi_; b.npairs = 1; b.K[0] = ho;
        b.A[0] = delta; b.a_rs[0] = ho; b.a_ks[0] = 1;
        b.B[0] = theta + oW[l]; b.b_cs[0] = 1; b.b_ks[0] = hi_;
        if (Wt[l] && wt_theta == theta && N >= 2 * GP_BM) { b.B[0] = Wt[l]; b.b_cs[0] = ho; b.b_ks[0] = 1; }      // W_l^T: B(k, j) = Wt[j][k]
        b.C = T[l - 1]; b.ldc = hi_; b.c_zs = 0;
        b.aux = H[l - 1]; b.ld_aux = hi_;
        b.epi = EPI_BACK;
        b.rows_padded = (delta != d3) ? 1 : 0;         // (delta of a hidden layer lives in T[l]; the top delta d3 has no padding rows)
        b.colsum = (rowblocks == 1) ? grad + ob[l - 1] : cpart;
        launch_gemm(b, 1, st);
        if (rowblocks > 1024) {
          // thousands of row blocks (3 907 at 500 k samples): 16..32 workgroups walking them took 100-200 us per layer;
          // sum them in 64 row ranges first (k_colsum: 64 x hi/64 workgroups), then the 64 partial rows -- fixed order
          float* rpart = cpart + (int64_t)rowblocks * hi_;
          hipLaunchKernelGGL(k_colsum, dim3((hi_ + 63) / 64, rsplits), dim3(256), 0, st, cpart, (int64_t)rowblocks, hi_, (int64_t)hi_, rpart);
          hipLaunchKernelGGL(k_reduce_partials, dim3((hi_ + 15) / 16), dim3(256), 0, st, rpart, rsplits, hi_, grad + ob[l - 1],
                             (const float*)nullptr, (const float*)nullptr, 0, 0.f);
        } else if (rowblocks > 1)
          hipLaunchKernelGGL(k_reduce_partials, dim3((hi_ + 15) / 16), dim3(256), 0, st, cpart, rowblocks, hi_, grad + ob[l - 1],
                             (const float*)nullptr, (const float*)nullptr, 0, 0.f);
        bias_done = true;
        delta = T[l - 1];
        if (overlap) { (void)hipEventRecord(ev_a, main_st); (void)hipStreamWaitEvent(side, ev_a, 0); }   // the next layer's gradient needs this delta
      }
    }
    if (overlap) { (void)hipEventRecord(ev_b, side); (void)hipStreamWaitEvent(main_st, ev_b, 0); }      // join: the gradients are complete on the caller's stream
    return hipGetLastError() == hipSuccess ? 0 : 1;
  }

  int surr_vpg(const float* obs, const float* act, const float* adv, int64_t N, int64_t Ng, const float* th_new,
               const float* th_old, const float* tr_new, const float* tr_old, int old_is_new, float* grad, double* scal,
               hipStream_t st) {
    if (N > cap) return 1;
    const float* mo = nullptr;
    if (!old_is_new) {          // old net first (its hidden activations are scratch in T)
      forward(th_old, tr_old, obs, N, T, mu2, st);
      mo = mu2;
    }
    forward(th_new, tr_new, obs, N, H, mu, st);
    fwd_valid = true; fwd_rows = N;
    mu_valid = old_is_new != 0; mu_rows = N; mu_obs = obs;
    hipLaunchKernelGGL(k_head, dim3(HEAD_G), dim3(HEAD_NT), 0, st, 0, mu, mo, act, adv, N, m, th_new + oS, th_old + oS,
                       tr_new + 2 * n + m, (float)(1.0 / (double)Ng), d3, hpart);
    hipLaunchKernelGGL(k_reduce_head, dim3(1), dim3(256), 0, st, hpart, HEAD_G, m, 0, scal, grad + oS);
    return backward(th_new, N, grad, st);
  }

  int fvp(const float* obs, int64_t N, int64_t Ng, const float* theta, const float* tr, const float* v, float* out, hipStream_t st) {
    if (N > cap) return 1;
    if (!fwd_valid || N > fwd_rows) { forward(theta, tr, obs, N, H, mu, st); fwd_valid = true; fwd_rows = N; mu_valid = false; }
    // tangent pass
    const float* tin = nullptr;
    for (int l = 0; l < nL(); ++l) {
      const bool last = (l == nL() - 1);
      const float* in = (l == 0) ? Xn : H[l - 1];
      GemmArgs g{};
      g.M = (int)N; g.N = sizes[l + 1];
      g.npairs = tin ? 2 : 1;
      g.K[0] = sizes[l]; g.A[0] = in; g.a_rs[0] = ld_in(l); g.a_ks[0] = 1;
      g.B[0] = v + oW[l]; g.b_cs[0] = sizes[l]; g.b_ks[0] = 1;
      if (l == 0 && ldx() != n && V1p != nullptr) {       // padded rows: K = ldx() (the pad column of Xn and of V1p is zero)
        hipLaunchKernelGGL(k_pad_rows, dim3(ew_grid((int64_t)sizes[1] * ldx())), dim3(256), 0, st, v + oW[0], sizes[1], n, ldx(), V1p);
        g.K[0] = ldx(); g.B[0] = V1p; g.b_cs[0] = ldx();
      }
      if (tin) {
        g.K[1] = sizes[l]; g.A[1] = tin; g.a_rs[1] = sizes[l]; g.a_ks[1] = 1;
        g.B[1] = theta + oW[l]; g.b_cs[1] = sizes[l]; g.b_ks[1] = 1;
      }
      g.bias = v + ob[l];
      g.c_zs = 0;
      // the output layer's tangent goes straight to d3 = out_scale D mudot / N in the GEMM epilogue (no pass over N x m)
      if (last) { g.C = d3; g.ldc = m; g.epi = EPI_FVP_HEAD; g.osc = tr + 2 * n + m; g.ls = theta + oS; g.inv_N = (float)(1.0 / (double)Ng); }
      else { g.C = T[l]; g.ldc = sizes[l + 1]; g.epi = EPI_TANGENT; g.aux = H[l]; g.ld_aux = sizes[l + 1]; g.rows_padded = 1; }
      if (last && head_fused(theta, v)) break;
      launch_gemm(g, 1, st);
      tin = last ? nullptr : T[l];
    }
    hipLaunchKernelGGL(k_fvp_logstd, dim3(1), dim3(64), 0, st, theta + oS, v + oS, m, (float)((double)N / (double)Ng), out + oS);
    if (head_fused(theta, v)) {
      if (int rc = fvp_head(theta, tr, v, N, Ng, out, st)) return rc;
      return backward(theta, N, out, st, nL() - 2, T[nL() - 2]);
    }
    return backward(theta, N, out, st);
  }

  // the output layer of the product as one pass over the last hidden layer (lw_head.h); MJX_LW_HEAD=0 keeps the generic chain
  // (the pass reads theta / v with 16-byte loads: a direction that is a view at an odd offset of a larger tensor takes the
  //  generic chain instead of failing -- the fast path is an optimisation, not a precondition)
  bool head_fused(const float* theta, const float* v) const {
    const bool on = [] { const char* e = getenv("MJX_LW_HEAD"); return !(e && e[0] == '0'); }();
    if (!on || nL() < 2 || m > 32) return false;
    if ((((uintptr_t)theta | (uintptr_t)v) & 15) != 0) return false;
    const int hl = sizes[nL() - 1];
    return (hl % 128) == 0 && hl <= 512 && (oW[nL() - 1] % 4) == 0;
  }
  int fvp_head(const float* theta, const float* tr, const float* v, int64_t N, int64_t Ng, float* out, hipStream_t st) {
    const int L = nL() - 1, hl = sizes[L];
    const int ncu = lw_ncu();
    const int64_t ntile = (N + LH_R - 1) / LH_R;
    const int grid = (int)(ntile < ncu ? ntile : ncu);             // one workgroup per CU (the kernel takes the whole register file: lw_head.h)
    if (ensure_part((int64_t)grid * ((int64_t)m * hl + 32 + hl))) return 2;
    HeadArgs a{};
    a.H = H[L - 1]; a.T = T[L - 1]; a.V3 = v + oW[L]; a.W3 = theta + oW[L]; a.c3 = v + ob[L];
    a.osc = tr + 2 * n + m; a.ls = theta + oS; a.inv_N = (float)(1.0 / (double)Ng);
    a.N = N; a.h = hl; a.m = m;
    a.gw_part = part; a.gb_part = part + (int64_t)grid * m * hl; a.cs_part = a.gb_part + (int64_t)grid * 32;
    auto launch = [&](auto ch) {
      constexpr int CH = decltype(ch)::value;
      void (*const kern)(HeadArgs) = k_lw_head<CH>;
      lw_set_dyn_lds((const void*)kern, (int)lw_head_lds_bytes());
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lw_head_lds_bytes(), st, a);
    };
    auto launch8 = [&](auto ch) {             // 256 / 512 units: the eight-wave build (two waves per SIMD)
      constexpr int CH = decltype(ch)::value;
      // the delta product's contraction over the actions in ceil(m / 2) steps: 9 (m <= 18: Humanoid's 17), 12 (<= 24), else all 16
      const bool trim = [] { const char* e = getenv("MJX_LW_HEAD_KTRIM"); return !(e && e[0] == '0'); }();
      void (*const kern)(HeadArgs) = !trim ? k_lw_head8<CH, 16> : m <= 18 ? k_lw_head8<CH, 9> : m <= 24 ? k_lw_head8<CH, 12> : k_lw_head8<CH, 16>;
      lw_set_dyn_lds((const void*)kern, (int)lw_head8_lds_bytes());
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lw_head8_lds_bytes(), st, a);
    };
    const bool eight = [] { const char* e = getenv("MJX_LW_HEAD8"); return !(e && e[0] == '0'); }();
    // (r06: k_lw_head8 re-bases its buffer resources per row tile -- no 4 GB limit on the blocks any more)
    if (eight && hl == 256) launch8(std::integral_constant<int, 1>{});
    else if (eight && hl == 512) launch8(std::integral_constant<int, 2>{});
    else switch (hl / 128) {
      case 1: launch(std::integral_constant<int, 1>{}); break;
      case 2: launch(std::integral_constant<int, 2>{}); break;
      case 3: launch(std::integral_constant<int, 3>{}); break;
      default: launch(std::integral_constant<int, 4>{}); break;
    }
    reduce_split(a.gw_part, grid, (int64_t)m * hl, out + oW[L], st);
    hipLaunchKernelGGL(k_reduce_partials, dim3((m + 15) / 16), dim3(256), 0, st, a.gb_part, grid, m, out + ob[L], (const float*)nullptr,
                       (const float*)nullptr, 0, 0.f);
    hipLaunchKernelGGL(k_reduce_partials, dim3((hl + 15) / 16), dim3(256), 0, st, a.cs_part, grid, hl, out + ob[L - 1], (const float*)nullptr,
                       (const float*)nullptr, 0, 0.f);
    return 0;
  }

  // Exact Hessian-vector product of mean_kl(new, old) wrt theta_new for theta_new != theta_old
  // (npg_cg.py:62-81 in general position, e.g. under input_normalization :101-107):
  // forward + R-forward + backward + R-backward (Pearlmutter), without the damping term.
  int hvp_general(const float* obs, int64_t N, int64_t Ng, const float* th_new, const float* th_old, const float* tr_new,
                  const float* tr_old, const float* v, float* out, hipStream_t st) {
    if (N > cap) return 1;
    if (gen_cap < cap) {
      for (size_t l = 0; l < P.size(); ++l) {
        hipFree(P[l]); hipFree(RD[l]);
        if (hipMalloc(&P[l], (size_t)cap * sizes[l + 1] * 4) != hipSuccess) return 2;
        if (hipMalloc(&RD[l], (size_t)cap * sizes[l + 1] * 4) != hipSuccess) return 2;
      }
      hipFree(rd3);
      if (hipMalloc(&rd3, (size_t)cap * m * 4) != hipSuccess) return 2;
      gen_cap = cap;
    }
    const float inv_N = (float)(1.0 / (double)Ng);
    forward(th_old, tr_old, obs, N, T, mu2, st);          // old means (hidden activations are scratch)
    forward(th_new, tr_new, obs, N, H, mu, st);
    fwd_valid = true; fwd_rows = N; mu_valid = false;
    // R-forward: T_l = (in V_l^T + T_{l-1} W_l^T + c_l)(1 - H_l^2), rd3 = out_scale (.. + c_L)
    const float* tin = nullptr;
    for (int l = 0; l < nL(); ++l) {
      const bool last = (l == nL() - 1);
      const float* in = (l == 0) ? Xn : H[l - 1];
      GemmArgs g{};
      g.M = (int)N; g.N = sizes[l + 1]; g.npairs = tin ? 2 : 1;
      g.K[0] = sizes[l]; g.A[0] = in; g.a_rs[0] = ld_in(l); g.a_ks[0] = 1;
      g.B[0] = v + oW[l]; g.b_cs[0] = sizes[l]; g.b_ks[0] = 1;
      if (l == 0 && ldx() != n && V1p != nullptr) {       // padded rows: K = ldx() (the pad column of Xn and of V1p is zero)
        hipLaunchKernelGGL(k_pad_rows, dim3(ew_grid((int64_t)sizes[1] * ldx())), dim3(256), 0, st, v + oW[0], sizes[1], n, ldx(), V1p);
        g.K[0] = ldx(); g.B[0] = V1p; g.b_cs[0] = ldx();
      }
      if (tin) { g.K[1] = sizes[l]; g.A[1] = tin; g.a_rs[1] = sizes[l]; g.a_ks[1] = 1; g.B[1] = th_new + oW[l]; g.b_cs[1] = sizes[l]; g.b_ks[1] = 1; }
      g.bias = v + ob[l]; g.c_zs = 0;
      if (last) { g.C = rd3; g.ldc = m; g.epi = EPI_BIAS_AFFINE; g.osc = tr_new + 2 * n + m; g.osh = nullptr; }
      else { g.C = T[l]; g.ldc = sizes[l + 1]; g.epi = EPI_TANGENT; g.aux = H[l]; g.ld_aux = sizes[l + 1]; }
      launch_gemm(g, 1, st);
      tin = last ? nullptr : T[l];
    }
    hipLaunchKernelGGL(k_hvp_head, dim3(HEAD_G), dim3(256), 0, st, mu, mu2, rd3, d3, N, m, th_new + oS, th_old + oS, v + oS,
                       tr_new + 2 * n + m, inv_N, hpart);
    hipLaunchKernelGGL(k_reduce_hs, dim3(1), dim3(256), 0, st, hpart, HEAD_G, m, out + oS);
    // backward with R: delta (first order) and Rdelta
    const float* dl = d3; const float* rdl = rd3;
    for (int l = nL() - 1; l >= 0; --l) {
      const int ho = sizes[l + 1], hi_ = sizes[l];
      const float* in = (l == 0) ? Xn : H[l - 1];
      const float* tinl = (l == 0) ? nullptr : T[l - 1];
      const int tbm = hi_ <= 32 ? 128 : bm_of(ho, true);
      int tiles = ((ho + tbm - 1) / tbm) * col_blocks(hi_);
      int splits = (int)((N + wgrad_chain() - 1) / wgrad_chain());
      int maxs = (wgrad_wg_cap() + tiles - 1) / tiles;
      if (splits > maxs) splits = maxs;
      if (splits < 1) splits = 1;
      splits = pick_splits(N, (ho + tbm - 1) / tbm, hi_, splits);
      if (ensure_part((int64_t)splits * ho * hi_ + (int64_t)splits * ho)) return 2;
      GemmArgs g{};                                         // R{gW_l} = Rdelta^T in + delta^T Tin
      g.M = ho; g.N = hi_; g.npairs = tinl ? 2 : 1;
      g.K[0] = (int)N; g.A[0] = rdl; g.a_rs[0] = 1; g.a_ks[0] = ho; g.B[0] = in; g.b_cs[0] = 1; g.b_ks[0] = ld_in(l);
      if (tinl) { g.K[1] = (int)N; g.A[1] = dl; g.a_rs[1] = 1; g.a_ks[1] = ho; g.B[1] = tinl; g.b_cs[1] = 1; g.b_ks[1] = hi_; }
      g.C = part; g.ldc = hi_; g.c_zs = (int64_t)ho * hi_; g.epi = EPI_STORE;
      launch_gemm(g, splits, st, true);
      reduce_split(part, splits, (int64_t)ho * hi_, out + oW[l], st);
      float* bpart = part + (int64_t)splits * ho * hi_;
      hipLaunchKernelGGL(k_colsum, dim3((ho + 63) / 64, splits), dim3(256), 0, st, rdl, N, ho, (int64_t)ho, bpart);
      hipLaunchKernelGGL(k_reduce_split, dim3(ew_grid(ho)), dim3(256), 0, st, bpart, splits, (int64_t)ho, out + ob[l]);
      if (l > 0) {
        GemmArgs a{};                                       // pre = delta W_l
        a.M = (int)N; a.N = hi_; a.npairs = 1; a.K[0] = ho;
        a.A[0] = dl; a.a_rs[0] = ho; a.a_ks[0] = 1; a.B[0] = th_new + oW[l]; a.b_cs[0] = 1; a.b_ks[0] = hi_;
        a.C = P[l - 1]; a.ldc = hi_; a.c_zs = 0; a.epi = EPI_STORE;
        launch_gemm(a, 1, st);
        GemmArgs b{};                                       // Rdelta_l = (delta V_l + Rdelta W_l)(1-H^2) - 2 H T pre
        b.M = (int)N; b.N = hi_; b.npairs = 2;
        b.K[0] = ho; b.A[0] = dl; b.a_rs[0] = ho; b.a_ks[0] = 1; b.B[0] = v + oW[l]; b.b_cs[0] = 1; b.b_ks[0] = hi_;
        b.K[1] = ho; b.A[1] = rdl; b.a_rs[1] = ho; b.a_ks[1] = 1; b.B[1] = th_new + oW[l]; b.b_cs[1] = 1; b.b_ks[1] = hi_;
        b.C = RD[l - 1]; b.ldc = hi_; b.c_zs = 0;
        b.aux = H[l - 1]; b.ld_aux = hi_; b.aux2 = T[l - 1]; b.aux3 = P[l - 1]; b.epi = EPI_RBACK;
        launch_gemm(b, 1, st);
        hipLaunchKernelGGL(k_scale_dtanh, dim3(ew_grid(N * hi_)), dim3(256), 0, st, P[l - 1], H[l - 1], N * hi_);
        dl = P[l - 1]; rdl = RD[l - 1];
      }
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
  }

  // old_cached: the caller vouches that theta_old / tr_old / obs are what surr_vpg saw (the one-call updates: nothing outside the
  // library runs between their K1 and their evaluations); old_is_new: theta_new == theta_old as well (DAPG's surr_before)
  int eval(const float* obs, const float* act, const float* adv, int64_t N, const float* th_new, const float* th_old,
           const float* tr_new, const float* tr_old, double* scal, hipStream_t st, bool old_cached = false, bool old_is_new = false) {
    if (N > cap) return 1;
    static const bool reuse_on = [] { const char* e = getenv("MJX_LW_OLD_OUTPUTS"); return !(e && e[0] == '0'); }();
    const float *mnew = mu, *mold = mu2;
    if (reuse_on && old_cached && mu_valid && obs == mu_obs && N <= mu_rows) {
      if (old_is_new) { mnew = mu; mold = nullptr; }                    // LR = 1, KL = 0: no network runs at all; the caches stay
      else {
        forward(th_new, tr_new, obs, N, T, mu2, st);                    // (hidden activations are scratch; `mu` keeps the old outputs)
        fwd_valid = false;
        mnew = mu2; mold = mu;
      }
    } else {
      forward(th_old, tr_old, obs, N, T, mu2, st);
      forward(th_new, tr_new, obs, N, T, mu, st);       // hidden activations are scratch here
      fwd_valid = false; mu_valid = false;
    }
    hipLaunchKernelGGL(k_head, dim3(HEAD_G), dim3(HEAD_NT), 0, st, 2, mnew, mold, act, adv, N, m, th_new + oS, th_old + oS,
                       tr_new + 2 * n + m, 0.f, (float*)nullptr, hpart);
    hipLaunchKernelGGL(k_reduce_head, dim3(1), dim3(256), 0, st, hpart, HEAD_G, m, 2, scal, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 1;
  }
};

}  // namespace mjx
