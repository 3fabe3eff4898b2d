// lw_head.h -- the output layer of a layer-wise Fisher-vector product as ONE pass over the last hidden layer.
//
// The generic chain spends four launches on the action head (npg_cg.py:62-81 at the shapes of BASELINE configs[3] / [4]):
//   mudot = H V3^T + T W3^T + c3  -> d3 = out_scale D mudot / N      (128 x 32 GEMM, reads H and T)
//   gW3   = d3^T H, gb3 = colsum(d3)                                  (reads H again)
//   delta = (d3 W3) (1 - H^2), colsum(delta) = gb of the layer below  (reads H a third time, writes delta)
// 17..28 output columns make all of them HBM-bound: 10 GB of traffic per product at 1M x 512.  Here a persistent
// workgroup takes 64 rows at a time: phase 1 streams H and T once through LDS k-tiles into mudot (the four waves split
// 2 row blocks x 2 operand pairs; two k-tiles per thread in flight, the next row tile's first two requested before phase 2),
// d3 stays in LDS; phase 2 re-reads the H rows (just fetched: L2 / MALL hits) in accumulator layout, forms
// delta = (d3 W3)(1 - H^2) in place over T and accumulates gW3 / gb3 / colsum(delta) in registers for the whole run (each
// wave owns h / 4 columns).  6 GB instead of 10.
// Two builds:
//   k_lw_head<CH>   four waves, each owning h / 4 columns in phase 2 (h = 128 CH): 428-480 VGPRs, one wave per SIMD.  Measured
//                   (MI355X, rocprofv3): 1M x 512 x 28: 2.30 ms against 2.59 ms for the four launches; 500 k x 256 x 17: 0.55 against
//                   0.72 ms -- the padded matrix-core work (m -> 32, 131 GFLOP = 0.95 ms at 1M x 512) is not hidden behind anything.
//                   (The same kernel forced to two waves per SIMD spills 100-300 registers and runs 0.61 / 2.73 ms.)
//   k_lw_head8<CH>  eight waves, each owning h / 8 columns (h = 256 CH), phase 1 split over two wave quartets: 235-256 VGPRs, two
//                   waves per SIMD that hide each other's latency: 1.74 ms / 0.45 ms.  Used for h = 256 and 512.
//
// Requires h % 128 == 0, h <= 512, m <= 32, 16-byte aligned weight rows; everything else takes the generic chain.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fused_policy.h"

#ifndef MJX_HEAD_EXP
#define MJX_HEAD_EXP 0        // timing experiments (results WRONG): 1 = k_lw_head8 without phase 2, 2 = without phase 1's k-loop.
                              // r04: each phase alone runs 0.25 ms of the pass's 0.44 ms at the configs[3] shard (0.8 + 0.8 of 1.63 ms at
                              // configs[4]): one workgroup per CU (89 KB of LDS, 210 registers), its two phases back to back.  Picking phase 2's
                              // H values out of the operand buffer while their k-tile is in LDS (no re-read from L2) changed nothing:
                              // profiles/r04_lw/head_phases.log
#endif

namespace mjx {

struct HeadArgs {
  const float* H;        // [N x h] last hidden activations
  float* T;              // [N x h] their tangent on entry, delta of that layer on exit
  const float* V3;       // [m x h] direction (weights of the output layer)
  const float* W3;       // [m x h] weights of the output layer
  const float* c3;       // [m] direction (bias of the output layer)
  const float* osc;      // [m] out_scale
  const float* ls;       // [m] log_std
  float inv_N;
  int64_t N;
  int h, m;
  float* gw_part;        // [grid][m x h]
  float* gb_part;        // [grid][m]
  float* cs_part;        // [grid][h]  column sums of delta
};

constexpr int LH_R = 64, LH_LD = 36, LH_MS = 33;
constexpr size_t lw_head_lds_bytes() { return sizeof(float) * (size_t)(2 * 2 * LH_R * LH_LD + 2 * 2 * 32 * LH_LD + 2 * LH_R * LH_MS); }

template <int CH>       // h = 128 * CH
__global__ __launch_bounds__(256, 1) void k_lw_head(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lhs[];
  float* Abuf = lhs;                                   // [pair][buf][64][36]
  float* Bbuf = Abuf + 2 * 2 * LH_R * LH_LD;           // [pair][buf][32][36]
  float* mud = Bbuf + 2 * 2 * 32 * LH_LD;              // [pair][64][33]; [0] becomes d3
  constexpr int h = 128 * CH, NKT = h / 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hi = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int rb1 = wv & 1, pr1 = wv >> 1;               // phase 1: row block, operand pair of this wave
  const int m = a.m;
  const int64_t ntile = (a.N + LH_R - 1) / LH_R;

  // phase-2 state that lives for the whole kernel: this wave's W3 fragments, gW3 accumulators, column sums
  f32x16 gacc[CH];
  float csum[CH];
  float gbacc = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) { gacc[i] = (f32x16)(0.f); csum[i] = 0.f; }
  // per-column constants of d3 for the combine step (thread -> action tid & 31)
  const int ca = tid & 31;
  float c_osc = 0.f, c_dk = 0.f, c_b = 0.f;
  if (ca < m) {
    const float sg = expf(a.ls[ca]);
    c_osc = a.osc[ca]; c_dk = 2.0f / (2.0f * sg * sg + 1e-8f); c_b = a.c3[ca];
  }
  // lane offsets of the accumulator layout (row unit_of(r, hi), column j) at row stride h
  uint32_t offH[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) offH[r] = (uint32_t)(unit_of(r, hi)) * (uint32_t)h + (uint32_t)j;

  // operand k-tiles travel global -> registers (TWO tiles in flight per thread: 48 KB per workgroup, the kernel is bound by
  // the bytes it keeps in flight) -> LDS (double buffer); the next row tile's first two k-tiles are requested before phase 2
  auto gload = [&](f32x4 (&ra)[2][2], f32x4 (&rbq)[2], int64_t r0, int kt) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float* __restrict__ S = p ? (const float*)a.T : a.H;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int row = (tid >> 3) + 32 * c;
        const bool ok = r0 + row < a.N;
        const f32x4 v = *(const f32x4*)(S + (ok ? (r0 + row) : 0) * (int64_t)h + 32 * kt + 4 * (tid & 7));
        ra[p][c] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      const float* __restrict__ Wm = p ? a.W3 : a.V3;
      const int act = tid >> 3;
      const f32x4 w = *(const f32x4*)(Wm + (int64_t)(act < m ? act : 0) * h + 32 * kt + 4 * (tid & 7));
      rbq[p] = act < m ? w : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&](const f32x4 (&ra)[2][2], const f32x4 (&rbq)[2], int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int c = 0; c < 2; ++c) *(f32x4*)&Abuf[((p * 2 + buf) * LH_R + (tid >> 3) + 32 * c) * LH_LD + 4 * (tid & 7)] = ra[p][c];
      *(f32x4*)&Bbuf[((p * 2 + buf) * 32 + (tid >> 3)) * LH_LD + 4 * (tid & 7)] = rbq[p];
    }
  };
  f32x4 raA[2][2], rbA[2], raB[2][2], rbB[2];
  if ((int64_t)blockIdx.x < ntile) { gload(raA, rbA, (int64_t)blockIdx.x * LH_R, 0); gload(raB, rbB, (int64_t)blockIdx.x * LH_R, 1); }

  for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int64_t row0 = tile * LH_R;
    const bool full = row0 + LH_R <= a.N;
    // ---------------- phase 1: mudot partials, k-tiles of 32 through LDS ----------------
    f32x16 acc1 = (f32x16)(0.f);
    auto compute = [&](int buf) {
      const float* Ac = Abuf + ((pr1 * 2 + buf) * LH_R + 32 * rb1 + j) * LH_LD + 4 * hi;
      const float* Bc = Bbuf + ((pr1 * 2 + buf) * 32 + j) * LH_LD + 4 * hi;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 a4 = *(const f32x4*)(Ac + 8 * q), b4 = *(const f32x4*)(Bc + 8 * q);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc1 = MJX_MFMA(a4[t], b4[t], acc1);
      }
    };
    __syncthreads();                       // the previous tile's phase 2 no longer reads mud / the operand buffers
    lstore(raA, rbA, 0);
    gload(raA, rbA, row0, 2);              // (NKT >= 4)
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < NKT; kt += 2) {
      lstore(raB, rbB, 1);                 // tile kt + 1 (its buffer's last readers passed the barrier below)
      if (kt + 3 < NKT) gload(raB, rbB, row0, kt + 3);
      compute(0);
      __syncthreads();
      if (kt + 2 < NKT) {
        lstore(raA, rbA, 0);
        if (kt + 4 < NKT) gload(raA, rbA, row0, kt + 4);
      }
      compute(1);
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) mud[(pr1 * LH_R + 32 * rb1 + unit_of(r, hi)) * LH_MS + j] = acc1[r];
    __syncthreads();
    // d3 = out_scale D (mudot out_scale) / N   (the EPI_FVP_HEAD formula of the generic chain); zero outside m x N
#pragma unroll
    for (int c = 0; c < LH_R * 32 / 256; ++c) {
      const int row = (tid >> 5) + 8 * c;
      float v = mud[row * LH_MS + ca] + mud[(LH_R + row) * LH_MS + ca] + c_b;
      v = v * c_osc;
      v = c_osc * (c_dk * v * a.inv_N);
      mud[row * LH_MS + ca] = (ca < m && row0 + row < a.N) ? v : 0.f;
    }
    __syncthreads();
    if (tid < 32) {
      float s = 0.f;
#pragma unroll 8
      for (int row = 0; row < LH_R; ++row) s += mud[row * LH_MS + tid];
      gbacc += s;
    }
    if (tile + gridDim.x < ntile) {         // the next row tile's first k-tiles fly under phase 2
      gload(raA, rbA, (tile + gridDim.x) * LH_R, 0);
      gload(raB, rbB, (tile + gridDim.x) * LH_R, 1);
    }
    // ---------------- phase 2: delta = (d3 W3)(1 - H^2) in place over T, gW3 += d3^T H ----------------
    auto phase2 = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int cb = 32 * (4 * i + wv);
        __builtin_amdgcn_sched_barrier(0);   // (keeps the scheduler from hoisting every chunk's loads to the top of the phase)
        float w3f[16];                       // B fragments of W3[:, cb .. cb + 31]: lane (j, hi) of step s holds W3[2 s + hi][cb + j]
        const float* w3p = a.W3;             // (laundered: otherwise these loads are loop-invariant, get hoisted out of the tile
        asm volatile("" : "+s"(w3p));        //  loop and all CH x 16 fragments stay live for the whole kernel -- 300 spills)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int act = 2 * s + hi;
          const float w = w3p[(int64_t)(act < m ? act : 0) * h + cb + j];
          w3f[s] = act < m ? w : 0.f;
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          __builtin_amdgcn_sched_barrier(0);
          const float* __restrict__ Hb = a.H + (row0 + 32 * rb) * (int64_t)h + cb;
          float* __restrict__ Db = a.T + (row0 + 32 * rb) * (int64_t)h + cb;
          float y[16];
          if (FULL) {
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = Hb[offH[r]];
          } else {                           // rows past N: any finite value will do (their d3 rows are zero), so clamp the row
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int64_t row = row0 + 32 * rb + unit_of(r, hi);
              y[r] = a.H[(row < a.N ? row : a.N - 1) * (int64_t)h + cb + j];
            }
          }
          f32x16 dacc = (f32x16)(0.f);
#pragma unroll
          for (int s = 0; s < 16; ++s) dacc = MJX_MFMA(mud[(32 * rb + j) * LH_MS + 2 * s + hi], w3f[s], dacc);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float dv = dacc[r] * fmaf(-y[r], y[r], 1.0f);
            csum[i] += dv;
            if (FULL) Db[offH[r]] = dv;
            else if (row0 + 32 * rb + unit_of(r, hi) < a.N) Db[offH[r]] = dv;
            gacc[i] = MJX_MFMA(mud[(32 * rb + unit_of(r, hi)) * LH_MS + j], y[r], gacc[i]);
          }
        }
      }
    };
    if (full) phase2(std::true_type{});
    else phase2(std::false_type{});
  }
  // ---------------- partial results of this workgroup ----------------
  float* gw = a.gw_part + (int64_t)blockIdx.x * m * h;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int col = 32 * (4 * i + wv) + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int act = unit_of(r, hi);
      if (act < m) gw[(int64_t)act * h + col] = gacc[i][r];
    }
    const float t = half_sum(csum[i]);
    if (hi == 0) a.cs_part[(int64_t)blockIdx.x * h + col] = t;
  }
  if (tid < m) a.gb_part[(int64_t)blockIdx.x * m + tid] = gbacc;
}

// The same pass with EIGHT waves (512 threads, two per SIMD at <= 256 registers) for last hidden layers of 256 / 512 units:
// phase 1 splits every k-tile's four k-groups over two wave quartets (4 partial mudots), phase 2 gives each wave h / 8 columns
// (one or two 32-column chunks: 16..32 accumulator registers instead of 64), so two waves per SIMD fit without spills and hide
// each other's memory latency.
constexpr size_t lw_head8_lds_bytes() { return sizeof(float) * (size_t)(2 * 2 * LH_R * LH_LD + 2 * 2 * 32 * LH_LD + 4 * LH_R * LH_MS); }

// KS: k-steps (two actions each) of the delta product of phase 2, whose contraction runs over the ACTIONS: ceil(m / 2) of the 16
// steps a 32-action padding allows are enough -- the fragments of the actions >= m are zero (r05: 9 steps at the 17 actions of
// BASELINE configs[3] instead of 16: 14 of a tile's 128 matrix instructions per wave gone, and as many weight-fragment loads).
// Phase 1 (N dimension = actions) and the gW3 product (M dimension = actions) keep their 32-wide padding: the 32 x 32 matrix
// instruction has no narrower output.
template <int CH, int KS = 16>       // h = 256 * CH
__global__ __launch_bounds__(512, 2) void k_lw_head8(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lhs[];
  float* Abuf = lhs;                                   // [pair][buf][64][36]
  float* Bbuf = Abuf + 2 * 2 * LH_R * LH_LD;           // [pair][buf][32][36]
  float* mud = Bbuf + 2 * 2 * 32 * LH_LD;              // [4 partials][64][33]; [0] becomes d3
  constexpr int h = 256 * CH, NKT = h / 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hi = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int rb1 = wv & 1, pr1 = (wv >> 1) & 1, kh1 = wv >> 2;      // phase 1: row block, operand pair, half of the k-groups
  const int m = a.m;
  const int64_t ntile = (a.N + LH_R - 1) / LH_R;
  f32x16 gacc[CH];
  float csum[CH];
  float gbacc = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) { gacc[i] = (f32x16)(0.f); csum[i] = 0.f; }
  const int ca = tid & 31;
  float c_osc = 0.f, c_dk = 0.f, c_b = 0.f;
  if (ca < m) {
    const float sg = expf(a.ls[ca]);
    c_osc = a.osc[ca]; c_dk = 2.0f / (2.0f * sg * sg + 1e-8f); c_b = a.c3[ca];
  }
  const uint32_t laneoff = (uint32_t)(4 * hi) * (uint32_t)h + (uint32_t)j;
  auto rowof = [](int r) { return (r & 3) + 8 * (r >> 2); };

  // r04: buffer addressing throughout -- resource base (scalar) + scalar row / k-tile offset + ONE 32-bit lane offset per access
  // kind, formed once.  H and T are workspace blocks allocated in whole 128-row tiles (LayerwiseWS::reserve), so rows past N
  // are readable: their mudot rows are zeroed where d3 is formed (a select, so NaN garbage does not survive) and nothing else of
  // them is used -- the per-load row masks of the first version are gone.  The weight matrices are [m x h]: their resources
  // are bounded at m rows, reads of the rows m..31 return zero by the hardware's range check (no clamp, no select).
  // (The first version formed a 64-bit address and two selects per load: 1 092 vector-ALU instructions per 144 MFMAs in the
  //  tile loop, tools/isa_mix.py -- vector-ALU time that fp32 MFMAs do not hide.)
  // r06: the H / T resources are re-based at the row tile they serve (64-bit scalar arithmetic, four SGPRs each -- free next to
  // the matrix pipe), so every offset below is relative to the tile and fits 32 bits whatever the block size: BASELINE configs[4]
  // at its full 8M rows (16 GB per block) takes this kernel too (before: blocks >= 4 GB fell back to the four-wave k_lw_head).
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  auto rsrc_at = [&](const float* base, int64_t r0) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + r0 * (int64_t)h), 0, -1, 0x00020000);
  };
  const int wbytes = m * h * 4;
  const __amdgpu_buffer_rsrc_t rV3 = __builtin_amdgcn_make_buffer_rsrc((void*)a.V3, 0, wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW3 = __builtin_amdgcn_make_buffer_rsrc((void*)a.W3, 0, wbytes, 0x00020000);
  const uint32_t voA = ((uint32_t)(tid >> 3) * (uint32_t)h + 4u * (uint32_t)(tid & 7)) * 4u;           // row tid / 8 of the tile, 16 bytes of the k-tile
  const uint32_t voW = ((uint32_t)((tid >> 3) & 31) * (uint32_t)h + 4u * (uint32_t)(tid & 7)) * 4u;    // action (tid / 8) % 32
  const uint32_t lane4 = laneoff * 4u;
  auto ld128 = [](const __amdgpu_buffer_rsrc_t& r, uint32_t vo, uint32_t so) {
    return __builtin_bit_cast(f32x4, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(r, (int)vo, (int)so, 0));
  };
  // per k-tile: A 2 pairs x 64 rows x 8 float4 = 1024 -> two per thread; B 2 pairs x 32 rows x 8 = 512 -> one per thread
  auto gload = [&](f32x4 (&ra)[2], f32x4& rbq, int64_t r0, int kt) {
    const __amdgpu_buffer_rsrc_t rH = rsrc_at(a.H, r0), rT = rsrc_at((const float*)a.T, r0);
    const uint32_t so = (uint32_t)(32 * kt) * 4u;
    ra[0] = ld128(rH, voA, so);
    ra[1] = ld128(rT, voA, so);
    const uint32_t sw = (uint32_t)(32 * kt) * 4u;
    rbq = (tid >> 8) ? ld128(rW3, voW, sw) : ld128(rV3, voW, sw);
  };
  auto lstore = [&](const f32x4 (&ra)[2], const f32x4& rbq, int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p) *(f32x4*)&Abuf[((p * 2 + buf) * LH_R + (tid >> 3)) * LH_LD + 4 * (tid & 7)] = ra[p];
    *(f32x4*)&Bbuf[(((tid >> 8) * 2 + buf) * 32 + ((tid >> 3) & 31)) * LH_LD + 4 * (tid & 7)] = rbq;
  };
  f32x4 raA[2], rbA, raB[2], rbB;
  if ((int64_t)blockIdx.x < ntile) { gload(raA, rbA, (int64_t)blockIdx.x * LH_R, 0); gload(raB, rbB, (int64_t)blockIdx.x * LH_R, 1); }

  for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int64_t row0 = tile * LH_R;
    const bool full = row0 + LH_R <= a.N;
    f32x16 acc1 = (f32x16)(0.f);
    auto compute = [&](int buf) {
      const float* Ac = Abuf + ((pr1 * 2 + buf) * LH_R + 32 * rb1 + j) * LH_LD + 4 * hi + 16 * kh1;
      const float* Bc = Bbuf + ((pr1 * 2 + buf) * 32 + j) * LH_LD + 4 * hi + 16 * kh1;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 a4 = *(const f32x4*)(Ac + 8 * q), b4 = *(const f32x4*)(Bc + 8 * q);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc1 = MJX_MFMA(a4[t], b4[t], acc1);
      }
    };
#if MJX_HEAD_EXP == 2
    if (false) {
#else
    {
#endif
    __syncthreads();
    lstore(raA, rbA, 0);
    gload(raA, rbA, row0, 2);              // (NKT >= 8)
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < NKT; kt += 2) {
      lstore(raB, rbB, 1);
      if (kt + 3 < NKT) gload(raB, rbB, row0, kt + 3);
      compute(0);
      __syncthreads();
      if (kt + 2 < NKT) {
        lstore(raA, rbA, 0);
        if (kt + 4 < NKT) gload(raA, rbA, row0, kt + 4);
      }
      compute(1);
      __syncthreads();
    }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) mud[((kh1 * 2 + pr1) * LH_R + 32 * rb1 + unit_of(r, hi)) * LH_MS + j] = acc1[r];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < LH_R * 32 / 512; ++c) {
      const int row = (tid >> 5) + 16 * c;
      float v = (mud[row * LH_MS + ca] + mud[(LH_R + row) * LH_MS + ca]) + (mud[(2 * LH_R + row) * LH_MS + ca] + mud[(3 * LH_R + row) * LH_MS + ca]) + c_b;
      v = v * c_osc;
      v = c_osc * (c_dk * v * a.inv_N);
      mud[row * LH_MS + ca] = (ca < m && row0 + row < a.N) ? v : 0.f;
    }
    __syncthreads();
    if (tid < 32) {
      float s = 0.f;
#pragma unroll 8
      for (int row = 0; row < LH_R; ++row) s += mud[row * LH_MS + tid];
      gbacc += s;
    }
    if (tile + gridDim.x < ntile) { gload(raA, rbA, (tile + gridDim.x) * LH_R, 0); gload(raB, rbB, (tile + gridDim.x) * LH_R, 1); }
    auto phase2 = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      const __amdgpu_buffer_rsrc_t rH = rsrc_at(a.H, row0), rT = rsrc_at((const float*)a.T, row0);
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int cb = 32 * (8 * i + wv);
        __builtin_amdgcn_sched_barrier(0);
        float w3f[KS];
        uint32_t w3o = ((uint32_t)hi * (uint32_t)h + (uint32_t)j) * 4u;     // lane part of W3[2 s + hi][cb + j]; laundered: keeps these loads
        asm volatile("" : "+v"(w3o));                                       // inside the tile loop (hoisted, CH x 16 fragments stay live: spills)
#pragma unroll
        for (int s = 0; s < KS; ++s)
          w3f[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rW3, (int)w3o, (int)((uint32_t)(2 * s * h + cb) * 4u), 0));   // rows >= m: 0
#pragma unroll 1
        for (int rb = 0; rb < 2; ++rb) {
          __builtin_amdgcn_sched_barrier(0);
          float* __restrict__ Db = a.T + (row0 + 32 * rb) * (int64_t)h + cb;
          const uint32_t sob = (uint32_t)((32 * rb) * h + cb) * 4u;                     // (relative to the tile's resources)
          float y[16];
          if (FULL) {
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rH, (int)lane4, (int)(sob + (uint32_t)(rowof(r) * h) * 4u), 0));
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int64_t row = row0 + 32 * rb + unit_of(r, hi);
              y[r] = a.H[(row < a.N ? row : a.N - 1) * (int64_t)h + cb + j];
            }
          }
          f32x16 dacc = (f32x16)(0.f);
#pragma unroll
          for (int s = 0; s < KS; ++s) dacc = MJX_MFMA(mud[(32 * rb + j) * LH_MS + 2 * s + hi], w3f[s], dacc);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float dv = dacc[r] * fmaf(-y[r], y[r], 1.0f);
            csum[i] += dv;
            if (FULL) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dv), rT, (int)lane4, (int)(sob + (uint32_t)(rowof(r) * h) * 4u), 0);
            else if (row0 + 32 * rb + unit_of(r, hi) < a.N) (Db + rowof(r) * h)[laneoff] = dv;
            gacc[i] = MJX_MFMA(mud[(32 * rb + unit_of(r, hi)) * LH_MS + j], y[r], gacc[i]);
          }
        }
      }
    };
#if MJX_HEAD_EXP != 1
    if (full) phase2(std::true_type{});
    else phase2(std::false_type{});
#endif
  }
  float* gw = a.gw_part + (int64_t)blockIdx.x * m * h;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int col = 32 * (8 * i + wv) + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int act = unit_of(r, hi);
      if (act < m) gw[(int64_t)act * h + col] = gacc[i][r];
    }
    const float t = half_sum(csum[i]);
    if (hi == 0) a.cs_part[(int64_t)blockIdx.x * h + col] = t;
  }
  if (tid < m) a.gb_part[(int64_t)blockIdx.x * m + tid] = gbacc;
}

}  // namespace mjx
