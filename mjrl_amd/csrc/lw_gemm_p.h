// lw_gemm_p.h -- persistent interior-tile GEMM for the sample-major products of the layer-wise Fisher-vector product.
//
// The per-workgroup clocks of the general kernel (tools/lw_clock.py) show a 128 x 256 tile whose k-loop
// already runs at the matrix-core rate, wrapped in 2.5-3.5 us of prologue (first operand tiles: one exposed memory latency),
// 5-6 us of epilogue (the activation operand's latency + 2 us of store issue) and 0.3-0.8 us of dispatch gap -- 8-14 % of a
// K <= 512 product, with nothing to overlap it: one workgroup per CU (108 KB of LDS).  The general k_gemm cannot take
// another live register (256 VGPRs at two waves per SIMD; every addition re-allocates its hot loops, +-5 %), so this is a
// separate, lean kernel for exactly the steady-state case:
//   * one workgroup per CU walks the 128 x 256 tiles of the product (tickets from an atomic counter); the operands live in workspace buffers
//     whose row count is padded to a multiple of 128 (LayerwiseWS::reserve), so the last row tile is computed like the others
//     -- its surplus rows read and write padding, only the column sums mask them (a masked variant of the loads / stores cost
//     the hot loop 5-8 % through register allocation);
//   * A operands K-contiguous (activations), B either K-contiguous (LB = 0: weights as "NT", the tangent products) or
//     row-contiguous (LB = 1: weights as "NN", the backward products); K of every operand pair a multiple of 32, >= 64 (the
//     first layer gets there through observation rows padded to whole 128-byte k-tiles: LayerwiseWS::ldx);
//   * under the MFMAs of a tile's last two k-tiles it requests the epilogue's activation block (64 registers) and the NEXT
//     tile's first operand k-tile, so that the epilogue computes on data that has arrived and the next k-loop starts without a
//     cold prologue; the epilogue's stores drain under the next tile's MFMAs.
// Epilogues: tangent ((acc + c)(1 - y^2)), delta (acc (1 - y^2) + column sums), forward (tanh(acc + b)).
// Other shapes and other epilogues stay with the general kernel (layerwise.h).
// Same arithmetic in the same order as k_gemm's fast path: bit-identical products; the column sums of the delta products
// (bias gradients) can differ in the last bit (the compiler contracts `sum += acc * factor` differently around the row mask);
// tests/test_gpu_parity.py::test_persistent_gemm_bitwise_equals_general_kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "fused_policy.h"

// (included by layerwise.h after GemmArgs and the EPI_* kinds are declared)
namespace mjx {

constexpr int GP_BM = 128, GP_BN = 256, GP_BK = 32, GP_LD = GP_BK + 4, GP_NTH = 512;
constexpr int GP_ASZ = GP_BM * GP_LD;                                                  // [row][k] image of the A tile
template <int LB> constexpr int gp_bsz() { return LB ? GP_BK * (GP_BN + 4) : GP_BN * GP_LD; }
template <int LB> constexpr size_t gp_lds_bytes() { return sizeof(float) * (size_t)(2 * GP_ASZ + 2 * gp_bsz<LB>() + 2 * GP_BN); }

template <int LB, int EPI, class Args>
__global__ __launch_bounds__(GP_NTH, 2) void k_gemm_p(Args g, int row_tiles, int col_blocks, int* __restrict__ ticket) {
  constexpr int WN = 4, TM = 64, TN = 64, MT = 2, NT = 2;
  constexpr int BSZ = gp_bsz<LB>();
  constexpr int CA = GP_BM * 8 / GP_NTH, CB = GP_BN * 8 / GP_NTH;        // float4 per thread and operand k-tile: 2, 4
  extern __shared__ __attribute__((aligned(16))) float gps[];
  float* As = gps;                         // [2][GP_ASZ]
  float* Bs = gps + 2 * GP_ASZ;            // [2][BSZ]
  float* Cs = Bs + 2 * BSZ;                // [2][256]: column sums of the two wave rows (EPI_BACK)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hi = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wv / WN, wn = wv % WN;
  __shared__ int s_next;
  const int ntiles = row_tiles * col_blocks;
  int t = blockIdx.x;                      // the first tile of every workgroup is its block index, the following ones come from
                                           // a ticket counter: 3 907 tiles over 256 workgroups as a static stride would end in a
                                           // 16th round that only 67 workgroups run (+4.7 %).
  // The counter is zero when a launch starts and is put back to zero by the LAST workgroup that leaves (ticket[256] counts the
  // leavers; every workgroup has drawn its final ticket before it is counted): no memset launch in front of every product
  // (r04: three 5 us fill kernels per Fisher-vector product).
  auto leave = [&]() {
    if (tid == 0) {
      const int d = atomicAdd(ticket + 256, 1);
      if (d == (int)gridDim.x - 1) { atomicExch(ticket + 256, 0); atomicExch(ticket, 0); }
    }
  };
  if (t >= ntiles) { leave(); return; }
  const int KT0 = g.K[0] / GP_BK, KT = KT0 + (g.npairs > 1 ? g.K[1] / GP_BK : 0);

  // Operand addresses of the k-tile that is loaded next: ONE wave-uniform base per operand (scalar registers, bumped by one
  // k-tile per load on the scalar unit) + per-thread 32-bit byte offsets that stay put for a whole tile.  (r03: six 64-bit
  // per-thread pointers cost 12 vector adds per k-tile and wave -- fp32 MFMAs do not hide vector-ALU instructions,
  // tools/probe_fill.hip.  Tried on top and dropped: making the LDS buffer index a compile-time constant, to fold the
  // remaining 17 LDS-address adds into immediates -- the kernel sits at its 256-register limit, the duplicated k-tile bodies
  // spilled 324 bytes.)
  typedef const char __attribute__((address_space(1)))* gptr;
  gptr ua = nullptr, ub = nullptr;
  uint32_t oa[CA], ob[CB];
  int64_t sb = 0;
  auto rebase = [&](int tile, int p) {
    const int m0 = (tile / col_blocks) * GP_BM, n0 = (tile % col_blocks) * GP_BN;
    const float* __restrict__ Ap = g.A[p];
    const float* __restrict__ Bp = g.B[p];
    ua = (gptr)(Ap + (int64_t)m0 * g.a_rs[p]);
#pragma unroll
    for (int c = 0; c < CA; ++c) {
      const int idx = tid + GP_NTH * c;
      oa[c] = (uint32_t)(((int64_t)(idx >> 3) * g.a_rs[p] + 4 * (idx & 7)) * 4);
    }
    ub = LB ? (gptr)(Bp + n0) : (gptr)(Bp + (int64_t)n0 * g.b_cs[p]);
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const int idx = tid + GP_NTH * c;
      ob[c] = LB ? (uint32_t)(((int64_t)(idx / (GP_BN / 4)) * g.b_ks[p] + 4 * (idx % (GP_BN / 4))) * 4)
                 : (uint32_t)(((int64_t)(idx >> 3) * g.b_cs[p] + 4 * (idx & 7)) * 4);
    }
    sb = (LB ? (int64_t)GP_BK * g.b_ks[p] : (int64_t)GP_BK) * 4;
  };
  f32x4 ra[CA], rb[CB];
  auto gload = [&]() {
#pragma unroll
    for (int c = 0; c < CA; ++c) ra[c] = *(const f32x4 __attribute__((address_space(1)))*)(ua + oa[c]);
    ua += GP_BK * 4;
#pragma unroll
    for (int c = 0; c < CB; ++c) rb[c] = *(const f32x4 __attribute__((address_space(1)))*)(ub + ob[c]);
    ub += sb;
  };
  // LDS addresses: ONE pinned per-lane byte address per operand and use (store / fragment read), formed once per k-tile from
  // the buffer parity; everything else is an immediate offset.  (Left to the compiler, every fragment read re-derived
  // `lane offset * 4 + buffer base` with its own v_lshl_add_u32 between two MFMAs -- ~20 isolated vector-ALU instructions per
  // k-tile and wave at ~12 cycles each, tools/probe_fill.hip.)
  typedef __attribute__((address_space(3))) f32x4 lds_f4;
  typedef __attribute__((address_space(3))) float lds_f1;
  auto pin = [](const float* p) { uint32_t a = (uint32_t)(uintptr_t)(const lds_f1*)p; asm volatile("" : "+v"(a)); return a; };
  auto lstore = [&](int buf) {
    const uint32_t ad = pin(As + buf * GP_ASZ + (tid >> 3) * GP_LD + 4 * (tid & 7));
    const uint32_t bd = LB ? pin(Bs + buf * BSZ + (tid / (GP_BN / 4)) * (GP_BN + 4) + 4 * (tid % (GP_BN / 4)))
                           : pin(Bs + buf * BSZ + (tid >> 3) * GP_LD + 4 * (tid & 7));
#pragma unroll
    for (int c = 0; c < CA; ++c) *(lds_f4*)(uintptr_t)(ad + (uint32_t)((GP_NTH / 8) * c * GP_LD * 4)) = ra[c];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      if (LB) *(lds_f4*)(uintptr_t)(bd + (uint32_t)((GP_NTH / (GP_BN / 4)) * c * (GP_BN + 4) * 4)) = rb[c];
      else *(lds_f4*)(uintptr_t)(bd + (uint32_t)((GP_NTH / 8) * c * GP_LD * 4)) = rb[c];
    }
  };
  // operand fragments of k-group q (k-slot `hi` of step t carries k = 8 q + 4 hi + t for A and B alike)
  // (pa / pb: this wave's pinned fragment addresses in the buffer read -- row block `a` / column block `b`, k-group q)
  auto pinA = [&](int buf) { return pin(As + buf * GP_ASZ + (wm * TM + j) * GP_LD + 4 * hi); };
  auto pinB = [&](int buf) { return LB ? pin(Bs + buf * BSZ + (4 * hi) * (GP_BN + 4) + wn * TN + j) : pin(Bs + buf * BSZ + (wn * TN + j) * GP_LD + 4 * hi); };
  auto fragA = [&](uint32_t pa, int a, int q) { return *(const lds_f4*)(uintptr_t)(pa + (uint32_t)((32 * a * GP_LD + 8 * q) * 4)); };
  auto fragB = [&](uint32_t pb, int b, int q) {
    if (LB) {
      // (volatile: single ds_read_b32 with 16-bit offsets; as ds_read2_b32 pairs -- 8-bit offsets -- rows 1 040 bytes apart needed a
      //  v_add_u32 re-base per pair: 18 per k-tile and wave between the MFMAs of the delta products)
      const volatile lds_f1* p = (const volatile lds_f1*)(uintptr_t)(pb + (uint32_t)((8 * q * (GP_BN + 4) + 32 * b) * 4));
      return f32x4{p[0], p[GP_BN + 4], p[2 * (GP_BN + 4)], p[3 * (GP_BN + 4)]};
    }
    return *(const lds_f4*)(uintptr_t)(pb + (uint32_t)((32 * b * GP_LD + 8 * q) * 4));
  };

  // `kl` counts the k-tiles of the current output tile that have been REQUESTED so far (the loader runs two ahead)
  int kl = 0;
  auto load_next = [&]() {                 // request the next k-tile of this output tile; switches to the second operand pair
    if (kl == KT0) rebase(t, 1);
    gload();
    ++kl;
  };
  rebase(t, 0);
  load_next();                             // k-tile 0
  lstore(0);
  load_next();                             // k-tile 1 (KT >= 2)
  __syncthreads();

  const uint32_t laneC4 = ((uint32_t)(4 * hi) * (uint32_t)g.ldc + (uint32_t)j) * 4u, laneA4 = ((uint32_t)(4 * hi) * (uint32_t)g.ld_aux + (uint32_t)j) * 4u;
  auto rowof = [](int r) { return (r & 3) + 8 * (r >> 2); };

#ifdef MJX_PHASE_CLOCK
#define GP_STAMP(k) do { if (g.clk && tid == 0 && t < 16384) { __builtin_amdgcn_sched_barrier(0); \
  g.clk[8 + 8 * t + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); \
  if ((k) == 0 || (k) == 1) g.clk[8 + 8 * t + 6 + (k)] = (long long)__builtin_readcyclecounter(); \
  __builtin_amdgcn_sched_barrier(0); } } while (0)
  if (g.clk && tid == 0 && blockIdx.x == 0) {
    g.clk[0] = g.M; g.clk[1] = g.N; g.clk[2] = g.K[0]; g.clk[3] = g.npairs > 1 ? g.K[1] : 0; g.clk[4] = g.epi; g.clk[5] = 1000 + GP_BN;
    g.clk[6] = ntiles < 16384 ? ntiles : 16384; g.clk[7] = 1;
  }
#else
#define GP_STAMP(k) do {} while (0)
#endif
  for (;;) {
    const int m0 = (t / col_blocks) * GP_BM, n0 = (t % col_blocks) * GP_BN;
    // buffer resources at this tile's corner of C and of the epilogue's activation operand (raw buffers: stride 0, no bound)
    const __amdgpu_buffer_rsrc_t r_c = __builtin_amdgcn_make_buffer_rsrc((void*)(g.C + (int64_t)m0 * g.ldc + n0), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_aux = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(EPI == EPI_BIAS_TANH ? g.C : g.aux + (int64_t)m0 * g.ld_aux + n0), 0, -1, 0x00020000);
    // (the scalar offsets below do not depend on the tile: hoisted out of the tile loop they would occupy 128 scalar registers and
    //  come back as v_readlane spills -- laundering the strides keeps them recomputed per tile on the scalar unit, which is idle)
    uint32_t ldc4 = (uint32_t)g.ldc * 4u, lda4 = (uint32_t)g.ld_aux * 4u;
    asm volatile("" : "+s"(ldc4), "+s"(lda4));
    GP_STAMP(0);
#ifdef MJX_PHASE_CLOCK
    if (g.clk && tid == 0 && t < 16384) { g.clk[8 + 8 * t + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4); g.clk[8 + 8 * t + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20); }
#endif
    if (tid == 0) s_next = (int)gridDim.x + atomicAdd(ticket, 1);     // this workgroup's next tile; read after a k-tile barrier
    int tn = 0;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[a][b] = (f32x16)(0.f);
    float yall[MT * NT * 16];

    // One k-tile: 64 MFMAs per wave; `mid0` runs after the first, `mid1` after the second MFMA group.  The per-k-tile barrier
    // sits BEFORE the last MFMA group, and that group's shadow fetches the first fragments of the NEXT k-tile from the other
    // buffer (complete: every wave stored its share after this tile's first group) -- a barrier at the end of the k-tile left
    // all eight waves reading their first fragments with the matrix pipe idle (~9 400 instead of 8 192 cycles per k-tile).
    // Still safe for the buffer this tile reads: its last fragments (group 3) are in registers before the barrier, and it is
    // overwritten only after the next k-tile's first group.  `a4 / b4` enter with group 0 of k-tile kt; `more` = there is a next
    // k-tile in the other buffer.
    f32x4 a4[MT], b4[NT];
    auto first_frags = [&](int kt) {
      const uint32_t Ac = pinA(kt & 1), Bc = pinB(kt & 1);
#pragma unroll
      for (int a = 0; a < MT; ++a) a4[a] = fragA(Ac, a, 0);
#pragma unroll
      for (int b = 0; b < NT; ++b) b4[b] = fragB(Bc, b, 0);
    };
    auto body = [&](int kt, bool more, auto&& mid0, auto&& mid1) {
      const uint32_t Ac = pinA(kt & 1), Bc = pinB(kt & 1), An = pinA((kt + 1) & 1), Bn = pinB((kt + 1) & 1);
      f32x4 an[MT], bn[NT];
#pragma unroll
      for (int q = 0; q < GP_BK / 8; ++q) {
        if (q + 1 < GP_BK / 8) {
#pragma unroll
          for (int a = 0; a < MT; ++a) an[a] = fragA(Ac, a, q + 1);
#pragma unroll
          for (int b = 0; b < NT; ++b) bn[b] = fragB(Bc, b, q + 1);
        } else {
          __syncthreads();
          if (more) {
#pragma unroll
            for (int a = 0; a < MT; ++a) an[a] = fragA(An, a, 0);
#pragma unroll
            for (int b = 0; b < NT; ++b) bn[b] = fragB(Bn, b, 0);
          }
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] = MJX_MFMA(a4[a][tt], b4[b][tt], acc[a][b]);
        if (q == 0) mid0();
        if (q == 1) mid1();
#pragma unroll
        for (int a = 0; a < MT; ++a) a4[a] = an[a];
#pragma unroll
        for (int b = 0; b < NT; ++b) b4[b] = bn[b];
      }
    };
    auto nop = [] {};
    int kt = 0;
    first_frags(0);                        // (k-tile 0 is in buffer 0: the barrier before this tile's loop)
#pragma unroll 1
    for (; kt + 2 < KT; ++kt)              // steady state: k-tile kt + 1 -> the other buffer, k-tile kt + 2 -> registers
      body(kt, true, [&] { lstore((kt + 1) & 1); }, [&] { load_next(); });
    // the last two k-tiles: the epilogue's activation block is requested under the second-to-last one (its staging
    // registers are free once k-tile KT - 1 is in LDS), the NEXT output tile's first k-tile under the last one
    body(kt, true, [&] { lstore((kt + 1) & 1); },
         [&] {
           if (EPI == EPI_BIAS_TANH) return;       // (the forward products have no activation operand)
#pragma unroll
           for (int mt = 0; mt < MT; ++mt)
#pragma unroll
             for (int nt = 0; nt < NT; ++nt) {
               // buffer addressing: resource base (this tile's corner, scalar) + scalar row / block offset + ONE 32-bit lane offset --
               // no vector-ALU instruction per access (as global_load with 64-bit lane addresses: 137 v_lshl_add_u64 per tile and wave)
               const uint32_t so = (uint32_t)(wm * TM + mt * 32) * lda4 + (uint32_t)(wn * TN + nt * 32) * 4u;
#pragma unroll
               for (int r = 0; r < 16; ++r)
                 yall[(mt * NT + nt) * 16 + r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_aux, laneA4, so + (uint32_t)rowof(r) * lda4, 0));
             }
         });
    ++kt;
    tn = __builtin_amdgcn_readfirstlane(s_next);      // (wave-uniform: the tile's operand bases stay scalar)
    body(kt, false, [&] { if (tn < ntiles) { kl = 0; rebase(tn, 0); gload(); kl = 1; } }, nop);
    GP_STAMP(1); GP_STAMP(2);
    // (every wave is past the last k-tile's barrier and had all its fragments in registers before it: both operand buffers are free)
    if (tn < ntiles) {
      lstore(0);                            // next tile's k-tile 0 (requested a k-tile ago)
      // its k-tile 1 flies under the epilogue
      if (kl == KT0) rebase(tn, 1);
      gload();
      ++kl;
    }
    // ---- epilogue
    float csum[NT] = {0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + wn * TN + nt * 32 + j;
        const float bias = (EPI == EPI_TANGENT || EPI == EPI_BIAS_TANH) ? g.bias[col] : 0.f;
        const uint32_t so = (uint32_t)(wm * TM + mt * 32) * ldc4 + (uint32_t)(wn * TN + nt * 32) * 4u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float y = (EPI == EPI_BIAS_TANH) ? 0.f : yall[(mt * NT + nt) * 16 + r];
          float v = acc[mt][nt][r];
          if (EPI == EPI_BIAS_TANH) v = tanhf(v + bias);
          else if (EPI == EPI_TANGENT) v = (v + bias) * fmaf(-y, y, 1.0f);
          else v = v * fmaf(-y, y, 1.0f);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r_c, laneC4, so + (uint32_t)rowof(r) * ldc4, 0);
          if (EPI == EPI_BACK) csum[nt] += (m0 + wm * TM + mt * 32 + unit_of(r, hi) < g.M) ? v : 0.f;    // (padding rows of the last tile)
        }
      }
    if (EPI == EPI_BACK && g.colsum) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float s = half_sum(csum[nt]);
        if (hi == 0) Cs[wm * GP_BN + wn * TN + nt * 32 + j] = s;
      }
    }
    __syncthreads();                        // next tile's k-tile 0 is in buffer 0; the column sums are in Cs
    if (EPI == EPI_BACK && g.colsum && tid < GP_BN)
      g.colsum[(int64_t)(m0 / GP_BM) * (g.cs_ld ? g.cs_ld : g.N) + n0 + tid] = Cs[tid] + Cs[GP_BN + tid];
    GP_STAMP(3);
    if (tn >= ntiles) { leave(); break; }
    t = tn;
  }
}

}  // namespace mjx
