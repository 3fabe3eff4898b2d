// mlp_fit.h -- persistent single-workgroup minibatch-Adam trainer for the MLP value baseline
// (d_in -> H -> H -> 1, ReLU, H = 128, batch 64).
//
// Minibatch SGD is a strictly sequential chain of ~15k tiny steps per epoch per million samples
// (mjrl/utils/optimize_model.py:7-36, mlp_baseline.py:61-95): launched as kernels it is pure launch
// latency (14 launches / step).  Here ONE workgroup (4 waves, one per SIMD of a CU) runs the whole fit:
// weights live in LDS for the entire run, the Adam moments stream through L2, every step is
//   gather 64 rows -> forward -> MSE gradient -> backward -> Adam, in two 32-sample halves,
// with the GEMMs on v_mfma_f32_32x32x2_f32 in the same chained / operand-swapped formulation as the policy
// kernels (fused_policy.h): wave w owns units 32w..32w+31 of both hidden layers, activations are exchanged
// between waves through [unit][sample] LDS tiles, weight-gradient accumulators stay in registers until the
// owning thread applies Adam to "its" weights.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fused_policy.h"

namespace mjx {

#ifdef MJX_PHASE_CLOCK
#define MJX_FIT_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (tid == 0 && ep == 0 && mb == 100 && hb == 0) ((long long*)A.epoch_loss)[8 + (k)] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MJX_FIT_STAMP(k) do {} while (0)
#endif

// sum over the lanes 0..31 of a wave, valid in lane 0: DPP moves inside the 16-lane rows (vector-ALU latency) + one
// v_permlane16_swap instead of five ds_bpermute round trips through the LDS pipe (~100 cycles each, on the critical path of
// the wave every other wave is waiting for at the next barrier)
__device__ __forceinline__ float sum32_lane0(float v) {
  v = row16_sum(v);                                       // every lane holds its row's sum
  auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(s[0]) + __uint_as_float(s[1]);   // rows 0 + 1 (lanes 0..31), rows 2 + 3 (lanes 32..63)
}

// workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope release + acquire around s_barrier, and
// on gfx9 the release waits for vmcnt(0) -- i.e. for every global load in flight, including the next minibatch's rows, which
// are requested a whole step ahead precisely so that nobody has to wait for them (~1.5-2 k cycles from the memory-side cache at
// each of the first barriers after the request).  The trainer's barriers only ever publish LDS tiles and LDS-resident weights.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// sum32_lane0 of 16 independent values, stage by stage: each DPP step runs over all 16 registers before the next one starts, so a
// DPP instruction never reads the register the previous instruction wrote (that hazard costs two wait states -- the per-value form
// compiled to a v_add_f32_dpp / s_nop chain, ~110 s_nop per step of the one-pass trainer)
__device__ __forceinline__ void sum32_lane0_x16(float (&v)[16]) {
  auto dpp = [](float x, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xF, 0xF, true));
  };
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] += dpp(v[r], std::integral_constant<int, 0xB1>{});       // quad_perm [1,0,3,2]
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] += dpp(v[r], std::integral_constant<int, 0x4E>{});       // quad_perm [2,3,0,1]
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] += dpp(v[r], std::integral_constant<int, 0x141>{});      // row_half_mirror
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] += dpp(v[r], std::integral_constant<int, 0x140>{});      // row_mirror
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[r]), __float_as_uint(v[r]), false, false);
    v[r] = __uint_as_float(q[0]) + __uint_as_float(q[1]);
  }
}

struct MlpFitArgs {
  const float* feat;       // (N, d_in) fp32
  const float* y;          // (N)
  const int32_t* perm;     // epochs * N row indices
  int64_t N;
  int d_in, epochs;
  int64_t steps;           // minibatch steps per epoch = N / 64 - 1
  float* params;           // flat [W1 (H x d_in), b1, W2 (H x H), b2, W3 (H), b3], updated in place
  float* m;                // Adam first moment  (same layout)
  float* v;                // Adam second moment
  float* mv;               // workspace: interleaved (m, v) pairs, 2 * P floats
  int64_t step0;           // Adam steps taken before this call
  float lr, wd;
  double* epoch_loss;      // [epochs] sum over steps of the minibatch MSE
  // ---- several workgroups (k_mlp_fit<.., MULTI = true>: inputs wider than one workgroup's LDS holds; see the kernel)
  float* xch = nullptr;    // UNCACHED exchange block [2 parities][G][H][32]: every workgroup's partial first-layer pre-activations
  unsigned* bar = nullptr; // UNCACHED arrival counter (zero at launch); bar[1]: "a workgroup gave up on a barrier" (k_mlp_fit_verdict)
  int G = 1, FS = 0;       // workgroups; features per workgroup (the last one takes what is left + the bias column)
  int fault = 0;           // tests (MJX_FIT_FAULT=straggler): workgroup 0 starts 2.6 s late -- the others give up on the first barrier
};

template <int H>
struct MlpFitLayout {
  static constexpr int ST = 36, S2 = H + 4;
  int K1, S1;
  int oW1, oW2, oW3, oB2, oXS, oXT, oH1, oH2, oD2, oY, oPART, oDY, TOTAL;
  // wide (more than 31 inputs): no sample-major copy of the minibatch (layer 1 reads its operand from the transposed tile):
  // 7 KB that let 50 inputs -- the 46-wide Adroit hammer observations -- fit the 160 KB
  __host__ __device__ explicit MlpFitLayout(int d_in, bool wide = false) {
    K1 = (d_in + 1 + 3) & ~3; S1 = K1 + 2;
    oW1 = 0; oW2 = oW1 + H * S1; oW3 = oW2 + H * S2; oB2 = oW3 + H;
    oXS = ((oB2 + H + 4 + 3) / 4) * 4;            // [32][S1]  (b3 sits at oB2 + H)
    oXT = wide ? oXS : ((oXS + 32 * S1 + 3) / 4) * 4;          // [K1][ST]
    oH1 = oXT + K1 * ST;                          // [H][ST]
    oH2 = oH1 + H * ST;
    oD2 = oH2 + H * ST;
    oY = oD2 + H * ST;                            // [32]
    oPART = oY + 32;                              // [4][32]
    oDY = oPART + 128;                            // [32]
    TOTAL = oDY + 32;
  }
  __host__ __device__ size_t bytes() const { return (size_t)TOTAL * 4; }
};

// NF1: 32-feature blocks of the input layer's weight gradient -- 1 for d_in <= 31 (the MuJoCo locomotion observations + 4 time
// features), 2 for d_in <= 63 as far as the LDS layout fits 160 KB (d_in <= 55: all Adroit observations, 39..46 wide).
// REGMOM: the Adam moments of the weights a thread owns (83 (m, v) pairs at NF1 = 1) stay in its registers for the WHOLE run --
// loaded once before the first step, written back after the last -- instead of streaming 2 x 156 KB through the CU's L2 path in
// every step (r02 measured the Adam phase as bound by exactly that traffic, not by its arithmetic).  The compute phase needs
// ~250 of the 512 registers of a one-wave-per-SIMD kernel; the pairs take 166 more (the allocator parks them in AGPRs).
//
// MULTI (r05): inputs wider than one workgroup can hold (the 55-input limit above: Ant's 115, Humanoid's 380 = BASELINE configs[3])
// used to fall to ~14 launches per step.  Now G workgroups -- one per CU, launched as ONE grid -- share a step: workgroup g keeps
// the first-layer weights (and their Adam moments) of features [g FS, (g + 1) FS) in its LDS, forms the partial pre-activations
// W1[:, slice] x[slice] of every half, publishes them in an uncached exchange block, and after one grid barrier every workgroup
// sums all G partials IN WORKGROUP ORDER -- the same bits everywhere.  From there on the step is REPLICATED: every workgroup
// runs layers 2 / 3, the loss, delta2, delta1 and the Adam update of W2 / b2 / W3 / b3 on identical data with identical
// instructions (the copies stay bit-identical; no second exchange), and updates its own slice of W1 from delta1 x[slice]^T.
// One barrier per 32-sample half; the exchange block is double-buffered by barrier parity (a workgroup is at most one phase
// ahead of another).  The last workgroup carries the bias column b1.  A wait that exceeds ~2 s poisons the epoch losses with NaN
// and leaves (no hung GPU if a workgroup never gets a CU).
// MULTI's verdict (one wave, launched on the same stream behind the trainer): a workgroup that gave up on a grid barrier left a
// mark in the word behind the arrival counter -> every epoch loss becomes NaN, whatever the workgroups wrote there and in whatever
// order (the host rejects a fit with non-finite losses and keeps the previous parameters, baselines/mlp_baseline.py _settle)
__global__ void k_mlp_fit_verdict(const unsigned* __restrict__ gave_up, double* __restrict__ epoch_loss, int epochs) {
  if (__hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
  for (int e = threadIdx.x; e < (epochs > 0 ? epochs : 1); e += blockDim.x) epoch_loss[e] = (double)__builtin_nanf("");
}

template <int H, int NF1 = 1, bool REGMOM = false, bool MULTI = false>
__global__ __launch_bounds__(256, 1) void k_mlp_fit(MlpFitArgs A) {
  static_assert(H == 128, "wave w owns unit tile w: 4 waves x 32 units");
  using LT = MlpFitLayout<H>;
  constexpr int ST = LT::ST, S2 = LT::S2, NT = H / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int g_id = MULTI ? (int)blockIdx.x : 0;
  const int dG = A.d_in;                                              // width of a feature row / of a W1 row in global memory
  const int f_off = MULTI ? g_id * A.FS : 0;                          // first feature of this workgroup's slice
  const bool has_b1 = !MULTI || g_id == A.G - 1;                      // who carries the bias column of layer 1
  const LT L(MULTI ? A.FS : A.d_in, NF1 > 1);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, j = lane & 31, hi = lane >> 5;
  const int d_in = MULTI ? (dG - f_off < A.FS ? dG - f_off : A.FS) : dG;      // LOCAL width from here on
  const int K1 = L.K1, S1 = L.S1;
  unsigned phase = 0;                                                 // grid barriers passed (MULTI)
  bool timed_out = false;
  if constexpr (MULTI) {
    if (A.fault == 1 && g_id == 0) {                                  // fault injection: the straggler of ADVICE r05 (it then passes every abandoned barrier at once)
      if (tid == 0) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < 260000000ull) __builtin_amdgcn_s_sleep(8); }
      __syncthreads();
    }
  }
  float* sW1 = lds + L.oW1; float* sW2 = lds + L.oW2; float* sW3 = lds + L.oW3; float* sB2 = lds + L.oB2;
  float* sB3 = sB2 + H;
  float* xs = lds + L.oXS; float* xT = lds + L.oXT; float* h1T = lds + L.oH1; float* h2T = lds + L.oH2; float* d2T = lds + L.oD2;
  float* sY = lds + L.oY; float* sPart = lds + L.oPART; float* sDY = lds + L.oDY;
  const int64_t oW1g = 0, oB1g = (int64_t)H * dG, oW2g = oB1g + H, oB2g = oW2g + (int64_t)H * H, oW3g = oB2g + H, oB3g = oW3g + H;

  // ---- load parameters into LDS (b1 rides as the "ones" column of W1)
  for (int i = tid; i < L.TOTAL; i += 256) lds[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < H * (d_in + 1); i += 256) {
    int u = i / (d_in + 1), f = i - u * (d_in + 1);
    sW1[u * S1 + f] = (f < d_in) ? A.params[oW1g + (int64_t)u * dG + f_off + f] : (has_b1 ? A.params[oB1g + u] : 0.f);
  }
  for (int i = tid; i < H * H; i += 256) sW2[(i / H) * S2 + (i % H)] = A.params[oW2g + i];
  for (int i = tid; i < H; i += 256) { sW3[i] = A.params[oW3g + i]; sB2[i] = A.params[oB2g + i]; }
  if (tid == 0) sB3[0] = A.params[oB3g];
  if (tid < 32) { if (NF1 == 1) xs[tid * S1 + d_in] = 1.0f; xT[d_in * ST + tid] = has_b1 ? 1.0f : 0.f; }
  const int64_t Ptot = oB3g + 1;
  float* const mvbase = A.mv + (MULTI ? (int64_t)g_id * 2 * Ptot : 0);          // (MULTI: every workgroup its own copy of the pairs)
  for (int64_t i = tid; i < Ptot; i += 256) { mvbase[2 * i] = A.m[i]; mvbase[2 * i + 1] = A.v[i]; }
  __syncthreads();

  const float b1c = 0.9f, b2c = 0.999f, eps = 1e-8f;
  double pw1 = pow((double)b1c, (double)A.step0), pw2 = pow((double)b2c, (double)A.step0);
  constexpr int GL = (32 * 32 * NF1 + 255) / 256;   // gather elements per thread (d_in <= 32 NF1 - 1)
  float gx[GL];
  float gy = 0.f;

  // Two-stage prefetch: the permutation entries (row indices) are fetched one half-step before the
  // rows they address, the rows one half-step before they are staged, so neither latency is exposed.
  int gidx[GL];
  int gyi = 0;
  // (sample, feature) of this thread's gather elements: fixed for the whole run -- the division by the run-time d_in was
  // re-done for every element in each of the three lambdas below, twice per step
  int gs[GL], gf[GL];
#pragma unroll
  for (int c = 0; c < GL; ++c) {
    const int e = c * 256 + tid;
    gs[c] = (e < 32 * d_in) ? e / d_in : -1;
    gf[c] = e - (e / d_in) * d_in;
  }
  auto index_load = [&](int ep, int64_t mb, int hb) {
    const int32_t* idx = A.perm + (int64_t)ep * A.N + mb * 64 + 32 * hb;
#pragma unroll
    for (int c = 0; c < GL; ++c) gidx[c] = idx[gs[c] >= 0 ? gs[c] : 0];
    gyi = idx[tid & 31];
  };
  auto gather_load = [&]() {
#pragma unroll
    for (int c = 0; c < GL; ++c) gx[c] = A.feat[gs[c] >= 0 ? (int64_t)gidx[c] * dG + f_off + gf[c] : 0];
    gy = A.y[gyi];
  };
  auto gather_store = [&]() {
#pragma unroll
    for (int c = 0; c < GL; ++c)
      if (gs[c] >= 0) { if (NF1 == 1) xs[gs[c] * S1 + gf[c]] = gx[c]; xT[gf[c] * ST + gs[c]] = gx[c]; }
    if (tid < 32) sY[tid] = gy;
  };

  // ---- which weights this thread applies Adam to, and where their moment pairs live.  Every owned element sits at a
  // compile-time offset from one per-lane base.  Without REGMOM ALL moment pairs a thread owns -- its 64 W2
  // weights, its 16 W1 / b1 entries, its b2 / W3 / b3 entry -- are requested at the top of every Adam phase (the backward pass's
  // registers are free by then), so the L2 latency is paid once per step and not once per block; then update, write the weight to
  // LDS and the pair back.  Threads that do not own an entry of a block read a valid dummy pair and store nothing.
  f32x2* __restrict__ MV = (f32x2*)mvbase;
  const int64_t gbase2 = oW2g + (int64_t)(32 * w + 4 * hi) * H + j;
  f32x2* __restrict__ mvW2 = MV + gbase2;
  float* pW2 = sW2 + (32 * w + 4 * hi) * S2 + j;
  // W1 rows of this wave (feature f = 32 fb + j < d_in) and b1 (f == d_in): one base pointer + a small per-register stride
  bool own1[NF1];
  int stg[NF1];
  f32x2* mvW1[NF1];
  float* pW1[NF1];
#pragma unroll
  for (int fb = 0; fb < NF1; ++fb) {
    const int f = 32 * fb + j;
    const bool isw = f < d_in;
    own1[fb] = isw || (f == d_in && has_b1);
    stg[fb] = isw ? dG : 1;
    const int64_t gbase1 = isw ? oW1g + (int64_t)(32 * w + 4 * hi) * dG + f_off + f : oB1g + 32 * w + 4 * hi;
    mvW1[fb] = MV + (own1[fb] ? gbase1 : 0);
    pW1[fb] = sW1 + (32 * w + 4 * hi) * S1 + (isw ? f : d_in);
  }
  const bool ownb2 = hi == 0, ownw3 = tid < H, ownb3 = tid == 0;
  const int64_t gb2i = oB2g + 32 * w + j, gw3i = oW3g + (ownw3 ? tid : 0), gb3i = oB3g;
  f32x2 q2[NT][16], q1[NF1][16], qb2, qw3, qb3;
  if constexpr (REGMOM) {
    // the pairs this thread owns, for the whole run (MV was filled above, barrier passed), re-paired for packed math:
    // entry r (even) = the first moments of weights r, r + 1, entry r + 1 = their second moments
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 a = mvW2[unit_of(r, 0) * H + 32 * nt], b = mvW2[unit_of(r + 1, 0) * H + 32 * nt];
        q2[nt][r] = f32x2{a.x, b.x}; q2[nt][r + 1] = f32x2{a.y, b.y};
      }
#pragma unroll
    for (int fb = 0; fb < NF1; ++fb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 a = mvW1[fb][own1[fb] ? unit_of(r, 0) * stg[fb] : 0], b = mvW1[fb][own1[fb] ? unit_of(r + 1, 0) * stg[fb] : 0];
        q1[fb][r] = f32x2{a.x, b.x}; q1[fb][r + 1] = f32x2{a.y, b.y};
      }
    qb2 = MV[gb2i]; qw3 = MV[gw3i]; qb3 = MV[gb3i];
  }

  for (int ep = 0; ep < A.epochs; ++ep) {
    double ep_loss = 0.0;
    if (A.steps > 0) { index_load(ep, 0, 0); gather_load(); index_load(ep, 0, 1); }
#pragma unroll 1
    for (int64_t mb = 0; mb < A.steps; ++mb) {
      f32x16 gW2[NT], gW1[NF1];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) gW2[nt] = (f32x16)(0.f);
#pragma unroll
      for (int fb = 0; fb < NF1; ++fb) gW1[fb] = (f32x16)(0.f);
      float gb2 = 0.f, gw3 = 0.f, gb3 = 0.f;
      // this lane's 16 output-layer weights (units 32 w + unit_of(r, hi)), fetched once per step: read from LDS where they are
      // used -- next to the h2^T / delta2^T stores, one element at a time -- every element paid an LDS round trip of its own
      // (store, load, wait: 2 x 16 serialised trips per half)
      float w3v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) w3v[r] = sW3[32 * w + unit_of(r, hi)];
#pragma unroll 1
      for (int hb = 0; hb < 2; ++hb) {
        MJX_FIT_STAMP(0);
        // (r04, REGMOM builds) this lane's byte address per tile and use, re-laundered every half: every LDS access of the MFMA phases
        // below is then `base + immediate` -- no hoisted address parked in an AGPR, no v_add_u32 re-basing ds_read2 pairs between
        // MFMAs (k_mlp_fit1p tells the story; the r03 kernels, REGMOM = false, keep the compiler's addressing)
        const uint32_t h1w = lds_pin(&h1T[(32 * w + 4 * hi) * ST + j]), d2w = lds_pin(&d2T[(32 * w + 4 * hi) * ST + j]);
        const uint32_t h2w = lds_pin(&h2T[(32 * w + 4 * hi) * ST + j]);
        const uint32_t h1c = lds_pin(&h1T[(4 * hi) * ST + j]), d2c = lds_pin(&d2T[(4 * hi) * ST + j]);
        const uint32_t h1r = lds_pin(&h1T[(32 * w + j) * ST + 4 * hi]), d2r = lds_pin(&d2T[(32 * w + j) * ST + 4 * hi]);
        const uint32_t h1n = lds_pin(&h1T[j * ST + 4 * hi]);
        const uint32_t w2r = lds_pin(&sW2[(32 * w + j) * S2 + 4 * hi]), w2c = lds_pin(&sW2[(4 * hi) * S2 + 32 * w + j]);
        typedef __attribute__((address_space(3))) f32x4 lds_f4;
#define FIT2_LD1(base, off) (*(const volatile lds_float*)(uintptr_t)((base) + (uint32_t)((off) * 4)))
#define FIT2_LD4(base, off) (*(const lds_f4*)(uintptr_t)((base) + (uint32_t)((off) * 4)))
        gather_store();
        __syncthreads();
        // prefetch the next half's rows while this half computes
        if (hb == 0) { gather_load(); if (mb + 1 < A.steps) index_load(ep, mb + 1, 0); }       // rows of half 1; indices of the next step
        else if (mb + 1 < A.steps) { gather_load(); index_load(ep, mb + 1, 1); }
        MJX_FIT_STAMP(1);
        // ---- layer 1: z1[unit 32w+., sample] = W1a x~a ; h1 = relu
        f32x16 z1 = (f32x16)(0.f);
        for (int q = 0; q < K1 / 4; ++q) {
          const int f0 = 4 * q + 2 * hi;
          f32x2 a = *(const f32x2*)&sW1[(32 * w + j) * S1 + f0];
          f32x2 b;
          if (NF1 == 1) b = *(const f32x2*)&xs[j * S1 + f0];
          else b = f32x2{xT[f0 * ST + j], xT[(f0 + 1) * ST + j]};
          z1 = MJX_MFMA(a.x, b.x, z1);
          z1 = MJX_MFMA(a.y, b.y, z1);
        }
        if constexpr (MULTI) {
          // this workgroup's partial W1[:, slice] x[slice] -> its slot of the exchange block; one grid barrier; the sum of all
          // G partials in workgroup order (identical bits on every workgroup: the replicated rest of the step depends on it)
          // slot layout [wave][quad q][lane][4]: register r = 4 q + t of lane `lane` of wave w -- the same matrix element on every
          // workgroup -- so a lane moves its 16 values as four 16-byte accesses (coalesced 1 KB per wave and quad) instead of
          // sixteen 4-byte ones: the first version read G x 16 dwords per lane one after the other, 7 us per workgroup and step
          const size_t slot = (size_t)H * 32;
          f32x4* xo = (f32x4*)(A.xch + ((size_t)(phase & 1u) * (size_t)A.G + (size_t)g_id) * slot) + (size_t)(w * 4) * 64 + lane;
#pragma unroll
          for (int q = 0; q < 4; ++q) xo[q * 64] = f32x4{z1[4 * q], z1[4 * q + 1], z1[4 * q + 2], z1[4 * q + 3]};
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // uncached stores: acknowledged by memory, nothing holds them back
          __syncthreads();
          if (tid == 0) {
            __hip_atomic_fetch_add(A.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)A.G * (phase + 1u);
            const unsigned long long t0 = wall_clock64();             // 100 MHz
            while ((int)(__hip_atomic_load(A.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
              __builtin_amdgcn_s_sleep(1);
              if (timed_out || wall_clock64() - t0 > 200000000ull) {                                  // ~2 s: give up (once), poison the losses
                // ... through a word of its own next to the counter (uncached, OR-ed: no other writer can undo it).  The epoch
                // losses alone are not a safe place: workgroup 0 -- if IT was the straggler -- passes every abandoned barrier at
                // once later and overwrites the NaNs with finite sums of stale partials.  k_mlp_fit_verdict, launched
                // behind this kernel, turns the word into NaN losses once every workgroup has left.
                if (!timed_out) __hip_atomic_fetch_or(A.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                timed_out = true; break;
              }
            }
          }
          __syncthreads();
          const f32x4* xi = (const f32x4*)(A.xch + (size_t)(phase & 1u) * (size_t)A.G * slot) + (size_t)(w * 4) * 64 + lane;
          asm volatile("" : "+v"(xi) :: "memory");                    // (a fresh look at memory every phase: nothing cached in registers)
          f32x16 zs = (f32x16)(0.f);
          for (int g0 = 0; g0 < A.G; g0 += 4) {                       // four workgroups' partials in flight, summed in workgroup order
            f32x4 t[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int gg = g0 + u < A.G ? g0 + u : g0;              // (surplus entries re-read a valid slot and are not added)
#pragma unroll
              for (int q = 0; q < 4; ++q) t[u][q] = xi[(size_t)gg * (slot / 4) + q * 64];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (g0 + u < A.G) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { zs[4 * q] += t[u][q].x; zs[4 * q + 1] += t[u][q].y; zs[4 * q + 2] += t[u][q].z; zs[4 * q + 3] += t[u][q].w; }
              }
          }
          z1 = zs;
          ++phase;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if constexpr (REGMOM) LDS_AT(h1w)[unit_of(r, 0) * ST] = fmaxf(z1[r], 0.f);
          else h1T[(32 * w + unit_of(r, hi)) * ST + j] = fmaxf(z1[r], 0.f);
        }
        __syncthreads();
        MJX_FIT_STAMP(2);
        // ---- layer 2: K = all 128 h1 units (B operand from the shared h1^T tile)
        f32x16 z2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 b = *(const f32x4*)&sB2[32 * w + 8 * q + 4 * hi];
          z2[4 * q] = b.x; z2[4 * q + 1] = b.y; z2[4 * q + 2] = b.z; z2[4 * q + 3] = b.w;
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          f32x4 ac, an;
          float bc[4], bn[4];
          if constexpr (REGMOM) ac = FIT2_LD4(w2r, 0); else ac = *(const f32x4*)&sW2[(32 * w + j) * S2 + 4 * hi];
#pragma unroll
          for (int t = 0; t < 4; ++t) { if constexpr (REGMOM) bc[t] = FIT2_LD1(h1c, t * ST); else bc[t] = h1T[(4 * hi + t) * ST + j]; }
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            if (g + 1 < 16) {
              const int k1 = 32 * ((g + 1) >> 2) + 8 * ((g + 1) & 3) + 4 * hi;
              const int k1c = 32 * ((g + 1) >> 2) + 8 * ((g + 1) & 3);
              if constexpr (REGMOM) an = FIT2_LD4(w2r, k1c); else an = *(const f32x4*)&sW2[(32 * w + j) * S2 + k1];
#pragma unroll
              for (int t = 0; t < 4; ++t) { if constexpr (REGMOM) bn[t] = FIT2_LD1(h1c, (k1c + t) * ST); else bn[t] = h1T[(k1 + t) * ST + j]; }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) z2 = MJX_MFMA(ac[t], bc[t], z2);
            ac = an;
#pragma unroll
            for (int t = 0; t < 4; ++t) bc[t] = bn[t];
          }
          // in-order single wave: keep group g+1's five LDS fetches ahead of group g's four MFMAs
          __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            if (g + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float hv = fmaxf(z2[r], 0.f);
          z2[r] = hv;
          if constexpr (REGMOM) LDS_AT(h2w)[unit_of(r, 0) * ST] = hv; else h2T[(32 * w + unit_of(r, hi)) * ST + j] = hv;
          part = fmaf(w3v[r], hv, part);
        }
        part = half_sum(part);                            // (+ the other lane half: v_permlane32_swap, the bits of part + shfl_xor(part, 32))
        if (hi == 0) sPart[w * 32 + j] = part;
        __syncthreads();
        MJX_FIT_STAMP(3);
        // ---- output + MSE gradient (every wave redundantly, lane j = sample)
        const float yhat = (sPart[j] + sPart[32 + j]) + (sPart[64 + j] + sPart[96 + j]) + sB3[0];
        const float err = yhat - sY[j];
        const float dy = 2.0f * err / 64.0f;              // MSELoss(mean) over the 64-row minibatch
        if (w == 0 && hi == 0) {
          sDY[j] = dy;
          const float e2 = sum32_lane0(err * err);
          if (j == 0) ep_loss += (double)e2 / 64.0;
        }
        // delta2 (lane = sample) -> d2T [unit][sample]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int u = 32 * w + unit_of(r, hi);
          if constexpr (REGMOM) LDS_AT(d2w)[unit_of(r, 0) * ST] = (z2[r] > 0.f) ? w3v[r] * dy : 0.f;
          else d2T[u * ST + j] = (z2[r] > 0.f) ? w3v[r] * dy : 0.f;
        }
        __syncthreads();
        MJX_FIT_STAMP(4);
        // ---- grad W3 / b3 (thread = unit), grad b2, delta2 in lane = unit layout straight from d2T
        if (tid < H) {
          float a = 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            f32x4 hv = *(const f32x4*)&h2T[tid * ST + 4 * q];
            f32x4 dv = *(const f32x4*)&sDY[4 * q];
            a += hv.x * dv.x + hv.y * dv.y + hv.z * dv.z + hv.w * dv.w;
          }
          gw3 += a;
        }
        if (tid == 0) { float a = 0.f; for (int s = 0; s < 32; ++s) a += sDY[s]; gb3 += a; }
        f32x16 d2u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 t4;
          if constexpr (REGMOM) t4 = FIT2_LD4(d2r, 8 * q); else t4 = *(const f32x4*)&d2T[(32 * w + j) * ST + 8 * q + 4 * hi];
          d2u[4 * q] = t4.x; d2u[4 * q + 1] = t4.y; d2u[4 * q + 2] = t4.z; d2u[4 * q + 3] = t4.w;
        }
        {
          float s2 = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) s2 += d2u[r];
          gb2 += half_sum(s2);
        }
        MJX_FIT_STAMP(5);
        // grad W2 rows of this wave: A = delta2u (registers), B = h1^T tiles
        {
          f32x4 bc[NT], bn[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) { if constexpr (REGMOM) bc[nt] = FIT2_LD4(h1n, 32 * nt * ST); else bc[nt] = *(const f32x4*)&h1T[(32 * nt + j) * ST + 4 * hi]; }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (q + 1 < 4) {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) { if constexpr (REGMOM) bn[nt] = FIT2_LD4(h1n, 32 * nt * ST + 8 * (q + 1)); else bn[nt] = *(const f32x4*)&h1T[(32 * nt + j) * ST + 8 * (q + 1) + 4 * hi]; }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) gW2[nt] = MJX_MFMA(d2u[4 * q + t], bc[nt][t], gW2[nt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bc[nt] = bn[nt];
          }
        }
        MJX_FIT_STAMP(6);
        // delta1 (lane = h1 unit of this wave's tile): A = delta2 [sample][k] from d2T, B = W2[k][unit]
        f32x16 d1u = (f32x16)(0.f);
        __builtin_amdgcn_sched_barrier(0);
        {
          float ac[4], an[4], bc[4], bn[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if constexpr (REGMOM) { ac[t] = FIT2_LD1(d2c, t * ST); bc[t] = FIT2_LD1(w2c, t * S2); }
            else { ac[t] = d2T[(4 * hi + t) * ST + j]; bc[t] = sW2[(4 * hi + t) * S2 + 32 * w + j]; }
          }
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            if (g + 1 < 16) {
              const int k1 = 32 * ((g + 1) >> 2) + 8 * ((g + 1) & 3) + 4 * hi;
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const int k1c = 32 * ((g + 1) >> 2) + 8 * ((g + 1) & 3);
                if constexpr (REGMOM) { an[t] = FIT2_LD1(d2c, (k1c + t) * ST); bn[t] = FIT2_LD1(w2c, (k1c + t) * S2); }
                else { an[t] = d2T[(k1 + t) * ST + j]; bn[t] = sW2[(k1 + t) * S2 + 32 * w + j]; }
              }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) d1u = MJX_MFMA(ac[t], bc[t], d1u);
#pragma unroll
            for (int t = 0; t < 4; ++t) { ac[t] = an[t]; bc[t] = bn[t]; }
          }
          __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            if (g + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {                       // relu'(z1) mask from h1^T (same layout)
          f32x4 hv;
          if constexpr (REGMOM) hv = FIT2_LD4(h1r, 8 * q); else hv = *(const f32x4*)&h1T[(32 * w + j) * ST + 8 * q + 4 * hi];
          d1u[4 * q] = hv.x > 0.f ? d1u[4 * q] : 0.f; d1u[4 * q + 1] = hv.y > 0.f ? d1u[4 * q + 1] : 0.f;
          d1u[4 * q + 2] = hv.z > 0.f ? d1u[4 * q + 2] : 0.f; d1u[4 * q + 3] = hv.w > 0.f ? d1u[4 * q + 3] : 0.f;
        }
        MJX_FIT_STAMP(7);
        // grad W1a rows of this wave (column d_in = grad b1)
#pragma unroll
        for (int fb = 0; fb < NF1; ++fb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int f = 32 * fb + j;
            f32x4 b4 = *(const f32x4*)&xT[(f < K1 ? f : 0) * ST + 8 * q + 4 * hi];
            if (f >= K1) b4 = (f32x4)(0.f);
#pragma unroll
            for (int t = 0; t < 4; ++t) gW1[fb] = MJX_MFMA(d1u[4 * q + t], b4[t], gW1[fb]);
          }
        __syncthreads();                                    // tiles are rewritten by the next half
        MJX_FIT_STAMP(8);
      }
      { const int hb = 0; MJX_FIT_STAMP(9); }
      // ---- Adam (torch.optim.Adam: L2 weight decay folded into the gradient, bias-corrected)
      pw1 *= (double)b1c; pw2 *= (double)b2c;
      const float bc1 = (float)(1.0 - pw1), bc2s = (float)sqrt(1.0 - pw2);
      const float step_size = A.lr / bc1;
      // torch.optim.Adam's update; the two divides use v_rcp_f32 + one Newton step (<= 1 ulp from IEEE),
      // which keeps the 19 457-weight update at ~18 VALU ops per weight
      const float inv_bc2s = 1.0f / bc2s;
      auto adam_math = [&](float p, float g, f32x2& q) {
        g += A.wd * p;
        float mi = q.x, vi = q.y;
        mi = mi + (g - mi) * (1.0f - b1c);
        vi = vi * b2c + g * g * (1.0f - b2c);
        q = f32x2{mi, vi};
        const float denom = fmaf(__builtin_amdgcn_sqrtf(vi), inv_bc2s, eps);
        float r = __builtin_amdgcn_rcpf(denom);
        r = r * fmaf(-denom, r, 2.0f);
        return fmaf(-step_size * mi, r, p);
      };
      // the same update on TWO weights per instruction (packed fp32): the arithmetic of adam_math element by element -- the
      // vector ALU is paid in full on this chip (fp32 MFMAs hide none of it), the 19 457-weight update is ~12 instructions per weight
      auto adam_math2 = [&](f32x2 p, f32x2 g, f32x2& qa, f32x2& qb) {
        g = __builtin_elementwise_fma((f32x2)(A.wd), p, g);
        f32x2 mi = {qa.x, qb.x}, vi = {qa.y, qb.y};
        mi = __builtin_elementwise_fma(g - mi, (f32x2)(1.0f - b1c), mi);
        vi = __builtin_elementwise_fma(g * g, (f32x2)(1.0f - b2c), vi * (f32x2)(b2c));
        qa = f32x2{mi.x, vi.x}; qb = f32x2{mi.y, vi.y};
        const f32x2 denom = __builtin_elementwise_fma(f32x2{__builtin_amdgcn_sqrtf(vi.x), __builtin_amdgcn_sqrtf(vi.y)}, (f32x2)(inv_bc2s), (f32x2)(eps));
        f32x2 r = {__builtin_amdgcn_rcpf(denom.x), __builtin_amdgcn_rcpf(denom.y)};
        r = r * __builtin_elementwise_fma(-denom, r, (f32x2)(2.0f));
        return __builtin_elementwise_fma(mi * (f32x2)(-step_size), r, p);
      };
      // ... and on moments that already sit as (m, m) / (v, v) pairs (REGMOM): no re-pairing moves
      auto adam_math2p = [&](f32x2 p, f32x2 g, f32x2& mi, f32x2& vi) {
        g = __builtin_elementwise_fma((f32x2)(A.wd), p, g);
        mi = __builtin_elementwise_fma(g - mi, (f32x2)(1.0f - b1c), mi);
        vi = __builtin_elementwise_fma(g * g, (f32x2)(1.0f - b2c), vi * (f32x2)(b2c));
        const f32x2 denom = __builtin_elementwise_fma(f32x2{__builtin_amdgcn_sqrtf(vi.x), __builtin_amdgcn_sqrtf(vi.y)}, (f32x2)(inv_bc2s), (f32x2)(eps));
        f32x2 r = {__builtin_amdgcn_rcpf(denom.x), __builtin_amdgcn_rcpf(denom.y)};
        r = r * __builtin_elementwise_fma(-denom, r, (f32x2)(2.0f));
        return __builtin_elementwise_fma(mi * (f32x2)(-step_size), r, p);
      };
      {
        if constexpr (!REGMOM) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) q2[nt][r] = mvW2[unit_of(r, 0) * H + 32 * nt];
#pragma unroll
          for (int fb = 0; fb < NF1; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) q1[fb][r] = mvW1[fb][own1[fb] ? unit_of(r, 0) * stg[fb] : 0];
          qb2 = MV[gb2i]; qw3 = MV[gw3i]; qb3 = MV[gb3i];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int o0 = unit_of(r, 0) * H + 32 * nt, l0 = unit_of(r, 0) * S2 + 32 * nt;
            const int o1 = unit_of(r + 1, 0) * H + 32 * nt, l1 = unit_of(r + 1, 0) * S2 + 32 * nt;
            if constexpr (REGMOM) {
              const f32x2 pn = adam_math2p(f32x2{pW2[l0], pW2[l1]}, f32x2{gW2[nt][r], gW2[nt][r + 1]}, q2[nt][r], q2[nt][r + 1]);
              pW2[l0] = pn.x; pW2[l1] = pn.y;
            } else if constexpr (NF1 == 1) {
              const f32x2 pn = adam_math2(f32x2{pW2[l0], pW2[l1]}, f32x2{gW2[nt][r], gW2[nt][r + 1]}, q2[nt][r], q2[nt][r + 1]);
              pW2[l0] = pn.x; pW2[l1] = pn.y;
            } else {                                      // (the two-block variant already spills: the packed form's temporaries cost it 2 %)
              pW2[l0] = adam_math(pW2[l0], gW2[nt][r], q2[nt][r]);
              pW2[l1] = adam_math(pW2[l1], gW2[nt][r + 1], q2[nt][r + 1]);
            }
            if constexpr (!REGMOM) { mvW2[o0] = q2[nt][r]; mvW2[o1] = q2[nt][r + 1]; }
          }
#pragma unroll
        for (int fb = 0; fb < NF1; ++fb)
          if (own1[fb]) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const int o0 = unit_of(r, 0) * stg[fb], l0 = unit_of(r, 0) * S1, o1 = unit_of(r + 1, 0) * stg[fb], l1 = unit_of(r + 1, 0) * S1;
              if constexpr (REGMOM) {
                const f32x2 pn = adam_math2p(f32x2{pW1[fb][l0], pW1[fb][l1]}, f32x2{gW1[fb][r], gW1[fb][r + 1]}, q1[fb][r], q1[fb][r + 1]);
                pW1[fb][l0] = pn.x; pW1[fb][l1] = pn.y;
              } else if constexpr (NF1 == 1) {
                const f32x2 pn = adam_math2(f32x2{pW1[fb][l0], pW1[fb][l1]}, f32x2{gW1[fb][r], gW1[fb][r + 1]}, q1[fb][r], q1[fb][r + 1]);
                pW1[fb][l0] = pn.x; pW1[fb][l1] = pn.y;
              } else {
                pW1[fb][l0] = adam_math(pW1[fb][l0], gW1[fb][r], q1[fb][r]);
                pW1[fb][l1] = adam_math(pW1[fb][l1], gW1[fb][r + 1], q1[fb][r + 1]);
              }
              if constexpr (!REGMOM) { mvW1[fb][o0] = q1[fb][r]; mvW1[fb][o1] = q1[fb][r + 1]; }
            }
          }
        if (ownb2) { sB2[32 * w + j] = adam_math(sB2[32 * w + j], gb2, qb2); if constexpr (!REGMOM) MV[gb2i] = qb2; }
        if (ownw3) { sW3[tid] = adam_math(sW3[tid], gw3, qw3); if constexpr (!REGMOM) MV[gw3i] = qw3; }
        if (ownb3) { sB3[0] = adam_math(sB3[0], gb3, qb3); if constexpr (!REGMOM) MV[gb3i] = qb3; }
      }
      __syncthreads();
      { const int hb = 0; MJX_FIT_STAMP(10); }
    }
    if (tid == 0 && g_id == 0) A.epoch_loss[ep] = ep_loss;
    if (MULTI && tid == 0 && timed_out) A.epoch_loss[ep] = (double)__builtin_nanf("");      // (any workgroup that gave up says so)
  }
  // ---- write the trained parameters and the moments back
  if constexpr (REGMOM) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        mvW2[unit_of(r, 0) * H + 32 * nt] = f32x2{q2[nt][r].x, q2[nt][r + 1].x};
        mvW2[unit_of(r + 1, 0) * H + 32 * nt] = f32x2{q2[nt][r].y, q2[nt][r + 1].y};
      }
#pragma unroll
    for (int fb = 0; fb < NF1; ++fb)
      if (own1[fb]) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          mvW1[fb][unit_of(r, 0) * stg[fb]] = f32x2{q1[fb][r].x, q1[fb][r + 1].x};
          mvW1[fb][unit_of(r + 1, 0) * stg[fb]] = f32x2{q1[fb][r].y, q1[fb][r + 1].y};
        }
      }
    if (ownb2) MV[gb2i] = qb2;
    if (ownw3) MV[gw3i] = qw3;
    if (ownb3) MV[gb3i] = qb3;
  }
  __syncthreads();
  // (MULTI: a workgroup writes what it alone holds -- its W1 slice, b1 on the last one -- and workgroup 0 the replicated rest)
  for (int64_t i = tid; i < Ptot; i += 256) {
    bool mine = true;
    if constexpr (MULTI) {
      if (i < oB1g) { const int f = (int)(i % dG); mine = f >= f_off && f < f_off + d_in; }
      else if (i < oW2g) mine = has_b1;
      else mine = g_id == 0;
    }
    if (mine) { A.m[i] = mvbase[2 * i]; A.v[i] = mvbase[2 * i + 1]; }
  }
  for (int i = tid; i < H * (d_in + 1); i += 256) {
    int u = i / (d_in + 1), f = i - u * (d_in + 1);
    if (f < d_in) A.params[oW1g + (int64_t)u * dG + f_off + f] = sW1[u * S1 + f];
    else if (has_b1) A.params[oB1g + u] = sW1[u * S1 + d_in];
  }
  if (!MULTI || g_id == 0) {
    for (int i = tid; i < H * H; i += 256) A.params[oW2g + i] = sW2[(i / H) * S2 + (i % H)];
    for (int i = tid; i < H; i += 256) { A.params[oW3g + i] = sW3[i]; A.params[oB2g + i] = sB2[i]; }
    if (tid == 0) A.params[oB3g] = sB3[0];
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// One pass per step (r04): the 64 rows of a minibatch as TWO accumulator chains of the same instruction stream instead of two
// sequential 32-sample halves.  Every weight fragment is fetched from LDS once per step and feeds both chains (the per-half
// kernel fetched it twice), a step crosses 4 workgroup barriers instead of 10, and each LDS hand-over's latency is followed by
// twice the matrix work.  The [unit][sample] tiles are 64 samples wide (stride 68); to fit 160 KB the h2^T tile is gone -- the
// output layer's weight gradient is formed from the accumulator registers (lane = sample) with DPP row sums -- and so is the
// sample-major copy of the minibatch (layer 1 reads x^T).  That fits up to 23 inputs (K1 <= 24: every MuJoCo locomotion
// observation up to 19 wide + 4 time features; HalfCheetah's 21); wider inputs run k_mlp_fit.  Adam moments are register-resident
// (REGMOM above).  Same minibatches, same update rule; the gradient sums run in a different order than the per-half kernel's
// (both halves interleaved per k-group), so parameters agree with it to round-off, not bit for bit.
template <int H>
struct MlpFit1pLayout {
  static constexpr int ST = 68, S2 = H + 4;
  int K1, S1;
  int oW1, oW2, oW3, oB2, oXT, oH1, oD2, oY, oPART, oGW3, TOTAL;
  __host__ __device__ explicit MlpFit1pLayout(int d_in) {
    K1 = (d_in + 1 + 3) & ~3; S1 = K1 + 2;
    oW1 = 0; oW2 = oW1 + H * S1; oW3 = oW2 + H * S2; oB2 = oW3 + H;
    oXT = ((oB2 + H + 4 + 3) / 4) * 4;            // [K1][ST]   (b3 sits at oB2 + H)
    oH1 = oXT + K1 * ST;                          // [H][ST]
    oD2 = oH1 + H * ST;
    oY = oD2 + H * ST;                            // [64]
    oPART = oY + 64;                              // [4][64]
    oGW3 = oPART + 256;                           // [H]
    TOTAL = oGW3 + H;
  }
  __host__ __device__ size_t bytes() const { return (size_t)TOTAL * 4; }
};

template <int H>
__global__ __launch_bounds__(256, 1) void k_mlp_fit1p(MlpFitArgs A) {
  static_assert(H == 128, "wave w owns unit tile w: 4 waves x 32 units");
  using LT = MlpFit1pLayout<H>;
  constexpr int ST = LT::ST, S2 = LT::S2, NT = H / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const LT L(A.d_in);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, j = lane & 31, hi = lane >> 5;
  const int d_in = A.d_in, K1 = L.K1, S1 = L.S1;
  float* sW1 = lds + L.oW1; float* sW2 = lds + L.oW2; float* sW3 = lds + L.oW3; float* sB2 = lds + L.oB2;
  float* sB3 = sB2 + H;
  float* xT = lds + L.oXT; float* h1T = lds + L.oH1; float* d2T = lds + L.oD2;
  float* sY = lds + L.oY; float* sPart = lds + L.oPART; float* sGW3 = lds + L.oGW3;
  const int64_t oW1g = 0, oB1g = (int64_t)H * d_in, oW2g = oB1g + H, oB2g = oW2g + (int64_t)H * H, oW3g = oB2g + H, oB3g = oW3g + H;

  // ---- parameters into LDS (b1 rides as the "ones" column of W1), moments into the interleaved workspace
  for (int i = tid; i < L.TOTAL; i += 256) lds[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < H * (d_in + 1); i += 256) {
    int u = i / (d_in + 1), f = i - u * (d_in + 1);
    sW1[u * S1 + f] = (f < d_in) ? A.params[oW1g + (int64_t)u * d_in + f] : A.params[oB1g + u];
  }
  for (int i = tid; i < H * H; i += 256) sW2[(i / H) * S2 + (i % H)] = A.params[oW2g + i];
  for (int i = tid; i < H; i += 256) { sW3[i] = A.params[oW3g + i]; sB2[i] = A.params[oB2g + i]; }
  if (tid == 0) sB3[0] = A.params[oB3g];
  if (tid < 64) xT[d_in * ST + tid] = 1.0f;
  const int64_t Ptot = oB3g + 1;
  for (int64_t i = tid; i < Ptot; i += 256) { A.mv[2 * i] = A.m[i]; A.mv[2 * i + 1] = A.v[i]; }
  __syncthreads();

  const float b1c = 0.9f, b2c = 0.999f, eps = 1e-8f;
  double pw1 = pow((double)b1c, (double)A.step0), pw2 = pow((double)b2c, (double)A.step0);
  constexpr int GL = (64 * 24 + 255) / 256;          // gather elements per thread (d_in <= 23)
  float gx[GL];
  float gy = 0.f;
  int gidx[GL], gyi = 0, gs[GL], gf[GL];
#pragma unroll
  for (int c = 0; c < GL; ++c) {
    const int e = c * 256 + tid;
    gs[c] = (e < 64 * d_in) ? e / d_in : -1;
    gf[c] = e - (e / d_in) * d_in;
  }
  // rows are requested one step before they are staged, the permutation entries that address them one step before that
  auto index_load = [&](int ep, int64_t mb) {
    const int32_t* idx = A.perm + (int64_t)ep * A.N + mb * 64;
#pragma unroll
    for (int c = 0; c < GL; ++c) gidx[c] = idx[gs[c] >= 0 ? gs[c] : 0];
    gyi = idx[tid & 63];
  };
  auto gather_load = [&]() {
#pragma unroll
    for (int c = 0; c < GL; ++c) gx[c] = A.feat[gs[c] >= 0 ? (int64_t)gidx[c] * d_in + gf[c] : 0];
    gy = A.y[gyi];
  };
  auto gather_store = [&]() {
#pragma unroll
    for (int c = 0; c < GL; ++c)
      if (gs[c] >= 0) xT[gf[c] * ST + gs[c]] = gx[c];
    if (tid < 64) sY[tid] = gy;
  };

  // ---- Adam ownership (as k_mlp_fit): the 64 W2 weights, 16 W1 / b1 entries and the b2 / W3 / b3 entry of this thread, their
  // moments in registers for the whole run as (m, m) / (v, v) pairs of neighbouring weights
  f32x2* __restrict__ MV = (f32x2*)A.mv;
  f32x2* __restrict__ mvW2 = MV + oW2g + (int64_t)(32 * w + 4 * hi) * H + j;
  float* pW2 = sW2 + (32 * w + 4 * hi) * S2 + j;
  const bool isw1 = j < d_in, own1 = j <= d_in;
  const int stg = isw1 ? d_in : 1;
  f32x2* mvW1 = MV + (own1 ? (isw1 ? oW1g + (int64_t)(32 * w + 4 * hi) * d_in + j : oB1g + 32 * w + 4 * hi) : 0);
  float* pW1 = sW1 + (32 * w + 4 * hi) * S1 + (isw1 ? j : d_in);
  const bool ownb2 = hi == 0, ownw3 = tid < H, ownb3 = tid == 0;
  const int64_t gb2i = oB2g + 32 * w + j, gw3i = oW3g + (ownw3 ? tid : 0), gb3i = oB3g;
  f32x2 q2[NT][16], q1[16], qb2, qw3, qb3;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 a = mvW2[unit_of(r, 0) * H + 32 * nt], b = mvW2[unit_of(r + 1, 0) * H + 32 * nt];
      q2[nt][r] = f32x2{a.x, b.x}; q2[nt][r + 1] = f32x2{a.y, b.y};
    }
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const f32x2 a = mvW1[own1 ? unit_of(r, 0) * stg : 0], b = mvW1[own1 ? unit_of(r + 1, 0) * stg : 0];
    q1[r] = f32x2{a.x, b.x}; q1[r + 1] = f32x2{a.y, b.y};
  }
  qb2 = MV[gb2i]; qw3 = MV[gw3i]; qb3 = MV[gb3i];

  for (int ep = 0; ep < A.epochs; ++ep) {
    double ep_loss = 0.0;
    if (A.steps > 0) { index_load(ep, 0); gather_load(); if (A.steps > 1) index_load(ep, 1); }
#pragma unroll 1
    for (int64_t mb = 0; mb < A.steps; ++mb) {
      f32x16 gW2[NT], gW1 = (f32x16)(0.f);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) gW2[nt] = (f32x16)(0.f);
      float w3v[16];                                     // this lane's 16 output-layer weights (units 32 w + unit_of(r, hi))
#pragma unroll
      for (int r = 0; r < 16; ++r) w3v[r] = sW3[32 * w + unit_of(r, hi)];
      const int hb = 0;
      // this lane's byte address in row group (32 w + 4 hi) of the two [unit][sample] tiles, re-laundered every step: every access
      // below is then `base + immediate`.  (Left to the compiler, the 16 + 32 row addresses of the h1^T / delta2^T stores were hoisted
      // out of the step loop as invariants, parked in AGPRs by the register allocator and fetched back with v_accvgpr_read + s_nop
      // in front of every store: ~75 isolated vector-ALU instructions per step that fp32 MFMAs do not hide.)
      uint32_t h1w = lds_pin(&h1T[(32 * w + 4 * hi) * ST + j]), d2w = lds_pin(&d2T[(32 * w + 4 * hi) * ST + j]);
      // ... and of every fragment read of the MFMA phases (the same story: 41 + 114 v_add_u32 between the MFMAs of layer 2 and of the
      // backward pass, re-deriving `lane offset + buffer base + constant`)
      const uint32_t h1c = lds_pin(&h1T[(4 * hi) * ST + j]), d2c = lds_pin(&d2T[(4 * hi) * ST + j]);                 // rows k (+ const), sample 32 h + j
      const uint32_t h1r = lds_pin(&h1T[(32 * w + j) * ST + 4 * hi]), d2r = lds_pin(&d2T[(32 * w + j) * ST + 4 * hi]); // row = this lane's unit, 4 samples per read
      const uint32_t h1n = lds_pin(&h1T[j * ST + 4 * hi]);                                                           // rows 32 nt + j
      const uint32_t w2r = lds_pin(&sW2[(32 * w + j) * S2 + 4 * hi]), w2c = lds_pin(&sW2[(4 * hi) * S2 + 32 * w + j]);
      const uint32_t x1c = lds_pin(&xT[(2 * hi) * ST + j]), w1r = lds_pin(&sW1[(32 * w + j) * S1 + 2 * hi]);
      const uint32_t xgr = lds_pin(&xT[(j < K1 ? j : 0) * ST + 4 * hi]);
      typedef __attribute__((address_space(3))) f32x4 lds_f4;
      typedef __attribute__((address_space(3))) f32x2 lds_f2;
// (volatile: keeps the load a single ds_read_b32 with its 16-bit byte offset.  Paired into ds_read2_b32 -- 8-bit offsets, 1 020
//  bytes of reach -- every k-group of the MFMA loops needed a v_add_u32 to re-base: 33 + 64 per step between MFMAs)
#define FIT_LD1(base, off) (*(const volatile lds_float*)(uintptr_t)((base) + (uint32_t)((off) * 4)))
#define FIT_LD2(base, off) (*(const lds_f2*)(uintptr_t)((base) + (uint32_t)((off) * 4)))
#define FIT_LD4(base, off) (*(const lds_f4*)(uintptr_t)((base) + (uint32_t)((off) * 4)))
      MJX_FIT_STAMP(0);
      gather_store();
      lds_barrier();                                                                                // (1) minibatch staged
      MJX_FIT_STAMP(1);
      if (mb + 1 < A.steps) { gather_load(); if (mb + 2 < A.steps) index_load(ep, mb + 2); }   // next step's rows; indices of the one after
      // ---- layer 1: z1[unit 32w+., sample] = W1a x~a for both halves; h1 = relu -> h1^T
      {
        f32x16 z1[2] = {(f32x16)(0.f), (f32x16)(0.f)};
        // K1 <= 24: at most 6 k-groups; every operand fragment is requested before the first MFMA (one LDS latency per step
        // instead of one per group -- the wave issues in order)
        const int nq = K1 / 4;
        f32x2 a1[6];
        float b1x[6][2], b1y[6][2];
#pragma unroll
        for (int q = 0; q < 6; ++q)
          if (q < nq) {
            a1[q] = FIT_LD2(w1r, 4 * q);                                   // W1[32 w + j][4 q + 2 hi .. + 1]
#pragma unroll
            for (int h = 0; h < 2; ++h) { b1x[q][h] = FIT_LD1(x1c, (4 * q) * ST + 32 * h); b1y[q][h] = FIT_LD1(x1c, (4 * q + 1) * ST + 32 * h); }
          }
#pragma unroll
        for (int q = 0; q < 6; ++q)
          if (q < nq) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              z1[h] = MJX_MFMA(a1[q].x, b1x[q][h], z1[h]);
              z1[h] = MJX_MFMA(a1[q].y, b1y[q][h], z1[h]);
            }
          }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 16; ++r) LDS_AT(h1w)[unit_of(r, 0) * ST + 32 * h] = fmaxf(z1[h][r], 0.f);     // row 32 w + unit_of(r, hi)
      }
      lds_barrier();                                                                                // (2) h1^T complete
      MJX_FIT_STAMP(2);
      // ---- layer 2: K = all 128 h1 units; one W2 fragment per k-group, two B fragments (one per half)
      f32x16 z2[2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = *(const f32x4*)&sB2[32 * w + 8 * q + 4 * hi];
#pragma unroll
        for (int h = 0; h < 2; ++h) { z2[h][4 * q] = b.x; z2[h][4 * q + 1] = b.y; z2[h][4 * q + 2] = b.z; z2[h][4 * q + 3] = b.w; }
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        f32x4 ac = FIT_LD4(w2r, 0), an;
        float bc[2][4], bn[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int t = 0; t < 4; ++t) bc[h][t] = FIT_LD1(h1c, t * ST + 32 * h);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          if (g + 1 < 16) {
            const int k1c = 32 * ((g + 1) >> 2) + 8 * ((g + 1) & 3);          // (+ 4 hi: in the pinned bases)
            an = FIT_LD4(w2r, k1c);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int t = 0; t < 4; ++t) bn[h][t] = FIT_LD1(h1c, (k1c + t) * ST + 32 * h);
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) { z2[0] = MJX_MFMA(ac[t], bc[0][t], z2[0]); z2[1] = MJX_MFMA(ac[t], bc[1][t], z2[1]); }
          ac = an;
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < 4; ++t) bc[h][t] = bn[h][t];
        }
        // in-order single wave: group g+1's nine LDS fetches go out ahead of group g's eight MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          if (g + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float hv = fmaxf(z2[h][r], 0.f); z2[h][r] = hv; part = fmaf(w3v[r], hv, part); }
        part = half_sum(part);                                // + the other lane half
        if (hi == 0) sPart[w * 64 + 32 * h + j] = part;
      }
      lds_barrier();                                                                                // (3) output partials complete
      MJX_FIT_STAMP(3);
      // ---- output + MSE gradient: every lane for the two samples j, 32 + j (all waves redundantly)
      float dy[2], gw3p[16];
      {
        float e2 = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int s = 32 * h + j;
          const float yhat = (sPart[s] + sPart[64 + s]) + (sPart[128 + s] + sPart[192 + s]) + sB3[0];
          const float err = yhat - sY[s];
          dy[h] = 2.0f * err / 64.0f;                         // MSELoss(mean) over the 64-row minibatch
          e2 = fmaf(err, err, e2);
        }
        if (w == 0 && hi == 0) { const float t = sum32_lane0(e2); if (j == 0) ep_loss += (double)t / 64.0; }
      }
      // delta2 (lane = sample) -> d2^T; the output layer's weight gradient from the same registers: sum over this lane half's
      // 32 lanes (= samples j, both halves folded first) of h2 * dy, per accumulator register (= unit)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        LDS_AT(d2w)[unit_of(r, 0) * ST] = (z2[0][r] > 0.f) ? w3v[r] * dy[0] : 0.f;                          // row 32 w + unit_of(r, hi)
        LDS_AT(d2w)[unit_of(r, 0) * ST + 32] = (z2[1][r] > 0.f) ? w3v[r] * dy[1] : 0.f;
        gw3p[r] = fmaf(z2[1][r], dy[1], z2[0][r] * dy[0]);
      }
      sum32_lane0_x16(gw3p);                                  // valid in lane 0 of each half
      if (j == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sGW3[32 * w + unit_of(r, hi)] = gw3p[r];
      }
      float gb3 = 0.f;
      if (w == 0) { const float t = sum32_lane0(dy[0] + dy[1]); gb3 = t; }     // (valid in lane 0; only thread 0 uses it)
      lds_barrier();                                                                                // (4) d2^T, sGW3 complete
      MJX_FIT_STAMP(4);
      // ---- delta2 in lane = unit layout, grad b2
      f32x16 d2u[2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 t4 = FIT_LD4(d2r, 32 * h + 8 * q);
          d2u[h][4 * q] = t4.x; d2u[h][4 * q + 1] = t4.y; d2u[h][4 * q + 2] = t4.z; d2u[h][4 * q + 3] = t4.w;
        }
      float gb2;
      {
        float s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s2 += d2u[0][r] + d2u[1][r];
        gb2 = half_sum(s2);
      }
      MJX_FIT_STAMP(5);
      // ---- grad W2 rows of this wave: A = delta2u (registers), B = h1^T tiles
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 bc[NT], bn[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bc[nt] = FIT_LD4(h1n, 32 * nt * ST + 32 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (q + 1 < 4) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bn[nt] = FIT_LD4(h1n, 32 * nt * ST + 32 * h + 8 * (q + 1));
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) gW2[nt] = MJX_MFMA(d2u[h][4 * q + t], bc[nt][t], gW2[nt]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bc[nt] = bn[nt];
        }
      }
      MJX_FIT_STAMP(6);
      // ---- delta1 (lane = h1 unit of this wave's tile): A = delta2 [sample][k] from d2^T (per half), B = W2[k][unit] (shared)
      f32x16 d1u[2] = {(f32x16)(0.f), (f32x16)(0.f)};
      __builtin_amdgcn_sched_barrier(0);
      {
        float ac[2][4], an[2][4], bc[4], bn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          bc[t] = FIT_LD1(w2c, t * S2);
          ac[0][t] = FIT_LD1(d2c, t * ST); ac[1][t] = FIT_LD1(d2c, t * ST + 32);
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          if (g + 1 < 16) {
            const int k1c = 32 * ((g + 1) >> 2) + 8 * ((g + 1) & 3);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              bn[t] = FIT_LD1(w2c, (k1c + t) * S2);
              an[0][t] = FIT_LD1(d2c, (k1c + t) * ST); an[1][t] = FIT_LD1(d2c, (k1c + t) * ST + 32);
            }
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) { d1u[0] = MJX_MFMA(ac[0][t], bc[t], d1u[0]); d1u[1] = MJX_MFMA(ac[1][t], bc[t], d1u[1]); }
#pragma unroll
          for (int t = 0; t < 4; ++t) { ac[0][t] = an[0][t]; ac[1][t] = an[1][t]; bc[t] = bn[t]; }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          if (g + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) {                       // relu'(z1) mask from h1^T (same layout)
          const f32x4 hv = FIT_LD4(h1r, 32 * h + 8 * q);
          d1u[h][4 * q] = hv.x > 0.f ? d1u[h][4 * q] : 0.f; d1u[h][4 * q + 1] = hv.y > 0.f ? d1u[h][4 * q + 1] : 0.f;
          d1u[h][4 * q + 2] = hv.z > 0.f ? d1u[h][4 * q + 2] : 0.f; d1u[h][4 * q + 3] = hv.w > 0.f ? d1u[h][4 * q + 3] : 0.f;
        }
      MJX_FIT_STAMP(7);
      // ---- grad W1a rows of this wave (column d_in = grad b1)
      {
        f32x4 b4[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            b4[h][q] = FIT_LD4(xgr, 32 * h + 8 * q);
            if (j >= K1) b4[h][q] = (f32x4)(0.f);
          }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) gW1 = MJX_MFMA(d1u[h][4 * q + t], b4[h][q][t], gW1);
      }
      const float gw3 = ownw3 ? sGW3[tid] : 0.f;
      lds_barrier();                                                                                // (5) tiles are rewritten by the next step
      MJX_FIT_STAMP(8);
      // ---- Adam (torch.optim.Adam: L2 weight decay folded into the gradient, bias-corrected), moments in registers
      pw1 *= (double)b1c; pw2 *= (double)b2c;
      const float bc1 = (float)(1.0 - pw1), bc2s = (float)sqrt(1.0 - pw2);
      const float step_size = A.lr / bc1;
      const float inv_bc2s = 1.0f / bc2s;
      auto adam_math = [&](float p, float g, f32x2& q) {
        g += A.wd * p;
        float mi = q.x, vi = q.y;
        mi = mi + (g - mi) * (1.0f - b1c);
        vi = vi * b2c + g * g * (1.0f - b2c);
        q = f32x2{mi, vi};
        const float denom = fmaf(__builtin_amdgcn_sqrtf(vi), inv_bc2s, eps);
        float r = __builtin_amdgcn_rcpf(denom);
        r = r * fmaf(-denom, r, 2.0f);
        return fmaf(-step_size * mi, r, p);
      };
      auto adam_math2p = [&](f32x2 p, f32x2 g, f32x2& mi, f32x2& vi) {
        g = __builtin_elementwise_fma((f32x2)(A.wd), p, g);
        mi = __builtin_elementwise_fma(g - mi, (f32x2)(1.0f - b1c), mi);
        vi = __builtin_elementwise_fma(g * g, (f32x2)(1.0f - b2c), vi * (f32x2)(b2c));
        const f32x2 denom = __builtin_elementwise_fma(f32x2{__builtin_amdgcn_sqrtf(vi.x), __builtin_amdgcn_sqrtf(vi.y)}, (f32x2)(inv_bc2s), (f32x2)(eps));
        f32x2 r = {__builtin_amdgcn_rcpf(denom.x), __builtin_amdgcn_rcpf(denom.y)};
        r = r * __builtin_elementwise_fma(-denom, r, (f32x2)(2.0f));
        return __builtin_elementwise_fma(mi * (f32x2)(-step_size), r, p);
      };
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int l0 = unit_of(r, 0) * S2 + 32 * nt, l1 = unit_of(r + 1, 0) * S2 + 32 * nt;
          const f32x2 pn = adam_math2p(f32x2{pW2[l0], pW2[l1]}, f32x2{gW2[nt][r], gW2[nt][r + 1]}, q2[nt][r], q2[nt][r + 1]);
          pW2[l0] = pn.x; pW2[l1] = pn.y;
        }
      if (own1) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int l0 = unit_of(r, 0) * S1, l1 = unit_of(r + 1, 0) * S1;
          const f32x2 pn = adam_math2p(f32x2{pW1[l0], pW1[l1]}, f32x2{gW1[r], gW1[r + 1]}, q1[r], q1[r + 1]);
          pW1[l0] = pn.x; pW1[l1] = pn.y;
        }
      }
      if (ownb2) sB2[32 * w + j] = adam_math(sB2[32 * w + j], gb2, qb2);
      if (ownw3) sW3[tid] = adam_math(sW3[tid], gw3, qw3);
      if (ownb3) sB3[0] = adam_math(sB3[0], gb3, qb3);
      lds_barrier();                                                                                // (6) weights updated
      MJX_FIT_STAMP(9);
    }
    if (tid == 0) A.epoch_loss[ep] = ep_loss;
  }
  // ---- write the moments and the trained parameters back
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      mvW2[unit_of(r, 0) * H + 32 * nt] = f32x2{q2[nt][r].x, q2[nt][r + 1].x};
      mvW2[unit_of(r + 1, 0) * H + 32 * nt] = f32x2{q2[nt][r].y, q2[nt][r + 1].y};
    }
  if (own1) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      mvW1[unit_of(r, 0) * stg] = f32x2{q1[r].x, q1[r + 1].x};
      mvW1[unit_of(r + 1, 0) * stg] = f32x2{q1[r].y, q1[r + 1].y};
    }
  }
  if (ownb2) MV[gb2i] = qb2;
  if (ownw3) MV[gw3i] = qw3;
  if (ownb3) MV[gb3i] = qb3;
  __syncthreads();
  for (int64_t i = tid; i < Ptot; i += 256) { A.m[i] = A.mv[2 * i]; A.v[i] = A.mv[2 * i + 1]; }
  for (int i = tid; i < H * (d_in + 1); i += 256) {
    int u = i / (d_in + 1), f = i - u * (d_in + 1);
    if (f < d_in) A.params[oW1g + (int64_t)u * d_in + f] = sW1[u * S1 + f]; else A.params[oB1g + u] = sW1[u * S1 + d_in];
  }
  for (int i = tid; i < H * H; i += 256) A.params[oW2g + i] = sW2[(i / H) * S2 + (i % H)];
  for (int i = tid; i < H; i += 256) { A.params[oW3g + i] = sW3[i]; A.params[oB2g + i] = sB2[i]; }
  if (tid == 0) A.params[oB3g] = sB3[0];
}

}  // namespace mjx
