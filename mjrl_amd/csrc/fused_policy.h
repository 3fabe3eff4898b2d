// fused_policy.h -- single-launch fused kernels for small tanh-MLP Gaussian policies
// (hidden sizes 32 / 64) on gfx950.
//
// One persistent 4-wave workgroup per CU; every wave owns 32-sample tiles and keeps the
// whole forward / tangent / backward chain of its tile in registers:
//
//   * all products are formed "transposed" -- units on the MFMA M/K dims, the 32 samples
//     of the tile on the N dim -- so the accumulator of layer l (lane = sample, reg = unit)
//     is *directly* the B operand of layer l+1 (v_mfma_f32_32x32x2_f32; the weight A
//     operand is read from LDS in the matching k-permuted order with ds_read_b128);
//   * weight-gradient products reduce over samples, so they need lane = unit: the two
//     operands (delta^T, activation^T) take one trip through per-wave LDS scratch;
//   * weight gradients accumulate in MFMA accumulators across all of a wave's tiles and
//     are written once per workgroup as a partial (deterministic 2-stage reduction).
//
// Replaces (per launch) FCNetwork.forward x2, mean_LL, likelihood_ratio, mean_kl and the
// torch.autograd single / double backward of
//   mjrl/algos/batch_reinforce.py:40-58  (MODE_VPG)
//   mjrl/algos/npg_cg.py:62-81           (MODE_FVP, Gauss-Newton form valid at new==old)
//   mjrl/algos/batch_reinforce.py:40-52  (MODE_EVAL)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <vector>

namespace mjx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { MODE_VPG = 0, MODE_FVP = 1, MODE_EVAL = 2 };

struct FusedArgs {
  const float* obs;      // (N, n)
  const float* act;      // (N, m)   VPG / EVAL
  const float* adv;      // (N)      VPG / EVAL
  int64_t N;             // local samples
  float inv_N;           // 1 / N_global
  const float* thetaA;   // new parameters (flat)
  const float* thetaB;   // FVP: the vector v (flat);  VPG/EVAL: old parameters
  const float* trA;      // packed transforms of the new net (never null)
  const float* trB;      // packed transforms of the old net (never null)
  int old_is_new;        // VPG: skip the old forward (LR == 1 exactly)
  float* partials;       // [gridDim.x][d]   VPG / FVP
  double* spartials;     // [gridDim.x][4]
  float* dbg;            // optional dump of tile 0 (block 0, wave 0)
  long long* clk;        // optional: workgroup 0 stamps {cycle, real time} at entry / exit
  int reverse;           // cached FVP: walk the tiles from the far end (alternate launches: what the previous sweep touched
                         // last is still in the memory-side cache when this one starts there)
  float* hcache;         // forward-activation cache [tile][MT1+MT2][4][64 lanes][4]: written by MODE_VPG, read by cached FVP
  float* ocache;         // old-policy outputs [tile][MP + 1][32]: mean per action + log-likelihood; written by MODE_VPG
                         // (old == new), read by MODE_EVAL when thetaB / trB still equal `snap`
  const float* snap;     // [d + 2n + 2m] parameters and transforms the ocache was computed with
  int snap_trusted;      // MODE_EVAL: the caller vouches that thetaB / trB / trA still equal `snap` (the one-call updates: nothing
                         // outside the library ran since K1) -- the bit-exact compare in the prologue is skipped (~4 us per launch)
  float* snap_out;       // MODE_VPG: the kernel writes that snapshot (thetaB | trB) while it fills the ocache
  int n, m;
  int raw_dr;            // > 0: the gradient partial of a workgroup is written in ACCUMULATOR order (RawSlab<..>::DR floats per
                         // workgroup, see RawSlab below); k_reduce_partials4 maps the columns to the flat order (perm)
};

#ifdef MJX_PHASE_CLOCK
#define MJX_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (A.dbg && blockIdx.x == 0 && wave == 0 && lane == 0 && tile == tstride) ((long long*)A.dbg)[k] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MJX_STAMP(k) do {} while (0)
#endif
#ifdef MJX_PHASE_CLOCK
#define MJX_GSTAMP(k) do { if (A.dbg && blockIdx.x == 0 && threadIdx.x == 0) { ((long long*)A.dbg)[k] = (long long)__builtin_readcyclecounter(); ((long long*)A.dbg)[(k) + 8] = (long long)__builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define MJX_GSTAMP(k) do {} while (0)
#endif
#define MJX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// Accumulators that live across ALL tiles of a wave (the weight gradients of the cached Fisher-vector product) stay in the
// accumulation registers: with -amdgpu-mfma-vgpr-form the builtins put every MFMA result in an architectural VGPR and hipcc then
// shuttles the persistent ones to AGPRs and back around each use (120 v_accvgpr moves per tile, 6 cycles each: probe_fill.hip).
// Written as inline assembly with an "a" constraint they are AGPR operands of the instruction itself.  The compiler does not see an
// MFMA in them, so it inserts none of the wait states MFMA hazards need: (i) their results are only read after the tile loop;
// (ii) `volatile` keeps them in program order, which walks the accumulators round-robin -- measured (r03): with the statements
// free to move, hipcc grouped the 4x4x1 ones by accumulator, back to back with one s_nop, and the sums came out wrong (a
// dependent 4x4x1 needs more wait states than that; the builtin form gets them from the hazard recogniser).  Used only where
// >= 4 accumulators alternate (gW2, the 4x4x1 form of gW1) or the chain is made of 64-cycle 32x32x2 instructions.
#define MJX_MFMA_ACC(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define MJX_MFMA4_ACC(acc, a, b) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
// 4- and 8-byte LDS accesses of the cached Fisher-vector product.  (Measured, r03: making them volatile LDS-space accesses keeps
// hipcc from pairing them into ds_read2 / ds_write2 -- whose 8-bit offset fields cost a vector add per pair to re-base the
// address, 48 per tile -- and saves 2.5 % of the kernel's cycles, but the chip gives the same 2.5 % back in clock: no
// change in time, and hipcc 7.2 crashes on the volatile form in one instance.  Plain accesses it is.)
typedef __attribute__((address_space(3))) float lds_float;
// LDS byte address of a shared-memory pointer as an opaque 32-bit value: the compiler must keep it in a register (it would
// otherwise re-derive `base + constant` with a v_add_u32 next to every ds_write2 / ds_read2 whose 8-bit offsets do not reach --
// ~25 single vector-ALU instructions inside the MFMA phases of a cached Fisher-vector-product tile, 12 cycles each there)
__device__ __forceinline__ uint32_t lds_pin(const float* p) {
  uint32_t a = (uint32_t)(uintptr_t)(const lds_float*)p;
  asm volatile("" : "+v"(a));
  return a;
}
#define LDS_AT(addr) ((lds_float*)(uintptr_t)(addr))
#define LDS_ST(p, v) (*(p) = (v))
#define LDS_LD(p) (*(p))
#define LDS_LD2(p) (*(const f32x2*)(p))

__device__ __forceinline__ void wave_sync() {
  // LDS traffic between lanes of ONE wave: hardware executes a wave's DS ops in order,
  // this only stops the compiler from moving them across the hand-off.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// tanh(x) = 1 - 2 / (1 + exp2(2 log2(e) x)): branch-free, v_exp_f32 + v_rcp_f32 + one FMA; no |x|, no sign transfer, no
// Newton step (exp2 -> inf gives rcp -> 0 -> +1, exp2 -> 0 gives -1; a Newton step would form inf * 0 there).  Max abs error
// ~1.2e-7 (1 ulp of the reciprocal doubled + the exponential's ulp; ocml tanhf: 6e-8, branchy).  fp32 MFMAs do not hide
// vector-ALU work (DESIGN "what an fp32 MFMA hides"), so every instruction here is paid in full: r02's form -- sign(x)(1 - t)/(1 + t),
// t = exp2(-2 log2(e)|x|), Newton-refined -- cost 6 packed + 4 quarter-rate + 2 v_bfi per register pair, this one 3 + 4
// (K1 0.379 -> 0.361 ms, K3 0.177 -> 0.155 ms; step direction vs the reference at 1M: 2.90e-6 -> 2.95e-6, bar 1e-5).
__device__ __forceinline__ float fast_tanh(float x) {
  const float d = 1.0f + __builtin_amdgcn_exp2f(x * 2.885390081777927f);
  return fmaf(__builtin_amdgcn_rcpf(d), -2.0f, 1.0f);
}

// the same on a 16-register tile, two registers per packed instruction (scale, 1 + e, the final FMA; v_exp_f32 / v_rcp_f32
// stay per register)
__device__ __forceinline__ void fast_tanh16(f32x16& o, const f32x16& x) {
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const f32x2 s = f32x2{x[r], x[r + 1]} * (f32x2)(2.885390081777927f);
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(s.x), __builtin_amdgcn_exp2f(s.y)} + (f32x2)(1.0f);
    const f32x2 rr = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2 y = __builtin_elementwise_fma(rr, (f32x2)(-2.0f), (f32x2)(1.0f));
    o[r] = y.x; o[r + 1] = y.y;
  }
}

// a / b with v_rcp_f32 + one Newton step (<= 2 ulp; exact for b == 1): the per-sample divisions of the likelihood
// head and of the input normalisation cost 4 VALU ops instead of the ~12 of the IEEE sequence
__device__ __forceinline__ float fast_div(float a, float b) {
  float r = __builtin_amdgcn_rcpf(b);
  r = r * fmaf(-b, r, 2.0f);
  return a * r;
}

// mask ? a : b on bit patterns (mask = all ones / all zeros per lane)
__device__ __forceinline__ float half_select(uint32_t mask, float a, float b) {
  return __uint_as_float((__float_as_uint(a) & mask) | (__float_as_uint(b) & ~mask));
}

// x[lane] + x[lane ^ 32] in every lane: v_permlane32_swap exchanges the upper half of one copy with the
// lower half of the other (one VALU op, no LDS round trip like ds_bpermute)
__device__ __forceinline__ float half_sum(float p) {
  auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false);
  return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

// every lane: the sum over its 16-lane row, by DPP moves (vector-ALU latency; the same operand pairs, hence the same bits, as
// `v += __shfl_xor(v, 1); ... 2; ... 4; ... 8` -- after each step all lanes of a group hold the group's sum -- without the four
// ~100-cycle trips through the LDS pipe that ds_bpermute costs)
__device__ __forceinline__ float row16_sum(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xF, 0xF, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});       // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});       // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});      // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});      // row_mirror
  return v;
}

// packed fp32 element-wise helpers on 16-register tiles (register pairs -> v_pk_fma_f32 / v_pk_mul_f32)
__device__ __forceinline__ f32x16 pk_1mh2(const f32x16& h) {                      // 1 - h^2
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const f32x2 hh = {h[r], h[r + 1]};
    const f32x2 ff = __builtin_elementwise_fma(-hh, hh, (f32x2)(1.0f));
    o[r] = ff.x; o[r + 1] = ff.y;
  }
  return o;
}
__device__ __forceinline__ void pk_mul(f32x16& t, const f32x16& f) {              // t *= f
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    f32x2 tt = {t[r], t[r + 1]};
    tt *= f32x2{f[r], f[r + 1]};
    t[r] = tt.x; t[r + 1] = tt.y;
  }
}
__device__ __forceinline__ void pk_mul_1mh2(f32x16& t, const f32x16& h) {         // t *= 1 - h^2
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const f32x2 hh = {h[r], h[r + 1]};
    f32x2 tt = {t[r], t[r + 1]};
    tt *= __builtin_elementwise_fma(-hh, hh, (f32x2)(1.0f));
    t[r] = tt.x; t[r + 1] = tt.y;
  }
}
// The same two as ONE packed FMA per register pair with the negation in the instruction's modifiers: left to the compiler, -h
// becomes a v_xor_b32 per register (32 extra vector-ALU instructions per cached Fisher-vector-product tile) or the pair is split
// into scalar v_fma_f32 / v_mul_f32.  Inline asm is invisible to the compiler's hazard recogniser: NO wait states are inserted
// between an MFMA and an asm instruction that reads its result -- use these only on registers whose producing MFMA retired long
// ago (a whole MFMA phase in between), never directly behind the producer (the non-cached tile body does that: 2 % wrong and
// nondeterministic when it was tried there).
__device__ __forceinline__ f32x2 pk_1mh2_pair_far(f32x2 hh) {
  f32x2 ff;
  asm("v_pk_fma_f32 %0, %1, %1, 1.0 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(ff) : "v"(hh));
  return ff;
}
__device__ __forceinline__ f32x16 pk_1mh2_far(const f32x16& h) {
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const f32x2 ff = pk_1mh2_pair_far(f32x2{h[r], h[r + 1]});
    o[r] = ff.x; o[r + 1] = ff.y;
  }
  return o;
}
// a fresh MFMA result times 1 - h^2 of values that came from the vector ALU or from LDS: the factor as one packed FMA with
// modifiers (asm: its operands are not MFMA results), the product as a builtin (the compiler sees the MFMA -> VALU hazard)
__device__ __forceinline__ f32x2 mul_1mh2_pair(float a0, float a1, float h0, float h1) {
  const f32x2 ff = pk_1mh2_pair_far(f32x2{h0, h1});
  f32x2 tt = {a0, a1};
  tt *= ff;
  return tt;
}
__device__ __forceinline__ void pk_mul_1mh2_far(f32x16& t, const f32x16& h) {
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    f32x2 tt = {t[r], t[r + 1]};
    const f32x2 ff = pk_1mh2_pair_far(f32x2{h[r], h[r + 1]});
    asm("v_pk_mul_f32 %0, %0, %1" : "+v"(tt) : "v"(ff));
    t[r] = tt.x; t[r + 1] = tt.y;
  }
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<N, I + 1>(f); }
}

// unit index (within a 32-block) held by accumulator register r of lane-half hi
__device__ __forceinline__ constexpr int unit_of(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <int H1, int H2, int NT1, int MP>
struct FusedLayout {
  static constexpr int MT1 = H1 / 32, MT2 = H2 / 32;
  static constexpr int S2 = H1 + 4;             // row stride of W2 (ds_read_b128, S2/4 odd)
  static constexpr int ST = 36;                 // row stride of [unit][sample] scratch
  static constexpr int S3 = H2 + 4;             // row stride of W3 ([MP][S3])
  static constexpr int HM = (H1 > H2 ? H1 : H2);
  int NP;                                       // features + ones column, padded to 4
  int S1;                                       // row stride of xs / W1a (ds_read_b64, == 2 mod 4)
  int oW1, oW2, oW3, oB2, oB3, SLOT;            // weight slot (one per parameter set)
  int oXS, oXT, oD3, oBA, oBB, WAVE;            // per-wave scratch
  int oTR, oCST, oWAVES, TOTAL;
  static constexpr int NCST = 13;               // osc, osh, sigma, log_std (new) ; the same for old ; Dk = 2/(2 sigma^2 + 1e-8) ; {c3, osc^2 Dk / N} pairs ; 1/sigma new, old
                                                // (MODE_EVAL keeps mean_kl's per-action constants in the rows only the other modes read: 2, 8, 9, 10 --
                                                //  the 16-action 64 x 64 instance with 17 observations sits exactly AT the 160 KB limit, no row to spare)
  // eval_only: the layout of MODE_EVAL launches -- no backward pass, so a wave's scratch is just the raw image of its
  // observation tile (3.2 KB instead of 25 KB at HalfCheetah shapes); the workgroup then needs ~64 KB and TWO of them
  // share a CU, i.e. two waves per SIMD: one wave's tanh / likelihood VALU work runs under the other's MFMAs
  __host__ __device__ explicit FusedLayout(int n, bool eval_only = false) {
    NP = (n + 1 + 3) & ~3;
    S1 = NP + 2;
    oW1 = 0;
    oW2 = oW1 + H1 * S1;
    oW3 = oW2 + H2 * S2;                        // [MP][S3]
    oB2 = oW3 + MP * S3;
    oB3 = oB2 + H2;
    SLOT = ((oB3 + MP + 3) / 4) * 4;
    oXS = 0;                                    // raw image of the tile: 32*n floats (+ float4 slack)
    oXT = ((oXS + 32 * n + 4 * 64 + 3) / 4) * 4; // [NP][ST]
    oD3 = oXT + NP * ST;                        // [MP][ST]
    oBA = oD3 + MP * ST;                        // [HM][ST]
    oBB = oBA + HM * ST;                        // [HM][ST]
    WAVE = eval_only ? ((oXT + 3) / 4) * 4 : ((oBB + HM * ST + 3) / 4) * 4;
    oTR = 2 * SLOT;                             // in_shift / in_scale of A and B: 4 * NP
    oCST = oTR + 4 * NP;                        // per-action constants [NCST][MP]
    oWAVES = oCST + NCST * MP;
    TOTAL = oWAVES + 4 * WAVE;
  }
  __host__ __device__ size_t bytes() const { return (size_t)TOTAL * 4; }
  __host__ __device__ bool fits(int n) const { return n + 1 <= 32 * NT1; }
};

// offsets into the flat parameter vector
struct FlatOff {
  int W1, b1, W2, b2, W3, b3, S, d;
  __host__ __device__ FlatOff(int n, int m, int h1, int h2) {
    W1 = 0; b1 = W1 + h1 * n; W2 = b1 + h1; b2 = W2 + h2 * h1; W3 = b2 + h2; b3 = W3 + m * h2; S = b3 + m; d = S + m;
  }
};

// The gradient partial of a workgroup in ACCUMULATOR order (r06).  The kernels keep the weight gradients in MFMA accumulator
// layout; writing a workgroup's partial in the flat parameter order cost every launch ~8 400 cycles (4 us: per-element index
// arithmetic with the runtime observation width, scattered LDS writes) -- 4 % of a 1M-sample product, 8 % of a 125 k-sample one
// (one rank of eight).  Now every wave drops register r of accumulator X at slot [X][r][lane] of its LDS copy (one ds_write with
// an immediate offset), the four copies are added as they lie, and the COLUMN -> flat index map (a table made once per context
// on the host: perm[], -1 for the padding slots) is applied where each column's 256 partials have been summed
// (k_reduce_partials4).  Same additions in the same order: the same bits as the flat-order epilogue (MJX_RAW_SLAB=0).
template <int H1, int H2, int NT1, int MP, int NPC>
struct RawSlab {
  static constexpr int MT1 = H1 / 32, MT2 = H2 / 32, RA = MP / 2;
  static constexpr int QPI3 = 16 / (MP / 4), UPI3 = 4 * QPI3, NT3 = H2 / UPI3;
  static constexpr int NFQ = NPC ? NPC / 4 : 1;
  static constexpr int oW2 = 0, nW2 = MT2 * MT1 * 16 * 64;                         // [mt][nt][r][lane]
  static constexpr int oW1 = oW2 + nW2, nW1 = NPC ? MT1 * NFQ * 4 * 32 : MT1 * NT1 * 16 * 64;   // [mt][fq][r][lane < 32] | [mt][nt][r][lane]
  static constexpr int oW3 = oW1 + nW1, nW3 = NT3 * 4 * 64;                        // [nt][r][lane]
  static constexpr int oB2 = oW3 + nW3, nB2 = MT2 * 32;                            // [nt][j]
  static constexpr int oB3 = oB2 + nB2;                                            // [r][hi]
  static constexpr int oS = oB3 + 2 * RA;                                          // [r][hi]
  static constexpr int DR = ((oS + 2 * RA) + 3) & ~3;
  static int unit(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
  // perm[slot] = flat parameter index (mirrors the flat-order epilogue of k_fused) or -1; -> true when every flat index is hit exactly once
  static bool fill_perm(int* perm, int n, int m) {
    const FlatOff fo(n, m, H1, H2);
    for (int i = 0; i < DR; ++i) perm[i] = -1;
    auto w1 = [&](int u, int f) { return f < n ? fo.W1 + u * n + f : f == n ? fo.b1 + u : -1; };
    for (int mt = 0; mt < MT2; ++mt) for (int nt = 0; nt < MT1; ++nt) for (int r = 0; r < 16; ++r) for (int l = 0; l < 64; ++l)
      perm[oW2 + ((mt * MT1 + nt) * 16 + r) * 64 + l] = fo.W2 + (32 * mt + unit(r, l >> 5)) * H1 + 32 * nt + (l & 31);
    if (NPC) {
      for (int mt = 0; mt < MT1; ++mt) for (int fq = 0; fq < NFQ; ++fq) for (int r = 0; r < 4; ++r) for (int l = 0; l < 32; ++l)
        perm[oW1 + ((mt * NFQ + fq) * 4 + r) * 32 + l] = w1(32 * mt + 4 * ((l >> 2) & 7) + r, 4 * fq + (l & 3));
    } else {
      for (int mt = 0; mt < MT1; ++mt) for (int nt = 0; nt < NT1; ++nt) for (int r = 0; r < 16; ++r) for (int l = 0; l < 64; ++l)
        perm[oW1 + ((mt * NT1 + nt) * 16 + r) * 64 + l] = w1(32 * mt + unit(r, l >> 5), 32 * nt + (l & 31));
    }
    for (int nt = 0; nt < NT3; ++nt) for (int r = 0; r < 4; ++r) for (int l = 0; l < 64; ++l) {
      const int blk = l >> 2, a = 4 * (blk / QPI3) + r, u = 4 * (blk % QPI3) + (l & 3) + UPI3 * nt;
      perm[oW3 + (nt * 4 + r) * 64 + l] = a < m ? fo.W3 + a * H2 + u : -1;
    }
    for (int nt = 0; nt < MT2; ++nt) for (int jj = 0; jj < 32; ++jj) perm[oB2 + nt * 32 + jj] = fo.b2 + 32 * nt + jj;
    for (int r = 0; r < RA; ++r) for (int hi = 0; hi < 2; ++hi) {
      const int a = unit(r, hi);
      perm[oB3 + 2 * r + hi] = a < m ? fo.b3 + a : -1;
      perm[oS + 2 * r + hi] = a < m ? fo.S + a : -1;
    }
    std::vector<int> hits(fo.d, 0);
    for (int i = 0; i < DR; ++i) if (perm[i] >= 0) { if (perm[i] >= fo.d) return false; ++hits[perm[i]]; }
    for (int i = 0; i < fo.d; ++i) if (hits[i] != 1) return false;
    return true;
  }
};

template <int H1, int H2, int NT1, int MP, int MODE, bool DBG = false, int NPC = 0, bool CACHED = false>
__global__ __launch_bounds__(256, (MODE == MODE_EVAL && !DBG && MP <= 8) ? 2 : 1) void k_fused(FusedArgs A) {
  using LT = FusedLayout<H1, H2, NT1, MP>;
  constexpr int MT1 = LT::MT1, MT2 = LT::MT2, S2 = LT::S2, S3 = LT::S3, ST = LT::ST;
  constexpr int RA = MP / 2;                      // accumulator rows per lane that map to actions: a = unit_of(r, hi), r < RA
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hi = lane >> 5;
  // (wave-uniform by construction: as a scalar, the tile index and everything derived from it -- cache addresses, the
  //  wave's LDS base -- is computed on the scalar unit instead of per lane)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t himask = hi ? 0xffffffffu : 0u;
  const int n = A.n, m = A.m;
  // small per-wave scratch, two workgroups per CU (up to 8 actions: the 16- / 32-action instances do not fit 256 registers)
  constexpr bool EV2 = (MODE == MODE_EVAL) && !DBG && MP <= 8;
  // (instances with a compile-time feature count lay their LDS out for the widest observation they serve, NPC - 1 features:
  //  every offset is then a compile-time constant and folds into the DS instructions' immediate fields instead of costing a
  //  vector add per access -- vector-ALU instructions are not hidden by fp32 MFMAs, see the cached Fisher-vector product)
  const LT L(NPC ? NPC - 1 : n, EV2);
  const int NP = NPC ? NPC : L.NP;                // compile-time when the variant is specialised for the obs dim
  const int S1 = NP + 2;
  const FlatOff fo(n, m, H1, H2);

  float* slotA = lds;
  float* slotB = lds + L.SLOT;
  float* trs = lds + L.oTR;                    // [A shift][A scale][B shift][B scale] x NPMAX
  float* ws = lds + L.oWAVES + wave * L.WAVE;
  float* xs = ws + L.oXS;
  float* xT = ws + L.oXT;
  float* d3T = ws + L.oD3;
  float* bufA = ws + L.oBA;
  float* bufB = ws + L.oBB;

  MJX_GSTAMP(16);
  if (A.clk && blockIdx.x == 0 && threadIdx.x == 0) { A.clk[0] = (long long)__builtin_readcyclecounter(); A.clk[1] = (long long)__builtin_amdgcn_s_memrealtime(); }
  // ---------------- stage weights (whole workgroup) ----------------
  // Every global value is requested first (one batch of independent loads per thread), the LDS zero-fill runs
  // while they are in flight, then the values are scattered to their padded LDS rows: one memory round trip for
  // the whole prologue instead of one per staging loop.
  constexpr bool XCACHED = (MODE == MODE_FVP) && CACHED;      // observations / activations come from the K1 cache
  constexpr int C1 = NPC ? NPC / 4 : 8 * NT1;                 // W1a: 4 threads per row, features p, p + 4, ...
  constexpr int C2 = (H2 * H1 / 4 + 255) / 256;               // W2: 16-byte granules (rows are 16-byte aligned in theta)
  constexpr int C3 = (MP * H2 + 255) / 256;
  const int u1 = tid >> 2, p1 = tid & 3;
  float w1r[2][C1], w3r[2][C3], b2r[2], b3r[2], trr[4], csr[7];
  f32x4 w2r[2][C2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float* th = s ? A.thetaB : A.thetaA;
    if (!(XCACHED && s == 0)) {                               // (the cached FVP never reads W1a of the parameter slot)
#pragma unroll
      for (int c = 0; c < C1; ++c) {
        const int f = p1 + 4 * c;
        const bool ok = (u1 < H1) && (f <= n);
        w1r[s][c] = th[ok ? ((f < n) ? fo.W1 + u1 * n + f : fo.b1 + u1) : 0];
      }
    }
#pragma unroll
    for (int c = 0; c < C2; ++c) { const int i4 = c * 256 + tid; w2r[s][c] = *(const f32x4*)&th[fo.W2 + 4 * (i4 < H2 * H1 / 4 ? i4 : 0)]; }
#pragma unroll
    for (int c = 0; c < C3; ++c) { const int idx = c * 256 + tid; w3r[s][c] = th[fo.W3 + (idx < m * H2 ? idx : 0)]; }
    b2r[s] = th[fo.b2 + (tid < H2 ? tid : 0)];
    b3r[s] = th[fo.b3 + (tid < m ? tid : 0)];
  }
  {
    const int i = tid < n ? tid : 0;
    trr[0] = A.trA[i]; trr[1] = A.trA[n + i]; trr[2] = A.trB[i]; trr[3] = A.trB[n + i];
    const int a = tid < m ? tid : 0;
    csr[0] = A.thetaA[fo.S + a]; csr[1] = A.thetaB[fo.S + a];
    csr[2] = A.trA[2 * n + m + a]; csr[3] = A.trA[2 * n + a];
    csr[4] = A.trB[2 * n + m + a]; csr[5] = A.trB[2 * n + a];
    csr[6] = A.thetaB[fo.b3 + a];
  }
  // MODE_EVAL: the old policy's per-sample outputs of this batch may still be around from MODE_VPG (same update);
  // they are used only if the old parameters and transforms are bit-identical to the ones they were computed with.
  int mism = 0, mismx = 0;
  if (MODE == MODE_EVAL && A.ocache && !A.snap_trusted) {
    for (int idx = tid; idx < fo.d; idx += 256) mism |= (A.thetaB[idx] != A.snap[idx]);
    for (int idx = tid; idx < 2 * (n + m); idx += 256) mism |= (A.trB[idx] != A.snap[fo.d + idx]);
    // K1's normalised-observation image (in the forward-activation cache) may stand in for staging + normalising the raw
    // observations again -- if the NEW policy's input transform is still the one K1 normalised with (= the snapshot's: K1
    // fills the caches only when old == new)
    if (A.hcache) for (int idx = tid; idx < 2 * n; idx += 256) mismx |= (A.trA[idx] != A.snap[fo.d + idx]);
  }
  if (MODE == MODE_VPG && A.snap_out) {           // every block copies a slice: at most a few elements per thread
    const int nsnap = fo.d + 2 * (n + m);
    for (int idx = blockIdx.x * 256 + tid; idx < nsnap; idx += gridDim.x * 256)
      A.snap_out[idx] = idx < fo.d ? A.thetaB[idx] : A.trB[idx - fo.d];
  }
  // zero what is read before (or without) being written: the weight slots' pads, the constants, and each wave's
  // xs slack / xT / d3T.  bufA / bufB are fully rewritten every tile before they are read.
  {
    f32x4* z4 = (f32x4*)lds;
    for (int i = tid; i < L.oWAVES / 4; i += 256) z4[i] = (f32x4)(0.f);
    f32x4* w4 = (f32x4*)ws;
    for (int i = lane; i < (EV2 ? L.WAVE : L.oBA) / 4; i += 64) w4[i] = (f32x4)(0.f);
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float* slot = s ? slotB : slotA;
    if (!(XCACHED && s == 0)) {
#pragma unroll
      for (int c = 0; c < C1; ++c) {
        const int f = p1 + 4 * c;
        if ((u1 < H1) && (f <= n)) slot[L.oW1 + u1 * S1 + f] = w1r[s][c];
      }
    }
#pragma unroll
    for (int c = 0; c < C2; ++c) {
      const int i4 = c * 256 + tid;
      if (i4 < H2 * H1 / 4) *(f32x4*)&slot[L.oW2 + ((4 * i4) / H1) * S2 + (4 * i4) % H1] = w2r[s][c];
    }
#pragma unroll
    for (int c = 0; c < C3; ++c) {
      const int idx = c * 256 + tid;
      if (idx < m * H2) slot[L.oW3 + (idx / H2) * S3 + idx % H2] = w3r[s][c];
    }
    if (tid < H2) slot[L.oB2 + tid] = b2r[s];
    if (tid < m) slot[L.oB3 + tid] = b3r[s];
  }
  // (input transforms as shift and RECIPROCAL scale, 1 / (scale + 1e-8): x~ = (x - shift) * inv is two vector-ALU instructions
  //  per value where the division sequence was six -- vector-ALU time is not hidden by fp32 MFMAs; exact for the identity
  //  transform like the reference's division, <= 1.5 ulp otherwise.  Entries past the n features stay 0: padding columns read 0)
  if (tid < n) { trs[tid] = trr[0]; trs[NP + tid] = 1.0f / (trr[1] + 1e-8f); trs[2 * NP + tid] = trr[2]; trs[3 * NP + tid] = 1.0f / (trr[3] + 1e-8f); }
  // constant "ones" feature (bias column) of every wave's staging buffers
  if (!EV2 && lane < 32) xT[n * ST + lane] = 1.0f;

  // ---------------- per-action constants (LDS, broadcast reads) ----------------
  float* cst = lds + L.oCST;
  enum { C_OSC = 0, C_OSH = 1, C_SG = 2, C_LS = 3, C_OSCB = 4, C_OSHB = 5, C_SGB = 6, C_LSB = 7, C_DK = 8, C_ISG = 11, C_ISGB = 12,   // (9, 10: the FVP's {c3, scale} pairs)
         C_RD = 8, C_SO2 = 9, C_SN2 = 10, C_KD = 2 };       // MODE_EVAL only: 1 / Dr, sigma_old^2, sigma_new^2, log_std_new - log_std_old (over C_DK, the FVP pairs, C_SG)
  if (tid < MP) {
    const int a = tid;
    const bool ok = a < m;
    const float lsa = ok ? csr[0] : 0.f;
    const float lsb = (ok && MODE != MODE_FVP) ? csr[1] : lsa;
    const float sga = ok ? expf(lsa) : 1.0f;
    const float dk = 2.0f / (2.0f * sga * sga + 1e-8f);
    cst[C_OSC * MP + a] = ok ? csr[2] : 0.f;
    cst[C_OSH * MP + a] = ok ? csr[3] : 0.f;
    cst[C_SG * MP + a] = sga;
    cst[C_LS * MP + a] = lsa;
    cst[C_OSCB * MP + a] = ok ? csr[4] : 0.f;
    cst[C_OSHB * MP + a] = ok ? csr[5] : 0.f;
    cst[C_SGB * MP + a] = ok ? expf(lsb) : 1.0f;
    cst[C_LSB * MP + a] = lsb;
    cst[C_DK * MP + a] = dk;
    // (0 for the padding actions a >= m: their z is then 0 whatever the action registers hold -- the likelihood head takes
    //  the loaded values as they are, no select per action and sample)
    cst[C_ISG * MP + a] = ok ? 1.0f / sga : 0.f;
    cst[C_ISGB * MP + a] = ok ? 1.0f / cst[C_SGB * MP + a] : 0.f;
    if (MODE == MODE_EVAL) {
      // mean_kl (gaussian_mlp.py:135-145): what does not depend on the sample is formed once -- 1 / Dr with fast_div's own
      // reciprocal + Newton step (the per-sample quotient keeps its bits), sigma^2 of both nets, the log_std difference.
      // (r06: the head recomputed them per action and sample, 8 quarter-rate reciprocals and ~40 instructions per tile and wave.)
      const float sgb = cst[C_SGB * MP + a];
      const float Dr = 2.0f * sga * sga + 1e-8f;
      float r = __builtin_amdgcn_rcpf(Dr);
      r = r * fmaf(-Dr, r, 2.0f);
      cst[C_RD * MP + a] = r;
      cst[C_SO2 * MP + a] = sgb * sgb;
      cst[C_SN2 * MP + a] = sga * sga;
      cst[C_KD * MP + a] = lsa - lsb;
    }
    if (MODE == MODE_FVP) {                       // FVP epilogue: d3 = (md + c3) * osc^2 * Dk / N
      const float osc = ok ? csr[2] : 0.f;
      cst[9 * MP + 2 * a] = ok ? csr[6] : 0.f;
      cst[9 * MP + 2 * a + 1] = osc * (dk * (osc * A.inv_N));
    }
  }
  // (flag word in wave 0's staging area, zero since the fill above and rewritten by the first tile's staging;
  //  no __syncthreads_or: its static LDS word would not fit next to a 160 KB dynamic allocation)
  if (MODE == MODE_EVAL && mism) lds[L.oWAVES] = 1.0f;
  if (MODE == MODE_EVAL && mismx) lds[L.oWAVES + 1] = 1.0f;
  __syncthreads();
  bool use_oc = false, use_xi = false;
  if (MODE == MODE_EVAL && A.ocache) {
    use_oc = (lds[L.oWAVES] == 0.0f);
    use_xi = use_oc && NPC != 0 && A.hcache != nullptr && (lds[L.oWAVES + 1] == 0.0f);
    __syncthreads();
  }
  // FVP epilogue constants as wave-uniform scalars (no LDS round trip on the d3 critical path)
  float kc3[MP], kcs[MP];
  if (MODE == MODE_FVP) {
#pragma unroll
    for (int a = 0; a < MP; ++a) {
      kc3[a] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cst[9 * MP + 2 * a])));
      kcs[a] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cst[9 * MP + 2 * a + 1])));
    }
  }
  float kc3r[RA], kcsr[RA];                       // the same for the actions this lane half owns: a = unit_of(r, hi)
  if (MODE == MODE_FVP) {
#pragma unroll
    for (int r = 0; r < RA; ++r) { kc3r[r] = cst[9 * MP + 2 * unit_of(r, hi)]; kcsr[r] = cst[9 * MP + 2 * unit_of(r, hi) + 1]; }
  }
  // One wave-uniform base (scalar registers, formed on the scalar unit) + the lane's 32-bit byte offset + an immediate: no
  // vector-ALU address arithmetic per access.  (readfirstlane of a uniform value is a scalar move; it keeps the compiler from
  // folding the steps back into 64-bit vector adds.)
  typedef const char __attribute__((address_space(1)))* gcptr;
  typedef char __attribute__((address_space(1)))* gwptr;
  typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));          // a register pair at a 4-byte-aligned address
  auto ubase = [&](const float* p) {
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return (gcptr)(((uint64_t)hi32 << 32) | lo);
  };
  constexpr int OC_TILE = (MP + 1) * 32;
  float ocn[MP + 1];                              // next tile's old-policy outputs (MODE_EVAL)
  auto load_oc = [&](int64_t t) {
    gcptr b = ubase(A.ocache + t * OC_TILE);
#pragma unroll
    for (int a = 0; a <= MP; ++a) ocn[a] = *(const float __attribute__((address_space(1)))*)(b + (uint32_t)(j * 4) + a * 128);
  };

  MJX_GSTAMP(17);
  // ---------------- persistent accumulators ----------------
  // gW3 lives in 4x4 blocks of v_mfma_f32_4x4x1: lane l = (block b = l >> 2, column l & 3), block b = (action group
  // g = b / QPI3, unit quad uq = b % QPI3); accumulator i, register r: gW3[4 g + r][4 uq + (l & 3) + UPI3 i]
  constexpr int QPI3 = 16 / (MP / 4), UPI3 = 4 * QPI3, NT3 = H2 / UPI3;
  f32x16 gW1[MT1][NT1], gW2[MT2][MT1];
  // compile-time feature count (NPC != 0): gW1 on v_mfma_f32_4x4x1 instead -- block b = lane >> 2 holds
  // gW1[32 mt + 4 (b & 7) + r][4 fq + (lane & 3)] of accumulator (mt, fq), register r, summed over the samples of the
  // lane's half (the two halves are added once, after the tile loop): NPC / 4 instructions per sample pair instead of a
  // 32-column tile for NPC = 20 columns
  constexpr int NFQ = NPC ? NPC / 4 : 1;
  f32x4 gW1q[MT1][NFQ];
  f32x4 gW3[NT3];
  constexpr bool G3B = XCACHED || (MODE == MODE_VPG && !DBG && NT3 < 4);
  f32x4 gW3b[G3B ? NT3 : 1];          // cached FVP, K1 (r06): a second set (odd sample quads), so that 4 chains of 4x4x1 MFMAs are in flight
  float sb2[MT2], sb3r[RA], gls[RA];      // grad b2[32*nt + j] (every lane); grad b3 / grad log_std of action unit_of(r, hi), this lane's samples
#pragma unroll
  for (int a = 0; a < MT1; ++a)
#pragma unroll
    for (int b = 0; b < NT1; ++b) gW1[a][b] = (f32x16)(0.f);
#pragma unroll
  for (int a = 0; a < MT1; ++a)
#pragma unroll
    for (int b = 0; b < NFQ; ++b) gW1q[a][b] = (f32x4)(0.f);
#pragma unroll
  for (int a = 0; a < MT2; ++a)
#pragma unroll
    for (int b = 0; b < MT1; ++b) gW2[a][b] = (f32x16)(0.f);
#pragma unroll
  for (int a = 0; a < NT3; ++a) gW3[a] = (f32x4)(0.f);
#pragma unroll
  for (int a = 0; a < (G3B ? NT3 : 1); ++a) gW3b[a] = (f32x4)(0.f);
#pragma unroll
  for (int a = 0; a < RA; ++a) { gls[a] = 0.f; sb3r[a] = 0.f; }
#pragma unroll
  for (int a = 0; a < MT2; ++a) sb2[a] = 0.f;
  double s_surr = 0.0, s_kl = 0.0, s_cnt = 0.0;

  const int64_t ntiles = (A.N + 31) / 32;
  const int64_t tstride = (int64_t)gridDim.x * 4;
  // One tile's observations are 32*n contiguous floats = 8*n aligned float4 (128*n bytes per tile, so
  // every tile starts 16-byte aligned when obs is): each lane fetches up to XL4 float4, one tile ahead,
  // straight into its final registers (no select on the loaded value, so the loads issue back to back).
  constexpr int XL4 = 4 * NT1;
  f32x4 xr[XL4];
  const float inv_n = 1.0f / (float)n;

  auto load_x = [&](int64_t tile) {
    const float* base = A.obs + tile * 32 * (int64_t)n;
    const int64_t rem4 = ((A.N - tile * 32) * (int64_t)n + 3) >> 2;      // float4 left in the batch (rounded up)
#pragma unroll
    for (int c = 0; c < XL4; ++c) {
      const int e4 = c * 64 + lane;
      const bool ok = (e4 < 8 * n) && (e4 < rem4);
      xr[c] = *(const f32x4*)(base + 4 * (ok ? e4 : 0));
    }
  };

  // forward-activation cache (sample-lane accumulator layout, 16-byte granules: [tile][mt][q][lane][4])
  // followed by the normalised observations as the layer-1 B-operand image [q][lane][2] = x~[sample j][4q + 2hi + {0,1}]
  // (ones column and zero pad included), so the cached FVP neither stages nor normalises the tile again.
  constexpr int HC_H = (MT1 + MT2) * 4 * 64 * 4;              // floats per tile: activations ...
  const int HC_TILE = HC_H + (NP / 4) * 128;                  // ... + observation image
  constexpr int NQC = NPC ? NPC / 4 : 1;
  // cached FVP: h1 / h2 / observation image of the tile being processed.  The next tile's copies are fetched into the
  // same registers as soon as the current ones are dead (behind the weight-gradient products), so nothing is copied.
  f32x16 hn1[MT1], hn2[MT2];
  f32x2 xc[NQC];
  // (one wave-uniform base per 4 KB of the tile's cache lines: ubase above)
  auto load_h = [&](int64_t t) {
    const float* tb = A.hcache + t * HC_TILE;
    if (NPC) {
      gcptr xb = ubase(tb + HC_H);
#pragma unroll
      for (int q = 0; q < NQC; ++q) xc[q] = *(const f32x2 __attribute__((address_space(1)))*)(xb + (uint32_t)(lane * 8) + q * 512);
    }
#pragma unroll
    for (int mt = 0; mt < MT1; ++mt) {
      gcptr b = ubase(tb + mt * 1024);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = *(const f32x4 __attribute__((address_space(1)))*)(b + (uint32_t)(lane * 16) + q * 1024);
        hn1[mt][4 * q] = v.x; hn1[mt][4 * q + 1] = v.y; hn1[mt][4 * q + 2] = v.z; hn1[mt][4 * q + 3] = v.w;
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) {
      gcptr b = ubase(tb + (MT1 + mt) * 1024);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = *(const f32x4 __attribute__((address_space(1)))*)(b + (uint32_t)(lane * 16) + q * 1024);
        hn2[mt][4 * q] = v.x; hn2[mt][4 * q + 1] = v.y; hn2[mt][4 * q + 2] = v.z; hn2[mt][4 * q + 3] = v.w;
      }
    }
  };

  auto load_xi = [&](int64_t t) {                 // MODE_EVAL with K1's observation image: the layer-1 operand pairs of tile t
    gcptr xb = ubase(A.hcache + t * HC_TILE + HC_H);
#pragma unroll
    for (int q = 0; q < NQC; ++q) xc[q] = *(const f32x2 __attribute__((address_space(1)))*)(xb + (uint32_t)(lane * 8) + q * 512);
  };

  // K1 of the instances with a compile-time observation width (r06): the lane's layer-1 operand pairs come STRAIGHT from the
  // observation block -- lane (sample j, half hi) needs features 4 q + 2 hi, + 1 of its row, a 4-byte-aligned pair per q at a byte
  // offset that is the same in every tile (scalar tile base + pinned lane offset: no address arithmetic in the loop) -- one tile
  // ahead into registers, normalised in ONE packed burst at the end of the previous tile.  The route through LDS it replaces
  // (raw image, wave hand-over, per-value ds_read + subtract + multiply + two selects under divergent branches) cost K1 ~4 000 of
  // its 26 900 cycles per tile (`tools/phase_clock_k1.py`: staging 2 500, layer 1 3 120 for 1 280 of MFMA).
  // Only the last pair can hold the ones column (f == n) or padding (f > n): its two elements are fetched one by one, from the
  // row's first element where f >= n (in bounds; shift and reciprocal scale are 0 there, the value drops out), and the 1 comes in
  // as the addend of a packed FMA.  Rows past the batch end (last tile only, a uniform branch): fetched from the tile's first row, zeroed.
  constexpr bool DIRX = (MODE == MODE_VPG) && NPC != 0 && !DBG;
  constexpr int NQX = DIRX ? NPC / 4 : 1;
  f32x2 xn[NQX], xw[NQX];                         // this tile's normalised pairs ; the next tile's raw pairs
  uint32_t xo[NQX + 1];
  f32x2 onep = {0.f, 0.f};
  auto set_xoff = [&](uint32_t (&o)[NQX + 1], uint32_t row) {
#pragma unroll
    for (int q = 0; q + 1 < NQX; ++q) o[q] = (row * (uint32_t)n + 4 * q + 2 * hi) * 4;
    const int f0 = NPC - 4 + 2 * hi;
    o[NQX - 1] = (row * (uint32_t)n + (f0 < n ? f0 : 0)) * 4;
    o[NQX] = (row * (uint32_t)n + (f0 + 1 < n ? f0 + 1 : 0)) * 4;
  };
  auto load_xw = [&](f32x2 (&w)[NQX], const uint32_t (&o)[NQX + 1], int64_t t) {
    gcptr b = ubase(A.obs + t * 32 * (int64_t)n);
#pragma unroll
    for (int q = 0; q + 1 < NQX; ++q) { const f32x2u v = *(const f32x2u __attribute__((address_space(1)))*)(b + o[q]); w[q] = f32x2{v.x, v.y}; }
    w[NQX - 1].x = *(const float __attribute__((address_space(1)))*)(b + o[NQX - 1]);
    w[NQX - 1].y = *(const float __attribute__((address_space(1)))*)(b + o[NQX]);
  };
  // x~ = (x - shift) * (1 / (scale + 1e-8)) pair by pair (the same two roundings as the scalar form); `t` = the tile the raw pairs belong to
  auto norm_xw = [&](f32x2 (&dst)[NQX], const f32x2 (&w)[NQX], const float* tsh, const float* tsc, int64_t t) {
#pragma unroll
    for (int q = 0; q < NQX; ++q) {
      const f32x2 sh = *(const f32x2*)&tsh[4 * q + 2 * hi], sc = *(const f32x2*)&tsc[4 * q + 2 * hi];
      const f32x2 d = w[q] - sh;
      dst[q] = (q == NQX - 1) ? __builtin_elementwise_fma(d, sc, onep) : d * sc;
    }
    if ((t + 1) * 32 > A.N) {                     // the batch's last, partial tile
      const bool vl = t * 32 + j < A.N;
#pragma unroll
      for (int q = 0; q < NQX; ++q) { dst[q].x = vl ? dst[q].x : (q == NQX - 1 ? onep.x : 0.f); dst[q].y = vl ? dst[q].y : (q == NQX - 1 ? onep.y : 0.f); }
    }
  };

  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  // cached FVP and K1 (the instances with registers to spare): this lane's address in row group g (rows 8 g + 4 hi .. + 3) of
  // the two transposed tiles, pinned in registers
  constexpr bool PINNED = XCACHED || (MODE == MODE_VPG && !DBG);
  uint32_t pinA[PINNED ? 4 * MT2 : 1], pinB[PINNED ? 4 * MT1 : 1];
  if constexpr (PINNED) {
#pragma unroll
    for (int g = 0; g < 4 * MT2; ++g) pinA[g] = lds_pin(&bufA[(8 * g + 4 * hi) * ST + j]);
#pragma unroll
    for (int g = 0; g < 4 * MT1; ++g) pinB[g] = lds_pin(&bufB[(8 * g + 4 * hi) * ST + j]);
  }
  // ... of the transposed observation tile (rows 8 g + 2 hi: the cached FVP stores rows 4 q + 2 hi, + 1 per layer-1 k-step)
  // (instances with a compile-time observation width)
  constexpr bool PINX = XCACHED && NPC != 0;
  uint32_t pinX[PINX ? (NPC / 4 + 1) / 2 : 1];
  if constexpr (PINX) {
#pragma unroll
    for (int g = 0; g < (NPC / 4 + 1) / 2; ++g) pinX[g] = lds_pin(&xT[(8 * g + 2 * hi) * ST + j]);
  }
  // ... and in the k-groups of W2's rows the delta1 phase reads (rows 8 g + 4 hi .. + 3 of the NEW parameters' slot, columns j, 32 + j)
  uint32_t pinW[PINNED ? 4 * MT2 : 1];
  if constexpr (PINNED) {
#pragma unroll
    for (int g = 0; g < 4 * MT2; ++g) pinW[g] = lds_pin(&slotA[L.oW2 + (8 * g + 4 * hi) * S2 + j]);
  }
  float sum_lsA = 0.f, sum_lsB = 0.f;             // sum of log_std over the actions (new, old): constants of the likelihood head
  if (MODE != MODE_FVP) {
#pragma unroll
    for (int a = 0; a < MP; ++a) { sum_lsA += cst[C_LS * MP + a]; sum_lsB += cst[C_LSB * MP + a]; }
  }
  uint32_t aoff = (uint32_t)(j * m * 4);          // the lane's row in a tile of the action block
  asm volatile("" : "+v"(aoff));
  const bool rev = XCACHED && A.reverse;
  auto ptile = [&](int64_t t) { return rev ? ntiles - 1 - t : t; };          // logical -> physical tile of this launch
  if constexpr (DIRX) {
    const int f0 = NPC - 4 + 2 * hi;
    onep = f32x2{f0 == n ? 1.0f : 0.f, f0 + 1 == n ? 1.0f : 0.f};
    set_xoff(xo, (uint32_t)j);
    if (tile < ntiles) {
      if ((tile + 1) * 32 > A.N) set_xoff(xo, (tile * 32 + j < A.N) ? (uint32_t)j : 0u);
      load_xw(xw, xo, tile);
      norm_xw(xn, xw, trs, trs + NP, tile);
    }
  }
  if (!DIRX && !XCACHED && !use_xi && tile < ntiles) load_x(tile);
  if (XCACHED && tile < ntiles) load_h(ptile(tile));
  if (MODE == MODE_EVAL && use_oc && tile < ntiles) load_oc(tile);
  if (MODE == MODE_EVAL && use_xi && tile < ntiles) load_xi(tile);

  // (the tile loop is instantiated twice for MODE_EVAL: XI = K1's normalised-observation image replaces the staging and
  //  normalisation of the raw observations; the choice is made once per launch, above)
  auto run_tiles = [&](auto xi_tag) {
  constexpr bool XI = decltype(xi_tag)::value;
  for (; tile < ntiles; tile += tstride) {
    const int64_t s0 = (XCACHED ? ptile(tile) : tile) * 32;
    const bool valid = (s0 + j) < A.N;
    MJX_STAMP(0);
    if constexpr (XCACHED) {
      // ================= cached-forward Fisher-vector product: its own tile schedule (r03) =================
      // Measured on gfx950 (tools/probe_fill.hip): a v_mfma_f32_32x32x2_f32 does NOT hide vector-ALU instructions -- fp32
      // matrix products run at the vector-FMA rate on a shared datapath: every VALU instruction between two MFMAs costs its
      // 4 cycles, and the first one of a gap 8 more; LDS reads / stores, s_waitcnt, s_nop and scalar instructions are free
      // in the 64-cycle shadow.  So the tile is a sequence of MFMA phases that carry ONLY LDS traffic, separated by a few
      // VALU bursts (one entry penalty each), VALU instructions are counted, and whatever can go through LDS does:
      //   R1  t1  = V1a x~                      (20 MFMAs)   LDS: x~^T -> xT
      //   R2  t2  = c2 + V2 h1                  (64)         LDS: h1^T -> bufB, h2^T -> bufA
      //   VA  t1 *= 1 - h1^2 ; f2 = 1 - h2^2                 (96 VALU)
      //   R3  t2 += W2 t1                       (64)
      //   R4a out = V3 h2 (4x4x1; fills the drain of R3)   VB  t2 *= f2 (32)   R4b out += W3 t2 (4x4x1)
      //   VC  d3 (epilogue, ~35 VALU), d3^T -> LDS
      //   R6  delta2 = W3^T d3, lane = sample   (8)          -- ONE layout: the lane = unit copy takes a trip through LDS
      //   R7  gW3 += d3^T h2 (4x4x1; fills the drain of R6)   VD  delta2 *= f2 (32)
      //   R8  delta1 = W2^T delta2, lane = unit (64)         LDS: delta2^T -> bufA (h2^T is dead) -> lane = unit registers;
      //                                                       next tile's cache lines requested
      //   R9  gW2 += delta2^T h1                (64)         LDS: h1^T fetches
      //   VE  delta1 *= 1 - h1^2 (64, from the h1^T fetches) ; b2 sums (32)
      //   R10 gW1 += delta1^T x~  (4x4x1 / 32x32x2)
      constexpr int NGRP = MP / 4;
      f32x16 (&h1)[MT1] = hn1;
      f32x16 (&h2)[MT2] = hn2;
      f32x16 t1[MT1], t2[MT2];
      f32x16 f2s[MT2];
      // ---------------- R1
#pragma unroll
      for (int mt = 0; mt < MT1; ++mt) t1[mt] = (f32x16)(0.f);
      {
        f32x2 vc[MT1];
        const int f00 = 2 * hi;
        const float* ximg = A.hcache + ptile(tile) * HC_TILE + HC_H + lane * 2;
        auto xpair = [&](int q) {
          if constexpr (NPC != 0) return xc[q]; else return *(const f32x2*)(ximg + q * 128);
        };
        f32x2 xb = xpair(0);
        float xb0 = xb.x, xb1 = xb.y;
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt) vc[mt] = LDS_LD2(&slotB[L.oW1 + (32 * mt + j) * S1 + f00]);
#pragma unroll NPC ? NPC / 4 : 1
        for (int q = 0; q < (NPC ? NPC / 4 : NP / 4); ++q) {
          const int f0 = 4 * q + 2 * hi;
          const int f1 = (q + 1 < NP / 4) ? f0 + 4 : f0;
          f32x2 vn[MT1];
          const f32x2 xnx = xpair((q + 1 < NP / 4) ? q + 1 : q);
#pragma unroll
          for (int mt = 0; mt < MT1; ++mt) vn[mt] = LDS_LD2(&slotB[L.oW1 + (32 * mt + j) * S1 + f1]);
          if constexpr (PINX) { LDS_AT(pinX[q >> 1])[(4 * (q & 1)) * ST] = xb0; LDS_AT(pinX[q >> 1])[(4 * (q & 1) + 1) * ST] = xb1; }     // rows f0 = 4 q + 2 hi, f0 + 1
          else { LDS_ST(&xT[f0 * ST + j], xb0); LDS_ST(&xT[(f0 + 1) * ST + j], xb1); }
#pragma unroll
          for (int mt = 0; mt < MT1; ++mt) t1[mt] = MJX_MFMA(vc[mt].x, xb0, t1[mt]);
#pragma unroll
          for (int mt = 0; mt < MT1; ++mt) t1[mt] = MJX_MFMA(vc[mt].y, xb1, t1[mt]);
          xb0 = xnx.x; xb1 = xnx.y;
#pragma unroll
          for (int mt = 0; mt < MT1; ++mt) vc[mt] = vn[mt];
        }
      }
      MJX_STAMP(1);
      // ---------------- R2: t2 = c2 + V2 h1 ; the [unit][sample] copies of h1 / h2 (LDS stores: free) ride along
      constexpr int NS = MT1 * 4;                               // (kb, q) operand groups
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 c = *(const f32x4*)&slotB[L.oB2 + 32 * mt + 8 * q + 4 * hi];
          t2[mt][4 * q + 0] = c.x; t2[mt][4 * q + 1] = c.y; t2[mt][4 * q + 2] = c.z; t2[mt][4 * q + 3] = c.w;
        }
      {
        __builtin_amdgcn_sched_barrier(0);
        f32x4 vc[MT2], vn[MT2];
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) vc[mt] = *(const f32x4*)&slotB[L.oW2 + (32 * mt + j) * S2 + 4 * hi];
        constexpr int R1 = 16 * MT1 / NS, R2 = 16 * MT2 / NS;   // registers of h1 / h2 stored per step
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          const int kb = st >> 2, q = st & 3;
          if (st + 1 < NS) {
            const int kb1 = (st + 1) >> 2, q1 = (st + 1) & 3;
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) vn[mt] = *(const f32x4*)&slotB[L.oW2 + (32 * mt + j) * S2 + 32 * kb1 + 8 * q1 + 4 * hi];
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) t2[mt] = MJX_MFMA(vc[mt][t], h1[kb][4 * q + t], t2[mt]);
#pragma unroll
          for (int e = 0; e < R1; ++e) {
            const int idx = st * R1 + e, mt = idx >> 4, r = idx & 15;
            LDS_AT(pinB[4 * mt + (r >> 2)])[(r & 3) * ST] = h1[mt][r];          // row 32 mt + unit_of(r, hi) = 8 (4 mt + r / 4) + 4 hi + r % 4
          }
#pragma unroll
          for (int e = 0; e < R2; ++e) {
            const int idx = st * R2 + e, mt = idx >> 4, r = idx & 15;
            LDS_AT(pinA[4 * mt + (r >> 2)])[(r & 3) * ST] = h2[mt][r];
          }
#pragma unroll
          for (int mt = 0; mt < MT2; ++mt) vc[mt] = vn[mt];
        }
        // pipeline: fragment prefetch, then the MFMAs with the LDS stores between them -- and no vector-ALU instruction
        __builtin_amdgcn_sched_group_barrier(0x100, MT2, 0);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          if (st + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, MT2, 0);
#pragma unroll
          for (int i = 0; i < 4 * MT2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      MJX_STAMP(2);
      // (the first operands of the next MFMA phases are requested before each burst, so that they land during it)
      f32x4 wc3[MT2];
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt) wc3[mt] = *(const f32x4*)&slotA[L.oW2 + (32 * mt + j) * S2 + 4 * hi];
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- VA: t1 *= 1 - h1^2 ; f2 = 1 - h2^2   (one burst; R1 retired long ago, R2 does not touch t1)
      // (packed: v_pk_fma_f32 / v_pk_mul_f32 handle two registers per instruction at the price of one in a burst -- probe_fill.hip)
#pragma unroll
      for (int mt = 0; mt < MT1; ++mt) pk_mul_1mh2_far(t1[mt], h1[mt]);     // (t1: R1's result, the 64 MFMAs of R2 ago; h1 / h2: loaded)
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt) f2s[mt] = pk_1mh2_far(h2[mt]);
      __builtin_amdgcn_sched_barrier(0);
      MJX_STAMP(3);
      // ---------------- R3: t2 += W2 t1
      {
        f32x4 wc[MT2], wn[MT2];
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) wc[mt] = wc3[mt];
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          const int kb = st >> 2, q = st & 3;
          if (st + 1 < NS) {
            const int kb1 = (st + 1) >> 2, q1 = (st + 1) & 3;
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) wn[mt] = *(const f32x4*)&slotA[L.oW2 + (32 * mt + j) * S2 + 32 * kb1 + 8 * q1 + 4 * hi];
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) t2[mt] = MJX_MFMA(wc[mt][t], t1[kb][4 * q + t], t2[mt]);
#pragma unroll
          for (int mt = 0; mt < MT2; ++mt) wc[mt] = wn[mt];
        }
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          if (st + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, MT2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT2, 0);
        }
      }
      // (output-layer fragments of both weight sets: requested under R3's last MFMAs)
      f32x4 wa[MP / 4], wb[MP / 4];
      {
        const int b = (lane >> 2) & 7;
        const int woff = L.oW3 + (lane & 3) * S3 + 32 * ((b >> 2) % MT2) + 8 * (b & 3) + 4 * hi;
#pragma unroll
        for (int gp = 0; gp < MP / 4; ++gp) { wa[gp] = *(const f32x4*)(slotB + woff + 4 * gp * S3); wb[gp] = *(const f32x4*)(slotA + woff + 4 * gp * S3); }
      }
      __builtin_amdgcn_sched_barrier(0);
      MJX_STAMP(4);
      // ---------------- R4: output layer on v_mfma_f32_4x4x1_16b_f32 (operands through the A-broadcast, see out_small below).
      // V3 h2 does not depend on t2: its instructions keep the matrix pipe busy while R3 drains; then the t2 *= f2 burst.
      constexpr int CH = (NGRP >= 4) ? 1 : 2;                   // independent accumulator chains per action group and product
      f32x4 og[NGRP];
      float w6[RA][MT2];
      {
        f32x4 oa[NGRP][CH], ob[NGRP][CH];
#pragma unroll
        for (int gp = 0; gp < NGRP; ++gp)
#pragma unroll
          for (int c = 0; c < CH; ++c) { oa[gp][c] = (f32x4)(0.f); ob[gp][c] = (f32x4)(0.f); }
        // (chains alternate by k-step: with NGRP = 2 an accumulator recurs every 4th instruction -- a dependent 4x4x1 two
        //  instructions behind its producer still stalls)
        static_for<MT2 * 4>([&](auto st) {
          constexpr int mt = decltype(st)::value >> 2, q = decltype(st)::value & 3;
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int gp = 0; gp < NGRP; ++gp)
              oa[gp][t % CH] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[gp][t], h2[mt][4 * q + t], oa[gp][t % CH], 3, mt * 4 + q, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) pk_mul(t2[mt], f2s[mt]);
        __builtin_amdgcn_sched_barrier(0);
        static_for<MT2 * 4>([&](auto st) {
          constexpr int mt = decltype(st)::value >> 2, q = decltype(st)::value & 3;
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int gp = 0; gp < NGRP; ++gp)
              ob[gp][t % CH] = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[gp][t], t2[mt][4 * q + t], ob[gp][t % CH], 3, mt * 4 + q, 0);
        });
        // (R6's weight operands: requested before the epilogue burst)
#pragma unroll
        for (int sidx = 0; sidx < RA; ++sidx)
#pragma unroll
          for (int mt = 0; mt < MT2; ++mt) w6[sidx][mt] = LDS_LD(&slotA[L.oW3 + unit_of(sidx, hi) * S3 + 32 * mt + j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gp = 0; gp < NGRP; ++gp) {
          og[gp] = oa[gp][0] + ob[gp][0];
          if (CH == 2) og[gp] += oa[gp][CH - 1] + ob[gp][CH - 1];
        }
      }
      MJX_STAMP(5);
      // ---------------- VC: d3 = (md + c3) * out_scale^2 Dk / N for the actions this lane half owns (0 past the batch end)
      float d3r[RA];
      {
        const float vmask = valid ? 1.0f : 0.0f;
#pragma unroll
        for (int r = 0; r < RA; ++r) {
          const int g0 = 2 * (r >> 2), c = r & 3;
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(og[g0][c]), __float_as_uint(og[g0 + 1][c]), false, false);
          const float mdr = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
          d3r[r] = (mdr + kc3r[r]) * (kcsr[r] * vmask);
        }
#pragma unroll
        for (int r = 0; r < RA; ++r) { LDS_ST(&d3T[unit_of(r, hi) * ST + j], d3r[r]); sb3r[r] += d3r[r]; }
      }
      __builtin_amdgcn_sched_barrier(0);
      MJX_STAMP(6);
      // ---------------- R6: delta2 (lane = sample) = W3^T d3 ; R7: gW3[a][k] += sum_s d3[s][a] h2[s][k] on 4x4x1 blocks (see the
      // VPG branch below; two persistent accumulator sets = 4 chains) right behind it, filling R6's drain
      f32x16 dl2s[MT2];
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt) dl2s[mt] = (f32x16)(0.f);
#pragma unroll
      for (int sidx = 0; sidx < RA; ++sidx)
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) dl2s[mt] = MJX_MFMA(w6[sidx][mt], d3r[sidx], dl2s[mt]);
      float wc8[4][MT1];
      {
        const int blk = lane >> 2, g3 = blk / QPI3, uq3 = blk % QPI3;
        const float* arow = &d3T[(4 * g3 + (lane & 3)) * ST];
        const float* brow = &bufA[(4 * uq3 + (lane & 3)) * ST];
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {                 // sample quads 2 sp (-> gW3) and 2 sp + 1 (-> gW3b), instruction by instruction
          const f32x4 ave = *(const f32x4*)(arow + 8 * sp), avo = *(const f32x4*)(arow + 8 * sp + 4);
          f32x4 bve[NT3], bvo[NT3];
#pragma unroll
          for (int nt = 0; nt < NT3; ++nt) { bve[nt] = *(const f32x4*)(brow + UPI3 * nt * ST + 8 * sp); bvo[nt] = *(const f32x4*)(brow + UPI3 * nt * ST + 8 * sp + 4); }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < NT3; ++nt) {
              gW3[nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(ave[t], bve[nt][t], gW3[nt], 0, 0, 0);
              gW3b[nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(avo[t], bvo[nt][t], gW3b[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nt = 0; nt < MT1; ++nt) wc8[t][nt] = LDS_AT(pinW[0])[t * S2 + 32 * nt];
      }
      __builtin_amdgcn_sched_barrier(0);
      MJX_STAMP(7);
      // ---------------- VD: delta2 *= f2
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt) pk_mul(dl2s[mt], f2s[mt]);
      // ---------------- R8: delta1 (lane = unit) = W2^T delta2 ; delta2^T -> bufA -> lane = unit registers in its shadow
      // (LDS executes one wave's operations in order and the compiler keeps may-alias accesses in program order)
      f32x16 dl1u[MT1];
      float dl2u[MT2][16];
#pragma unroll
      for (int nt = 0; nt < MT1; ++nt) dl1u[nt] = (f32x16)(0.f);
      {
        constexpr int NG = MT2 * 4;                 // groups of 4 k-steps over the h2 units
        constexpr int NGH = NG / 2;
        float wc[4][MT1], wn[4][MT1];
        __builtin_amdgcn_sched_barrier(0);
        // h1 / h2 / the operand image are dead since R4: request the next tile's copies under these MFMAs, two weight-gradient
        // phases ahead of their first use (branch-free -- the last tile of a wave fetches its own lines again -- so that the
        // loads stay inside this scheduling region)
        load_h(ptile((tile + tstride < ntiles) ? tile + tstride : tile));
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nt = 0; nt < MT1; ++nt) wc[t][nt] = wc8[t][nt];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int kb = g >> 2, q = g & 3;
          if (g + 1 < NG) {
            const int kb1 = (g + 1) >> 2, q1 = (g + 1) & 3;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int nt = 0; nt < MT1; ++nt) wn[t][nt] = LDS_AT(pinW[4 * kb1 + q1])[t * S2 + 32 * nt];
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < MT1; ++nt) dl1u[nt] = MJX_MFMA(dl2s[kb][4 * q + t], wc[t][nt], dl1u[nt]);
          if (g < NGH) {
            // this group's share of the transposed copy: 16 MT2 / NGH registers of delta2
            constexpr int RS = 16 * MT2 / NGH;
#pragma unroll
            for (int e = 0; e < RS; ++e) {
              const int idx = g * RS + e, mt = idx >> 4, r = idx & 15;
              LDS_AT(pinA[4 * mt + (r >> 2)])[(r & 3) * ST] = dl2s[mt][r];
            }
          } else {
            // ... and of the read-back: delta2[sample unit_of(4 qq + t, hi)][unit 32 nt + j], one ds_read_b128 per (nt, qq)
            constexpr int RQ = (4 * MT2 + NGH - 1) / NGH;
#pragma unroll
            for (int e = 0; e < RQ; ++e) {
              const int idx = (g - NGH) * RQ + e;
              if (idx < 4 * MT2) {
                const int nt = idx >> 2, qq = idx & 3;
                const f32x4 v = *(const f32x4*)&bufA[(32 * nt + j) * ST + 8 * qq + 4 * hi];
                dl2u[nt][4 * qq] = v.x; dl2u[nt][4 * qq + 1] = v.y; dl2u[nt][4 * qq + 2] = v.z; dl2u[nt][4 * qq + 3] = v.w;
              }
            }
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < MT1; ++nt) wc[t][nt] = wn[t][nt];
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT1, 0);      // (the ds_read_b32 pair up as ds_read2_b32)
#pragma unroll
          for (int i = 0; i < 4 * MT1; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (g < NGH) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            else if (i < (4 * MT2 + NGH - 1) / NGH) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
      }
      // ---------------- R9: gW2[u2][u1] += sum_s delta2[s][u2] h1[s][u1]   (B operand: h1^T from bufB, kept for the burst below)
      f32x4 bcs[4][MT1];
#pragma unroll
      for (int nt = 0; nt < MT1; ++nt) bcs[0][nt] = *(const f32x4*)&bufB[(32 * nt + j) * ST + 4 * hi];
      __builtin_amdgcn_sched_barrier(0);
      MJX_STAMP(8);
      {
#pragma unroll
        for (int q = 1; q < 4; ++q)
#pragma unroll
          for (int nt = 0; nt < MT1; ++nt) bcs[q][nt] = *(const f32x4*)&bufB[(32 * nt + j) * ST + 8 * q + 4 * hi];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
              for (int nt = 0; nt < MT1; ++nt) MJX_MFMA_ACC(gW2[mt][nt], dl2u[mt][4 * q + t], bcs[q][nt][t]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      MJX_STAMP(9);
      // ---------------- VE: delta1 *= 1 - h1^2 (h1^T values of R9's fetches) ; grad b2[32 nt + j] += sum over the tile's samples
      // (16 registers x 2 lane halves; the halves are added after the tile loop)
#pragma unroll
      for (int nt = 0; nt < MT1; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t = 0; t < 4; t += 2) {
            const f32x2 hh = {bcs[q][nt][t], bcs[q][nt][t + 1]};
            f32x2 dd = {dl1u[nt][4 * q + t], dl1u[nt][4 * q + t + 1]};
            dd *= __builtin_elementwise_fma(-hh, hh, (f32x2)(1.0f));
            dl1u[nt][4 * q + t] = dd.x; dl1u[nt][4 * q + t + 1] = dd.y;
          }
#pragma unroll
      for (int nt = 0; nt < MT2; ++nt) {
        f32x2 s2 = {dl2u[nt][0], dl2u[nt][1]};
#pragma unroll
        for (int r = 2; r < 16; r += 2) s2 += f32x2{dl2u[nt][r], dl2u[nt][r + 1]};
        sb2[nt] += s2.x + s2.y;
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- R10: gW1a[u1][f] += sum_s delta1[s][u1] x~a[s][f]   (column n = bias gradient)
      if constexpr (NPC != 0) {
        const float* xrow = &xT[(lane & 3) * ST + 4 * hi];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 bx[NFQ];
#pragma unroll
          for (int fq = 0; fq < NFQ; ++fq) bx[fq] = *(const f32x4*)(xrow + 4 * fq * ST + 8 * q);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int fq = 0; fq < NFQ; ++fq)
#pragma unroll
              for (int mt = 0; mt < MT1; ++mt)
                MJX_MFMA4_ACC(gW1q[mt][fq], dl1u[mt][4 * q + t], bx[fq][t]);
        }
      } else {
        f32x4 bc[NT1], bn[NT1];
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) { int f = 32 * nt + j; bc[nt] = *(const f32x4*)&xT[(f < NP ? f : 0) * ST + 4 * hi]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (q + 1 < 4) {
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt) { int f = 32 * nt + j; bn[nt] = *(const f32x4*)&xT[(f < NP ? f : 0) * ST + 8 * (q + 1) + 4 * hi]; }
          }
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt) {
            f32x4 b4 = (32 * nt + j < NP) ? bc[nt] : (f32x4)(0.f);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int mt = 0; mt < MT1; ++mt) gW1[mt][nt] = MJX_MFMA(dl1u[mt][4 * q + t], b4[t], gW1[mt][nt]);
          }
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt) bc[nt] = bn[nt];
        }
      }
      MJX_STAMP(13);
      __builtin_amdgcn_sched_barrier(0);
    } else {
    // ---- 0. stage the tile's observations: xs is the raw memory image (sample-major, row stride n),
    // written with the same float4 granules it was fetched in.  Rows past the batch end are masked when read.
    // this tile's actions / advantage: requested now, consumed by the likelihood head two layers later
    float actr[MP], advr = 0.f;
    if (MODE != MODE_FVP) {
      if (s0 + 32 + MP <= A.N) {
        // every tile but the batch's last one or two: the lane's MP values from its row on (a >= m: the next row's, dropped below) at
        // scalar tile base + pinned lane offset + immediate -- no address arithmetic, no selects
        gcptr ab = ubase(A.act + s0 * m);
        gcptr db = ubase(A.adv + s0);
#pragma unroll
        for (int a = 0; a < MP; ++a) actr[a] = *(const float __attribute__((address_space(1)))*)(ab + aoff + 4 * a);
        advr = *(const float __attribute__((address_space(1)))*)(db + (uint32_t)(j * 4));
      } else {
#pragma unroll
        for (int a = 0; a < MP; ++a) actr[a] = A.act[(valid && a < m) ? (s0 + j) * m + a : 0];
        advr = A.adv[valid ? s0 + j : 0];
      }
    }
    if constexpr (DIRX) {
      // the next tile's raw pairs: in flight under this whole tile, normalised behind its last MFMA phase
      const int64_t tn = tile + tstride;
      if (tn < ntiles) {
        if ((tn + 1) * 32 > A.N) set_xoff(xo, (tn * 32 + j < A.N) ? (uint32_t)j : 0u);
        load_xw(xw, xo, tn);
      }
      if (A.hcache) {                               // the operand image for the Fisher-vector products / K3: one batch of stores
        gwptr xb = (gwptr)ubase(A.hcache + tile * HC_TILE + HC_H);
#pragma unroll
        for (int q = 0; q < NQX; ++q) *(f32x2 __attribute__((address_space(1)))*)(xb + (uint32_t)(lane * 8) + q * 512) = xn[q];
      }
    }
    if (!DIRX && !XCACHED && !XI) {
#pragma unroll
      for (int c = 0; c < XL4; ++c) {
        const int e4 = c * 64 + lane;
        *(f32x4*)&xs[4 * ((e4 < 8 * n) ? e4 : 8 * n + lane)] = xr[c];   // out-of-range lanes hit the slack area
      }
      if (tile + tstride < ntiles) load_x(tile + tstride);
      wave_sync();
    }

    // ---- layers 1 and 2 of one parameter set (and, for the FVP, the tangent pass riding on the
    // same operand fetches).  Every MFMA loop prefetches the next LDS operand group before issuing
    // the current group's MFMAs, so ds_read latency hides under the matrix pipe.
    //   TAN == false: h1 = tanh(W1a x~), h2 = tanh(W2 h1 + b2)                      (slot)
    //   TAN == true : additionally t1 = (V1a x~)(1-h1^2), t2 = (V2 h1 + W2 t1 + c2)(1-h2^2) (slotB)
    auto layers12 = [&](auto tan_tag, const float* slot, const float* tsh, const float* tsc, bool writeT,
                        f32x16 (&h1)[MT1], f32x16 (&h2)[MT2], f32x16 (&t1)[MT1], f32x16 (&t2)[MT2], const f32x2 (&xdir)[NQX]) {
      constexpr bool TAN = decltype(tan_tag)::value;
      constexpr bool FWD = !(TAN && CACHED);          // cached FVP: h1 / h2 arrive from HBM, only the tangent products run
      f32x16 z1[MT1];
#pragma unroll
      for (int mt = 0; mt < MT1; ++mt) { z1[mt] = (f32x16)(0.f); if (TAN) t1[mt] = (f32x16)(0.f); }
      // layer 1: K = features (+ ones column carrying the bias), operands by ds_read_b64
      {
        f32x2 wc[MT1], vc[MT1];
        const int f00 = 2 * hi;
        // x~[j][f] = (x - shift)/(scale + 1e-8) for f < n (fc_network.py:46); the bias column f == n reads 1, the
        // pad reads 0; rows past the batch end read 0.  Computed one group ahead, so the divide overlaps the MFMAs.
        auto xnorm = [&](int f) {
          // (f >= n: shift and reciprocal scale are 0, the value read -- a neighbouring row's / the zeroed slack -- drops out)
          const float v = (xs[j * n + f] - tsh[f]) * tsc[f];
          return (f == n) ? 1.0f : (valid ? v : 0.0f);
        };
        const float* ximg = A.hcache + tile * HC_TILE + HC_H + lane * 2;
        // group q's pair of features for this lane: computed, or (cached FVP) read back as K1 stored it
        auto xpair = [&](int q) {
          if constexpr (TAN && CACHED) {
            if constexpr (NPC != 0) return xc[q]; else return *(const f32x2*)(ximg + q * 128);
          } else if constexpr (XI) {
            return xc[q];
          } else if constexpr (DIRX) {
            return xdir[q];
          } else {
            const int f = 4 * q + 2 * hi;
            return f32x2{xnorm(f), xnorm(f + 1)};
          }
        };
        f32x2 xb = xpair(0);
        float xb0 = xb.x, xb1 = xb.y;
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt) {
          if (FWD) wc[mt] = *(const f32x2*)&slot[L.oW1 + (32 * mt + j) * S1 + f00];
          if (TAN) vc[mt] = *(const f32x2*)&slotB[L.oW1 + (32 * mt + j) * S1 + f00];
        }
#pragma unroll NPC ? NPC / 4 : 1
        for (int q = 0; q < (NPC ? NPC / 4 : NP / 4); ++q) {
          const int f0 = 4 * q + 2 * hi;
          const int f1 = (q + 1 < NP / 4) ? f0 + 4 : f0;        // next group (clamped on the last trip)
          f32x2 wn[MT1], vn[MT1];
          const f32x2 xnx = xpair((q + 1 < NP / 4) ? q + 1 : q);
          const float xn0 = xnx.x, xn1 = xnx.y;
          if (!DIRX && MODE == MODE_VPG && writeT && A.hcache) *(f32x2*)(A.hcache + tile * HC_TILE + HC_H + (q * 64 + lane) * 2) = f32x2{xb0, xb1};
#pragma unroll
          for (int mt = 0; mt < MT1; ++mt) {
            if (FWD) wn[mt] = *(const f32x2*)&slot[L.oW1 + (32 * mt + j) * S1 + f1];
            if (TAN) vn[mt] = *(const f32x2*)&slotB[L.oW1 + (32 * mt + j) * S1 + f1];
          }
          const float x0 = xb0, x1 = xb1;
          if (writeT) { xT[f0 * ST + j] = x0; xT[(f0 + 1) * ST + j] = x1; }
#pragma unroll
          for (int mt = 0; mt < MT1; ++mt) {
            if (FWD) z1[mt] = MJX_MFMA(wc[mt].x, x0, z1[mt]);
            if (TAN) t1[mt] = MJX_MFMA(vc[mt].x, x0, t1[mt]);
          }
#pragma unroll
          for (int mt = 0; mt < MT1; ++mt) {
            if (FWD) z1[mt] = MJX_MFMA(wc[mt].y, x1, z1[mt]);
            if (TAN) t1[mt] = MJX_MFMA(vc[mt].y, x1, t1[mt]);
          }
          xb0 = xn0; xb1 = xn1;
#pragma unroll
          for (int mt = 0; mt < MT1; ++mt) { if (FWD) wc[mt] = wn[mt]; if (TAN) vc[mt] = vn[mt]; }
        }
      }
      MJX_STAMP(2);
      if (XI && tile + tstride < ntiles) load_xi(tile + tstride);      // xc is dead: the next tile's image flies under the rest of this one
      // (forward-only passes: tanh as ONE burst between the layers' matrix instructions -- hipcc otherwise feeds layer 2 "just in
      //  time", a few exp / rcp between every four MFMAs, and on this chip a vector-ALU instruction in an MFMA gap costs 8 cycles
      //  on top of its own; with two waves per SIMD (K3) the burst runs at raised priority so that the other wave's MFMAs do not
      //  cut it into such gaps either)
      constexpr bool BURST = !TAN && !DBG;
      if constexpr (BURST) { __builtin_amdgcn_sched_barrier(0); if (EV2) __builtin_amdgcn_s_setprio(3); }
#pragma unroll
      for (int mt = 0; mt < MT1; ++mt) {
        if (FWD) fast_tanh16(h1[mt], z1[mt]);
        if (TAN) pk_mul_1mh2(t1[mt], h1[mt]);
      }
      if constexpr (BURST) { if (EV2) __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); }
      if (MODE == MODE_VPG && writeT && A.hcache) {
        // keep h1 for the Fisher-vector products of this update (theta is fixed during CG); issued here so that the
        // stores drain under the layer-2 MFMAs
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt) {
          gwptr base = (gwptr)ubase(A.hcache + tile * HC_TILE + mt * 1024);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *(f32x4 __attribute__((address_space(1)))*)(base + (uint32_t)(lane * 16) + q * 1024) = f32x4{h1[mt][4 * q], h1[mt][4 * q + 1], h1[mt][4 * q + 2], h1[mt][4 * q + 3]};
        }
      }
      MJX_STAMP(3);
      // layer 2: accumulators start at the bias (b2 / c2), K = h1 units chained from registers
      f32x16 z2[MT2];
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (FWD) {
            f32x4 b = *(const f32x4*)&slot[L.oB2 + 32 * mt + 8 * q + 4 * hi];
            z2[mt][4 * q + 0] = b.x; z2[mt][4 * q + 1] = b.y; z2[mt][4 * q + 2] = b.z; z2[mt][4 * q + 3] = b.w;
          }
          if (TAN) {
            f32x4 c = *(const f32x4*)&slotB[L.oB2 + 32 * mt + 8 * q + 4 * hi];
            t2[mt][4 * q + 0] = c.x; t2[mt][4 * q + 1] = c.y; t2[mt][4 * q + 2] = c.z; t2[mt][4 * q + 3] = c.w;
          }
        }
      constexpr int NS = MT1 * 4;                               // (kb, q) operand groups
      {
        // pass A: z2 += W2 h1   and (TAN)  t2 += W2 t1   -- one W2 fragment feeds both
        f32x4 wc[MT2], wn[MT2];
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) wc[mt] = *(const f32x4*)&slot[L.oW2 + (32 * mt + j) * S2 + 4 * hi];
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          const int kb = st >> 2, q = st & 3;
          if (st + 1 < NS) {
            const int kb1 = (st + 1) >> 2, q1 = (st + 1) & 3;
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) wn[mt] = *(const f32x4*)&slot[L.oW2 + (32 * mt + j) * S2 + 32 * kb1 + 8 * q1 + 4 * hi];
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
              if (FWD) z2[mt] = MJX_MFMA(wc[mt][t], h1[kb][4 * q + t], z2[mt]);
              if (TAN) t2[mt] = MJX_MFMA(wc[mt][t], t1[kb][4 * q + t], t2[mt]);
            }
#pragma unroll
          for (int mt = 0; mt < MT2; ++mt) wc[mt] = wn[mt];
        }
      }
      MJX_STAMP(4);
      if (TAN) {
        // pass B: t2 += V2 h1.  z2 is final after pass A, so tanh(z2), and the [unit][sample] copies of h2 / h1
        // the backward pass needs (bufA / bufB), are issued step by step in the shadow of these MFMAs.
        __builtin_amdgcn_sched_barrier(0);
        f32x4 vc[MT2], vn[MT2];
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) vc[mt] = *(const f32x4*)&slotB[L.oW2 + (32 * mt + j) * S2 + 4 * hi];
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          const int kb = st >> 2, q = st & 3;
          if (st + 1 < NS) {
            const int kb1 = (st + 1) >> 2, q1 = (st + 1) & 3;
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) vn[mt] = *(const f32x4*)&slotB[L.oW2 + (32 * mt + j) * S2 + 32 * kb1 + 8 * q1 + 4 * hi];
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) t2[mt] = MJX_MFMA(vc[mt][t], h1[kb][4 * q + t], t2[mt]);
          // this step's share of the VALU / LDS-store work: 16*MT2/NS registers of z2 and 16*MT1/NS of h1
          constexpr int R2 = 16 * MT2 / NS, R1 = 16 * MT1 / NS;
#pragma unroll
          for (int e = 0; e < R2; ++e) {
            const int idx = st * R2 + e, mt = idx >> 4, r = idx & 15;
            if (FWD) h2[mt][r] = fast_tanh(z2[mt][r]);
            if constexpr (PINNED) LDS_AT(pinA[4 * mt + (r >> 2)])[(r & 3) * ST] = h2[mt][r];
            else bufA[(32 * mt + unit_of(r, hi)) * ST + j] = h2[mt][r];
          }
#pragma unroll
          for (int e = 0; e < R1; ++e) {
            const int idx = st * R1 + e, mt = idx >> 4, r = idx & 15;
            if constexpr (PINNED) LDS_AT(pinB[4 * mt + (r >> 2)])[(r & 3) * ST] = h1[mt][r];
            else bufB[(32 * mt + unit_of(r, hi)) * ST + j] = h1[mt][r];
          }
#pragma unroll
          for (int mt = 0; mt < MT2; ++mt) vc[mt] = vn[mt];
        }
        // one wave per SIMD issues in order: interleave, per MFMA, a slice of the tanh / LDS-store work
        __builtin_amdgcn_sched_group_barrier(0x100, MT2, 0);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          if (st + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, MT2, 0);
#pragma unroll
          for (int i = 0; i < 4 * MT2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) t2[mt][r] *= fmaf(-h2[mt][r], h2[mt][r], 1.0f);
      } else {
        if constexpr (BURST) { __builtin_amdgcn_sched_barrier(0); if (EV2) __builtin_amdgcn_s_setprio(3); }
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) fast_tanh16(h2[mt], z2[mt]);
        if constexpr (BURST) { if (EV2) __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); }
      }
    };

    // Output layer (M = #actions is tiny) on v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer-product
    // blocks per instruction; lane l = (block l/4, column l%4) gets D[r][l%4] += A[4*(l/4)+r] * B[l].
    // Here block = the lane's sample quad, B = the lane's own activation register (unit k of its half),
    // A = W[a = 4*grp + r][k]: each lane accumulates og[grp][r] = partial out[a = 4*grp + r] of ITS
    // sample over the 32 units its half owns.  8 cycles per instruction instead of padding M to 32.
    // The A values are the same for all 8 blocks of a lane half, so they are fetched ONCE per tile: block b
    // of a half keeps the fragment of k-step group (mt, q) = (b >> 2, b & 3) and the instruction's A-broadcast
    // control (cbsz = 3, abid = b) hands it to the other 7 blocks -- 1 ds_read_b128 per (matrix, grp) and
    // tile instead of one per 4 MFMAs.
    constexpr int NGRP = MP / 4;
    auto out_frag = [&](f32x4 (&w)[NGRP], const float* slot) {
      const int b = (lane >> 2) & 7;
      const float* wbase = &slot[L.oW3 + (lane & 3) * S3 + 32 * ((b >> 2) % MT2) + 8 * (b & 3) + 4 * hi];
#pragma unroll
      for (int gp = 0; gp < NGRP; ++gp) w[gp] = *(const f32x4*)(wbase + 4 * gp * S3);
    };
    // (r06: two accumulator chains per action group where there are fewer than four groups -- with NGRP = 2 an accumulator
    //  recurred every second instruction and every dependent 4x4x1 waited for its producer: the cached product's R4 rule)
    auto out_small = [&](f32x4 (&og)[NGRP], const float* slot, const f32x16 (&v)[MT2]) {
      constexpr int CH = (NGRP >= 4) ? 1 : 2;
      f32x4 w[NGRP], o2[NGRP];
      out_frag(w, slot);
#pragma unroll
      for (int gp = 0; gp < NGRP; ++gp) o2[gp] = (f32x4)(0.f);
      static_for<MT2 * 4>([&](auto st) {
        constexpr int mt = decltype(st)::value >> 2, q = decltype(st)::value & 3;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int gp = 0; gp < NGRP; ++gp) {
            if (CH == 2 && (t & 1)) o2[gp] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[gp][t], v[mt][4 * q + t], o2[gp], 3, mt * 4 + q, 0);
            else og[gp] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[gp][t], v[mt][4 * q + t], og[gp], 3, mt * 4 + q, 0);
          }
      });
      if (CH == 2) {
#pragma unroll
        for (int gp = 0; gp < NGRP; ++gp) og[gp] += o2[gp];
      }
    };
    // the same with two independent accumulator sets (out = Wa va + Wb vb): twice the distance between
    // dependent 4x4x1 MFMAs, which otherwise stall on their own 8-cycle predecessors
    auto out_small2 = [&](f32x4 (&oa)[NGRP], f32x4 (&ob)[NGRP], const float* sa, const f32x16 (&va)[MT2],
                          const float* sb, const f32x16 (&vb)[MT2]) {
      f32x4 wa[NGRP], wb[NGRP];
      out_frag(wa, sa);
      out_frag(wb, sb);
      static_for<MT2 * 4>([&](auto st) {
        constexpr int mt = decltype(st)::value >> 2, q = decltype(st)::value & 3;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int gp = 0; gp < NGRP; ++gp) {
            oa[gp] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[gp][t], va[mt][4 * q + t], oa[gp], 3, mt * 4 + q, 0);
            ob[gp] = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[gp][t], vb[mt][4 * q + t], ob[gp], 3, mt * 4 + q, 0);
          }
      });
    };
    // both lane halves end up with the full sums for all MP actions
    auto out_finish = [&](f32x4 (&og)[NGRP], float (&o)[MP]) {
#pragma unroll
      for (int gp = 0; gp < NGRP; ++gp)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[4 * gp + r] = half_sum(og[gp][r]);
    };

    MJX_STAMP(1);
    f32x16 h1l[MT1], h2l[MT2];
    f32x16 (&h1)[MT1] = XCACHED ? hn1 : h1l;
    f32x16 (&h2)[MT2] = XCACHED ? hn2 : h2l;
    f32x16 t1[MT1], t2[MT2];                        // tangent activations (FVP only)
    if (MODE == MODE_FVP) layers12(std::true_type{}, slotA, trs, trs + NP, true, h1, h2, t1, t2, xn);
    else layers12(std::false_type{}, slotA, trs, trs + NP, MODE != MODE_EVAL, h1, h2, t1, t2, xn);
    if (MODE == MODE_VPG && A.hcache) {
      // ... and h2 (h1 was stored right after its tanh)
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt) {
        gwptr base = (gwptr)ubase(A.hcache + tile * HC_TILE + (MT1 + mt) * 1024);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(f32x4 __attribute__((address_space(1)))*)(base + (uint32_t)(lane * 16) + q * 1024) = f32x4{h2[mt][4 * q], h2[mt][4 * q + 1], h2[mt][4 * q + 2], h2[mt][4 * q + 3]};
      }
    }

    MJX_STAMP(6);
    float d3r[RA];                                  // cotangent on the pre-scale output, rows a = unit_of(r, hi)
    if (MODE == MODE_FVP) {
      f32x4 og[NGRP];
#pragma unroll
      for (int gp = 0; gp < NGRP; ++gp) og[gp] = (f32x4)(0.f);
      {
        f32x4 og2[NGRP];
#pragma unroll
        for (int gp = 0; gp < NGRP; ++gp) og2[gp] = (f32x4)(0.f);
        out_small2(og, og2, slotB, h2, slotA, t2);  // V3 h2 + W3 t2
#pragma unroll
        for (int gp = 0; gp < NGRP; ++gp) og[gp] += og2[gp];
      }
      MJX_STAMP(7);
      float md[MP];
      if (DBG) out_finish(og, md);
      if (DBG && A.dbg && blockIdx.x == 0 && wave == 0 && tile == 0) {
        float* g = A.dbg;
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            g[(32 * mt + unit_of(r, hi)) * 32 + j] = h1[mt][r];
            g[2048 * 2 + (32 * mt + unit_of(r, hi)) * 32 + j] = t1[mt][r];
          }
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            g[2048 + (32 * mt + unit_of(r, hi)) * 32 + j] = h2[mt][r];
            g[2048 * 3 + (32 * mt + unit_of(r, hi)) * 32 + j] = t2[mt][r];
          }
      }
      if (DBG) {
#pragma unroll
        for (int a = 0; a < MP; ++a)
          if (A.dbg && blockIdx.x == 0 && wave == 0 && tile == 0 && hi == 0) A.dbg[2048 * 4 + a * 32 + j] = cst[C_OSC * MP + a] * (md[a] + kc3[a]);
      }
      // Each lane half needs the full sums of ITS actions only (a = unit_of(r, hi): group 2(r>>2) + hi, column r&3):
      // v_permlane32_swap of the two groups' partial sums hands the lower half both halves' values of group 2(r>>2)
      // and the upper half both of group 2(r>>2)+1 -- one swap and one add per owned action, no selects.
      const float vmask = valid ? 1.0f : 0.0f;
#pragma unroll
      for (int r = 0; r < RA; ++r) {
        const int g0 = 2 * (r >> 2), c = r & 3;
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(og[g0][c]), __float_as_uint(og[g0 + 1][c]), false, false);
        const float mdr = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        d3r[r] = (mdr + kc3r[r]) * (kcsr[r] * vmask);               // d3 = (md + c3) * out_scale^2 Dk / N, 0 past the batch end
      }
    } else {
      // ---- likelihoods (mean_LL, gaussian_mlp.py:99-115); each lane owns the actions a = unit_of(r, hi)
      const float llc = 0.5f * (float)m * 1.8378770664093453f;
      // K1 (one wave per SIMD): the head's per-action constants are requested BEFORE the output layer's matrix instructions and
      // tanh(z2) is finished before them -- a vector-ALU instruction between two MFMAs costs 8 cycles on top of its own, and a
      // ds_read consumed right behind its issue is a full LDS round trip (r06: this phase took 2 556 cycles for 512 of MFMA
      // and ~110 vector-ALU instructions).  K3 keeps the reads at their use: two waves per SIMD, 231 of 256 registers.
      constexpr bool HK = (MODE == MODE_VPG) && !DBG;
      f32x4 kb3[HK ? NGRP : 1], kosc[HK ? NGRP : 1], kosh[HK ? NGRP : 1], kisg[HK ? NGRP : 1];
      if constexpr (HK) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gp = 0; gp < NGRP; ++gp) {
          kb3[gp] = *(const f32x4*)&slotA[L.oB3 + 4 * gp]; kosc[gp] = *(const f32x4*)&cst[C_OSC * MP + 4 * gp];
          kosh[gp] = *(const f32x4*)&cst[C_OSH * MP + 4 * gp]; kisg[gp] = *(const f32x4*)&cst[C_ISG * MP + 4 * gp];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      f32x4 og[NGRP];
#pragma unroll
      for (int gp = 0; gp < NGRP; ++gp) og[gp] = (f32x4)(0.f);
      out_small(og, slotA, h2);
      if constexpr (HK) __builtin_amdgcn_sched_barrier(0);
      float oa[MP];
      out_finish(og, oa);
      float z[MP], muv[MP], av[MP];
      float llA = 0.f;
#pragma unroll
      for (int a = 0; a < MP; ++a) {
        av[a] = actr[a];                              // (a >= m: 1 / sigma reads 0; rows past the batch end: every sum below is masked)
        if constexpr (HK) {
          muv[a] = (oa[a] + kb3[a >> 2][a & 3]) * kosc[a >> 2][a & 3] + kosh[a >> 2][a & 3];
          z[a] = (av[a] - muv[a]) * kisg[a >> 2][a & 3];
        } else {
          muv[a] = (oa[a] + slotA[L.oB3 + a]) * cst[C_OSC * MP + a] + cst[C_OSH * MP + a];
          z[a] = (av[a] - muv[a]) * cst[C_ISG * MP + a];
        }
        llA = fmaf(-0.5f * z[a], z[a], llA);
      }
      llA = llA - sum_lsA - llc;
      float llB = llA, muB[MP];
#pragma unroll
      for (int a = 0; a < MP; ++a) muB[a] = muv[a];
      if (MODE == MODE_VPG && A.ocache && A.old_is_new && hi == 0) {
        float* oc = A.ocache + tile * OC_TILE + j;
#pragma unroll
        for (int a = 0; a < MP; ++a) oc[a * 32] = muv[a];
        oc[MP * 32] = llA;
      }
      if (MODE == MODE_EVAL && use_oc) {
#pragma unroll
        for (int a = 0; a < MP; ++a) muB[a] = ocn[a];
        llB = ocn[MP];
        if (tile + tstride < ntiles) load_oc(tile + tstride);
      } else if (MODE == MODE_EVAL || !A.old_is_new) {
        f32x16 g1[MT1], g2[MT2];
        f32x2 xnB[NQX];
        if constexpr (DIRX) {                       // the old network's input transform on the same rows (fetched again: L2)
          uint32_t xoB[NQX + 1];
          f32x2 xwB[NQX];
          set_xoff(xoB, valid ? (uint32_t)j : 0u);
          load_xw(xwB, xoB, tile);
          norm_xw(xnB, xwB, trs + 2 * NP, trs + 3 * NP, tile);
        }
        layers12(std::false_type{}, slotB, trs + 2 * NP, trs + 3 * NP, false, g1, g2, t1, t2, xnB);
#pragma unroll
        for (int gp = 0; gp < NGRP; ++gp) og[gp] = (f32x4)(0.f);
        out_small(og, slotB, g2);
        float ob[MP];
        out_finish(og, ob);
        llB = 0.f;
#pragma unroll
        for (int a = 0; a < MP; ++a) {
          muB[a] = (ob[a] + slotB[L.oB3 + a]) * cst[C_OSCB * MP + a] + cst[C_OSHB * MP + a];
          float zb = (av[a] - muB[a]) * cst[C_ISGB * MP + a];
          llB = fmaf(-0.5f * zb, zb, llB);
        }
        llB = llB - sum_lsB - llc;
      }
      const float advv = valid ? advr : 0.f;
      // (old == new in K1: llB IS llA, the ratio is exp(0) = 1 exactly -- a uniform branch instead of the exponential's ~12 instructions)
      float LR = (MODE == MODE_VPG && A.old_is_new) ? 1.0f : expf(llA - llB);
      if (valid && hi == 0) { s_surr += (double)(LR * advv); s_cnt += 1.0; }
      if (MODE == MODE_EVAL) {
        // mean_kl(new, old), gaussian_mlp.py:135-145
        float kl = 0.f;
#pragma unroll
        for (int a = 0; a < MP; ++a) {
          const float dm = muB[a] - muv[a];
          const float Nr = (dm * dm + cst[C_SO2 * MP + a]) - cst[C_SN2 * MP + a];
          kl += Nr * cst[C_RD * MP + a] + cst[C_KD * MP + a];
        }
        if (valid && hi == 0) s_kl += (double)kl;
      } else {
        float w = valid ? advv * LR * A.inv_N : 0.f;
        float d3a[MP];
#pragma unroll
        for (int a = 0; a < MP; ++a) {
          if constexpr (HK) d3a[a] = kosc[a >> 2][a & 3] * ((w * z[a]) * kisg[a >> 2][a & 3]);
          else d3a[a] = cst[C_OSC * MP + a] * ((w * z[a]) * cst[C_ISG * MP + a]);
        }
#pragma unroll
        for (int r = 0; r < RA; ++r) {
          // (bit-select on the lane-half mask: one v_bfi each; a ?: on array elements becomes a dynamically indexed
          //  register array, i.e. a chain of 8 compares and selects per value)
          d3r[r] = half_select(himask, d3a[unit_of(r, 1)], d3a[unit_of(r, 0)]);
          const float zr = half_select(himask, z[unit_of(r, 1)], z[unit_of(r, 0)]);
          gls[r] += w * (zr * zr - 1.0f);
        }
        if (DBG && A.dbg && blockIdx.x == 0 && wave == 0 && tile == 0 && hi == 0) {
#pragma unroll
          for (int a = 0; a < MP; ++a) A.dbg[2048 * 4 + a * 32 + j] = muv[a];
          A.dbg[2048 * 4 + MP * 32 + j] = llA;
        }
      }
    }

    __builtin_amdgcn_sched_barrier(0);
    if (MODE != MODE_EVAL) {
      // ================= backward (shared by VPG and FVP) =================
      // Park h2^T in bufA, h1^T in bufB ([unit][sample]) and d3^T in d3T ([action][sample]).  The deltas
      // are produced directly in the layout each consumer wants (swapping the MFMA operand roles
      // transposes the product), so they never pass through LDS:
      //   delta2s / (lane = sample)  feeds the next back-propagation step as a chained operand,
      //   delta2u, delta1u (lane = unit) feed the weight-gradient products as A operands.
#pragma unroll
      for (int r = 0; r < RA; ++r) d3T[unit_of(r, hi) * ST + j] = d3r[r];
      if (MODE != MODE_FVP) {                        // (the FVP's pass B already parked them)
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if constexpr (PINNED) LDS_AT(pinA[4 * mt + (r >> 2)])[(r & 3) * ST] = h2[mt][r];
            else bufA[(32 * mt + unit_of(r, hi)) * ST + j] = h2[mt][r];
          }
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if constexpr (PINNED) LDS_AT(pinB[4 * mt + (r >> 2)])[(r & 3) * ST] = h1[mt][r];
            else bufB[(32 * mt + unit_of(r, hi)) * ST + j] = h1[mt][r];
          }
      }
      MJX_STAMP(8);
      // delta2 in both layouts: K = actions, the W3 column fragment serves as A (-> lane = sample) and as B (-> lane = unit)
      // (K1, r06: ONE layout like the cached product -- the lane = unit copy the gW2 product needs takes a trip through LDS in
      //  the shadow of the delta1 phase, into bufA where h2^T is dead by then, and arrives with its (1 - h2^2) factor applied:
      //  8 matrix instructions, 32 packed factor instructions and 8 fragment reads less per tile)
      constexpr bool T2L = (MODE == MODE_VPG) && PINNED;
      f32x16 dl2s[MT2], dl2u[MT2];
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt) { dl2s[mt] = (f32x16)(0.f); dl2u[mt] = (f32x16)(0.f); }
#pragma unroll
      for (int sidx = 0; sidx < RA; ++sidx) {
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) {
          float w = slotA[L.oW3 + unit_of(sidx, hi) * S3 + 32 * mt + j];
          dl2s[mt] = MJX_MFMA(w, d3r[sidx], dl2s[mt]);
          if (!T2L) dl2u[mt] = MJX_MFMA(d3r[sidx], w, dl2u[mt]);
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) { const f32x2 v = mul_1mh2_pair(dl2s[mt][r], dl2s[mt][r + 1], h2[mt][r], h2[mt][r + 1]); dl2s[mt][r] = v.x; dl2s[mt][r + 1] = v.y; }
      // h1 / h2 are dead from here on: fetch the next tile's copies behind the weight-gradient products
      if (MODE == MODE_FVP && CACHED && tile + tstride < ntiles) load_h(tile + tstride);
      wave_sync();
      MJX_STAMP(9);
      // gW3[a][k] += sum_s d3[s][a] * h2[s][k] on v_mfma_f32_4x4x1_16b_f32: one instruction = 16 independent 4 x 4 outer
      // products (8 cycles), here one SAMPLE's contribution to (MP / 4 action groups) x (QPI3 unit quads) blocks -- no
      // padding of the action dimension (6 -> 8 instead of 6 -> 16 on a 16x16x4 tile: half the matrix-pipe time).
      // Operands straight from the [action][sample] / [unit][sample] LDS copies: one ds_read_b128 = 4 samples = 4 steps.
      {
        const int blk = lane >> 2, g3 = blk / QPI3, uq3 = blk % QPI3;
        const float* arow = &d3T[(4 * g3 + (lane & 3)) * ST];
        const float* brow = &bufA[(4 * uq3 + (lane & 3)) * ST];
        // delta2u *= (1 - h2^2): h2[sample unit_of(4q+t, hi)][unit 32nt + j] from the [unit][sample] copy
        f32x4 fc[MT2][4];
        if constexpr (!T2L) {
#pragma unroll
        for (int nt = 0; nt < MT2; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) fc[nt][q] = *(const f32x4*)&bufA[(32 * nt + j) * ST + 8 * q + 4 * hi];
        }
        if constexpr (G3B) {
#pragma unroll
          for (int sp = 0; sp < 4; ++sp) {                 // sample quads 2 sp (-> gW3) and 2 sp + 1 (-> gW3b), instruction by instruction
            const f32x4 ave = *(const f32x4*)(arow + 8 * sp), avo = *(const f32x4*)(arow + 8 * sp + 4);
            f32x4 bve[NT3], bvo[NT3];
#pragma unroll
            for (int nt = 0; nt < NT3; ++nt) { bve[nt] = *(const f32x4*)(brow + UPI3 * nt * ST + 8 * sp); bvo[nt] = *(const f32x4*)(brow + UPI3 * nt * ST + 8 * sp + 4); }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int nt = 0; nt < NT3; ++nt) {
                gW3[nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(ave[t], bve[nt][t], gW3[nt], 0, 0, 0);
                gW3b[G3B ? nt : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(avo[t], bvo[nt][t], gW3b[G3B ? nt : 0], 0, 0, 0);
              }
          }
        } else {
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
          const f32x4 av = *(const f32x4*)(arow + 4 * s4);
          f32x4 bv[NT3];
#pragma unroll
          for (int nt = 0; nt < NT3; ++nt) bv[nt] = *(const f32x4*)(brow + UPI3 * nt * ST + 4 * s4);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < NT3; ++nt) gW3[nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[t], bv[nt][t], gW3[nt], 0, 0, 0);
        }
        }
        if constexpr (!T2L) {
#pragma unroll
        for (int nt = 0; nt < MT2; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < 4; t += 2) {
              const f32x2 v = mul_1mh2_pair(dl2u[nt][4 * q + t], dl2u[nt][4 * q + t + 1], fc[nt][q][t], fc[nt][q][t + 1]);
              dl2u[nt][4 * q + t] = v.x; dl2u[nt][4 * q + t + 1] = v.y;
            }
        }
      }
#pragma unroll
      for (int r = 0; r < RA; ++r) sb3r[r] += d3r[r];   // grad b3[a] = sum_s d3[s][a]: per-lane partial sums, reduced over the lanes after the tile loop
      // grad b2[32nt + j] = sum over this tile's samples (16 registers x 2 lane halves)
      if constexpr (!T2L) {
#pragma unroll
      for (int nt = 0; nt < MT2; ++nt) {
        float sacc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += dl2u[nt][r];
        sb2[nt] += sacc;                            // the two lane halves are summed once, after the tile loop
      }
      }
      MJX_STAMP(10);
      // gW2[u2][u1] += sum_s delta2[s][u2] * h1[s][u1]; delta1u = (delta2 W2)(1 - h1^2), lane = h1 unit.
      // Both walk h1^T (bufB) group by group, so the (1 - h1^2) factors ride on the gW2 operand fetches.
      f32x16 dl1u[MT1];
#pragma unroll
      for (int nt = 0; nt < MT1; ++nt) dl1u[nt] = (f32x16)(0.f);
      {
        constexpr int NG = MT2 * 4;                 // groups of 4 k-steps over the h2 units
        float wc[4][MT1], wn[4][MT1];
        __builtin_amdgcn_sched_barrier(0);          // own scheduling region: pin "next group's reads, then this group's MFMAs"
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nt = 0; nt < MT1; ++nt) {
            if constexpr (PINNED) wc[t][nt] = LDS_AT(pinW[0])[t * S2 + 32 * nt];
            else wc[t][nt] = slotA[L.oW2 + (4 * hi + t) * S2 + 32 * nt + j];
          }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int kb = g >> 2, q = g & 3;
          if (g + 1 < NG) {
            const int kb1 = (g + 1) >> 2, q1 = (g + 1) & 3;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int nt = 0; nt < MT1; ++nt) {
                if constexpr (PINNED) wn[t][nt] = LDS_AT(pinW[4 * kb1 + q1])[t * S2 + 32 * nt];
                else wn[t][nt] = slotA[L.oW2 + (32 * kb1 + 8 * q1 + 4 * hi + t) * S2 + 32 * nt + j];
              }
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < MT1; ++nt) dl1u[nt] = MJX_MFMA(dl2s[kb][4 * q + t], wc[t][nt], dl1u[nt]);
          if constexpr (T2L) {
            constexpr int NGH = NG / 2;
            if (g < NGH) {
              // this group's share of the transposed copy: 16 MT2 / NGH registers of delta2 (LDS executes one wave's operations in order)
              constexpr int RS = 16 * MT2 / NGH;
#pragma unroll
              for (int e = 0; e < RS; ++e) {
                const int idx = g * RS + e, mt = idx >> 4, r = idx & 15;
                LDS_AT(pinA[4 * mt + (r >> 2)])[(r & 3) * ST] = dl2s[mt][r];
              }
            } else {
              // ... and of the read-back: delta2[sample unit_of(4 qq + t, hi)][unit 32 nt + j], one ds_read_b128 per (nt, qq)
              constexpr int RQ = (4 * MT2 + NGH - 1) / NGH;
#pragma unroll
              for (int e = 0; e < RQ; ++e) {
                const int idx = (g - NGH) * RQ + e;
                if (idx < 4 * MT2) {
                  const int nt = idx >> 2, qq = idx & 3;
                  const f32x4 v = *(const f32x4*)&bufA[(32 * nt + j) * ST + 8 * qq + 4 * hi];
                  dl2u[nt][4 * qq] = v.x; dl2u[nt][4 * qq + 1] = v.y; dl2u[nt][4 * qq + 2] = v.z; dl2u[nt][4 * qq + 3] = v.w;
                }
              }
            }
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < MT1; ++nt) wc[t][nt] = wn[t][nt];
        }
        // pipeline: prologue reads, then per group [reads of g+1][MFMAs of g]
        if constexpr (T2L) {
          constexpr int NGH = NG / 2;
          __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT1, 0);
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT1, 0);      // (the ds_read_b32 pair up as ds_read2_b32)
#pragma unroll
            for (int i = 0; i < 4 * MT1; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if (g < NGH) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
              else if (i < (4 * MT2 + NGH - 1) / NGH) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
          }
        } else {
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * MT1, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, 4 * MT1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT1, 0);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (T2L) {
          // grad b2[32 nt + j] += sum over the tile's samples (16 registers x 2 lane halves; the halves are added after the tile loop)
#pragma unroll
          for (int nt = 0; nt < MT2; ++nt) {
            f32x2 s2 = {dl2u[nt][0], dl2u[nt][1]};
#pragma unroll
            for (int r = 2; r < 16; r += 2) s2 += f32x2{dl2u[nt][r], dl2u[nt][r + 1]};
            sb2[nt] += s2.x + s2.y;
          }
        }
      }
      MJX_STAMP(11);
      {
        f32x4 bc[MT1], bn[MT1];
#pragma unroll
        for (int nt = 0; nt < MT1; ++nt) bc[nt] = *(const f32x4*)&bufB[(32 * nt + j) * ST + 4 * hi];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (q + 1 < 4) {
#pragma unroll
            for (int nt = 0; nt < MT1; ++nt) bn[nt] = *(const f32x4*)&bufB[(32 * nt + j) * ST + 8 * (q + 1) + 4 * hi];
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
              for (int nt = 0; nt < MT1; ++nt) MJX_MFMA_ACC(gW2[mt][nt], dl2u[mt][4 * q + t], bc[nt][t]);
#pragma unroll
          for (int nt = 0; nt < MT1; ++nt)
#pragma unroll
            for (int t = 0; t < 4; t += 2) {
              const f32x2 v = mul_1mh2_pair(dl1u[nt][4 * q + t], dl1u[nt][4 * q + t + 1], bc[nt][t], bc[nt][t + 1]);
              dl1u[nt][4 * q + t] = v.x; dl1u[nt][4 * q + t + 1] = v.y;
            }
#pragma unroll
          for (int nt = 0; nt < MT1; ++nt) bc[nt] = bn[nt];
        }
      }
      if (DBG && A.dbg && blockIdx.x == 0 && wave == 0 && tile == 0) {
        float* g = A.dbg + 2048 * 5;
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) g[(32 * mt + unit_of(r, hi)) * 32 + j] = dl2s[mt][r];
#pragma unroll
        for (int nt = 0; nt < MT1; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) g[2048 + (32 * nt + j) * 32 + unit_of(r, hi)] = dl1u[nt][r];
      }
      MJX_STAMP(12);
      // gW1a[u1][f] += sum_s delta1[s][u1] * x~a[s][f]   (column n = bias gradient); A from registers
      if constexpr (NPC != 0) {
        // 4x4x1 blocks: A = delta1 of the block's 4 units (lanes 4b .. 4b+3, this half's sample of register r),
        // B = x~ of that sample for the lane's feature 4 fq + (lane & 3); one ds_read_b128 of x~^T covers 4 registers
        const float* xrow = &xT[(lane & 3) * ST + 4 * hi];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 bx[NFQ];
#pragma unroll
          for (int fq = 0; fq < NFQ; ++fq) bx[fq] = *(const f32x4*)(xrow + 4 * fq * ST + 8 * q);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int fq = 0; fq < NFQ; ++fq)
#pragma unroll
              for (int mt = 0; mt < MT1; ++mt)
                MJX_MFMA4_ACC(gW1q[mt][fq], dl1u[mt][4 * q + t], bx[fq][t]);
        }
      } else {
        f32x4 bc[NT1], bn[NT1];
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) { int f = 32 * nt + j; bc[nt] = *(const f32x4*)&xT[(f < NP ? f : 0) * ST + 4 * hi]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (q + 1 < 4) {
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt) { int f = 32 * nt + j; bn[nt] = *(const f32x4*)&xT[(f < NP ? f : 0) * ST + 8 * (q + 1) + 4 * hi]; }
          }
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt) {
            f32x4 b4 = (32 * nt + j < NP) ? bc[nt] : (f32x4)(0.f);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int mt = 0; mt < MT1; ++mt) gW1[mt][nt] = MJX_MFMA(dl1u[mt][4 * q + t], b4[t], gW1[mt][nt]);
          }
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt) bc[nt] = bn[nt];
        }
      }
    }
    MJX_STAMP(13);
    if constexpr (DIRX) { if (tile + tstride < ntiles) norm_xw(xn, xw, trs, trs + NP, tile + tstride); }
    wave_sync();                                  // everything read before the next tile's staging
    }   // !XCACHED
  }
  };
  if constexpr (MODE == MODE_EVAL && NPC != 0) {
    if (use_xi) run_tiles(std::true_type{}); else run_tiles(std::false_type{});
  } else {
    run_tiles(std::false_type{});
  }

  MJX_GSTAMP(18);
  // ---------------- workgroup reduction + partial write ----------------
  __syncthreads();
  MJX_GSTAMP(19);
  if (MODE != MODE_EVAL) {
    // log_std gradient: reduce over the 32 samples (lanes j) within each half (the halves own different actions)
#pragma unroll
    for (int off = 1; off < 32; off <<= 1)
#pragma unroll
      for (int a = 0; a < RA; ++a) {
        if (MODE == MODE_VPG) gls[a] += __shfl_xor(gls[a], off);
        sb3r[a] += __shfl_xor(sb3r[a], off);
      }
    if (A.raw_dr > 0) {
      // accumulator-order partial (RawSlab above): one ds_write per register at [X][r][lane], immediates only
      using RS = RawSlab<H1, H2, NT1, MP, NPC>;
      float* red = lds;                           // 4 * DR floats <= TOTAL (checked on host)
      float* mine = red + wave * RS::DR + lane;
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int nt = 0; nt < MT1; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) mine[RS::oW2 + ((mt * MT1 + nt) * 16 + r) * 64] = gW2[mt][nt][r];
      if constexpr (NPC != 0) {
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
          for (int fq = 0; fq < NFQ; ++fq)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = half_sum(gW1q[mt][fq][r]);
              if (hi == 0) mine[RS::oW1 + ((mt * NFQ + fq) * 4 + r) * 32] = v;
            }
      } else {
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[RS::oW1 + ((mt * NT1 + nt) * 16 + r) * 64] = gW1[mt][nt][r];
      }
#pragma unroll
      for (int nt = 0; nt < NT3; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[RS::oW3 + (nt * 4 + r) * 64] = G3B ? gW3[nt][r] + gW3b[G3B ? nt : 0][r] : gW3[nt][r];
#pragma unroll
      for (int nt = 0; nt < MT2; ++nt) {
        const float v = half_sum(sb2[nt]);
        if (hi == 0) mine[RS::oB2 + nt * 32] = v;
      }
      if (j == 0) {
        float* w0 = red + wave * RS::DR;
#pragma unroll
        for (int r = 0; r < RA; ++r) {
          w0[RS::oB3 + 2 * r + hi] = sb3r[r];
          w0[RS::oS + 2 * r + hi] = (MODE == MODE_VPG) ? gls[r] : 0.f;
        }
      }
      if (lane < RS::DR - (RS::oS + 2 * RA)) red[wave * RS::DR + RS::oS + 2 * RA + lane] = 0.f;      // (the alignment pad)
      __syncthreads();
      float* outp = A.partials + (size_t)blockIdx.x * RS::DR;
      const f32x4* r4 = (const f32x4*)red;
      constexpr int d4 = RS::DR >> 2;
      for (int idx = tid; idx < d4; idx += 256)
        ((f32x4*)outp)[idx] = (r4[idx] + r4[d4 + idx]) + (r4[2 * d4 + idx] + r4[3 * d4 + idx]);
    } else {
    // each wave drops its partial gradient into its own LDS region [wave][d], then the
    // workgroup sums the four copies.  (weights in LDS are dead by now)
    float* red = lds;                             // 4 * d floats <= TOTAL (checked on host)
    float* mine = red + wave * fo.d;
    // (every entry of the wave's copy is written below: W1a/b1, W2, W3, b2, b3 and the log_std block)
    if constexpr (NPC != 0) {
#pragma unroll
      for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
        for (int fq = 0; fq < NFQ; ++fq)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = half_sum(gW1q[mt][fq][r]);
            const int u = 32 * mt + 4 * ((lane >> 2) & 7) + r, f = 4 * fq + (lane & 3);
            if (hi == 0) {
              if (f < n) mine[fo.W1 + u * n + f] = v;
              else if (f == n) mine[fo.b1 + u] = v;
            }
          }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int u = 32 * mt + unit_of(r, hi), f = 32 * nt + j;
            if (f < n) mine[fo.W1 + u * n + f] = gW1[mt][nt][r];
            else if (f == n) mine[fo.b1 + u] = gW1[mt][nt][r];
          }
    }
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
      for (int nt = 0; nt < MT1; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          mine[fo.W2 + (32 * mt + unit_of(r, hi)) * H1 + 32 * nt + j] = gW2[mt][nt][r];
#pragma unroll
    for (int nt = 0; nt < NT3; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int blk = lane >> 2, a = 4 * (blk / QPI3) + r, u = 4 * (blk % QPI3) + (lane & 3) + UPI3 * nt;
        if (a < m) mine[fo.W3 + a * H2 + u] = G3B ? gW3[nt][r] + gW3b[G3B ? nt : 0][r] : gW3[nt][r];
      }
    float sb2f[MT2];
#pragma unroll
    for (int nt = 0; nt < MT2; ++nt) sb2f[nt] = half_sum(sb2[nt]);
    if (hi == 0) {
#pragma unroll
      for (int nt = 0; nt < MT2; ++nt) mine[fo.b2 + 32 * nt + j] = sb2f[nt];
    }
    if (j == 0) {
#pragma unroll
      for (int r = 0; r < RA; ++r) {
        const int a = unit_of(r, hi);
        if (a < m) mine[fo.b3 + a] = sb3r[r];
        if (a < m) mine[fo.S + a] = (MODE == MODE_VPG) ? gls[r] : 0.f;
      }
    }
    __syncthreads();
    float* outp = A.partials + (size_t)blockIdx.x * fo.d;
    if ((fo.d & 3) == 0) {
      const f32x4* r4 = (const f32x4*)red;
      const int d4 = fo.d >> 2;
      for (int idx = tid; idx < d4; idx += 256)
        ((f32x4*)outp)[idx] = (r4[idx] + r4[d4 + idx]) + (r4[2 * d4 + idx] + r4[3 * d4 + idx]);
    } else {
      for (int idx = tid; idx < fo.d; idx += 256)
        outp[idx] = (red[idx] + red[fo.d + idx]) + (red[2 * fo.d + idx] + red[3 * fo.d + idx]);
    }
    }
  }
  MJX_GSTAMP(20);
  if (A.clk && blockIdx.x == 0 && threadIdx.x == 0) { A.clk[2] = (long long)__builtin_readcyclecounter(); A.clk[3] = (long long)__builtin_amdgcn_s_memrealtime(); }
  if (MODE != MODE_FVP) {
    // scalar partials: wave reduce (fp64) -> LDS -> one thread
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      s_surr += __shfl_xor(s_surr, off);
      s_kl += __shfl_xor(s_kl, off);
      s_cnt += __shfl_xor(s_cnt, off);
    }
    __syncthreads();
    double* sred = (double*)(lds + (EV2 ? 0 : 4 * (A.raw_dr > fo.d ? A.raw_dr : fo.d) + 4));     // (EV2: the weight slots are dead by now)
    sred = (double*)(((uintptr_t)sred + 7) & ~(uintptr_t)7);
    if (lane == 0) { sred[wave * 3 + 0] = s_surr; sred[wave * 3 + 1] = s_kl; sred[wave * 3 + 2] = s_cnt; }
    __syncthreads();
    if (tid == 0) {
      double a = 0, b = 0, c = 0;
      for (int w = 0; w < 4; ++w) { a += sred[w * 3]; b += sred[w * 3 + 1]; c += sred[w * 3 + 2]; }
      double* sp = A.spartials + (size_t)blockIdx.x * 4;
      sp[0] = a; sp[1] = (MODE == MODE_EVAL) ? b : c; sp[2] = c; sp[3] = 0.0;
    }
  }
}

}  // namespace mjx
