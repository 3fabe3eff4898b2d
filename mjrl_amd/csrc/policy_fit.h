// policy_fit.h -- persistent single-workgroup minibatch-Adam trainer for small tanh-MLP policies
// (two hidden layers of equal width H in {32, 64}, minibatches of up to 64 rows).
//
// The torch-optimizer loops of behaviour cloning and PPO (mjrl/algos/behavior_cloning.py:107-136,
// mjrl/algos/ppo_clip.py:85-95) are chains of tens of thousands of tiny dependent steps: a 64-row minibatch through a
// 5.7 k-parameter net is ~2 MFLOP.  As separate launches (gather, three GEMMs, loss head, five GEMMs, reductions, Adam)
// a step costs ~150 us of launches; here the whole chain runs inside ONE launch on one 1024-thread workgroup:
// parameters (in a padded compute layout), their gradients and every activation of the minibatch live in LDS, the Adam
// moments of a thread's parameters in its registers, and the phases of a step are separated by workgroup barriers only.
// The layer products run on the matrix cores (16 x 16 x 4 fp32 MFMA tiles straight out of LDS, one tile per wave); the
// loss head runs one thread per (row, action).  What bounds a step (~36 k cycles = 17 us at 64 rows x 64 x 64) is plain
// instruction issue -- 16 waves x ~2 k instructions over 4 SIMDs -- not the matrix pipe (~9 k cycles) and not LDS:
// sixteen waves hide each other's LDS / MFMA latencies (4 waves with the same code: 2x slower), every address that does
// not depend on the step is derived afresh from an opaque thread index (hoisted out of the step loop such values spilled
// to scratch), and the step body is one instance of each product walked over the layers (it has to fit the
// instruction cache).
//
// Same losses and the same torch.optim.Adam update as the launch-based path (k_minibatch_head / k_adam in layerwise.h,
// baseline.h), which remains the path for every other shape and batch size.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fused_policy.h"
#include "vecops.h"

namespace mjx {

#ifdef MJX_PFIT_CLOCK
#define PFIT_STAMP(k) do { if (tid == 0 && step == 3 && A.loss_trace) A.loss_trace[100 + (k)] = (double)__builtin_readcyclecounter(); } while (0)
#else
#define PFIT_STAMP(k) do {} while (0)
#endif

struct PolicyFitArgs {
  const float* obs; const float* act; const float* adv;      // batch (N, n), (N, m), (N)
  const int32_t* idx;                                         // [steps][B] row indices
  int64_t steps; int B, n, m;
  float* theta;                                               // in / out, flat reference order
  const float* theta_old;                                     // PPO
  const float* tr; const float* tr_old;                       // packed transforms (never null)
  int loss;                                                   // 0 MSE, 1 MLE, 2 PPO clip
  int old_tracks_new;                                         // PPO: old network evaluated with the current weights (see mjx.h)
  float* adam_m; float* adam_v; int64_t step0;
  float lr, clip;
  double* loss_trace;                                         // optional [steps]
  int lds_floats;                                             // size of the dynamic LDS block (zero-filled once: row pads)
};

// padded compute layout of the parameters in LDS (rows padded with zeros to a multiple of 16 floats + 4: whole MFMA
// contraction blocks, strides == 4 mod 32 so that the operand reads of different rows spread over the banks)
template <int H>
struct PolicyFitLayout {
  int n, m, S1, S2;                 // S1: row stride of W1 (n inputs), S2: row stride of W2 / W3 (H inputs)
  int oW1, ob1, oW2, ob2, oW3, ob3, oS, P;     // offsets in the padded parameter block, P = its size
  int XS;                           // row stride of the normalised observation block
  __host__ __device__ PolicyFitLayout(int n_, int m_) {
    n = n_; m = m_;
    S1 = ((n + 15) & ~15) + 4; S2 = H + 4;
    XS = S1;
    oW1 = 0; ob1 = oW1 + H * S1; oW2 = ob1 + H; ob2 = oW2 + H * S2; oW3 = ob2 + H; ob3 = oW3 + m * S2;
    oS = ob3 + ((m + 3) & ~3); P = oS + ((m + 3) & ~3);
  }
  // flat (reference order [W1, b1, W2, b2, W3, b3, log_std]) index -> padded offset
  __host__ __device__ int pad_of(int i) const {
    int k = i;
    if (k < H * n) return oW1 + (k / n) * S1 + k % n;
    k -= H * n; if (k < H) return ob1 + k;
    k -= H; if (k < H * H) return oW2 + (k / H) * S2 + k % H;
    k -= H * H; if (k < H) return ob2 + k;
    k -= H; if (k < m * H) return oW3 + (k / H) * S2 + k % H;
    k -= m * H; if (k < m) return ob3 + k;
    k -= m; return oS + k;
  }
  __host__ __device__ int d() const { return H * n + H + H * H + H + m * H + m + m; }
  // LDS floats: parameters + gradients (+ old parameters) + activations of B rows
  __host__ __device__ size_t lds_floats(int B, bool old_net) const {
    const int MS = 20;                                                       // row stride of the per-action blocks: 16 (one contraction block) + 4
    const int Bp = (B + 15) & ~15;                                           // sample blocks are padded with zero rows to whole MFMA tiles
    size_t act = (size_t)Bp * (XS + 4 * (size_t)S2 + 3 * (size_t)MS + 4);   // X, H1, H2, D1, D2, MU, MUo, D3, per-row scalars
    return (size_t)P * (old_net ? 3 : 2) + act + 512 + (size_t)B * MS + 128;   // + reduction scratch, transform table, this step's actions, row ids
  }
};

// The three products of a layer on v_mfma_f32_16x16x4_f32 (16 x 16 output tile, 4 contraction steps, 32 cycles): lane l
// supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15] and receives D[i = 4 (l >> 4) + r][j = l & 15].
// The four waves take the 16 x 16 output tiles round-robin, two at a time (two independent accumulators).  The
// contraction is walked 16 indices at a time -- k-slot q of step t carries index 16 kb + 4 q + t for both operands, i.e.
// 4 values per lane, operand and block: one 16-byte read where the contraction index is contiguous in LDS ("row"
// operand), four 4-byte reads down a column where it is the row index ("column" operand) -- with the next block's
// operands requested before the current block's MFMAs are issued (two register sets, no copies).
// Every operand array has its rows padded with zeros to a multiple of 16 floats (+ 4: row strides == 4 (mod 32) spread
// both read patterns over the banks) and the sample blocks are padded with zero rows to a multiple of 16, so partial
// tiles need no masks: whatever a tile reads past the real extent (zeros, or the neighbouring row / array: finite numbers)
// only reaches outputs that are not stored.
// ONE instance of each product serves all layers (the step loop walks the layers with runtime descriptors): the step body
// has to stay inside the instruction cache.
// an opaque copy of the thread index: everything the layer routines derive from it (lane roles, LDS addresses) is loop
// invariant, and hoisted out of the step loop it occupies hundreds of registers / scratch slots
__device__ __forceinline__ int fresh(int v) { asm volatile("" : "+v"(v)); return v; }

// sum over the 16 lanes of a DPP row (every lane gets the total): quad butterflies, then the two mirror swaps -- register
// cross-lane moves, no LDS round trips
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// (row16_sum -- the sum over the 16 lanes of a DPP row -- lives in fused_policy.h)

#define MJX_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <bool COL>
__device__ __forceinline__ f32x4 mma_operand(const float* p, int stride) {
  if (COL) return f32x4{p[0], p[stride], p[2 * stride], p[3 * stride]};
  return *(const f32x4*)p;
}

// D tiles (TR x TC of 16 x 16) = A (row tile r) x B (column tile c) over nblk blocks of 16 contraction indices;
// ep(r, c, acc) stores one tile.  The workgroup's 16 waves take the tiles round-robin (at most 16 tiles for the shapes
// this kernel accepts: one tile per wave, four waves per SIMD hide each other's LDS and MFMA latencies).
constexpr int PFIT_THREADS = 1024, PFIT_WAVES = PFIT_THREADS / 64;

template <bool ACOL, bool BCOL, class EP>
__device__ __forceinline__ void mma_tiles(int tid, const float* __restrict__ Ab, int AS, const float* __restrict__ Bb, int BS,
                                          int TR, int TC, int nblk, EP ep) {
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r16 = lane & 15, q = lane >> 4;
  const int T = TR * TC;
  const int mul = TR == 1 ? 65537 : TR == 2 ? 32769 : TR == 3 ? 21846 : 16385;     // t / TR for t < 16 without a division
  const float* al = Ab + (ACOL ? 4 * q * AS + r16 : r16 * AS + 4 * q);
  const float* bl = Bb + (BCOL ? 4 * q * BS + r16 : r16 * BS + 4 * q);
  const int astep = ACOL ? 16 * AS : 16, bstep = BCOL ? 16 * BS : 16;
  for (int t = wave; t < T; t += PFIT_WAVES) {
    const int c = (t * mul) >> 16, r = t - c * TR;
    const float* a = al + 16 * r * (ACOL ? 1 : AS);
    const float* b = bl + 16 * c * (BCOL ? 1 : BS);
    f32x4 acc = (f32x4)(0.f);
    f32x4 pa = mma_operand<ACOL>(a, AS), pb = mma_operand<BCOL>(b, BS);
    f32x4 qa = pa, qb = pb;
    int kb = 0;
    while (true) {                                     // two register sets: the next block's operands are in flight during the MFMAs
      if (kb + 1 < nblk) { a += astep; b += bstep; qa = mma_operand<ACOL>(a, AS); qb = mma_operand<BCOL>(b, BS); }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = MJX_MFMA16(pa[i], pb[i], acc);
      if (++kb >= nblk) break;
      if (kb + 1 < nblk) { a += astep; b += bstep; pa = mma_operand<ACOL>(a, AS); pb = mma_operand<BCOL>(b, BS); }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = MJX_MFMA16(qa[i], qb[i], acc);
      if (++kb >= nblk) break;
    }
    ep(r, c, acc);
  }
}

// out[s][u] = f(sum_k in[s][k] W[u][k] + b[u]), s < B, u < OUT; K16: K rounded up to 16.  act: 0 none, 1 tanh.
__device__ __forceinline__ void fit_layer(const float* __restrict__ in, int IS, const float* __restrict__ W, int WS,
                                          const float* __restrict__ b, int B, int OUT, int K16, float* __restrict__ out, int OS,
                                          int act, int tid) {
  const int lane = tid & 63, r16 = lane & 15, q = lane >> 4;
  mma_tiles<false, false>(tid, in, IS, W, WS, (B + 15) >> 4, (OUT + 15) >> 4, K16 >> 4, [&](int r, int c, f32x4 acc) {
    const int u = 16 * c + r16, s0 = 16 * r + 4 * q;
    const float bu = b[u];                              // (past the end for a partial tile: unused)
    float v[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) v[rr] = acc[rr] + bu;
    if (act) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) v[rr] = fast_tanh(v[rr]);          // (the fused kernels' tanh: branch-free, 1.2e-7)
    }
    float* o = out + s0 * OS + u;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) if (u < OUT && s0 + rr < B) o[rr * OS] = v[rr];
  });
}

// delta_in[s][k] = (sum_u delta[s][u] W[u][k]) (1 - h[s][k]^2) for k < K (the layer's inputs, a multiple of 16); O16: number
// of the layer's outputs u rounded up to 16 (delta's pad columns are zero)
__device__ __forceinline__ void fit_back(const float* __restrict__ delta, int DS, const float* __restrict__ W, int WS, int B, int O16,
                                         int K, const float* __restrict__ h, int HS, float* __restrict__ din, int DIS, int tid) {
  const int lane = tid & 63, r16 = lane & 15, q = lane >> 4;
  mma_tiles<false, true>(tid, delta, DS, W, WS, (B + 15) >> 4, K >> 4, O16 >> 4, [&](int r, int c, f32x4 acc) {
    const int s0 = 16 * r + 4 * q, k = 16 * c + r16;
    const float* hp = h + s0 * HS + k;                  // (rows >= B of h exist and are finite: the blocks are padded)
    float hv[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) hv[rr] = hp[rr * HS];
    float* dp = din + s0 * DIS + k;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) if (s0 + rr < B) dp[rr * DIS] = acc[rr] * fmaf(-hv[rr], hv[rr], 1.0f);
  });
}

// gW[u][k] = sum_s delta[s][u] in[s][k] (u < OUT, k < K), gb[u] = sum_s delta[s][u]; written in the padded parameter layout.
// (delta's rows s >= B are zero: they are never written after the initial fill)
__device__ __forceinline__ void fit_wgrad(const float* __restrict__ delta, int DS, const float* __restrict__ in, int IS, int B, int OUT,
                                          int K, float* __restrict__ gW, int WS, float* __restrict__ gb, int tid) {
  const int lane = tid & 63, r16 = lane & 15, q = lane >> 4;
  mma_tiles<true, true>(tid, delta, DS, in, IS, (OUT + 15) >> 4, (K + 15) >> 4, (B + 15) >> 4, [&](int r, int c, f32x4 acc) {
    const int u0 = 16 * r + 4 * q, k = 16 * c + r16;
    float* g = gW + u0 * WS + k;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) if (k < K && u0 + rr < OUT) g[rr * WS] = acc[rr];
  });
  // bias gradients: sixteen partial sums per unit, combined inside 16 neighbouring lanes
  {
    const int u = tid >> 4, part = tid & 15;           // 1024 threads: 64 units x 16 parts
    float g = 0.f;
    if (u < OUT) for (int sr = part; sr < B; sr += 16) g += delta[sr * DS + u];
    g = row16_sum(g);
    if (part == 0 && u < OUT) gb[u] = g;
  }
}

template <int H>
__global__ __launch_bounds__(PFIT_THREADS) void k_policy_fit(PolicyFitArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const PolicyFitLayout<H> L(A.n, A.m);
  const int tid = threadIdx.x, n = A.n, m = A.m, B = A.B, d = L.d();
  const int MS = 20, S1 = L.S1, S2 = L.S2, XS = L.XS;         // MS: row stride of the per-action blocks (m <= 16, + 4)
  const bool old_net = (A.loss == 2) && !A.old_tracks_new;
  float* W = lds;                          // parameters (padded layout)
  float* G = W + L.P;                      // gradients (same layout)
  float* Wo = G + L.P;                     // old parameters (PPO with a fixed old policy)
  const int Bp = (B + 15) & ~15;           // rows of every activation block (rows >= B stay zero)
  float* X = Wo + (old_net ? L.P : 0);     // [Bp][XS]
  float* H1 = X + (size_t)Bp * XS;          // [B][S2]
  float* H2 = H1 + (size_t)Bp * S2;
  float* D1 = H2 + (size_t)Bp * S2;
  float* D2 = D1 + (size_t)Bp * S2;
  float* MU = D2 + (size_t)Bp * S2;         // [B][MS]
  float* MUo = MU + (size_t)Bp * MS;
  float* D3 = MUo + (size_t)Bp * MS;
  float* ROW = D3 + (size_t)Bp * MS;        // [B][4]: adv, row weight w, -, -
  // (offsets are rounded as indices: a pointer -> integer -> pointer round trip would drop the LDS address space and turn
  //  every access below into a FLAT one that waits on the outstanding global prefetches)
  const int o_sh = (int)((ROW + (size_t)B * 4) - lds);
  double* sh = (double*)(lds + ((o_sh + 1) & ~1));            // 17 doubles of reduction scratch

  // ---- load parameters (zero pads), Adam moments of the parameters this thread owns
  for (int i = tid; i < A.lds_floats; i += PFIT_THREADS) lds[i] = 0.f;      // incl. the zero pads every 16-byte operand read relies on
  __syncthreads();
  // Global <-> LDS traffic goes through ROLLED loops and the thread's register copies are filled from / drained to LDS:
  // with the unrolled per-element global accesses the compiler kept ~120 precomputed 64-bit addresses alive across the
  // whole step loop (registers the step needs; they ended up in scratch).
  constexpr int EPT = 10;                  // d <= 1024 * EPT  (64 x 64 with 63 observations and 16 actions: 9.4 k)
  int poff[EPT]; float am[EPT], av[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) { const int i = tid + PFIT_THREADS * e; poff[e] = (i < d) ? L.pad_of(i) : 0; }
#pragma nounroll
  for (int i = tid; i < d; i += PFIT_THREADS) {
    const int po = L.pad_of(i);
    W[po] = A.theta[i]; G[po] = A.adam_m[i];
    if (old_net) Wo[po] = A.theta_old[i];
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < EPT; ++e) am[e] = (tid + PFIT_THREADS * e < d) ? G[poff[e]] : 0.f;
  __syncthreads();
#pragma nounroll
  for (int i = tid; i < d; i += PFIT_THREADS) G[L.pad_of(i)] = A.adam_v[i];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < EPT; ++e) av[e] = (tid + PFIT_THREADS * e < d) ? G[poff[e]] : 0.f;
  __syncthreads();
  // transforms: in_shift / in_scale (new and old), out_shift / out_scale (new and old), fixed old log_std
  float* TR = (float*)(sh + 17);           // [4][64] in-transforms, [4][16] out-transforms, [16] old log_std
  for (int f = tid; f < n; f += PFIT_THREADS) { TR[f] = A.tr[f]; TR[64 + f] = A.tr[n + f]; TR[128 + f] = A.tr_old[f]; TR[192 + f] = A.tr_old[n + f]; }
  for (int a = tid; a < m; a += PFIT_THREADS) {
    TR[256 + a] = A.tr[2 * n + a]; TR[272 + a] = A.tr[2 * n + m + a];
    TR[288 + a] = A.tr_old[2 * n + a]; TR[304 + a] = A.tr_old[2 * n + m + a];
    TR[320 + a] = (A.loss == 2) ? A.theta_old[d - m + a] : 0.f;
  }
  __syncthreads();

  const float b1c = 0.9f, b2c = 0.999f, eps = 1e-8f;
  double pw1 = pow((double)b1c, (double)A.step0), pw2 = pow((double)b2c, (double)A.step0);
  const float invB = 1.0f / (float)B;
  const float llc = 0.5f * (float)m * 1.8378770664093453f;

  // Minibatch pipeline, two stages deep so that neither global round trip sits on a step's critical path: the row ids of
  // step s + 2 are requested at the top of step s (one register, parked in an LDS double buffer later in the step), and the
  // rows of step s + 1 (observations, actions, advantage) are requested right after into registers, addressed through the
  // row ids that were parked during step s - 1.
  constexpr int XR = 4, AR = 1;            // B n <= 64 * 63 and B m <= 64 * 16 elements over 1024 threads
  float xraw[XR], araw[AR], advraw = 0.f;
  int xsf[XR], asf[AR];                    // (row | column << 8) of the elements this thread moves, -1: none
#pragma unroll
  for (int c = 0; c < XR; ++c) { const int e = tid + PFIT_THREADS * c, r = e / n; xsf[c] = (e < B * n) ? (r | ((e - r * n) << 8)) : -1; }
#pragma unroll
  for (int c = 0; c < AR; ++c) { const int e = tid + PFIT_THREADS * c, r = e / m; asf[c] = (e < B * m) ? (r | ((e - r * m) << 8)) : -1; }
  float* ACT = (float*)(TR + 336);         // [B][MS] this step's actions
  int* IDXL = (int*)(ACT + (size_t)B * MS);        // [2][64] row ids of the next two steps
  auto fetch = [&](int64_t st) {
    const int* rows = IDXL + 64 * (int)(st & 1);
#pragma unroll
    for (int c = 0; c < XR; ++c) { if (PFIT_THREADS * c >= B * n) break; if (xsf[c] >= 0) xraw[c] = A.obs[(int64_t)rows[xsf[c] & 255] * n + (xsf[c] >> 8)]; }
#pragma unroll
    for (int c = 0; c < AR; ++c) if (asf[c] >= 0) araw[c] = A.act[(int64_t)rows[asf[c] & 255] * m + (asf[c] >> 8)];
    if (tid < B && A.loss == 2) advraw = A.adv[rows[tid]];
  };
  if (tid < B) {
    if (A.steps > 0) IDXL[tid] = A.idx[tid];
    if (A.steps > 1) IDXL[64 + tid] = A.idx[B + tid];
  }
  __syncthreads();
  if (A.steps > 0) fetch(0);
  for (int64_t step = 0; step < A.steps; ++step) {
    PFIT_STAMP(0);
    int next_row = 0;
    if (tid == 0) {                                        // this step's bias corrections, in double like torch's Python scalars
      pw1 *= (double)b1c; pw2 *= (double)b2c;
      ((float*)sh)[0] = 1.0f / (float)sqrt(1.0 - pw2);
      ((float*)sh)[1] = A.lr / (float)(1.0 - pw1);
    }
    if (tid < B && step + 2 < A.steps) next_row = A.idx[(step + 2) * B + tid];
    // ---- 1. the minibatch: normalised observations, actions, advantages from the prefetched registers ----
#pragma unroll
    for (int c = 0; c < XR; ++c) {
      if (PFIT_THREADS * c >= B * n) break;                // (uniform)
      if (xsf[c] >= 0) {
        const int r = xsf[c] & 255, f = xsf[c] >> 8;
        X[r * XS + f] = fast_div(xraw[c] - TR[f], TR[64 + f] + 1e-8f);
        if (old_net) D1[r * S2 + f] = fast_div(xraw[c] - TR[128 + f], TR[192 + f] + 1e-8f);   // the old network's own input transforms
      }
    }
#pragma unroll
    for (int c = 0; c < AR; ++c) if (asf[c] >= 0) ACT[(asf[c] & 255) * MS + (asf[c] >> 8)] = araw[c];
    if (tid < B) ROW[tid * 4] = advraw;
    if (step + 1 < A.steps) fetch(step + 1);
    __syncthreads();
    PFIT_STAMP(1);
    // ---- 2. forward: the new parameters, then (PPO with a fixed old policy) the old ones on the same rows.  D1 / D2 are
    //         free until the backward pass: D1 holds the old network's inputs in its first n columns (written with the
    //         minibatch above; n <= H, checked on the host), its hidden activations go D2 -> D1.
    const int K16 = (n + 15) & ~15;
#pragma nounroll
    for (int pass = 0; pass < (old_net ? 2 : 1); ++pass) {
      const float* Wp = pass ? Wo : W;
#pragma nounroll
      for (int l = 0; l < 3; ++l) {
        const float* in = l == 0 ? (pass ? D1 : X) : l == 1 ? (pass ? D2 : H1) : (pass ? D1 : H2);
        float* out = l == 0 ? (pass ? D2 : H1) : l == 1 ? (pass ? D1 : H2) : (pass ? MUo : MU);
        fit_layer(in, (l == 0 && !pass) ? XS : S2, Wp + (l == 0 ? L.oW1 : l == 1 ? L.oW2 : L.oW3), l == 0 ? S1 : S2,
                  Wp + (l == 0 ? L.ob1 : l == 1 ? L.ob2 : L.ob3), B, l == 2 ? m : H, l == 0 ? K16 : H, out, l == 2 ? MS : S2, l != 2,
                  fresh(tid));
        if (pass == 0) PFIT_STAMP(10 + 2 * l);
        __syncthreads();
        if (pass == 0) PFIT_STAMP(11 + 2 * l);
      }
      if (pass == 0) {
        PFIT_STAMP(2);
        if (tid < B && step + 2 < A.steps) IDXL[64 * (int)(step & 1) + tid] = next_row;   // last read by fetch(step), a step ago
      }
    }
    PFIT_STAMP(3);
    // ---- 3. loss head: one thread per (row, action) -- 64 x 16 = the whole workgroup; the per-row sums over the actions
    //         are butterfly sums inside 16 neighbouring lanes (every lane of a row ends up with the same totals) ----
    {
      const int s = tid >> 4, a = tid & 15;
      float rloss = 0.f;                                  // this row's loss term (all 16 lanes of the row agree)
      const bool on = s < B && a < m;
      const int am = a < m ? a : 0, sm = s < B ? s : 0;
      const float x = ACT[sm * MS + am];
      const float mu = MU[sm * MS + am] * TR[272 + am] + TR[256 + am];
      if (A.loss == 0) {
        const float e = mu - x;
        if (on) D3[s * MS + a] = TR[272 + a] * (2.0f * e * invB / (float)m);
        rloss = row16_sum(on ? e * e : 0.f);
      } else {
        const float ls = W[L.oS + am], sg = expf(ls);
        const float zn = on ? (x - mu) / sg : 0.f;
        const float lln = row16_sum(-0.5f * zn * zn) - row16_sum(on ? ls : 0.f) - llc;
        float wrow;
        if (A.loss == 1) { wrow = -invB; rloss = lln; }
        else {
          const float lso = TR[320 + am];
          const float muo = old_net ? MUo[sm * MS + am] * TR[304 + am] + TR[288 + am] : mu;
          const float zo = on ? (x - muo) / expf(lso) : 0.f;
          const float llo = row16_sum(-0.5f * zo * zo) - row16_sum(on ? lso : 0.f) - llc;
          const float LR = expf(lln - llo), ad = ROW[sm * 4];
          const float s1 = LR * ad, s2 = fminf(fmaxf(LR, 1.0f - A.clip), 1.0f + A.clip) * ad;
          rloss = fminf(s1, s2);
          const bool inside = (LR >= 1.0f - A.clip) && (LR <= 1.0f + A.clip);
          wrow = (inside || s1 < s2) ? -ad * LR * invB : 0.f;
        }
        if (on) {
          D3[s * MS + a] = TR[272 + a] * (wrow * zn / sg);
          MUo[s * MS + a] = wrow * (zn * zn - 1.0f);            // per-row contribution to dLoss/dlog_std (MUo is free now)
        }
      }
      if (on && a == 0) ROW[s * 4 + 1] = rloss;
    }
    __syncthreads();
    if (A.loss_trace && (tid >> 6) == PFIT_WAVES - 1) {     // the minibatch loss: one wave adds the rows' terms (fixed order)
      const int lane = tid & 63;
      const double lt = wave_sum(lane < B ? (double)ROW[lane * 4 + 1] : 0.0);
      if (lane == 0) A.loss_trace[step] = (A.loss == 0) ? lt / ((double)B * (double)m) : -lt / (double)B;
    }
    PFIT_STAMP(4);
    // ---- 4. backward: layers 3, 2, 1 (weight gradients; the cotangent of the layer below for 3 and 2) ----
#pragma nounroll
    for (int l = 3; l >= 1; --l) {
      const float* dl = l == 3 ? D3 : l == 2 ? D2 : D1;
      const float* in = l == 3 ? H2 : l == 2 ? H1 : X;
      const int DS = l == 3 ? MS : S2, OUT = l == 3 ? m : H;
      fit_wgrad(dl, DS, in, l == 1 ? XS : S2, B, OUT, l == 1 ? n : H, G + (l == 3 ? L.oW3 : l == 2 ? L.oW2 : L.oW1), l == 1 ? S1 : S2,
                G + (l == 3 ? L.ob3 : l == 2 ? L.ob2 : L.ob1), fresh(tid));
      if (l > 1) fit_back(dl, DS, W + (l == 3 ? L.oW3 : L.oW2), S2, B, l == 3 ? 16 : H, H, in, S2, l == 3 ? D2 : D1, S2, fresh(tid));
      if (l == 3) {
        if (A.loss != 0 && tid < 8 * m) {                   // dLoss/dlog_std[a] = sum_rows w (z^2 - 1): 8 partial sums per action ...
          const int a = tid >> 3, part = tid & 7;
          float g = 0.f;
          for (int s = part; s < B; s += 8) g += MUo[s * MS + a];
          MU[a * 8 + part] = g;                             // (the means are dead once D3 holds dLoss/dmean; B * MS >= 8 m for B >= 8)
        }
        __syncthreads();
        if (A.loss != 0 && tid < m) {                       // ... added in a fixed order
          float g = 0.f;
#pragma unroll
          for (int part = 0; part < 8; ++part) g += MU[tid * 8 + part];
          G[L.oS + tid] = g;
        }
        PFIT_STAMP(5);
      } else {
        __syncthreads();
        if (l == 2) PFIT_STAMP(6); else PFIT_STAMP(7);
      }
    }
    // ---- 5. torch.optim.Adam on the parameters this thread owns (MSE: log_std has no gradient -> untouched) ----
    const int dlim = (A.loss == 0) ? d - m : d;
    const float inv_bc2s = ((const float*)sh)[0], step_size = ((const float*)sh)[1];     // (thread 0, top of the step)
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      if (PFIT_THREADS * e >= dlim) break;                 // (uniform: the unused register slots cost nothing)
      const int i = tid + PFIT_THREADS * e;
      if (i < dlim) {
        const float gi = G[poff[e]];
        const float mi = am[e] + (gi - am[e]) * (1.0f - b1c);
        const float vi = av[e] * b2c + gi * gi * (1.0f - b2c);
        am[e] = mi; av[e] = vi;
        const float denom = fmaf(__builtin_amdgcn_sqrtf(vi), inv_bc2s, eps);
        W[poff[e]] = fmaf(-step_size, fast_div(mi, denom), W[poff[e]]);
      }
    }
    __syncthreads();
    PFIT_STAMP(8);
  }
  // ---- write back (moments through the gradient block, see the prologue) ----
#pragma unroll
  for (int e = 0; e < EPT; ++e) if (tid + PFIT_THREADS * e < d) G[poff[e]] = am[e];
  __syncthreads();
#pragma nounroll
  for (int i = tid; i < d; i += PFIT_THREADS) { const int po = L.pad_of(i); A.theta[i] = W[po]; A.adam_m[i] = G[po]; }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < EPT; ++e) if (tid + PFIT_THREADS * e < d) G[poff[e]] = av[e];
  __syncthreads();
#pragma nounroll
  for (int i = tid; i < d; i += PFIT_THREADS) A.adam_v[i] = G[L.pad_of(i)];
}

}  // namespace mjx
