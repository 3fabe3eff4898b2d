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
  int n, m;
};

#define MJX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void wave_sync() {
  // LDS traffic between lanes of ONE wave: hardware executes a wave's DS ops in order,
  // this only stops the compiler from moving them across the hand-off.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// tanh(x) = sign(x) (1 - t)/(1 + t), t = exp(-2|x|): branch-free, v_exp_f32 + v_rcp_f32 with one
// Newton step; max abs error ~6e-8 (same class as ocml tanhf, which is branchy).
__device__ __forceinline__ float fast_tanh(float x) {
  float t = __builtin_amdgcn_exp2f(fabsf(x) * -2.885390081777927f);
  float d = 1.0f + t;
  float r = __builtin_amdgcn_rcpf(d);
  r = r * fmaf(-d, r, 2.0f);
  return copysignf((1.0f - t) * r, x);
}

// unit index (within a 32-block) held by accumulator register r of lane-half hi
__device__ __forceinline__ constexpr int unit_of(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <int H1, int H2, int NT1, int MP>
struct FusedLayout {
  static constexpr int MT1 = H1 / 32, MT2 = H2 / 32;
  static constexpr int S2 = H1 + 4;             // row stride of W2 (ds_read_b128, S2/4 odd)
  static constexpr int ST = 36;                 // row stride of [unit][sample] scratch
  static constexpr int HM = (H1 > H2 ? H1 : H2);
  int NP;                                       // features + ones column, padded to 4
  int S1;                                       // row stride of xs / W1a (ds_read_b64, == 2 mod 4)
  int oW1, oW2, oW3, oB2, oB3, SLOT;            // weight slot (one per parameter set)
  int oXS, oXT, oD3, oBA, oBB, WAVE;            // per-wave scratch
  int oTR, oCST, oWAVES, TOTAL;
  static constexpr int NCST = 8;                // osc, osh, sigma, log_std (new) ; osc, osh, sigma, log_std (old)
  __host__ __device__ explicit FusedLayout(int n) {
    NP = (n + 1 + 3) & ~3;
    S1 = NP + 2;
    oW1 = 0;
    oW2 = oW1 + H1 * S1;
    oW3 = oW2 + H2 * S2;                        // [MP][H2]
    oB2 = oW3 + MP * H2;
    oB3 = oB2 + H2;
    SLOT = ((oB3 + MP + 3) / 4) * 4;
    oXS = 0;                                    // [32][S1]
    oXT = ((oXS + 32 * S1 + 3) / 4) * 4;        // [NP][ST]
    oD3 = oXT + NP * ST;                        // [MP][ST]
    oBA = oD3 + MP * ST;                        // [HM][ST]
    oBB = oBA + HM * ST;                        // [HM][ST]
    WAVE = ((oBB + HM * ST + 3) / 4) * 4;
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

template <int H1, int H2, int NT1, int MP, int MODE, bool DBG = false>
__global__ __launch_bounds__(256, 1) void k_fused(FusedArgs A) {
  using LT = FusedLayout<H1, H2, NT1, MP>;
  constexpr int MT1 = LT::MT1, MT2 = LT::MT2, S2 = LT::S2, ST = LT::ST;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, hi = lane >> 5;
  const int n = A.n, m = A.m;
  const LT L(n);
  const int NP = L.NP, S1 = L.S1;
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

  // ---------------- stage weights (whole workgroup) ----------------
  for (int idx = tid; idx < L.TOTAL; idx += 256) lds[idx] = 0.0f;
  __syncthreads();
  for (int s = 0; s < 2; ++s) {
    float* slot = s ? slotB : slotA;
    const float* th = s ? A.thetaB : A.thetaA;
    for (int idx = tid; idx < H1 * (n + 1); idx += 256) {
      int u = idx / (n + 1), f = idx - u * (n + 1);
      slot[L.oW1 + u * S1 + f] = (f < n) ? th[fo.W1 + u * n + f] : th[fo.b1 + u];
    }
    for (int idx = tid; idx < H2 * H1; idx += 256) {
      int u = idx / H1, k = idx - u * H1;
      slot[L.oW2 + u * S2 + k] = th[fo.W2 + idx];
    }
    for (int idx = tid; idx < m * H2; idx += 256) slot[L.oW3 + idx] = th[fo.W3 + idx];
    for (int idx = tid; idx < H2; idx += 256) slot[L.oB2 + idx] = th[fo.b2 + idx];
    for (int idx = tid; idx < m; idx += 256) slot[L.oB3 + idx] = th[fo.b3 + idx];
  }
  for (int idx = tid; idx < n; idx += 256) {
    trs[idx] = A.trA[idx];
    trs[NP + idx] = A.trA[n + idx];
    trs[2 * NP + idx] = A.trB[idx];
    trs[3 * NP + idx] = A.trB[n + idx];
  }
  // constant "ones" feature (bias column) of every wave's staging buffers
  if (lane < 32) xs[lane * S1 + n] = 1.0f;
  if (lane < 32) xT[n * ST + lane] = 1.0f;
  __syncthreads();

  // ---------------- per-action constants (LDS, broadcast reads) ----------------
  float* cst = lds + L.oCST;
  enum { C_OSC = 0, C_OSH = 1, C_SG = 2, C_LS = 3, C_OSCB = 4, C_OSHB = 5, C_SGB = 6, C_LSB = 7 };
  if (tid < MP) {
    const int a = tid;
    const bool ok = a < m;
    float lsa = ok ? A.thetaA[fo.S + a] : 0.f;
    float lsb = (ok && MODE != MODE_FVP) ? A.thetaB[fo.S + a] : lsa;
    cst[C_OSC * MP + a] = ok ? A.trA[2 * n + m + a] : 0.f;
    cst[C_OSH * MP + a] = ok ? A.trA[2 * n + a] : 0.f;
    cst[C_SG * MP + a] = ok ? expf(lsa) : 1.0f;
    cst[C_LS * MP + a] = lsa;
    cst[C_OSCB * MP + a] = ok ? A.trB[2 * n + m + a] : 0.f;
    cst[C_OSHB * MP + a] = ok ? A.trB[2 * n + a] : 0.f;
    cst[C_SGB * MP + a] = ok ? expf(lsb) : 1.0f;
    cst[C_LSB * MP + a] = lsb;
  }
  __syncthreads();

  // ---------------- persistent accumulators ----------------
  f32x16 gW1[MT1][NT1], gW2[MT2][MT1], gW3[MT2];
  float sb2 = 0.f, sb3 = 0.f, gls[MP];   // lane u < H2: grad b2[u]; lane a < m: grad b3[a]
#pragma unroll
  for (int a = 0; a < MT1; ++a)
#pragma unroll
    for (int b = 0; b < NT1; ++b) gW1[a][b] = (f32x16)(0.f);
#pragma unroll
  for (int a = 0; a < MT2; ++a) {
    gW3[a] = (f32x16)(0.f);
#pragma unroll
    for (int b = 0; b < MT1; ++b) gW2[a][b] = (f32x16)(0.f);
  }
#pragma unroll
  for (int a = 0; a < MP; ++a) gls[a] = 0.f;
  double s_surr = 0.0, s_kl = 0.0, s_cnt = 0.0;

  const int64_t ntiles = (A.N + 31) / 32;
  const int64_t tstride = (int64_t)gridDim.x * 4;
  constexpr int XL = 16 * NT1;                   // obs loads per lane per tile (32*n/64 <= XL)
  float xr[XL];
  const float inv_n = 1.0f / (float)n;

  auto load_x = [&](int64_t tile) {
    const int64_t base = tile * 32 * (int64_t)n;
    const int64_t lim = A.N * (int64_t)n;
#pragma unroll
    for (int c = 0; c < XL; ++c) {
      int e = c * 64 + lane;
      bool ok = (e < 32 * n) && (base + e < lim);
      float v = A.obs[ok ? base + e : 0];
      xr[c] = ok ? v : 0.0f;
    }
  };

  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  if (tile < ntiles) load_x(tile);

  for (; tile < ntiles; tile += tstride) {
    const int64_t s0 = tile * 32;
    const bool valid = (s0 + j) < A.N;
    // ---- 0. stage observations (raw) into xs [sample][feature] and xT [feature][sample]
#pragma unroll
    for (int c = 0; c < XL; ++c) {
      int e = c * 64 + lane;
      if (e < 32 * n) {
        int sidx = (int)(((float)e + 0.5f) * inv_n);
        int f = e - sidx * n;
        xs[sidx * S1 + f] = xr[c];
      }
    }
    if (tile + tstride < ntiles) load_x(tile + tstride);
    // per-sample action / advantage loads for this tile (used late; issue early)
    float av[MP];
    float advv = 0.f;
    if (MODE != MODE_FVP) {
#pragma unroll
      for (int a = 0; a < MP; ++a) {
        bool ok = valid && (a < m);
        float v = A.act[ok ? (s0 + j) * m + a : 0];
        av[a] = ok ? v : 0.f;
      }
      {
        float v = A.adv[valid ? s0 + j : 0];
        advv = valid ? v : 0.f;
      }
    }
    wave_sync();

    // forward of one parameter set: fills h1 / h2 (activations, lane = sample)
    auto forward = [&](const float* slot, const float* tsh, const float* tsc, bool writeT,
                       f32x16 (&h1)[MT1], f32x16 (&h2)[MT2]) {
      // normalise this lane-pair's features on the fly: x~ = (x - shift)/(scale + 1e-8)
      f32x16 z1[MT1];
#pragma unroll
      for (int mt = 0; mt < MT1; ++mt) z1[mt] = (f32x16)(0.f);
      for (int q = 0; q < NP / 4; ++q) {
        int f0 = 4 * q + 2 * hi;
        f32x2 xb = *(const f32x2*)&xs[j * S1 + f0];
        // the ones column (f == n) and the zero pad must pass through unchanged
        float x0 = (f0 < n) ? (xb.x - tsh[f0]) / (tsc[f0] + 1e-8f) : xb.x;
        float x1 = (f0 + 1 < n) ? (xb.y - tsh[f0 + 1]) / (tsc[f0 + 1] + 1e-8f) : xb.y;
        if (writeT) { xT[f0 * ST + j] = x0; xT[(f0 + 1) * ST + j] = x1; }
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt) {
          f32x2 wa = *(const f32x2*)&slot[L.oW1 + (32 * mt + j) * S1 + f0];
          z1[mt] = MJX_MFMA(wa.x, x0, z1[mt]);
          z1[mt] = MJX_MFMA(wa.y, x1, z1[mt]);
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) h1[mt][r] = fast_tanh(z1[mt][r]);
      f32x16 z2[MT2];
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 b = *(const f32x4*)&slot[L.oB2 + 32 * mt + 8 * q + 4 * hi];
          z2[mt][4 * q + 0] = b.x; z2[mt][4 * q + 1] = b.y; z2[mt][4 * q + 2] = b.z; z2[mt][4 * q + 3] = b.w;
        }
#pragma unroll
      for (int kb = 0; kb < MT1; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int mt = 0; mt < MT2; ++mt) {
            f32x4 wa = *(const f32x4*)&slot[L.oW2 + (32 * mt + j) * S2 + 32 * kb + 8 * q + 4 * hi];
            z2[mt] = MJX_MFMA(wa.x, h1[kb][4 * q + 0], z2[mt]);
            z2[mt] = MJX_MFMA(wa.y, h1[kb][4 * q + 1], z2[mt]);
            z2[mt] = MJX_MFMA(wa.z, h1[kb][4 * q + 2], z2[mt]);
            z2[mt] = MJX_MFMA(wa.w, h1[kb][4 * q + 3], z2[mt]);
          }
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) h2[mt][r] = fast_tanh(z2[mt][r]);
    };

    // output layer on the VALU: out[a] = sum_k W3[a][k] * v[k]  (cross-half reduced)
    auto out_layer = [&](const float* slot, const f32x16 (&v)[MT2], float (&o)[MP]) {
#pragma unroll
      for (int a = 0; a < MP; ++a) {
        float acc = 0.f;
        {
#pragma unroll
          for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f32x4 w = *(const f32x4*)&slot[L.oW3 + a * H2 + 32 * mt + 8 * q + 4 * hi];
              acc = fmaf(w.x, v[mt][4 * q + 0], acc);
              acc = fmaf(w.y, v[mt][4 * q + 1], acc);
              acc = fmaf(w.z, v[mt][4 * q + 2], acc);
              acc = fmaf(w.w, v[mt][4 * q + 3], acc);
            }
        }
        o[a] = acc;
      }
#pragma unroll
      for (int a = 0; a < MP; ++a) o[a] += __shfl_xor(o[a], 32);
    };

    f32x16 h1[MT1], h2[MT2];
    forward(slotA, trs, trs + NP, MODE != MODE_EVAL, h1, h2);

    float d3[MP];                                 // delta on the (pre-scale) output layer
    if (MODE == MODE_FVP) {
      // ---- tangent pass: t1 = (V1 x~ + c1)(1-h1^2); t2 = (V2 h1 + W2 t1 + c2)(1-h2^2)
      wave_sync();                                // xT written above, read below
      f32x16 t1[MT1], t2[MT2];
#pragma unroll
      for (int mt = 0; mt < MT1; ++mt) t1[mt] = (f32x16)(0.f);
      for (int q = 0; q < NP / 4; ++q) {
        int f0 = 4 * q + 2 * hi;
        float x0 = xT[f0 * ST + j], x1 = xT[(f0 + 1) * ST + j];
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt) {
          f32x2 wa = *(const f32x2*)&slotB[L.oW1 + (32 * mt + j) * S1 + f0];
          t1[mt] = MJX_MFMA(wa.x, x0, t1[mt]);
          t1[mt] = MJX_MFMA(wa.y, x1, t1[mt]);
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) t1[mt][r] *= fmaf(-h1[mt][r], h1[mt][r], 1.0f);
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 b = *(const f32x4*)&slotB[L.oB2 + 32 * mt + 8 * q + 4 * hi];
          t2[mt][4 * q + 0] = b.x; t2[mt][4 * q + 1] = b.y; t2[mt][4 * q + 2] = b.z; t2[mt][4 * q + 3] = b.w;
        }
#pragma unroll
      for (int kb = 0; kb < MT1; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int mt = 0; mt < MT2; ++mt) {
            const int off = (32 * mt + j) * S2 + 32 * kb + 8 * q + 4 * hi;
            f32x4 va = *(const f32x4*)&slotB[L.oW2 + off];
            f32x4 wa = *(const f32x4*)&slotA[L.oW2 + off];
            t2[mt] = MJX_MFMA(va.x, h1[kb][4 * q + 0], t2[mt]);
            t2[mt] = MJX_MFMA(wa.x, t1[kb][4 * q + 0], t2[mt]);
            t2[mt] = MJX_MFMA(va.y, h1[kb][4 * q + 1], t2[mt]);
            t2[mt] = MJX_MFMA(wa.y, t1[kb][4 * q + 1], t2[mt]);
            t2[mt] = MJX_MFMA(va.z, h1[kb][4 * q + 2], t2[mt]);
            t2[mt] = MJX_MFMA(wa.z, t1[kb][4 * q + 2], t2[mt]);
            t2[mt] = MJX_MFMA(va.w, h1[kb][4 * q + 3], t2[mt]);
            t2[mt] = MJX_MFMA(wa.w, t1[kb][4 * q + 3], t2[mt]);
          }
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) t2[mt][r] *= fmaf(-h2[mt][r], h2[mt][r], 1.0f);
      float o1[MP], o2[MP];
      out_layer(slotB, h2, o1);                   // V3 h2
      out_layer(slotA, t2, o2);                   // W3 t2
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
#pragma unroll
      for (int a = 0; a < MP; ++a) {
        float c3 = slotB[L.oB3 + a];
        float mudot = cst[C_OSC * MP + a] * (o1[a] + o2[a] + c3);
        float u = cst[C_SG * MP + a] * cst[C_SG * MP + a];
        float Dk = 2.0f / (2.0f * u + 1e-8f);
        float dmu = valid ? Dk * mudot * A.inv_N : 0.f;
        d3[a] = cst[C_OSC * MP + a] * dmu;
        if (DBG && A.dbg && blockIdx.x == 0 && wave == 0 && tile == 0 && hi == 0) A.dbg[2048 * 4 + a * 32 + j] = mudot;
      }
    } else {
      // ---- likelihoods (mean_LL, gaussian_mlp.py:99-115)
      float o[MP], mu[MP];
      out_layer(slotA, h2, o);
      float llA = 0.f, sumls = 0.f;
      float z[MP];
#pragma unroll
      for (int a = 0; a < MP; ++a) {
        float b3 = slotA[L.oB3 + a];
        mu[a] = (o[a] + b3) * cst[C_OSC * MP + a] + cst[C_OSH * MP + a];
        z[a] = (av[a] - mu[a]) / cst[C_SG * MP + a];
        llA = fmaf(-0.5f * z[a], z[a], llA);
        sumls += cst[C_LS * MP + a];
      }
      llA = llA - sumls - 0.5f * (float)m * 1.8378770664093453f;
      float llB = llA, muB[MP];
#pragma unroll
      for (int a = 0; a < MP; ++a) muB[a] = mu[a];
      if (MODE == MODE_EVAL || !A.old_is_new) {
        f32x16 g1[MT1], g2[MT2];
        forward(slotB, trs + 2 * NP, trs + 3 * NP, false, g1, g2);
        float ob[MP];
        out_layer(slotB, g2, ob);
        llB = 0.f;
        float sumlsB = 0.f;
#pragma unroll
        for (int a = 0; a < MP; ++a) {
          float b3 = slotB[L.oB3 + a];
          muB[a] = (ob[a] + b3) * cst[C_OSCB * MP + a] + cst[C_OSHB * MP + a];
          float zb = (av[a] - muB[a]) / cst[C_SGB * MP + a];
          llB = fmaf(-0.5f * zb, zb, llB);
          sumlsB += cst[C_LSB * MP + a];
        }
        llB = llB - sumlsB - 0.5f * (float)m * 1.8378770664093453f;
      }
      float LR = expf(llA - llB);
      if (valid && hi == 0) { s_surr += (double)(LR * advv); s_cnt += 1.0; }
      if (MODE == MODE_EVAL) {
        // mean_kl(new, old), gaussian_mlp.py:135-145
        float kl = 0.f;
#pragma unroll
        for (int a = 0; a < MP; ++a) {
            float so = cst[C_SGB * MP + a], sn = cst[C_SG * MP + a];
            float Nr = (muB[a] - mu[a]) * (muB[a] - mu[a]) + so * so - sn * sn;
            float Dr = 2.0f * sn * sn + 1e-8f;
            kl += Nr / Dr + cst[C_LS * MP + a] - cst[C_LSB * MP + a];
          }
        if (valid && hi == 0) s_kl += (double)kl;
      } else {
        float w = valid ? advv * LR * A.inv_N : 0.f;
#pragma unroll
        for (int a = 0; a < MP; ++a) {
          float dmu = w * z[a] / cst[C_SG * MP + a];
          d3[a] = cst[C_OSC * MP + a] * dmu;
          if (hi == 0) gls[a] += w * (z[a] * z[a] - 1.0f);
        }
        if (DBG && A.dbg && blockIdx.x == 0 && wave == 0 && tile == 0 && hi == 0) {
#pragma unroll
          for (int a = 0; a < MP; ++a) A.dbg[2048 * 4 + a * 32 + j] = mu[a];
          A.dbg[2048 * 4 + MP * 32 + j] = llA;
        }
      }
    }

    if (MODE != MODE_EVAL) {
      // ================= backward (shared by VPG and FVP) =================
      // d3[a]: cotangent on the pre-scale output, lane = sample (both halves hold it).
      // Park h2^T in bufA and h1^T in bufB ([unit][sample]); the register copies die here and
      // the (1 - h^2) factors are read back from LDS in accumulator layout.
      if (hi == 0) {
#pragma unroll
        for (int a = 0; a < MP; ++a) d3T[a * ST + j] = d3[a];
      }
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) bufA[(32 * mt + unit_of(r, hi)) * ST + j] = h2[mt][r];
#pragma unroll
      for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) bufB[(32 * mt + unit_of(r, hi)) * ST + j] = h1[mt][r];
      wave_sync();
      // gW3[a][k] += sum_s d3[s][a] * h2[s][k]
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 a4 = *(const f32x4*)&d3T[(j < MP ? j : 0) * ST + 8 * q + 4 * hi];
        if (j >= MP) a4 = (f32x4)(0.f);
#pragma unroll
        for (int nt = 0; nt < MT2; ++nt) {
          f32x4 b4 = *(const f32x4*)&bufA[(32 * nt + j) * ST + 8 * q + 4 * hi];
          gW3[nt] = MJX_MFMA(a4.x, b4.x, gW3[nt]);
          gW3[nt] = MJX_MFMA(a4.y, b4.y, gW3[nt]);
          gW3[nt] = MJX_MFMA(a4.z, b4.z, gW3[nt]);
          gW3[nt] = MJX_MFMA(a4.w, b4.w, gW3[nt]);
        }
      }
      if (lane < MP) {                            // grad b3[a] = sum_s d3[s][a]
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          f32x4 v = *(const f32x4*)&d3T[lane * ST + 4 * q];
          sb3 += (v.x + v.y) + (v.z + v.w);
        }
      }
      // delta2 = (W3^T d3) (1 - h2^2)
      f32x16 dl2[MT2];
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt) dl2[mt] = (f32x16)(0.f);
#pragma unroll
      for (int s = 0; s < MP / 2; ++s) {
        {
          float b = hi ? d3[2 * s + 1] : d3[2 * s];
#pragma unroll
          for (int mt = 0; mt < MT2; ++mt) {
            float a = slotA[L.oW3 + (2 * s + hi) * H2 + 32 * mt + j];
            dl2[mt] = MJX_MFMA(a, b, dl2[mt]);
          }
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float y = bufA[(32 * mt + unit_of(r, hi)) * ST + j];
          dl2[mt][r] *= fmaf(-y, y, 1.0f);
        }
      wave_sync();                                // all reads of h2^T done: bufA is free
#pragma unroll
      for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) bufA[(32 * mt + unit_of(r, hi)) * ST + j] = dl2[mt][r];
      wave_sync();
      if (lane < H2) {                            // grad b2[u] = sum_s delta2[s][u]
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          f32x4 v = *(const f32x4*)&bufA[lane * ST + 4 * q];
          sb2 += (v.x + v.y) + (v.z + v.w);
        }
      }
      // gW2[u2][u1] += sum_s delta2[s][u2] * h1[s][u1]      (A = delta2^T in bufA, B = h1^T in bufB)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 b4[MT1];
#pragma unroll
        for (int nt = 0; nt < MT1; ++nt) b4[nt] = *(const f32x4*)&bufB[(32 * nt + j) * ST + 8 * q + 4 * hi];
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt) {
          f32x4 a4 = *(const f32x4*)&bufA[(32 * mt + j) * ST + 8 * q + 4 * hi];
#pragma unroll
          for (int nt = 0; nt < MT1; ++nt) {
            gW2[mt][nt] = MJX_MFMA(a4.x, b4[nt].x, gW2[mt][nt]);
            gW2[mt][nt] = MJX_MFMA(a4.y, b4[nt].y, gW2[mt][nt]);
            gW2[mt][nt] = MJX_MFMA(a4.z, b4[nt].z, gW2[mt][nt]);
            gW2[mt][nt] = MJX_MFMA(a4.w, b4[nt].w, gW2[mt][nt]);
          }
        }
      }
      // delta1 = (W2^T delta2) (1 - h1^2)   (A operand = W2 read column-wise, B = delta2 accumulators)
      f32x16 dl1[MT1];
#pragma unroll
      for (int mt = 0; mt < MT1; ++mt) dl1[mt] = (f32x16)(0.f);
#pragma unroll
      for (int kb = 0; kb < MT2; ++kb)
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
          for (int mt = 0; mt < MT1; ++mt) {
            float a = slotA[L.oW2 + (32 * kb + unit_of(s, hi)) * S2 + 32 * mt + j];
            dl1[mt] = MJX_MFMA(a, dl2[kb][s], dl1[mt]);
          }
#pragma unroll
      for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float y = bufB[(32 * mt + unit_of(r, hi)) * ST + j];
          dl1[mt][r] *= fmaf(-y, y, 1.0f);
        }
      if (DBG && A.dbg && blockIdx.x == 0 && wave == 0 && tile == 0) {
        float* g = A.dbg + 2048 * 5;
#pragma unroll
        for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) g[(32 * mt + unit_of(r, hi)) * 32 + j] = dl2[mt][r];
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) g[2048 + (32 * mt + unit_of(r, hi)) * 32 + j] = dl1[mt][r];
      }
      wave_sync();                                // gW2 reads of bufA / (1-h1^2) reads of bufB done
#pragma unroll
      for (int mt = 0; mt < MT1; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) bufA[(32 * mt + unit_of(r, hi)) * ST + j] = dl1[mt][r];
      wave_sync();
      // gW1a[u1][f] += sum_s delta1[s][u1] * x~a[s][f]   (column n = bias gradient)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) {
          int f = 32 * nt + j;
          f32x4 b4 = *(const f32x4*)&xT[(f < NP ? f : 0) * ST + 8 * q + 4 * hi];
          if (f >= NP) b4 = (f32x4)(0.f);
#pragma unroll
          for (int mt = 0; mt < MT1; ++mt) {
            f32x4 a4 = *(const f32x4*)&bufA[(32 * mt + j) * ST + 8 * q + 4 * hi];
            gW1[mt][nt] = MJX_MFMA(a4.x, b4.x, gW1[mt][nt]);
            gW1[mt][nt] = MJX_MFMA(a4.y, b4.y, gW1[mt][nt]);
            gW1[mt][nt] = MJX_MFMA(a4.z, b4.z, gW1[mt][nt]);
            gW1[mt][nt] = MJX_MFMA(a4.w, b4.w, gW1[mt][nt]);
          }
        }
      }
    }
    wave_sync();                                  // everything read before the next tile's staging
  }

  // ---------------- workgroup reduction + partial write ----------------
  __syncthreads();
  if (MODE != MODE_EVAL) {
    // log_std gradient: reduce over the 32 samples (lanes j) of the hi == 0 half
    if (MODE == MODE_VPG) {
#pragma unroll
      for (int off = 1; off < 32; off <<= 1)
#pragma unroll
        for (int a = 0; a < MP; ++a) gls[a] += __shfl_xor(gls[a], off);
    }
    // each wave drops its partial gradient into its own LDS region [wave][d], then the
    // workgroup sums the four copies.  (weights in LDS are dead by now)
    float* red = lds;                             // 4 * d floats <= TOTAL (checked on host)
    float* mine = red + wave * fo.d;
    for (int idx = lane; idx < fo.d; idx += 64) mine[idx] = 0.f;
    wave_sync();
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
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
      for (int nt = 0; nt < MT1; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          mine[fo.W2 + (32 * mt + unit_of(r, hi)) * H1 + 32 * nt + j] = gW2[mt][nt][r];
#pragma unroll
    for (int nt = 0; nt < MT2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int a = unit_of(r, hi);
        if (a < m) mine[fo.W3 + a * H2 + 32 * nt + j] = gW3[nt][r];
      }
    if (lane < H2) mine[fo.b2 + lane] = sb2;
    if (lane < m) mine[fo.b3 + lane] = sb3;
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < MP; ++a)
        if (a < m) mine[fo.S + a] = (MODE == MODE_VPG) ? gls[a] : 0.f;
    }
    __syncthreads();
    float* outp = A.partials + (size_t)blockIdx.x * fo.d;
    for (int idx = tid; idx < fo.d; idx += 256)
      outp[idx] = (red[idx] + red[fo.d + idx]) + (red[2 * fo.d + idx] + red[3 * fo.d + idx]);
  }
  if (MODE != MODE_FVP) {
    // scalar partials: wave reduce (fp64) -> LDS -> one thread
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      s_surr += __shfl_xor(s_surr, off);
      s_kl += __shfl_xor(s_kl, off);
      s_cnt += __shfl_xor(s_cnt, off);
    }
    __syncthreads();
    double* sred = (double*)(lds + 4 * fo.d + 4);
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
