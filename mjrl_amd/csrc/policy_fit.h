// policy_fit.h -- persistent single-workgroup minibatch-Adam trainer for small tanh-MLP policies
// (two hidden layers of equal width H in {32, 64}, minibatches of up to 64 rows).
//
// The torch-optimizer loops of behaviour cloning and PPO (mjrl/algos/behavior_cloning.py:107-136,
// mjrl/algos/ppo_clip.py:85-95) are chains of tens of thousands of tiny dependent steps: a 64-row minibatch through a
// 5.7 k-parameter net is ~2 MFLOP.  As separate launches (gather, three GEMMs, loss head, five GEMMs, reductions, Adam)
// a step costs ~150 us of dependent-dispatch latency; here the whole chain runs inside ONE launch on one workgroup:
// parameters (in a padded compute layout), their gradients and every activation of the minibatch live in LDS, the Adam
// moments of a thread's parameters in its registers, and the phases of a step are separated by workgroup barriers only.
// The arithmetic is plain fp32 FMA on 4 x 4 (sample x unit) register tiles with 16-byte LDS operand reads -- at this size
// the chain is latency-bound, not throughput-bound, so the matrix cores would not help.
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

// padded compute layout of the parameters in LDS (row strides are multiples of 4 floats and == 4 mod 8, so that 16-byte
// operand reads of different rows spread over the banks)
template <int H>
struct PolicyFitLayout {
  int n, m, S1, S2;                 // S1: row stride of W1 (n inputs), S2: row stride of W2 / W3 (H inputs)
  int oW1, ob1, oW2, ob2, oW3, ob3, oS, P;     // offsets in the padded parameter block, P = its size
  int XS;                           // row stride of the normalised observation block
  __host__ __device__ PolicyFitLayout(int n_, int m_) {
    n = n_; m = m_;
    S1 = ((n + 3) & ~3) + 4; S2 = H + 4;
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
    const int MS = (m + 3) & ~3;
    size_t act = (size_t)B * (XS + 4 * (size_t)S2 + 3 * (size_t)MS + 4);   // X, H1, H2, D1, D2, MU, MUo, D3, per-row scalars
    return (size_t)P * (old_net ? 3 : 2) + act + 512 + (size_t)B * MS + 128;   // + reduction scratch, transform table, this step's actions, row ids
  }
};

// out[s][u] = f(sum_k in[s][k] W[u][k] + b[u]) for s < B, u < OUT on 4 x 4 register tiles; K a multiple of 4 is walked with
// 16-byte reads (rows padded), a ragged K (layer 1) element-wise.  ACT: 0 none, 1 tanh.
template <int ACT>
__device__ __forceinline__ void fit_layer(const float* __restrict__ in, int IS, const float* __restrict__ W, int WS,
                                          const float* __restrict__ b, int B, int OUT, int K, float* __restrict__ out, int OS,
                                          int tid) {
  const int ug_n = (OUT + 3) >> 2, tiles = (B >> 2) * ug_n;
  for (int t = tid; t < tiles; t += 256) {
    const int sg = t / ug_n, ug = t - sg * ug_n;
    const int s0 = 4 * sg;
    // the thread's four units are ug, ug + UG, ug + 2 UG, ug + 3 UG: neighbouring lanes read neighbouring weight rows
    // (row stride == 4 mod 64 banks), so the 16-byte reads do not collide
    int urow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) urow[j] = ug + ug_n * j;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int K4 = K & ~3;
    for (int k = 0; k < K4; k += 4) {
      f32x4 a[4], w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *(const f32x4*)&in[(s0 + i) * IS + k];
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = *(const f32x4*)&W[(urow[j] < OUT ? urow[j] : 0) * WS + k];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = fmaf(a[i].w, w[j].w, fmaf(a[i].z, w[j].z, fmaf(a[i].y, w[j].y, fmaf(a[i].x, w[j].x, acc[i][j]))));
    }
    for (int k = K4; k < K; ++k) {
      float a[4], w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = in[(s0 + i) * IS + k];
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = W[(urow[j] < OUT ? urow[j] : 0) * WS + k];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (urow[j] >= OUT) continue;
      const float bj = b[urow[j]];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = acc[i][j] + bj;
        out[(s0 + i) * OS + urow[j]] = (ACT == 1) ? fast_tanh(v) : v;      // (the fused kernels' tanh: branch-free, 6e-8)
      }
    }
  }
}

// delta_in[s][k] = (sum_u delta[s][u] W[u][k]) (1 - h[s][k]^2) for k < K (the layer's inputs), u < OUT
__device__ __forceinline__ void fit_back(const float* __restrict__ delta, int DS, const float* __restrict__ W, int WS, int B, int OUT,
                                         int K, const float* __restrict__ h, int HS, float* __restrict__ din, int DIS, int tid) {
  const int kg_n = K >> 2, tiles = (B >> 2) * kg_n;          // K is a multiple of 4 (hidden widths)
  for (int t = tid; t < tiles; t += 256) {
    const int sg = t / kg_n, kg = t - sg * kg_n;
    const int s0 = 4 * sg, k0 = 4 * kg;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4)(0.f);
    for (int u = 0; u < OUT; ++u) {
      const f32x4 w = *(const f32x4*)&W[u * WS + k0];
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float dv = delta[(s0 + i) * DS + u]; acc[i] += dv * w; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 hv = *(const f32x4*)&h[(s0 + i) * HS + k0];
      f32x4 o;
      o.x = acc[i].x * fmaf(-hv.x, hv.x, 1.0f); o.y = acc[i].y * fmaf(-hv.y, hv.y, 1.0f);
      o.z = acc[i].z * fmaf(-hv.z, hv.z, 1.0f); o.w = acc[i].w * fmaf(-hv.w, hv.w, 1.0f);
      *(f32x4*)&din[(s0 + i) * DIS + k0] = o;
    }
  }
}

// gW[u][k] = sum_s delta[s][u] in[s][k] (k < K), gb[u] = sum_s delta[s][u]; written in the padded parameter layout.
// Both operands are row-major over samples with 16-byte aligned rows padded with zeros, so a 4 x 4 tile reads one 16-byte
// granule of each per sample; four samples per trip keep eight independent reads in flight (the loop is latency-bound).
__device__ __forceinline__ void fit_wgrad(const float* __restrict__ delta, int DS, const float* __restrict__ in, int IS, int B, int OUT,
                                          int K, float* __restrict__ gW, int WS, float* __restrict__ gb, int tid) {
  const int kq = (K + 3) >> 2, tiles = ((OUT + 3) >> 2) * kq;
  for (int t = tid; t < tiles; t += 256) {
    const int ug = t / kq, kg = t - ug * kq;
    const int u0 = 4 * ug, k0 = 4 * kg;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4)(0.f);
    f32x4 bs = (f32x4)(0.f);
    for (int s = 0; s < B; s += 4) {
      f32x4 dv[4], xv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { dv[q] = *(const f32x4*)&delta[(s + q) * DS + u0]; xv[q] = *(const f32x4*)&in[(s + q) * IS + k0]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bs += dv[q];
        acc[0] += dv[q].x * xv[q]; acc[1] += dv[q].y * xv[q]; acc[2] += dv[q].z * xv[q]; acc[3] += dv[q].w * xv[q];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (u0 + i >= OUT) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) if (k0 + j < K) gW[(u0 + i) * WS + k0 + j] = acc[i][j];
      if (kg == 0) gb[u0 + i] = bs[i];
    }
  }
}

template <int H>
__global__ __launch_bounds__(256) void k_policy_fit(PolicyFitArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const PolicyFitLayout<H> L(A.n, A.m);
  const int tid = threadIdx.x, n = A.n, m = A.m, B = A.B, d = L.d();
  const int MS = (m + 3) & ~3, S1 = L.S1, S2 = L.S2, XS = L.XS;
  const bool old_net = (A.loss == 2) && !A.old_tracks_new;
  float* W = lds;                          // parameters (padded layout)
  float* G = W + L.P;                      // gradients (same layout)
  float* Wo = G + L.P;                     // old parameters (PPO with a fixed old policy)
  float* X = Wo + (old_net ? L.P : 0);     // [B][XS]
  float* H1 = X + (size_t)B * XS;          // [B][S2]
  float* H2 = H1 + (size_t)B * S2;
  float* D1 = H2 + (size_t)B * S2;
  float* D2 = D1 + (size_t)B * S2;
  float* MU = D2 + (size_t)B * S2;         // [B][MS]
  float* MUo = MU + (size_t)B * MS;
  float* D3 = MUo + (size_t)B * MS;
  float* ROW = D3 + (size_t)B * MS;        // [B][4]: adv, row weight w, -, -
  // (offsets are rounded as indices: a pointer -> integer -> pointer round trip would drop the LDS address space and turn
  //  every access below into a FLAT one that waits on the outstanding global prefetches)
  const int o_sh = (int)((ROW + (size_t)B * 4) - lds);
  double* sh = (double*)(lds + ((o_sh + 1) & ~1));            // 17 doubles of reduction scratch

  // ---- load parameters (zero pads), Adam moments of the parameters this thread owns
  for (int i = tid; i < A.lds_floats; i += 256) lds[i] = 0.f;      // incl. the zero pads every 16-byte operand read relies on
  __syncthreads();
  constexpr int EPT = 40;                  // d <= 256 * EPT  (64 x 64 with 63 observations and 16 actions: 9.4 k)
  int poff[EPT]; float am[EPT], av[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = tid + 256 * e;
    poff[e] = (i < d) ? L.pad_of(i) : 0;
    am[e] = (i < d) ? A.adam_m[i] : 0.f;
    av[e] = (i < d) ? A.adam_v[i] : 0.f;
    if (i < d) { W[poff[e]] = A.theta[i]; if (old_net) Wo[poff[e]] = A.theta_old[i]; }
  }
  // transforms: in_shift / in_scale (new and old), out_shift / out_scale (new and old), fixed old log_std
  float* TR = (float*)(sh + 17);           // [4][64] in-transforms, [4][16] out-transforms, [16] old log_std
  for (int f = tid; f < n; f += 256) { TR[f] = A.tr[f]; TR[64 + f] = A.tr[n + f]; TR[128 + f] = A.tr_old[f]; TR[192 + f] = A.tr_old[n + f]; }
  for (int a = tid; a < m; a += 256) {
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
  constexpr int XR = 16, AR = 4;           // B n <= 64 * 63 and B m <= 64 * 16 elements over 256 threads
  float xraw[XR], araw[AR], advraw = 0.f;
  int xsf[XR], asf[AR];                    // (row | column << 8) of the elements this thread moves, -1: none
#pragma unroll
  for (int c = 0; c < XR; ++c) { const int e = tid + 256 * c, r = e / n; xsf[c] = (e < B * n) ? (r | ((e - r * n) << 8)) : -1; }
#pragma unroll
  for (int c = 0; c < AR; ++c) { const int e = tid + 256 * c, r = e / m; asf[c] = (e < B * m) ? (r | ((e - r * m) << 8)) : -1; }
  float* ACT = (float*)(TR + 336);         // [B][MS] this step's actions
  int* IDXL = (int*)(ACT + (size_t)B * MS);        // [2][64] row ids of the next two steps
  auto fetch = [&](int64_t st) {
    const int* rows = IDXL + 64 * (int)(st & 1);
#pragma unroll
    for (int c = 0; c < XR; ++c) if (xsf[c] >= 0) xraw[c] = A.obs[(int64_t)rows[xsf[c] & 255] * n + (xsf[c] >> 8)];
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
    if (tid < B && step + 2 < A.steps) next_row = A.idx[(step + 2) * B + tid];
    // ---- 1. the minibatch: normalised observations, actions, advantages from the prefetched registers ----
#pragma unroll
    for (int c = 0; c < XR; ++c) {
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
    // ---- 2. forward (new parameters) ----
    fit_layer<1>(X, XS, W + L.oW1, S1, W + L.ob1, B, H, n, H1, S2, tid);
    __syncthreads();
    fit_layer<1>(H1, S2, W + L.oW2, S2, W + L.ob2, B, H, H, H2, S2, tid);
    __syncthreads();
    fit_layer<0>(H2, S2, W + L.oW3, S2, W + L.ob3, B, m, H, MU, MS, tid);
    __syncthreads();
    PFIT_STAMP(2);
    if (tid < B && step + 2 < A.steps) IDXL[64 * (int)(step & 1) + tid] = next_row;   // last read by fetch(step), a step ago
    // ---- 2b. old policy on the same rows (PPO with a fixed old policy): D1 / D2 are free until the backward pass ----
    if (old_net) {
      // D1 holds the old network's inputs in its first n columns (written with the minibatch above; n <= H, checked on the host)
      fit_layer<1>(D1, S2, Wo + L.oW1, S1, Wo + L.ob1, B, H, n, D2, S2, tid);
      __syncthreads();
      fit_layer<1>(D2, S2, Wo + L.oW2, S2, Wo + L.ob2, B, H, H, D1, S2, tid);
      __syncthreads();
      fit_layer<0>(D1, S2, Wo + L.oW3, S2, Wo + L.ob3, B, m, H, MUo, MS, tid);
      __syncthreads();
    }
    PFIT_STAMP(3);
    // ---- 3. loss head: one thread per row ----
    double lpart = 0.0;
    if (tid < B) {
      const int s = tid;
      float wrow = 0.f;
      if (A.loss == 0) {
        for (int a = 0; a < m; ++a) {
          const float mu = MU[s * MS + a] * TR[272 + a] + TR[256 + a];
          const float e = mu - ACT[s * MS + a];
          D3[s * MS + a] = TR[272 + a] * (2.0f * e * invB / (float)m);
          lpart += (double)e * (double)e;
        }
      } else {
        float lln = 0.f, llo = 0.f, sumn = 0.f, sumo = 0.f;
        for (int a = 0; a < m; ++a) {
          const float ls = W[L.oS + a], lso = TR[320 + a];
          const float x = ACT[s * MS + a];
          const float mu = MU[s * MS + a] * TR[272 + a] + TR[256 + a];
          const float zn = (x - mu) / expf(ls);
          lln = fmaf(-0.5f * zn, zn, lln); sumn += ls;
          if (A.loss == 2) {
            const float muo = old_net ? MUo[s * MS + a] * TR[304 + a] + TR[288 + a] : mu;
            const float zo = (x - muo) / expf(lso);
            llo = fmaf(-0.5f * zo, zo, llo); sumo += lso;
          }
        }
        lln = lln - sumn - llc;
        if (A.loss == 1) { wrow = -invB; lpart += (double)lln; }
        else {
          llo = llo - sumo - llc;
          const float LR = expf(lln - llo), ad = ROW[s * 4];
          const float s1 = LR * ad, s2 = fminf(fmaxf(LR, 1.0f - A.clip), 1.0f + A.clip) * ad;
          lpart += (double)fminf(s1, s2);
          const bool inside = (LR >= 1.0f - A.clip) && (LR <= 1.0f + A.clip);
          wrow = (inside || s1 < s2) ? -ad * LR * invB : 0.f;
        }
        for (int a = 0; a < m; ++a) {
          const float sg = expf(W[L.oS + a]);
          const float x = ACT[s * MS + a];
          const float mu = MU[s * MS + a] * TR[272 + a] + TR[256 + a];
          const float zn = (x - mu) / sg;
          D3[s * MS + a] = TR[272 + a] * (wrow * zn / sg);
          MUo[s * MS + a] = wrow * (zn * zn - 1.0f);          // per-row contribution to dLoss/dlog_std (MUo is free now)
        }
      }
      ROW[s * 4 + 1] = wrow;
    }
    if (A.loss_trace) {
      const double lt = block_sum(lpart, sh);
      if (tid == 0) A.loss_trace[step] = (A.loss == 0) ? lt / ((double)B * (double)m) : -lt / (double)B;
    }
    __syncthreads();
    PFIT_STAMP(4);
    // ---- 4. backward ----
    fit_wgrad(D3, MS, H2, S2, B, m, H, G + L.oW3, S2, G + L.ob3, tid);
    fit_back(D3, MS, W + L.oW3, S2, B, m, H, H2, S2, D2, S2, tid);
    if (A.loss != 0 && tid < 8 * m) {                     // dLoss/dlog_std[a] = sum_rows w (z^2 - 1): 8 partial sums per action ...
      const int a = tid >> 3, part = tid & 7;
      float g = 0.f;
      for (int s = part; s < B; s += 8) g += MUo[s * MS + a];
      MU[a * 8 + part] = g;                               // (the means are dead once D3 holds dLoss/dmean; B * MS >= 8 m for B >= 8)
    }
    __syncthreads();
    if (A.loss != 0 && tid < m) {                         // ... added in a fixed order
      float g = 0.f;
#pragma unroll
      for (int part = 0; part < 8; ++part) g += MU[tid * 8 + part];
      G[L.oS + tid] = g;
    }
    PFIT_STAMP(5);
    fit_wgrad(D2, S2, H1, S2, B, H, H, G + L.oW2, S2, G + L.ob2, tid);
    fit_back(D2, S2, W + L.oW2, S2, B, H, H, H1, S2, D1, S2, tid);
    __syncthreads();
    PFIT_STAMP(6);
    fit_wgrad(D1, S2, X, XS, B, H, n, G + L.oW1, S1, G + L.ob1, tid);
    __syncthreads();
    PFIT_STAMP(7);
    // ---- 5. torch.optim.Adam on the parameters this thread owns (MSE: log_std has no gradient -> untouched) ----
    pw1 *= (double)b1c; pw2 *= (double)b2c;
    const float bc1 = (float)(1.0 - pw1), bc2s = (float)sqrt(1.0 - pw2);
    const int dlim = (A.loss == 0) ? d - m : d;
    const float inv_bc2s = 1.0f / bc2s, step_size = A.lr / bc1;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = tid + 256 * e;
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
  // ---- write back ----
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = tid + 256 * e;
    if (i < d) { A.theta[i] = W[poff[e]]; A.adam_m[i] = am[e]; A.adam_v[i] = av[e]; }
  }
}

}  // namespace mjx
