// baseline.h -- value-baseline kernels (K6): feature generation, fp64 normal equations for the
// linear / quadratic baselines, batched prediction, and the minibatch-Adam MLP regressor.
//
// Reference semantics:
//   mjrl/baselines/mlp_baseline.py:36-105   (features, fit via utils/optimize_model.py:7-36, predict)
//   mjrl/baselines/quadratic_baseline.py:11-74, linear_baseline.py:11-65 (features, ridge normal equations)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "layerwise.h"
#include "vecops.h"

namespace mjx {

enum { FEAT_MLP = 0, FEAT_LINEAR = 1, FEAT_QUADRATIC = 2 };

__host__ __device__ inline int bl_num_features(int kind, int n) {
  if (kind == FEAT_MLP) return n + 4;
  if (kind == FEAT_LINEAR) return n + 5;
  return n + n * (n + 1) / 2 + 5;
}

// feature c of one sample.  o[] = clip(obs, -10, 10) / 10, tau = t / 1000.
// layouts: MLP [o, tau..tau^4]; LINEAR [o, 1, tau..tau^4]; QUADRATIC [o, o_i*o_j (i<=j, row-major), 1, tau..tau^4]
struct FeatDesc { int16_t p, q; };   // p>=0,q<0: o[p];  p>=0,q>=0: o[p]*o[q];  p==-1: 1;  p==-2: tau^q

__device__ __forceinline__ double feat_value(FeatDesc fd, const double* o, double tau) {
  if (fd.p >= 0) return fd.q >= 0 ? o[fd.p] * o[fd.q] : o[fd.p];
  if (fd.p == -1) return 1.0;
  double t = tau;
  for (int k = 1; k < fd.q; ++k) t *= tau;          // tau^q by repeated product; numpy uses pow: agree to ~1 ulp
  return t;
}

inline std::vector<FeatDesc> build_feat_table(int kind, int n) {
  std::vector<FeatDesc> t;
  for (int i = 0; i < n; ++i) t.push_back({(int16_t)i, -1});
  if (kind == FEAT_QUADRATIC)
    for (int i = 0; i < n; ++i)
      for (int j = i; j < n; ++j) t.push_back({(int16_t)i, (int16_t)j});
  if (kind != FEAT_MLP) t.push_back({-1, -1});
  for (int k = 1; k <= 4; ++k) t.push_back({-2, (int16_t)k});
  return t;
}

// fp32 feature matrix of the MLP baseline (computed in fp64, then cast: mlp_baseline.py:65)
__global__ void k_bl_features_f32(const double* __restrict__ obs, const int32_t* __restrict__ tpos, int64_t N, int n,
                                  float* __restrict__ out) {
  const int F = n + 4;
  const int64_t tot = N * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t s = i / F; int c = (int)(i - s * F);
    double v;
    if (c < n) v = fmin(fmax(obs[s * n + c], -10.0), 10.0) / 10.0;
    else { double tau = (double)tpos[s] / 1000.0; v = pow(tau, (double)(c - n + 1)); }
    out[i] = (float)v;
  }
}

// ---- augmented Gram matrix  G' = [A y]^T [A y]  in fp64, features generated on the fly ------------
// grid: (tile pairs bi<=bj, sample chunks).  64x64 output tile per workgroup, 256 threads x 4x4 outputs.
constexpr int GT = 64, GKS = 32;
__global__ __launch_bounds__(256) void k_bl_gram(int nbt, const FeatDesc* __restrict__ table, int F, int n,
                                                 const double* __restrict__ obs, const int32_t* __restrict__ tpos,
                                                 const double* __restrict__ y, int64_t N, double* __restrict__ part) {
  extern __shared__ double sm[];
  double* so = sm;                       // [GKS][n] clipped obs
  double* fi = so + GKS * n;             // [GKS][GT+1]
  double* fj = fi + GKS * (GT + 1);      // [GKS][GT+1]
  __shared__ double stau[GKS];
  // decode the upper-triangular tile pair
  int pid = blockIdx.x, bi = 0;
  while (pid >= nbt - bi) { pid -= nbt - bi; ++bi; }
  const int bj = bi + pid;
  const int FA = F + 1;                  // augmented with y
  const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
  double acc[4][4] = {{0}};
  const int64_t chunk = (N + gridDim.y - 1) / gridDim.y;
  const int64_t lo = blockIdx.y * chunk, hi = (lo + chunk < N) ? lo + chunk : N;
  for (int64_t s0 = lo; s0 < hi; s0 += GKS) {
    const int ks = (int)((hi - s0 < GKS) ? hi - s0 : GKS);
    for (int i = tid; i < ks * n; i += 256) so[i] = fmin(fmax(obs[s0 * n + i], -10.0), 10.0) / 10.0;
    if (tid < ks) stau[tid] = (double)tpos[s0 + tid] / 1000.0;
    __syncthreads();
    for (int i = tid; i < GKS * GT; i += 256) {
      const int k = i / GT, c = i - k * GT;
      double vi = 0.0, vj = 0.0;
      if (k < ks) {
        const int ci = bi * GT + c, cj = bj * GT + c;
        if (ci < F) vi = feat_value(table[ci], so + k * n, stau[k]); else if (ci == F) vi = y[s0 + k];
        if (cj < F) vj = feat_value(table[cj], so + k * n, stau[k]); else if (cj == F) vj = y[s0 + k];
      }
      fi[k * (GT + 1) + c] = vi; fj[k * (GT + 1) + c] = vj;
    }
    __syncthreads();
    for (int k = 0; k < GKS; ++k) {
      double a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = fi[k * (GT + 1) + tr * 4 + u]; b[u] = fj[k * (GT + 1) + tc * 4 + u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int w = 0; w < 4; ++w) acc[u][w] = fma(a[u], b[w], acc[u][w]);
    }
    __syncthreads();
  }
  double* out = part + (size_t)blockIdx.y * FA * FA;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int r = bi * GT + tr * 4 + u, c = bj * GT + tc * 4 + w;
      if (r < FA && c < FA) out[(size_t)r * FA + c] = acc[u][w];
    }
}

// The same on the fp64 matrix cores for up to 176 augmented features (obs dim <= 17 quadratic, any linear baseline of that
// size): v_mfma_f64_16x16x4_f64, lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15] and receives
// D[4 r + (l >> 4)][l & 15] in register r (probed: not the 4 (l >> 4) + r of the fp32 16x16x4 form).  One workgroup walks its own sample range in chunks of 32: ALL features of a chunk are
// generated once into LDS ([sample][feature], row stride == 16 (mod 32) doubles: the 8-byte operand reads of the two
// 32-lane halves hit disjoint banks), then the 4 waves update the upper-triangle 16 x 16 tiles they own (round-robin,
// <= 17 accumulator tiles per wave, registers for the whole run).  The FMA kernel above re-generates a 64-column block
// per tile pair and is bound by LDS operand bandwidth (8 x 8 bytes per 16 FMAs): 5.0 ms at 1M x 175; this one ~0.7 ms.
typedef double f64x4 __attribute__((ext_vector_type(4)));
// (p, q) with column c == e[p] * e[q] for the sample's extended vector e = [o_0 .. o_{n-1}, 1, tau, tau^2, tau^3, tau^4, y, 0]
__device__ __forceinline__ void gb_pair(FeatDesc fd, int c, int F, int n, int& p, int& q) {
  const int one = n, zero = n + 6;
  if (c > F) { p = zero; q = one; return; }
  if (c == F) { p = n + 5; q = one; return; }
  if (fd.p >= 0) { p = fd.p; q = fd.q >= 0 ? fd.q : one; return; }
  if (fd.p == -1) { p = one; q = one; return; }
  p = n + fd.q; q = one;                                   // tau^q sits at n + q
}

constexpr int GM_TMAX = 11, GM_TPW = (GM_TMAX * (GM_TMAX + 1) / 2 + 3) / 4;      // 66 tiles / 4 waves -> 17
__global__ __launch_bounds__(256, 2) void k_bl_gram_mfma(const FeatDesc* __restrict__ table, int F, int n,
                                                         const double* __restrict__ obs, const int32_t* __restrict__ tpos,
                                                         const double* __restrict__ y, int64_t N, double* __restrict__ part) {
  extern __shared__ double sm[];
  const int FA = F + 1, T = (FA + 15) >> 4, FS = 16 * (T | 1), NTILE = T * (T + 1) / 2, NE = n + 7;
  double* so = sm;                       // [32][NE] extended vectors [o, 1, tau..tau^4, y, 0]
  double* ft = so + 32 * NE;             // [32][FS] features (+ y in column F, zeros beyond)
  __shared__ int16_t tile_i[GM_TMAX * (GM_TMAX + 1) / 2], tile_j[GM_TMAX * (GM_TMAX + 1) / 2];
  __shared__ int16_t spq[16 * GM_TMAX][2];   // column c == e[spq[c][0]] * e[spq[c][1]]: branch-free generation (see gb_pair)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, q = lane >> 4;
  if (tid < NTILE) {                     // row-major upper triangle
    int t = tid, i = 0;
    while (t >= T - i) { t -= T - i; ++i; }
    tile_i[tid] = (int16_t)i; tile_j[tid] = (int16_t)(i + t);
  }
  if (tid < 16 * T) { int p_, q_; gb_pair(table[tid < F ? tid : 0], tid, F, n, p_, q_); spq[tid][0] = (int16_t)p_; spq[tid][1] = (int16_t)q_; }
  __syncthreads();
  int ai[GM_TPW], bj[GM_TPW];            // this wave's tiles: LDS column offsets of the A / B operand of tile slot e
  f64x4 acc[GM_TPW];
#pragma unroll
  for (int e = 0; e < GM_TPW; ++e) {
    const int t = wave + 4 * e;
    ai[e] = 16 * tile_i[t < NTILE ? t : 0] + r16; bj[e] = 16 * tile_j[t < NTILE ? t : 0] + r16;
    acc[e] = (f64x4)(0.0);
  }
  const int nmine = wave < NTILE ? (NTILE - wave + 3) >> 2 : 0;
  int64_t chunk = (N + gridDim.x - 1) / gridDim.x;
  chunk = (chunk + 31) & ~(int64_t)31;
  const int64_t lo = blockIdx.x * chunk, hi = (lo + chunk < N) ? lo + chunk : N;
  // the next chunk's observations / time index / targets travel to registers under the current chunk's MFMAs
  constexpr int OPT = 3;                                       // 32 * n <= 256 * OPT: n <= 24 (T <= 11 means n <= 17 quadratic; linear: checked by the launcher)
  double ro[OPT], rtau = 0.0, ry = 0.0;
  auto prefetch = [&](int64_t s0) {
    const int ks = (int)((hi - s0 < 32) ? hi - s0 : 32);
#pragma unroll
    for (int c = 0; c < OPT; ++c) {
      const int i = tid + 256 * c, k = i / n;
      ro[c] = (i < 32 * n && k < ks) ? obs[s0 * n + i] : 0.0;
    }
    if (tid < 32) { rtau = tid < ks ? (double)tpos[s0 + tid] / 1000.0 : 0.0; ry = tid < ks ? y[s0 + tid] : 0.0; }
  };
  if (lo < hi) prefetch(lo);
  const int FC = 16 * T;
  for (int64_t s0 = lo; s0 < hi; s0 += 32) {
    const int ks = (int)((hi - s0 < 32) ? hi - s0 : 32);
#pragma unroll
    for (int c = 0; c < OPT; ++c) {
      const int i = tid + 256 * c;
      if (i < 32 * n) { const int k = i / n, f = i - k * n; so[k * NE + f] = fmin(fmax(ro[c], -10.0), 10.0) / 10.0; }
    }
    if (tid < 32) {
      const bool in = tid < ks;
      double* e = so + tid * NE + n;
      e[0] = in ? 1.0 : 0.0;
      double t = rtau;
      e[1] = t; t *= rtau; e[2] = t; t *= rtau; e[3] = t; t *= rtau; e[4] = t;
      e[5] = ry;
      e[6] = 0.0;
    }
    __syncthreads();
    if (s0 + 32 < hi) prefetch(s0 + 32);
    for (int i = tid; i < 32 * FC; i += 256) {
      const int k = i / FC, c = i - k * FC;
      const double* e = so + k * NE;
      ft[k * FS + c] = e[spq[c][0]] * e[spq[c][1]];
    }
    __syncthreads();
    for (int k0 = 0; k0 < 32; k0 += 4) {
      const double* row = ft + (k0 + q) * FS;
#pragma unroll
      for (int e = 0; e < GM_TPW; ++e)
        if (e < nmine) acc[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(row[ai[e]], row[bj[e]], acc[e], 0, 0, 0);
    }
    __syncthreads();
  }
  double* out = part + (size_t)blockIdx.x * FA * FA;
#pragma unroll
  for (int e = 0; e < GM_TPW; ++e) {
    const int c = bj[e];                                   // = 16 tj + (lane & 15)
    const int r0 = ai[e] - r16 + q;                        // 16 ti + (lane >> 4): register r holds row 4 r + (lane >> 4)
#pragma unroll
    for (int r = 0; r < 4; ++r) if (e < nmine && r0 + 4 * r < FA && c < FA) out[(size_t)(r0 + 4 * r) * FA + c] = acc[e][r];
  }
}

// More than 176 augmented features (the quadratic baseline of BASELINE configs[4]: obs 39 -> 824 features, whose N x F
// feature matrix would be 52.7 GB per 8M timesteps): the (F+1) x (F+1) matrix is cut into 128 x 128 feature blocks and
// one workgroup owns one upper-triangle block PAIR (bi <= bj) over one sample range.  Per 32-sample chunk it generates the
// 2 x 128 features it needs into LDS once ([sample][feature], row stride 144 doubles == 16 mod 32), then its 4 waves run
// the 8 x 8 tiles of v_mfma_f64_16x16x4_f64 (wave w: tile rows 2w, 2w+1 x all 8 tile columns: 16 accumulator tiles = 128
// VGPRs for the whole run; per 4-sample step 2 + 8 operand reads for 16 MFMAs).  Same operand / result mapping as above.
// Diagonal pairs compute their whole block (the reduction reads block (bi, bj) with bi <= bj and mirrors).
constexpr int GB_F = 128, GB_FS = GB_F + 16;
// Feature generation is branch-free: every column is a product e[p] * e[q] of two entries of the sample's EXTENDED vector
// e = [o_0 .. o_{n-1}, 1, tau, tau^2, tau^3, tau^4, y, 0] (o_p * 1 for a linear column, 1 * 1 for the constant, y * 1 for
// the augmented column, 0 * 1 beyond it) -- bit-identical to feat_value(), and the two index pairs of a thread's two
// columns are formed once: per value two LDS reads, one multiply, one LDS write (the first version looked the descriptor
// up and branched per value and spent more time generating features than multiplying them).
__global__ __launch_bounds__(256, 1) void k_bl_gram_mfma_blk(int nb, const FeatDesc* __restrict__ table, int F, int n,
                                                             const double* __restrict__ obs, const int32_t* __restrict__ tpos,
                                                             const double* __restrict__ y, int64_t N, double* __restrict__ part) {
  extern __shared__ double sm[];
  const int NE = n + 7;
  double* so = sm;                       // [32][NE] extended vectors
  double* fi = so + 32 * NE;             // [32][GB_FS] features of block bi
  double* fj = fi + 32 * GB_FS;          // [32][GB_FS] features of block bj (unused on the diagonal)
  int pid = blockIdx.x, bi = 0;          // row-major upper triangle of block pairs
  while (pid >= nb - bi) { pid -= nb - bi; ++bi; }
  const int bj = bi + pid;
  const bool diag = bi == bj;
  const double* fjr = diag ? fi : fj;
  const int FA = F + 1;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, q = lane >> 4;
  f64x4 acc[2][8];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[e][t] = (f64x4)(0.0);
  const int cl = tid & (GB_F - 1), kh = tid >> 7;
  const int ci = GB_F * bi + cl, cj = GB_F * bj + cl;
  int pi, qi, pj, qj;
  gb_pair(table[ci < F ? ci : 0], ci, F, n, pi, qi);
  gb_pair(table[cj < F ? cj : 0], cj, F, n, pj, qj);
  int64_t chunk = (N + gridDim.y - 1) / gridDim.y;
  chunk = (chunk + 31) & ~(int64_t)31;
  const int64_t lo = blockIdx.y * chunk, hi = (lo + chunk < N) ? lo + chunk : N;
  // the next chunk's observations / time index / targets are requested before the current chunk's MFMA loop and land in
  // registers under it (one wave per SIMD: a global-load round trip per chunk would otherwise sit in the open)
  constexpr int OPT = 8;                                       // 32 * n <= 256 * OPT, i.e. n <= 64 (checked by the launcher)
  double ro[OPT], rtau = 0.0, ry = 0.0;
  auto prefetch = [&](int64_t s0) {
    const int ks = (int)((hi - s0 < 32) ? hi - s0 : 32);
#pragma unroll
    for (int c = 0; c < OPT; ++c) {
      const int i = tid + 256 * c, k = i / n;
      ro[c] = (i < 32 * n && k < ks) ? obs[s0 * n + i] : 0.0;
    }
    if (tid < 32) { rtau = tid < ks ? (double)tpos[s0 + tid] / 1000.0 : 0.0; ry = tid < ks ? y[s0 + tid] : 0.0; }
  };
  if (lo < hi) prefetch(lo);
  for (int64_t s0 = lo; s0 < hi; s0 += 32) {
    const int ks = (int)((hi - s0 < 32) ? hi - s0 : 32);
#pragma unroll
    for (int c = 0; c < OPT; ++c) {
      const int i = tid + 256 * c;
      if (i < 32 * n) { const int k = i / n, f = i - k * n; so[k * NE + f] = fmin(fmax(ro[c], -10.0), 10.0) / 10.0; }
    }
    if (tid < 32) {
      const bool in = tid < ks;
      double* e = so + tid * NE + n;
      e[0] = in ? 1.0 : 0.0;                                  // (rows past the range end contribute zeros)
      double t = rtau;
      e[1] = t; t *= rtau; e[2] = t; t *= rtau; e[3] = t; t *= rtau; e[4] = t;  // tau^k by repeated product, like feat_value()
      e[5] = ry;
      e[6] = 0.0;
    }
    __syncthreads();
    if (s0 + 32 < hi) prefetch(s0 + 32);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int k = 2 * kk + kh;
      const double* e = so + k * NE;
      fi[k * GB_FS + cl] = e[pi] * e[qi];
      if (!diag) fj[k * GB_FS + cl] = e[pj] * e[qj];
    }
    __syncthreads();
#pragma unroll 2
    for (int k0 = 0; k0 < 32; k0 += 4) {
      const double* ra = fi + (k0 + q) * GB_FS + r16;
      const double* rb = fjr + (k0 + q) * GB_FS + r16;
      const double a0 = ra[16 * (2 * wave)], a1 = ra[16 * (2 * wave + 1)];
      double b[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) b[t] = rb[16 * t];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[t], acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[t], acc[1][t], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  double* out = part + (size_t)blockIdx.y * FA * FA;
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int c = GB_F * bj + 16 * t + r16;
      const int r0 = GB_F * bi + 16 * (2 * wave + e) + q;        // register r holds row 4 r + (lane >> 4) of the tile
#pragma unroll
      for (int r = 0; r < 4; ++r) if (r0 + 4 * r < FA && c < FA) out[(size_t)(r0 + 4 * r) * FA + c] = acc[e][t][r];
    }
}

// G[r][c] = sum_z part[z][min][max]  (upper-triangle tiles of TS x TS were computed; mirror)
__global__ void k_bl_gram_reduce(const double* __restrict__ part, int Z, int FA, double* __restrict__ G, int TS) {
  const int64_t tot = (int64_t)FA * FA;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(i / FA), c = (int)(i - (int64_t)r * FA);
    int rr = r, cc = c;
    if (r / TS > c / TS) { rr = c; cc = r; }       // tile (bi > bj) was not computed: take the mirrored element
    double a = 0.0;
    for (int z = 0; z < Z; ++z) a += part[(size_t)z * tot + (size_t)rr * FA + cc];
    G[i] = a;
  }
}

// out[s] = sum_c feat(s, c) * coef[c]   (fp64; quadratic_baseline.py:71-74)
__global__ __launch_bounds__(256) void k_bl_predict(const FeatDesc* __restrict__ table, int F, int n,
                                                    const double* __restrict__ obs, const int32_t* __restrict__ tpos,
                                                    const double* __restrict__ coef, int64_t N, double* __restrict__ out) {
  extern __shared__ double sm[];
  double* o = sm + (size_t)threadIdx.x * n;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < N; s += (int64_t)gridDim.x * blockDim.x) {
    for (int i = 0; i < n; ++i) o[i] = fmin(fmax(obs[s * n + i], -10.0), 10.0) / 10.0;
    const double tau = (double)tpos[s] / 1000.0;
    double a = 0.0;
    for (int c = 0; c < F; ++c) a += feat_value(table[c], o, tau) * coef[c];
    out[s] = a;
  }
}

// ---- MLP regressor (ReLU) : forward / minibatch Adam --------------------------------------------
__global__ void k_gather_rows(const float* __restrict__ X, const float* __restrict__ y, const int32_t* __restrict__ idx,
                              int bs, int dcols, float* __restrict__ Xb, float* __restrict__ yb) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < bs * dcols; i += gridDim.x * blockDim.x) {
    int r = i / dcols, c = i - r * dcols;
    Xb[i] = X[(int64_t)idx[r] * dcols + c];
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < bs; i += gridDim.x * blockDim.x) yb[i] = y[idx[i]];
}

// d = 2 (yhat - y) / bs   (MSELoss mean, optimize_model.py:31) ; loss_acc[0] += mean((yhat-y)^2)
__global__ void k_mse_grad(const float* __restrict__ yhat, const float* __restrict__ y, int bs, float* __restrict__ d,
                           double* __restrict__ loss_acc) {
  __shared__ double sh[17];
  double l = 0.0;
  for (int i = threadIdx.x; i < bs; i += blockDim.x) {
    float e = yhat[i] - y[i];
    d[i] = 2.0f * e / (float)bs;
    l += (double)e * (double)e;
  }
  l = block_sum(l, sh);
  if (threadIdx.x == 0) loss_acc[0] += l / (double)bs;
}

// torch.optim.Adam (non-amsgrad, L2 weight decay folded into the gradient), fp32 state
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       int64_t cnt, float lr, float wd, float b1, float b2, float eps, float bc1, float bc2_sqrt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i] + wd * p[i];
    float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    float vi = v[i] * b2 + gi * gi * (1.0f - b2);
    m[i] = mi; v[i] = vi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - (lr / bc1) * (mi / denom);
  }
}

struct MlpRegressor {
  std::vector<int> sizes;                 // d_in, h..., 1
  std::vector<int64_t> oW, ob;
  int64_t P = 0;
  int nL() const { return (int)sizes.size() - 1; }
  void init(int d_in, const int* hidden, int nh) {
    sizes.clear(); sizes.push_back(d_in); for (int i = 0; i < nh; ++i) sizes.push_back(hidden[i]); sizes.push_back(1);
    oW.clear(); ob.clear(); int64_t k = 0;
    for (int l = 0; l < nL(); ++l) { oW.push_back(k); k += (int64_t)sizes[l] * sizes[l + 1]; ob.push_back(k); k += sizes[l + 1]; }
    P = k;
  }
  // acts[l] (rows x sizes[l+1]) for hidden layers; out (rows)
  void forward(const float* params, const float* X, int64_t rows, const std::vector<float*>& acts, float* out, hipStream_t st) {
    const float* in = X;
    for (int l = 0; l < nL(); ++l) {
      const bool last = l == nL() - 1;
      GemmArgs g{};
      g.M = (int)rows; g.N = sizes[l + 1]; g.npairs = 1; g.K[0] = sizes[l];
      g.A[0] = in; g.a_rs[0] = sizes[l]; g.a_ks[0] = 1;
      g.B[0] = params + oW[l]; g.b_cs[0] = sizes[l]; g.b_ks[0] = 1;
      g.C = last ? out : acts[l]; g.ldc = sizes[l + 1]; g.c_zs = 0;
      g.bias = params + ob[l];
      g.epi = last ? EPI_BIAS : EPI_BIAS_RELU;
      LayerwiseWS::launch_gemm(g, 1, st);
      in = g.C;
    }
  }
};

// ---- the value network of mjrl's MLPBaseline, (n + 4) -> 128 -> 128 -> 1 ReLU (mjrl/baselines/mlp_baseline.py:21-28), evaluated in
// ONE launch (r06).  The layer-by-layer route (MlpRegressor::forward: 3 GEMM launches per 131 072-row chunk) spent 1.3 ms per 1M
// timesteps, most of it in the K = 21 first layer (rows of 84 bytes: neither 16-byte granules nor a whole 32-wide k-tile, i.e. the
// general kernel's element-wise loads) -- and baseline.predict sits on train_step's critical path (compute_advantages).  Same scheme as
// the fused policy kernels: units on the MFMA M / K dims, the tile's 32 samples on N, so layer 1's accumulators ARE layer 2's B
// operands; weights in LDS, read as one ds_read_b128 per four k-steps in the k-permuted order (k-slot `hi` of step (q, t) carries
// k = 8 q + 4 hi + t).  One wave = one 32-sample tile at a time; 8 waves per workgroup (two per SIMD), one workgroup per CU.
// Per tile: 4 x K1 / 2 + 256 MFMAs of 32x32x2; 1M timesteps: 8.9 GF on the matrix pipe.  d_in <= 64.
struct MlpPredictArgs { const float* feat; const float* params; float* out; int64_t N; int d_in; };
constexpr int MP_H = 128, MP_S2 = MP_H + 4;
__host__ __device__ inline int mp_k1(int d_in) { return (d_in + 7) & ~7; }
__host__ __device__ inline size_t mlp_predict_lds_bytes(int d_in) { return sizeof(float) * (size_t)(MP_H * (mp_k1(d_in) + 4) + MP_H * MP_S2 + 3 * MP_H + 4); }

__global__ __launch_bounds__(512, 1) void k_mlp_predict128(MlpPredictArgs a) {
  extern __shared__ __attribute__((aligned(16))) float mps[];
  const int d_in = a.d_in, K1 = mp_k1(d_in), S1 = K1 + 4, NQ = K1 / 8;
  float* W1s = mps;                       // [128][S1]   (columns d_in .. K1 - 1 zero)
  float* W2s = W1s + MP_H * S1;           // [128][132]
  float* b1s = W2s + MP_H * MP_S2;        // [128]
  float* b2s = b1s + MP_H;
  float* w3s = b2s + MP_H;
  float* b3s = w3s + MP_H;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t oB1 = (int64_t)MP_H * d_in, oW2 = oB1 + MP_H, oB2 = oW2 + (int64_t)MP_H * MP_H, oW3 = oB2 + MP_H, oB3 = oW3 + MP_H;
  for (int i = tid; i < MP_H * S1; i += 512) { const int u = i / S1, f = i - u * S1; W1s[i] = f < d_in ? a.params[(int64_t)u * d_in + f] : 0.f; }
  for (int i = tid; i < MP_H * MP_H / 4; i += 512) {
    const int u = (4 * i) / MP_H, c = (4 * i) % MP_H;
    *(f32x4*)&W2s[u * MP_S2 + c] = *(const f32x4*)&a.params[oW2 + 4 * (int64_t)i];       // (W2 starts 16-byte aligned when d_in * 128 + 128 is a multiple of 4: always)
  }
  if (tid < MP_H) { b1s[tid] = a.params[oB1 + tid]; b2s[tid] = a.params[oB2 + tid]; w3s[tid] = a.params[oW3 + tid]; }
  if (tid == 0) b3s[0] = a.params[oB3];
  __syncthreads();
  const float b3 = b3s[0];
  const int64_t ntiles = (a.N + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 8) {
    const int64_t s0 = tile * 32 + j;
    const bool valid = s0 < a.N;
    const float* __restrict__ xr = a.feat + (valid ? s0 : 0) * (int64_t)d_in;
    // layer 1: this lane's features in the k-permuted order of its half
    f32x16 h1[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) h1[mb] = (f32x16)(0.f);
    for (int q = 0; q < NQ; ++q) {
      float xb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { const int f = 8 * q + 4 * hi + t; const float v = xr[f < d_in ? f : 0]; xb[t] = (valid && f < d_in) ? v : 0.f; }
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const f32x4 a4 = *(const f32x4*)&W1s[(32 * mb + j) * S1 + 8 * q + 4 * hi];
#pragma unroll
        for (int t = 0; t < 4; ++t) h1[mb] = MJX_MFMA(a4[t], xb[t], h1[mb]);
      }
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) h1[mb][r] = fmaxf(h1[mb][r] + b1s[32 * mb + unit_of(r, hi)], 0.f);
    // layer 2 + output: register r of h1[mb] holds unit 32 mb + unit_of(r, hi) -- the B operand of k-step (mb, r) as it lies
    float ysum = 0.f;
#pragma unroll 1
    for (int ob = 0; ob < 4; ++ob) {
      f32x16 acc = (f32x16)(0.f);
      const float* wrow = &W2s[(32 * ob + j) * MP_S2 + 4 * hi];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const f32x4 a4 = *(const f32x4*)(wrow + 32 * mb + 8 * rq);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc = MJX_MFMA(a4[t], h1[mb][4 * rq + t], acc);
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int u = 32 * ob + unit_of(r, hi);
        ysum = fmaf(fmaxf(acc[r] + b2s[u], 0.f), w3s[u], ysum);
      }
    }
    const float y = half_sum(ysum) + b3;
    if (hi == 0 && valid) a.out[s0] = y;
  }
}

}  // namespace mjx
