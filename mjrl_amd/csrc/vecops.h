// vecops.h -- small deterministic reductions, the device-side CG bookkeeping (K4) and the
// trajectory scans (K5).  All reductions use a fixed order (no atomics) so that results do
// not depend on dispatch order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mjx {

// Sum over the 64 lanes of a wave, every lane gets the total.  Register-level cross-lane moves only (DPP inside the
// 16-lane rows, v_permlane16_swap / v_permlane32_swap across them): the ds_bpermute butterfly this replaces cost
// twelve LDS-crossbar round trips per fp64 sum.  Fixed order, identical on every lane.
template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
// the value of lane ^ 16 (ROWS == 16) or lane ^ 32 (ROWS == 32): after a swap of a register with itself every lane
// holds {own, partner} in the two results, so partner = r0 ^ r1 ^ own
template <int ROWS>
__device__ __forceinline__ double partner_f64(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
  unsigned plo, phi;
  if (ROWS == 16) {
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto c = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    plo = a[0] ^ a[1] ^ lo; phi = c[0] ^ c[1] ^ hi;
  } else {
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto c = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    plo = a[0] ^ a[1] ^ lo; phi = c[0] ^ c[1] ^ hi;
  }
  return __longlong_as_double((long long)(((unsigned long long)phi << 32) | plo));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move_f64<0xB1>(v);       // quad_perm [1,0,3,2]
  v += dpp_move_f64<0x4E>(v);       // quad_perm [2,3,0,1]
  v += dpp_move_f64<0x141>(v);      // row_half_mirror
  v += dpp_move_f64<0x140>(v);      // row_mirror
  v += partner_f64<16>(v);
  v += partner_f64<32>(v);
  return v;
}

// sum over a whole (<=1024-thread) block; every thread gets the result
__device__ __forceinline__ double block_sum(double v, double* sh /* >= 17 doubles */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double t = 0;                                   // every thread sums the wave totals in the same fixed order
  for (int i = 0; i < nw; ++i) t += sh[i];
  return t;
}

// out[c] = sum_g partials[g][c]  (fp64 accumulate, fixed order).  16 columns x 16 row-groups per
// 256-thread block (d/16 blocks: enough workgroups to pull the 5.8 MB of partials in a few us).
// FVP epilogue: the log_std block of the Hessian is diagonal, H_ss = c(sigma); its
// per-sample mean contributes frac*c*v_s locally.
__global__ __launch_bounds__(256) void k_reduce_partials(const float* __restrict__ partials, int G, int d,
                                                          float* __restrict__ out, const float* theta,
                                                          const float* v, int oS, float frac) {
  __shared__ double sh[16][17];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double acc = 0.0;
  if (c < d)
    for (int g = rg; g < G; g += 16) acc += (double)partials[(size_t)g * d + c];
  sh[rg][cl] = acc;
  __syncthreads();
  if (rg == 0 && c < d) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k][cl];
    if (v != nullptr && c >= oS) {
      float s = expf(theta[c]);
      float u = s * s, e = 1e-8f;
      float den = 2.0f * u + e;
      float cc = 16.0f * u * u / (den * den) - 4.0f * u / den;
      t = (double)(frac * cc * v[c]);
    }
    out[c] = (float)t;
  }
}

// Peer exchange (mjx_peer_*).  Every rank owns one uncached buffer [2 parities][world slots] + one arrival flag PER SOURCE RANK;
// the peers' buffers are mapped through hipIpcOpenMemHandle.  An all-reduce = every rank WRITES its vector into slot `rank` of
// every buffer (posted stores over xGMI), then stores the exchange number into ITS flag in every peer's buffer; the consumer
// kernel waits (bounded) on the flags of its OWN buffer -- local memory, one polling thread per source rank -- and sums its local
// slots in rank order: the same bits on every rank, no host in the loop.  A flag is written by exactly one rank over the same
// path as that rank's data stores, after they were acknowledged: no ordering between DIFFERENT peers' traffic is relied upon
// (an aggregate counter would need it for world > 2).
struct PeerSlots {                     // consumer: the local slots of this exchange (entries >= world point at a slot of zeros)
  const void* slot[16]; const unsigned* flags; unsigned* timeouts; unsigned long long ticks; unsigned seq; int world, rank;
};
// producer: slot `rank` in every buffer + the flag each peer polls for this rank; world == 0: off
struct PeerPush { void* dst[16]; unsigned* flag[16]; unsigned* ticket; unsigned seq; int world, rank; };

// tail of a producer kernel (all threads call it): once the LAST workgroup's stores are visible system-wide, one thread
// raises this rank's flag at every peer.  The ticket lives in ordinary device memory and is left at zero for the next kernel.
__device__ __forceinline__ void peer_signal_tail(const PeerPush& pp) {
  // the slots are uncached (MTYPE UC) memory: a store is acknowledged by the memory (or the link) it went to, no cache holds it
  // back -- waiting for the acknowledgements orders the stores before the flag stores without a system-scope release (which
  // on gfx950 writes back every dirty line of the L2, here the whole partials array: +10 us per launch)
#ifdef MJX_PEER_SYSTEM_FENCE
  __threadfence_system();
#else
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(pp.ticket, 1u);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(pp.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int q = 0; q < pp.world; ++q)
        if (q != pp.rank) __hip_atomic_store(pp.flag[q], pp.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// head of a consumer kernel (all threads call it): source rank r stores the exchange number into flag r once its vector has
// landed; no rank can be more than one exchange ahead of another, so flag r - seq is -1, 0 or +1 (32-bit wrap-around included).
// Thread r polls flag r.  A peer that never delivers must not hang the GPU: after `ticks` (100 MHz; MJX_PEER_TIMEOUT_MS, default
// 5 s) the wait gives up, counts the event in `timeouts` (mjx_peer_status) and the caller poisons its result with NaN.
__device__ __forceinline__ bool peer_arrived(const PeerSlots& ps) {
  __shared__ int arrived;
  if (threadIdx.x == 0) arrived = 1;
  __syncthreads();
  if ((int)threadIdx.x < ps.world && (int)threadIdx.x != ps.rank) {
    const unsigned long long t0 = wall_clock64();           // 100 MHz
    while ((int)(__hip_atomic_load(ps.flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - ps.seq) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > ps.ticks) {
        arrived = 0;
        if (blockIdx.x == 0) atomicAdd(ps.timeouts, 1u);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");              // system scope: nothing read below may predate the arrivals
  return arrived != 0;
}
template <typename T>
__global__ void k_peer_push(const T* __restrict__ src, PeerPush pp, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) {
    const T v = src[i];
    for (int q = 0; q < pp.world; ++q) ((T*)pp.dst[q])[i] = v;
  }
  peer_signal_tail(pp);
}
template <typename T>
__global__ void k_peer_sum(PeerSlots ps, T* __restrict__ out, int64_t count) {       // fixed rank order: the same bits on every rank
  const bool ok = peer_arrived(ps);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  T t[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] = ((const T*)ps.slot[r])[i];     // all loads in flight at once (uncached memory)
  T a = t[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) a += t[r];
  out[i] = ok ? a : (T)__builtin_nanf("");
}

// the 4 fp64 sums a fused kernel leaves per workgroup (surrogate / KL / ...), reduced by ONE EXTRA workgroup of the vector
// reduction below instead of a launch of their own (k_reduce_scalars: ~4.5 us as a dependent launch, r06); same arithmetic, same
// order.  peer_off >= 0: the sums also travel with the vector, as 4 doubles at byte `peer_off` of this rank's slot.
struct ScalTail { const double* sp; int G; double* out; int peer_off; };

// the same for d % 4 == 0 with 16-byte loads: 32 columns (8 float4) x 32 row groups per block, i.e. 128-byte row
// segments instead of 64-byte ones and a quarter of the load instructions (7 -> ~4.5 us for 256 x 5.7 k partials)
__global__ __launch_bounds__(256) void k_reduce_partials4(const float* __restrict__ partials, int G, int d,
                                                           float* __restrict__ out, const float* theta,
                                                           const float* v, int oS, float frac, PeerPush pp = PeerPush{},
                                                           ScalTail stl = ScalTail{nullptr, 0, nullptr, -1},
                                                           const int* __restrict__ perm = nullptr) {
  // perm (r06): the partials' columns are in the fused kernels' accumulator order (fused_policy.h RawSlab); column c belongs to
  // flat index perm[c] (-1: a padding slot) -- `d` is then the slab width, and theta / v / out / the peers' slots are indexed flat
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ double sh[32][33];
  if (stl.sp != nullptr && blockIdx.x == gridDim.x - 1) {            // the extra workgroup: k_reduce_scalars' sums (launched with grid + 1)
    for (int k = 0; k < 4; ++k) {
      double a = 0.0;
      for (int g = threadIdx.x; g < stl.G; g += blockDim.x) a += stl.sp[(size_t)g * 4 + k];
      a = block_sum(a, &sh[0][0]);
      if (threadIdx.x == 0) {
        stl.out[k] = a;
        if (stl.peer_off >= 0)
          for (int q = 0; q < pp.world; ++q)
            if (q != pp.rank) ((double*)((char*)pp.dst[q] + stl.peer_off))[k] = a;
      }
    }
    if (pp.world) peer_signal_tail(pp);
    return;
  }
  const int cq = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const int c0 = (blockIdx.x * 8 + cq) * 4;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (c0 < d)
    for (int g = rg; g < G; g += 32) {
      const f4 p = *(const f4*)(partials + (size_t)g * d + c0);
      a0 += (double)p.x; a1 += (double)p.y; a2 += (double)p.z; a3 += (double)p.w;
    }
  sh[rg][4 * cq] = a0; sh[rg][4 * cq + 1] = a1; sh[rg][4 * cq + 2] = a2; sh[rg][4 * cq + 3] = a3;
  __syncthreads();
  const int cl = threadIdx.x, c = blockIdx.x * 32 + cl;
  const int cf = (cl < 32 && c < d) ? (perm ? perm[c] : c) : -1;
  if (cf >= 0) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += sh[k][cl];
    if (v != nullptr && cf >= oS) {
      float s = expf(theta[cf]);
      float u = s * s, e = 1e-8f;
      float den = 2.0f * u + e;
      float cc = 16.0f * u * u / (den * den) - 4.0f * u / den;
      t = (double)(frac * cc * v[cf]);
    }
    out[cf] = (float)t;
    for (int q = 0; q < pp.world; ++q)                 // peer exchange: `out` is this rank's slot in its own buffer
      if (q != pp.rank) ((float*)pp.dst[q])[cf] = (float)t;
  }
  if (pp.world) peer_signal_tail(pp);
}

// a vector of d floats and 4 doubles in ONE exchange (the gradient and K1's sums of a rank that holds no samples: it has no
// reduction kernel to fold the push into, but must take part in the same exchanges as the others)
__global__ void k_peer_push_vs(const float* __restrict__ v, const double* __restrict__ s4, PeerPush pp, int d, int scal_off) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d) {
    const float x = v[i];
    for (int q = 0; q < pp.world; ++q) ((float*)pp.dst[q])[i] = x;
  }
  if (i < 4) {
    const double x = s4[i];
    for (int q = 0; q < pp.world; ++q) ((double*)((char*)pp.dst[q] + scal_off))[i] = x;
  }
  peer_signal_tail(pp);
}

// pp.world > 0: `out` is this rank's slot (4 doubles at its start) in its own buffer, the sums go to the same place in every
// peer's buffer and the arrival flags are raised -- the push of a 4-double exchange folded into the reduction (k_peer_sum<double>
// is the consumer)
__global__ void k_reduce_scalars(const double* __restrict__ sp, int G, double* __restrict__ out, PeerPush pp = PeerPush{}) {
  __shared__ double sh[17];
  for (int k = 0; k < 4; ++k) {
    double a = 0.0;
    for (int g = threadIdx.x; g < G; g += blockDim.x) a += sp[(size_t)g * 4 + k];
    a = block_sum(a, sh);
    if (threadIdx.x == 0) {
      out[k] = a;
      for (int q = 0; q < pp.world; ++q)
        if (q != pp.rank) ((double*)pp.dst[q])[k] = a;
    }
  }
  if (pp.world) peer_signal_tail(pp);
}

// ---- conjugate gradient (mjrl/utils/cg_solve.py:3-22); vectors fp32, dots fp64 ----
// scal: [0]=r.r  [1]=done flag  [2]=p.z  [3]=iterations performed
__global__ __launch_bounds__(1024) void k_cg_init(const float* __restrict__ b, float* x, float* r, float* p,
                                                   double* scal, int d) {
  __shared__ double sh[17];
  double a = 0.0;
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    float bi = b[i];
    x[i] = 0.f; r[i] = bi; p[i] = bi;
    a += (double)bi * (double)bi;
  }
  a = block_sum(a, sh);
  if (threadIdx.x == 0) { scal[0] = a; scal[1] = 0.0; scal[2] = 0.0; scal[3] = 0.0; }
}
// the same with b arriving as one slot per rank (the gradient's exchange folded in, r06): waits for the arrivals, sums the W
// slots in rank order -- vector AND the 4 doubles that travel with it at byte `scal_off` of every slot (K1's sums -> s4_out) --,
// leaves the summed gradient in b_out for everything that reads it later, and starts the solve.  One launch instead of
// k_peer_sum<float> + k_peer_sum<double> + k_cg_init.
template <int W>
__global__ __launch_bounds__(1024) void k_cg_init_w(PeerSlots ps, int scal_off, float* __restrict__ b_out, double* __restrict__ s4_out,
                                                     float* x, float* r, float* p, double* scal, int d) {
  __shared__ double sh[17];
  const bool ok = peer_arrived(ps);
  double a = 0.0;
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    float t[W];
#pragma unroll
    for (int q = 0; q < W; ++q) t[q] = ((const float*)ps.slot[q])[i];
    float bi = t[0];
#pragma unroll
    for (int q = 1; q < W; ++q) bi += t[q];
    if (!ok) bi = __builtin_nanf("");
    b_out[i] = bi; x[i] = 0.f; r[i] = bi; p[i] = bi;
    a += (double)bi * (double)bi;
  }
  if (threadIdx.x < 4) {
    double t[W];
#pragma unroll
    for (int q = 0; q < W; ++q) t[q] = q < ps.world ? ((const double*)((const char*)ps.slot[q] + scal_off))[threadIdx.x] : 0.0;   // (the zero slot is shorter than scal_off)
    double sres = t[0];
#pragma unroll
    for (int q = 1; q < W; ++q) sres += t[q];
    s4_out[threadIdx.x] = ok ? sres : (double)__builtin_nanf("");
  }
  a = block_sum(a, sh);
  if (threadIdx.x == 0) { scal[0] = a; scal[1] = 0.0; scal[2] = 0.0; scal[3] = 0.0; }
}

__global__ __launch_bounds__(1024) void k_cg_step(const float* __restrict__ Ap, float damping, double tol,
                                                   float* x, float* r, float* p, float* z, double* scal, int d) {
  __shared__ double sh[17];
  if (scal[1] != 0.0) return;                       // converged earlier: cg_solve.py:19-20 `break`
  double pz = 0.0;
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    float zi = Ap[i] + damping * p[i];              // npg_cg.py:81  hvp_flat + regu_coef*vector
    z[i] = zi;
    pz += (double)p[i] * (double)zi;
  }
  pz = block_sum(pz, sh);
  const double rr = scal[0];
  const float alpha = (float)(rr / pz);
  double nrr = 0.0;
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    x[i] = fmaf(alpha, p[i], x[i]);
    float ri = fmaf(-alpha, z[i], r[i]);
    r[i] = ri;
    nrr += (double)ri * (double)ri;
  }
  nrr = block_sum(nrr, sh);
  const float mu = (float)(nrr / rr);
  for (int i = threadIdx.x; i < d; i += blockDim.x) p[i] = fmaf(mu, p[i], r[i]);
  if (threadIdx.x == 0) {
    scal[0] = nrr; scal[2] = pz; scal[3] += 1.0;
    if (nrr < tol) scal[1] = 1.0;
  }
}

// The same step for LARGE d (the layer-wise policies: 166 690 parameters at BASELINE configs[3], 297 528 at configs[4]) on many
// workgroups (r06).  The single-workgroup kernel above walks five d-float vectors through one CU: 286 us per CG iteration at
// configs[3], 524 us at configs[4] -- 5.9 % / 2.4 % of an NPG update (rocprofv3 over tools/lw_update_trace.py).  Three launches of
// CGM_G workgroups instead, each phase ending in per-workgroup fp64 partials that EVERY workgroup of the next launch sums in the
// same fixed order (so all of them hold the same alpha / mu bit for bit); z = Ap + damping p is recomputed where it is used (the same
// two fp32 operations) instead of being stored.  Scalars that a later phase needs although workgroup 0 rewrites them at its end
// (rr, the convergence flag) are copied to scal[4..6] by the phase before.
constexpr int CGM_G = 64, CGM_T = 256;
__device__ __forceinline__ double cgm_total(const double* part, double* sh) {      // sum of the CGM_G partials, fixed order
  return block_sum(threadIdx.x < CGM_G ? part[threadIdx.x] : 0.0, sh);
}
__global__ __launch_bounds__(CGM_T) void k_cgm_pz(const float* __restrict__ Ap, const float* __restrict__ p, float damping,
                                                   const double* __restrict__ scal, double* __restrict__ part, int d) {
  __shared__ double sh[17];
  double pz = 0.0;
  if (scal[1] == 0.0)
    for (int i = blockIdx.x * CGM_T + threadIdx.x; i < d; i += CGM_G * CGM_T) {
      const float zi = Ap[i] + damping * p[i];        // npg_cg.py:81  hvp_flat + regu_coef*vector
      pz += (double)p[i] * (double)zi;
    }
  pz = block_sum(pz, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = pz;
}
__global__ __launch_bounds__(CGM_T) void k_cgm_xr(const float* __restrict__ Ap, const float* __restrict__ p, float damping,
                                                   float* __restrict__ x, float* __restrict__ r, double* scal, double* part, int d) {
  __shared__ double sh[17];
  const double done = scal[1], rr = scal[0];
  const double pz = cgm_total(part, sh);
  double nrr = 0.0;
  if (done == 0.0) {                                  // (converged earlier: cg_solve.py:19-20 `break` -- nothing moves any more)
    const float alpha = (float)(rr / pz);
    for (int i = blockIdx.x * CGM_T + threadIdx.x; i < d; i += CGM_G * CGM_T) {
      const float pi = p[i], zi = Ap[i] + damping * pi;
      x[i] = fmaf(alpha, pi, x[i]);
      const float ri = fmaf(-alpha, zi, r[i]);
      r[i] = ri;
      nrr += (double)ri * (double)ri;
    }
  }
  nrr = block_sum(nrr, sh);
  if (threadIdx.x == 0) {
    part[CGM_G + blockIdx.x] = nrr;
    if (blockIdx.x == 0) { scal[4] = rr; scal[5] = pz; scal[6] = done; }
  }
}
__global__ __launch_bounds__(CGM_T) void k_cgm_p(const float* __restrict__ r, float* __restrict__ p, double tol, double* scal,
                                                  const double* __restrict__ part, int d) {
  __shared__ double sh[17];
  const double rr = scal[4], pz = scal[5], done = scal[6];
  const double nrr = cgm_total(part + CGM_G, sh);
  if (done != 0.0) return;
  const float mu = (float)(nrr / rr);
  for (int i = blockIdx.x * CGM_T + threadIdx.x; i < d; i += CGM_G * CGM_T) p[i] = fmaf(mu, p[i], r[i]);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scal[0] = nrr; scal[2] = pz; scal[3] += 1.0;
    if (nrr < tol) scal[1] = 1.0;
  }
}

// the same step for d <= 1024 * EPT with every vector element held in registers: one round of loads, two block
// reductions, one round of stores (the looped version above pays a global round trip per phase).  Same arithmetic.
// W > 0: the Fisher-vector product arrives as one slot per rank (peer exchange, W = slots read: world rounded up to a power
// of two, the surplus entries point at zeros); the body waits for the arrivals and sums the slots in rank order.
// Called by all 1024 threads of one workgroup.
// what follows the LAST vector update of a solve, folded into its kernel (r06: k_cg_finish + k_apply_npg_step were two dependent
// ~4.7 us launches of a 3.8 ms -- or, on an eighth of the batch, 0.8 ms -- update): mode 1: x_out = x, bdotx = b.x (k_cg_finish);
// mode 2: ... and theta_out = theta + sqrt(|step_size / (b.x + 1e-20)|) x with the log_std clamp, the step length to alpha_out
// (k_apply_npg_step); mode 3: the same with the caller's constant step length.  Same arithmetic, same order, same bits.
struct CgFin {
  int mode = 0;
  const float* b = nullptr; float* x_out = nullptr; double* bdotx = nullptr;
  const float* theta = nullptr; float* theta_out = nullptr; double* alpha_out = nullptr;
  double step_size = 0.0; float const_alpha = 0.f, min_log_std = 0.f; int oS = 0;
};

template <int EPT, int W>
__device__ __forceinline__ void cg_step_body(const float* Ap, float damping, double tol, float* x, float* r, float* p,
                                             double* scal, int d, const PeerSlots& ps, double* sh /* 17 doubles */, const CgFin& fin) {
  bool ok = true;
  if (W) ok = peer_arrived(ps);
  const double done = scal[1], rr = scal[0];
  float ap[EPT], pv[EPT], xv[EPT], rv[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = threadIdx.x + e * 1024;
    const int ic = i < d ? i : 0;
    if (W) {
      float t[W ? W : 1];
#pragma unroll
      for (int q = 0; q < W; ++q) t[q] = ((const float*)ps.slot[q])[ic];
      float a = t[0];
#pragma unroll
      for (int q = 1; q < W; ++q) a += t[q];
      ap[e] = ok ? a : __builtin_nanf("");
    } else {
      ap[e] = Ap[ic];
    }
    pv[e] = p[ic]; xv[e] = x[ic]; rv[e] = r[ic];
  }
  if (done == 0.0) {                                  // (converged earlier: cg_solve.py:19-20 `break` -- nothing moves any more)
    double pz = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const float zi = ap[e] + damping * pv[e];       // npg_cg.py:81  hvp_flat + regu_coef*vector
      ap[e] = zi;
      if (threadIdx.x + e * 1024 < d) pz += (double)pv[e] * (double)zi;
    }
    pz = block_sum(pz, sh);
    const float alpha = (float)(rr / pz);
    double nrr = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      xv[e] = fmaf(alpha, pv[e], xv[e]);
      rv[e] = fmaf(-alpha, ap[e], rv[e]);
      if (threadIdx.x + e * 1024 < d) nrr += (double)rv[e] * (double)rv[e];
    }
    nrr = block_sum(nrr, sh);
    const float mu = (float)(nrr / rr);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = threadIdx.x + e * 1024;
      if (i < d) { x[i] = xv[e]; r[i] = rv[e]; p[i] = fmaf(mu, pv[e], rv[e]); }
    }
    if (threadIdx.x == 0) {
      scal[0] = nrr; scal[2] = pz; scal[3] += 1.0;
      if (nrr < tol) scal[1] = 1.0;
    }
  }
  if (fin.mode == 0) return;
  // ---- k_cg_finish: x_out, b.x (each thread adds its elements in increasing index order, like the strided loop there)
  double bx = 0.0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = threadIdx.x + e * 1024;
    if (i < d) {
      if (fin.x_out) fin.x_out[i] = xv[e];
      bx += (double)fin.b[i] * (double)xv[e];
    }
  }
  bx = block_sum(bx, sh);
  if (threadIdx.x == 0 && fin.bdotx) fin.bdotx[0] = bx;
  if (fin.mode == 1) return;
  // ---- k_apply_npg_step / k_apply_step
  const double a64 = fin.mode == 2 ? sqrt(fabs(fin.step_size / (bx + 1e-20))) : (double)fin.const_alpha;
  if (threadIdx.x == 0 && fin.alpha_out && fin.mode == 2) fin.alpha_out[0] = a64;
  const float alpha = fin.mode == 2 ? (float)a64 : fin.const_alpha;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = threadIdx.x + e * 1024;
    if (i < d) {
      float v = __fadd_rn(fin.theta[i], __fmul_rn(alpha, xv[e]));   // numpy: separate multiply and add
      if (i >= fin.oS) v = fmaxf(v, fin.min_log_std);
      fin.theta_out[i] = v;
    }
  }
}
template <int EPT, int W = 0>
__global__ __launch_bounds__(1024) void k_cg_step_reg(const float* __restrict__ Ap, float damping, double tol,
                                                       float* x, float* r, float* p, double* scal, int d, PeerSlots ps = PeerSlots{},
                                                       CgFin fin = CgFin{}) {
  __shared__ double sh[17];
  cg_step_body<EPT, W>(Ap, damping, tol, x, r, p, scal, d, ps, sh, fin);
}

__global__ __launch_bounds__(1024) void k_cg_finish(const float* __restrict__ b, const float* __restrict__ x,
                                                     float* x_out, double* bdotx, int d) {
  __shared__ double sh[17];
  double a = 0.0;
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    float xi = x[i];
    if (x_out) x_out[i] = xi;
    a += (double)b[i] * (double)xi;
  }
  a = block_sum(a, sh);
  if (threadIdx.x == 0 && bdotx) bdotx[0] = a;
}

// theta_out = theta + alpha*x ; log_std clamp (gaussian_mlp.py:73-75)
__global__ void k_apply_step(const float* __restrict__ theta, const float* __restrict__ x, float alpha,
                             float min_log_std, float* out, int d, int oS) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d) return;
  float v = __fadd_rn(theta[i], __fmul_rn(alpha, x[i]));   // numpy: separate multiply and add
  if (i >= oS) v = fmaxf(v, min_log_std);
  out[i] = v;
}

// x *= s (DAPG's sample_coef on the gradient, dapg.py:97-98: an fp32 product per element, like NumPy's float32 array x Python float)
__global__ void k_scale_f32(float* __restrict__ x, float s, int d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d) x[i] = __fmul_rn(x[i], s);
}

// TRPO's backtracking line search on the device (trpo.py:107-120).  res (doubles): [0..3] K3 sums of the last evaluation,
// [8] g.x, [9] step length of the last trial, [10] accepted flag, [11] trials so far, [12] step length of the next trial,
// [16 + 2 (k % 24)], [17 + 2 (k % 24)] surrogate / KL sums of trial k (a ring of 24: a call performs at most 24 trials).
// k_trpo_try: theta_out = theta_old + alpha x for the next trial -- unless a trial was accepted already (theta_out keeps
// the accepted parameters); init: first trial of an update, alpha = sqrt(|step_size / (g.x + 1e-20)|) (trpo.py:104).
__global__ void k_trpo_try(const float* __restrict__ theta, const float* __restrict__ x, double* res, double step_size, int init,
                           float min_log_std, float* out, int d, int oS) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double a64 = init ? sqrt(fabs(step_size / (res[8] + 1e-20))) : res[12];
  const bool done = !init && res[10] != 0.0;
  if (i == 0 && init) { res[12] = a64; res[10] = 0.0; res[11] = 0.0; }
  if (i >= d || done) return;
  const float alpha = (float)a64;
  float v = __fadd_rn(theta[i], __fmul_rn(alpha, x[i]));
  if (i >= oS) v = fmaxf(v, min_log_std);
  out[i] = v;
}
// k_trpo_check (one thread): accept the trial if its mean KL is below kl_dist, else alpha <- 0.9 alpha
__global__ void k_trpo_check(double* res, double kl_dist, double n_global) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || res[10] != 0.0) return;
  const int k = (int)res[11];
  res[16 + 2 * (k % 24)] = res[0]; res[17 + 2 * (k % 24)] = res[1];
  res[11] = (double)(k + 1);
  res[9] = res[12];
  if (res[1] / n_global < kl_dist) res[10] = 1.0;
  else res[12] = 0.9 * res[12];
}

// the same with the NPG step length formed on the device: alpha = sqrt(|delta / (g.x + 1e-20)|) in fp64 like
// npg_cg.py:133, so the host does not have to wait for g.x between the CG solve and the parameter step
__global__ void k_apply_npg_step(const float* __restrict__ theta, const float* __restrict__ x, const double* __restrict__ gdotx,
                                 double step_size, float min_log_std, float* out, double* alpha_out, int d, int oS) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double a64 = sqrt(fabs(step_size / (gdotx[0] + 1e-20)));
  if (i == 0 && alpha_out) alpha_out[0] = a64;
  if (i >= d) return;
  const float alpha = (float)a64;
  float v = __fadd_rn(theta[i], __fmul_rn(alpha, x[i]));
  if (i >= oS) v = fmaxf(v, min_log_std);
  out[i] = v;
}

// tpos[s] = position of sample s inside its trajectory (the np.arange(l) per path of the baselines' time features,
// mjrl/baselines/quadratic_baseline.py:28 / mlp_baseline.py:47): one block per trajectory, coalesced int32 stores
__global__ __launch_bounds__(256) void k_time_index(const int64_t* __restrict__ offsets, int* __restrict__ tpos) {
  const int64_t lo = offsets[blockIdx.x], len = offsets[blockIdx.x + 1] - lo;
  for (int64_t i = threadIdx.x; i < len; i += 256) tpos[lo + i] = (int)i;
}

// ---- K5: reverse discounted scans over ragged trajectories (process_samples.py:21-44) ----
// One 256-thread block per trajectory; the trajectory is cut into 256 contiguous segments,
// pass 1 reduces every segment with zero carry, thread 0 chains the 256 carries, pass 2
// replays each segment sequentially from its true carry (so within a segment the arithmetic
// is exactly the reference's recurrence).
// MODE 0: y[t] = x[t] + g*y[t+1]                       (discount_sum)
// MODE 1: x[t] := r[t] + gamma*b1[t+1] - b1[t]         (GAE td residual), g = gamma*lam
// MODE 2: y[t] = x[t] - b[t]                           (non-GAE branch, no scan)
template <int MODE>
__global__ __launch_bounds__(256) void k_traj_scan(const double* __restrict__ x, const double* __restrict__ bl,
                                                    const int64_t* __restrict__ off, const uint8_t* __restrict__ term,
                                                    double gamma, double g, double* __restrict__ y) {
  __shared__ double loc[256], pw[256], cin[256];
  const int64_t o = off[blockIdx.x];
  const int T = (int)(off[blockIdx.x + 1] - o);
  if (T <= 0) return;
  const double* xr = x + o;
  const double* br = (MODE != 0) ? bl + o : nullptr;
  double* yr = y + o;
  if (MODE == 2) {
    for (int t = threadIdx.x; t < T; t += 256) yr[t] = xr[t] - br[t];
    return;
  }
  const double blast = (MODE == 1) ? ((term && term[blockIdx.x]) ? 0.0 : br[T - 1]) : 0.0;
  const int seg = (T + 255) / 256;
  const int lo = min(T, (int)threadIdx.x * seg), hi = min(T, lo + seg);
  auto val = [&](int t) -> double {
    if (MODE == 0) return xr[t];
    double bn = (t + 1 < T) ? br[t + 1] : blast;
    return xr[t] + gamma * bn - br[t];
  };
  double run = 0.0, f = 1.0;
  for (int t = hi - 1; t >= lo; --t) { run = val(t) + g * run; f *= g; }
  loc[threadIdx.x] = run; pw[threadIdx.x] = f;
  __syncthreads();
  if (threadIdx.x == 0) {
    double c = 0.0;
    for (int k = 255; k >= 0; --k) { cin[k] = c; c = loc[k] + pw[k] * c; }
  }
  __syncthreads();
  run = cin[threadIdx.x];
  for (int t = hi - 1; t >= lo; --t) { run = val(t) + g * run; yr[t] = run; }
}

// ---- statistics / casts used by process_paths (batch_reinforce.py:178-197) ----
__global__ __launch_bounds__(256) void k_sum_stats_partial(const double* __restrict__ x, int64_t N, double shift,
                                                            double* __restrict__ part) {
  __shared__ double sh[17];
  double s = 0.0, q = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
    double v = x[i] - shift;
    s += v; q += v * v;
  }
  s = block_sum(s, sh);
  q = block_sum(q, sh);
  if (threadIdx.x == 0) { part[blockIdx.x * 2] = s; part[blockIdx.x * 2 + 1] = q; }
}
__global__ void k_sum_stats_final(const double* __restrict__ part, int G, int64_t N, double* out) {
  __shared__ double sh[17];
  double s = 0.0, q = 0.0;
  for (int g = threadIdx.x; g < G; g += blockDim.x) { s += part[g * 2]; q += part[g * 2 + 1]; }
  s = block_sum(s, sh);
  q = block_sum(q, sh);
  if (threadIdx.x == 0) { out[0] = s; out[1] = q; out[2] = (double)N; }
}
__global__ void k_whiten_cast(const double* __restrict__ a, int64_t N, double mean, double denom, float* __restrict__ o) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
    o[i] = (float)((a[i] - mean) / denom);
}
__global__ void k_cast_f64_f32(const double* __restrict__ a, int64_t N, float* __restrict__ o) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
    o[i] = (float)a[i];
}

}  // namespace mjx
