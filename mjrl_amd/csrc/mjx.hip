// mjx.hip -- C ABI (include/mjx.h) over the gfx950 kernels.  Host side only orchestrates:
// it owns the workspace, picks the kernel instance and enqueues launches on the caller's
// stream.  No host<->device synchronisation happens here unless an entry point says so.
#include "../../include/mjx.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <pthread.h>
#include <unistd.h>
#include <atomic>
#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "baseline.h"
#include "fused_policy.h"
#include "policy_fit.h"
#include "rccl_dyn.h"
#include "layerwise.h"
#include "mlp_fit.h"
#include "vecops.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) return fail((int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// ---- fork guard.  Sampler workers are FORKED from the training process (mjrl/samplers/core.py:189-210: mp.Pool under a parent
// that holds an mjx_ctx, page-locked staging blocks and libmjx's gather threads).  None of that survives a fork in a usable
// state, so a forked child of a process that touched the device through this library may not touch it again: every entry that
// creates device state refuses there, loudly, instead of hanging inside the HIP runtime.  mjx_process_state() lets a test (or
// a worker) prove that it never tried.
std::atomic<long long> g_device_calls{0};     // entries that reach the HIP runtime, made by THIS process (zeroed in a forked child)
std::atomic<int> g_forked_child{0};           // 1: this process was forked from one whose g_device_calls was > 0
void atfork_child() {
  if (g_device_calls.load() > 0) g_forked_child.store(1);
  g_device_calls.store(0);
}
struct ForkGuardInit { ForkGuardInit() { pthread_atfork(nullptr, nullptr, atfork_child); } } g_fork_guard_init;

#define MJX_DEVICE_ENTRY()                                                                                                          \
  do {                                                                                                                              \
    if (g_forked_child.load())                                                                                                      \
      return fail(MJX_ERR_STATE, "libmjx: this process was forked from one that holds HIP state; device work belongs to the "      \
                                 "training process (sampler workers stay on the host, mjrl/samplers/core.py)");                     \
    g_device_calls.fetch_add(1);                                                                                                    \
  } while (0)

}  // namespace

struct mjx_ctx {
  int device = 0;
  int n = 0, m = 0;
  std::vector<int> hidden;
  int64_t d = 0;
  int oS = 0;                      // offset of log_std in the flat vector
  int n_cu = 256;
  // fused path
  int fused = 0;                   // 0 = layer-wise, else variant id
  int grid = 256;
  size_t lds_bytes = 0;
  // bound inputs
  const float *obs = nullptr, *act = nullptr, *adv = nullptr;
  int64_t N_local = 0, N_global = 0;
  const float *theta_new = nullptr, *theta_old = nullptr, *tr_new = nullptr, *tr_old = nullptr;
  int old_is_new = 1;
  bool batch_bound = false;
  float* hcache = nullptr; size_t hcache_bytes = 0;   // forward-activation cache of the fused path
  bool hcache_valid = false; const float* hcache_obs = nullptr; int64_t hcache_rows = 0;
  bool ximg_ok = false; int64_t ximg_rows = 0;        // the cache's normalised-observation image outlives a parameter change (K3 reads it)
  int use_hcache = 1;
  float* ocache = nullptr; size_t ocache_bytes = 0;   // old-policy outputs of the batch (K1 -> K3)
  float* snap = nullptr;                              // parameters + transforms they were computed with
  bool ocache_valid = false; int64_t ocache_rows = 0;
  int64_t rows_bound = 0;                             // rows handed to the last mjx_bind_batch
  // workspace (device)
  float* partials = nullptr;       // [grid][max(d, raw_dr)]
  int raw_dr = 0; int* raw_perm = nullptr;   // fused path: workgroup partials in accumulator order + the column -> flat index table (fused_policy.h RawSlab)
  double* spartials = nullptr;     // [grid][4]
  float* ident_tr = nullptr;       // identity transforms
  float *cg_x = nullptr, *cg_r = nullptr, *cg_p = nullptr, *cg_z = nullptr, *cg_Ap = nullptr;
  double* cg_scal = nullptr;       // 8 doubles
  float* dbg = nullptr;
  long long* clk = nullptr;        // launch clock stamps (mjx_set_clock_buffer)
  unsigned fvp_seq = 0;            // products since the cache was filled / the last solve began: alternate sweep direction
  bool lw_old_ok = false;          // K1's outputs are the OLD policy's (old == new at K1; theta_old, its transform and the batch untouched since by this
                                   // library): set by mjx_surr_vpg, cleared by the public binding calls, consumed by the one-call updates' evaluations
                                   // (layer-wise path: reuse of the output block; fused path: the snapshot compare of K3's prologue is skipped)
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev;   // pairs
  size_t prof_used = 0;
  int prof_stride = 1;               // bracket every prof_stride-th launch
  bool prof_iter = false;            // bracket whole CG iterations (mjx_profile_enable(ctx, -k)) instead of product launches
  size_t prof_seen = 0;
  void* comm = nullptr; int comm_world = 0, comm_rank = 0;   // RCCL communicator (one process per GPU)
  mjx_reduce_fn reduce_cb = nullptr; void* reduce_user = nullptr;   // transport hook in its place (tests)
  // peer exchange (mjx_peer_*): one uncached device buffer per rank [2 parities][world slots] | flag block, the peers' mapped
  // through HIP IPC; every all-reduce = store the vector into slot `rank` of every buffer + raise this rank's flag at every
  // peer, one bounded in-kernel wait on the own flags, rank-ordered sum of the local slots (vecops.h "Peer exchange")
  struct Peer {
    bool on = false, loopback = false; int rank = 0, world = 0;
    char* buf = nullptr; char* map[16] = {nullptr};
    size_t slot_bytes = 0; uint32_t seq = 0;
    unsigned long long timeout_ticks = 500000000ull;   // 100 MHz ticks a consumer waits for a peer (MJX_PEER_TIMEOUT_MS; default 5 s)
    int fault = 0;                                     // MJX_PEER_FAULT (tests): 1 = "slot" (vectors land in the wrong slot at the peers), 2 = "flag" (arrival flags never raised)
  } peer;
  unsigned* ticket = nullptr;                      // workgroup ticket of the producer kernels (ordinary device memory, zero between launches)
  mjx::LayerwiseWS lw;             // layer-wise path workspace
  mjx::LayerwiseWS lwmb;           // minibatch trainer workspace (mjx_policy_minibatch_adam)
  float *mb_x = nullptr, *mb_a = nullptr, *mb_adv = nullptr, *mb_grad = nullptr; int mb_cap = 0;
};

namespace {

using namespace mjx;

template <int H1, int H2, int NT1, int MP, bool DBG = false, int NPC = 0>
int launch_fused(mjx_ctx* c, int mode, const FusedArgs& a, hipStream_t st) {
  const bool ev2 = (mode == MODE_EVAL) && MP <= 8;   // MODE_EVAL (always the non-debug instance): small layout, two workgroups per CU (fused_policy.h)
  FusedLayout<H1, H2, NT1, MP> L(NPC ? NPC - 1 : c->n, ev2);       // (the kernel's rule: fused_policy.h)
  size_t bytes = L.bytes();
  void (*k)(FusedArgs) = nullptr;
  const bool cached = (mode == MODE_FVP) && a.hcache != nullptr && !DBG;
  if (mode == MODE_VPG) k = k_fused<H1, H2, NT1, MP, MODE_VPG, DBG, NPC>;
  else if (mode == MODE_FVP) k = cached ? k_fused<H1, H2, NT1, MP, MODE_FVP, false, NPC, true> : k_fused<H1, H2, NT1, MP, MODE_FVP, DBG, NPC>;
  else k = k_fused<H1, H2, NT1, MP, MODE_EVAL, false, NPC>;
  if (cached) mode = 3;
  // the dynamic-LDS limit is a per-kernel, per-device attribute and FusedLayout::bytes() depends on the runtime observation
  // count (the generic NPC = 0 instances serve many): remember the largest size configured per (device, kernel) and raise it
  // whenever a context needs more
  static thread_local std::vector<std::pair<std::pair<int, const void*>, size_t>> configured;
  {
    size_t* have = nullptr;
    for (auto& e : configured) if (e.first.first == c->device && e.first.second == (const void*)k) have = &e.second;
    if (!have) { configured.push_back({{c->device, (const void*)k}, 0}); have = &configured.back().second; }
    if (*have < bytes) {
      HIPCHK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      *have = bytes;
    }
  }
  hipLaunchKernelGGL(k, dim3(ev2 ? 2 * c->grid : c->grid), dim3(256), bytes, st, a);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

// variant table: id -> (H1, H2, NT1, MP)
// the compile-time feature count a shape is served with (dispatch_fused's rule): 64 x 64 x <= 8 actions with 4..7 / 8..11 / 16..19 observations
int npc_of(int fused, int n) {
  const int NPr = (n + 1 + 3) & ~3;
#ifdef MJX_PHASE_CLOCK
  return (fused == 1 && NPr == 20) ? 20 : 0;       // (the timing build: NPC = 20 or the generic instance, debug buffer or not)
#else
  return (fused == 1 && (NPr == 20 || NPr == 12 || NPr == 8)) ? NPr : 0;
#endif
}
// -> RawSlab<...>::DR of the instance (variant id, NPC) and, with perm != nullptr, its column -> flat index table; 0: unknown instance / bad table
int raw_slab(int fused, int npc, int n, int m, std::vector<int>* perm) {
  auto one = [&](auto rs) -> int {
    using RS = decltype(rs);
    if (perm) { perm->assign(RS::DR, -1); if (!RS::fill_perm(perm->data(), n, m)) return 0; }
    return RS::DR;
  };
  switch (fused) {
    case 1: return npc == 20 ? one(RawSlab<64, 64, 1, 8, 20>{}) : npc == 12 ? one(RawSlab<64, 64, 1, 8, 12>{}) : npc == 8 ? one(RawSlab<64, 64, 1, 8, 8>{})
                                                                                                                             : one(RawSlab<64, 64, 1, 8, 0>{});
    case 2: return one(RawSlab<32, 32, 1, 8, 0>{});
    case 3: return one(RawSlab<64, 64, 1, 16, 0>{});
    case 4: return one(RawSlab<32, 32, 1, 16, 0>{});
    case 5: return one(RawSlab<32, 32, 2, 8, 0>{});
    case 6: return one(RawSlab<32, 32, 2, 32, 0>{});
  }
  return 0;
}

template <int H1, int H2, int NT1, int MP>
bool variant_fits(int n, int m, int h1, int h2, int64_t d, size_t* bytes) {
  if (h1 != H1 || h2 != H2 || m > MP || n + 1 > 32 * NT1) return false;
  FusedLayout<H1, H2, NT1, MP> L(n);
  if (L.bytes() > 160 * 1024) return false;
  if ((size_t)(4 * d + 64) * 4 > L.bytes()) return false;   // end-of-kernel reduction region
  *bytes = L.bytes();
  return true;
}

int pick_variant(int n, int m, const std::vector<int>& hid, int64_t d, size_t* bytes) {
  if (hid.size() != 2) return 0;
  int h1 = hid[0], h2 = hid[1];
  if (variant_fits<64, 64, 1, 8>(n, m, h1, h2, d, bytes)) return 1;
  if (variant_fits<32, 32, 1, 8>(n, m, h1, h2, d, bytes)) return 2;
  if (variant_fits<64, 64, 1, 16>(n, m, h1, h2, d, bytes)) return 3;
  if (variant_fits<32, 32, 1, 16>(n, m, h1, h2, d, bytes)) return 4;
  if (variant_fits<32, 32, 2, 8>(n, m, h1, h2, d, bytes)) return 5;
  if (variant_fits<32, 32, 2, 32>(n, m, h1, h2, d, bytes)) return 6;      // Adroit-class: 39-46 observations, 24-30 actions, 32 x 32 (hand_dapg)
  return 0;
}

int dispatch_fused(mjx_ctx* c, int mode, const FusedArgs& a, hipStream_t st) {
  // shape-specialised instances (compile-time feature count => fully unrolled first layer)
  const int NPr = (c->n + 1 + 3) & ~3;
#ifdef MJX_PHASE_CLOCK
  if (c->fused == 1 && NPr == 20) return launch_fused<64, 64, 1, 8, false, 20>(c, mode, a, st);
#endif
  if (c->fused == 1 && NPr == 20 && !c->dbg) return launch_fused<64, 64, 1, 8, false, 20>(c, mode, a, st);   // obs 16..19 (HalfCheetah 17)
  if (c->fused == 1 && NPr == 12 && !c->dbg) return launch_fused<64, 64, 1, 8, false, 12>(c, mode, a, st);   // obs 8..11 (Hopper 11, Reacher 11, Swimmer 8)
  if (c->fused == 1 && NPr == 8 && !c->dbg) return launch_fused<64, 64, 1, 8, false, 8>(c, mode, a, st);     // obs 4..7 (InvertedPendulum 4, point_mass 6)
  switch (c->fused) {
#ifdef MJX_PHASE_CLOCK
    case 1: return launch_fused<64, 64, 1, 8>(c, mode, a, st);          // stamps go to the debug buffer of the production kernel
#else
    case 1: return c->dbg ? launch_fused<64, 64, 1, 8, true>(c, mode, a, st) : launch_fused<64, 64, 1, 8>(c, mode, a, st);
#endif
    case 2: return launch_fused<32, 32, 1, 8>(c, mode, a, st);
    case 3: return launch_fused<64, 64, 1, 16>(c, mode, a, st);
    case 4: return launch_fused<32, 32, 1, 16>(c, mode, a, st);
    case 5: return launch_fused<32, 32, 2, 8>(c, mode, a, st);
    case 6: return launch_fused<32, 32, 2, 32>(c, mode, a, st);
  }
  return fail(MJX_ERR_STATE, "no fused variant");
}

FusedArgs make_args(mjx_ctx* c, const float* thetaB) {
  FusedArgs a;
  a.obs = c->obs; a.act = c->act; a.adv = c->adv;
  a.N = c->N_local;
  a.inv_N = (float)(1.0 / (double)c->N_global);
  a.thetaA = c->theta_new; a.thetaB = thetaB;
  a.trA = c->tr_new ? c->tr_new : c->ident_tr;
  a.trB = c->tr_old ? c->tr_old : c->ident_tr;
  a.old_is_new = c->old_is_new;
  a.partials = c->partials; a.spartials = c->spartials;
  a.dbg = c->dbg;
  a.clk = c->clk;
  a.reverse = 0;
  a.hcache = nullptr; a.ocache = nullptr; a.snap = nullptr; a.snap_out = nullptr; a.snap_trusted = 0;
  a.n = c->n; a.m = c->m;
#ifdef MJX_PHASE_CLOCK
  a.raw_dr = c->raw_perm ? c->raw_dr : 0;
#else
  a.raw_dr = (c->raw_perm && !c->dbg) ? c->raw_dr : 0;      // (the debug instances may be another NPC: flat-order partials)
#endif
  return a;
}

int check_bound(mjx_ctx* c, bool need_act) {
  if (!c) return fail(MJX_ERR_ARG, "null context");
  if (!c->batch_bound || c->N_local < 0 || c->N_global <= 0) return fail(MJX_ERR_STATE, "mjx_bind_batch has not been called");
  if (need_act && c->N_local > 0 && (!c->act || !c->adv)) return fail(MJX_ERR_STATE, "actions / advantages not bound");
  if (!c->theta_new || !c->theta_old) return fail(MJX_ERR_STATE, "mjx_bind_policy has not been called");
  return MJX_OK;
}

}  // namespace

extern "C" {

const char* mjx_last_error(void) { return g_err.c_str(); }
int mjx_version(void) { return 1; }

int mjx_process_state(int64_t* out2) {
  if (!out2) return fail(MJX_ERR_ARG, "null output");
  out2[0] = (int64_t)g_device_calls.load();
  out2[1] = (int64_t)g_forked_child.load();
  return MJX_OK;
}

int mjx_device_count(void) {
  if (g_forked_child.load()) return 0;            // (never fails: a forked child simply sees no device through this library)
  g_device_calls.fetch_add(1);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

int mjx_create(mjx_ctx** out, int device, int n, int m, const int* hidden, int n_hidden) {
  if (!out || n <= 0 || m <= 0 || n_hidden < 0 || (n_hidden > 0 && !hidden)) return fail(MJX_ERR_ARG, "bad arguments");
  MJX_DEVICE_ENTRY();
  if (mjx_device_count() <= device) return fail(MJX_ERR_NOGPU, "HIP device %d not available", device);
  HIPCHK(hipSetDevice(device));
  mjx_ctx* c = new mjx_ctx();
  c->device = device; c->n = n; c->m = m;
  c->hidden.assign(hidden, hidden + n_hidden);
  int prev = n; int64_t d = 0;
  for (int h : c->hidden) { if (h <= 0) { delete c; return fail(MJX_ERR_ARG, "bad hidden size"); } d += (int64_t)prev * h + h; prev = h; }
  d += (int64_t)prev * m + m;
  c->oS = (int)d;
  d += m;
  c->d = d;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  c->n_cu = prop.multiProcessorCount;
  c->grid = c->n_cu;
  c->fused = pick_variant(n, m, c->hidden, d, &c->lds_bytes);
  if (const char* e = getenv("MJX_FORCE_LAYERWISE")) if (e[0] == '1') c->fused = 0;
  if (const char* e = getenv("MJX_NO_HCACHE")) if (e[0] == '1') c->use_hcache = 0;
  if (c->fused) {
    // workgroup partials in accumulator order (RawSlab): the table is made here, once; any doubt about it (an instance this
    // switch does not know, a table that does not hit every flat index exactly once, no room for the four copies in LDS) keeps
    // the flat-order epilogue.  MJX_RAW_SLAB=0: A/B.
    const char* e = getenv("MJX_RAW_SLAB");
    std::vector<int> perm;
    const int dr = (e && e[0] == '0') ? 0 : raw_slab(c->fused, npc_of(c->fused, n), n, m, &perm);
    if (dr > 0 && (size_t)(4 * dr + 64) * 4 <= c->lds_bytes) {
      HIPCHK(hipMalloc((void**)&c->raw_perm, (size_t)dr * sizeof(int)));
      HIPCHK(hipMemcpy(c->raw_perm, perm.data(), (size_t)dr * sizeof(int), hipMemcpyHostToDevice));
      c->raw_dr = dr;
    }
  }
  HIPCHK(hipMalloc(&c->partials, (size_t)c->grid * (size_t)(c->raw_dr > d ? c->raw_dr : d) * sizeof(float)));
  HIPCHK(hipMalloc(&c->spartials, (size_t)2 * c->grid * 4 * sizeof(double)));       // (MODE_EVAL launches 2 workgroups per CU)
  HIPCHK(hipMalloc(&c->cg_x, d * 4)); HIPCHK(hipMalloc(&c->cg_r, d * 4)); HIPCHK(hipMalloc(&c->cg_p, d * 4));
  HIPCHK(hipMalloc(&c->cg_z, d * 4)); HIPCHK(hipMalloc(&c->cg_Ap, d * 4));
  HIPCHK(hipMalloc((void**)&c->ticket, 256)); HIPCHK(hipMemset(c->ticket, 0, 256));
  HIPCHK(hipMalloc(&c->cg_scal, 8 * sizeof(double)));
  std::vector<float> id(2 * n + 2 * m, 0.f);
  for (int i = 0; i < n; ++i) id[n + i] = 1.f;
  for (int i = 0; i < m; ++i) id[2 * n + m + i] = 1.f;
  HIPCHK(hipMalloc(&c->ident_tr, id.size() * 4));
  HIPCHK(hipMemcpy(c->ident_tr, id.data(), id.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(c->spartials, 0, (size_t)2 * c->grid * 4 * sizeof(double)));
  c->lw.init(n, m, c->hidden);
  c->lwmb.init(n, m, c->hidden);
  *out = c;
  return MJX_OK;
}

void mjx_destroy(mjx_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)mjx_comm_destroy(c);
  c->lw.release();
  c->lwmb.release();
  hipFree(c->mb_x); hipFree(c->mb_a); hipFree(c->mb_adv); hipFree(c->mb_grad);
  for (auto& e : c->prof_ev) hipEventDestroy(e);
  hipFree(c->hcache);
  hipFree(c->ocache);
  hipFree(c->snap);
  hipFree(c->partials); hipFree(c->spartials); hipFree(c->ident_tr); hipFree(c->raw_perm);
  hipFree(c->cg_x); hipFree(c->cg_r); hipFree(c->cg_p); hipFree(c->cg_z); hipFree(c->cg_Ap); hipFree(c->ticket); hipFree(c->cg_scal);
  delete c;
}

int64_t mjx_num_params(const mjx_ctx* c) { return c ? c->d : -1; }
int mjx_uses_fused_path(const mjx_ctx* c) { return c ? (c->fused != 0) : 0; }

int mjx_malloc(void** p, int64_t bytes) {
  MJX_DEVICE_ENTRY();
  if (!p || bytes < 0) return fail(MJX_ERR_ARG, "bad arguments");
  HIPCHK(hipMalloc(p, (size_t)bytes + 16));      // + one 16-byte granule: the tail reads of mjx_bind_batch's observation block stay inside
  return MJX_OK;
}
int mjx_free(void* p) { HIPCHK(hipFree(p)); return MJX_OK; }
int mjx_memcpy_h2d(void* dst, const void* src, int64_t bytes, void* stream) {
  HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream)); return MJX_OK; }
int mjx_memcpy_d2h(void* dst, const void* src, int64_t bytes, void* stream) {
  HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream)); return MJX_OK; }
int mjx_stream_sync(void* stream) { HIPCHK(hipStreamSynchronize((hipStream_t)stream)); return MJX_OK; }

int mjx_bind_batch(mjx_ctx* c, const float* obs, const float* act, const float* adv, int64_t N_local, int64_t N_global) {
  if (!c || (!obs && N_local > 0) || N_local < 0 || N_global < N_local || N_global <= 0) return fail(MJX_ERR_ARG, "bad batch");
  if (((uintptr_t)obs & 15) != 0) return fail(MJX_ERR_ARG, "obs must be 16-byte aligned");
  c->obs = obs; c->act = act; c->adv = adv; c->N_local = N_local; c->N_global = N_global;
  c->batch_bound = true;
  c->rows_bound = N_local;
  c->hcache_valid = false;                      // a new batch: nothing cached from earlier calls applies
  c->ocache_valid = false;
  c->ximg_ok = false;
  c->lw.invalidate();
  c->lw_old_ok = false;
  if (!c->fused) {
    // the layer-wise launches put one 128-row tile per grid row: gridDim.y <= 65 535 -> 8 388 480 rows per context (BASELINE
    // configs[4] at its full 8M + demonstrations fits; beyond that, shard the batch -- element and byte offsets are 64-bit throughout)
    if (N_local > (int64_t)65535 * 128) return fail(MJX_ERR_ARG, "layer-wise path: at most %lld rows per context (got %lld): bind the batch in shards", (long long)65535 * 128, (long long)N_local);
    int rc = c->lw.reserve(N_local); if (rc) return fail(rc, "layer-wise workspace allocation failed");
  }
  return MJX_OK;
}

int mjx_bind_rows(mjx_ctx* c, int64_t N_local, int64_t N_global, const float* adv) {
  if (!c || !c->batch_bound) return fail(MJX_ERR_STATE, "mjx_bind_batch has not been called");
  if (N_local < 0 || N_local > c->rows_bound || N_global < N_local || N_global <= 0) return fail(MJX_ERR_ARG, "bad row count");
  c->N_local = N_local; c->N_global = N_global;
  if (adv) c->adv = adv;
  c->lw.narrow(N_local);                        // (r06: a prefix of the bound rows keeps the activations cached for it -- DAPG's on-policy prefix)
  return MJX_OK;
}

static int bind_policy_impl(mjx_ctx* c, const float* theta_new, const float* theta_old, const float* tr_new,
                            const float* tr_old, int old_is_new, bool keep_old_outputs);
int mjx_bind_policy(mjx_ctx* c, const float* theta_new, const float* theta_old, const float* tr_new,
                    const float* tr_old, int old_is_new) {
  return bind_policy_impl(c, theta_new, theta_old, tr_new, tr_old, old_is_new, false);
}
// keep_old_outputs: the one-call updates re-bind (stepped parameters, the SAME old parameters) between their K1 and their
// evaluations -- the old policy's cached outputs stay usable; a caller's own binding may come with new contents under the same pointers
static int bind_policy_impl(mjx_ctx* c, const float* theta_new, const float* theta_old, const float* tr_new,
                            const float* tr_old, int old_is_new, bool keep_old_outputs) {
  if (!c || !theta_new || !theta_old) return fail(MJX_ERR_ARG, "bad policy");
  if (c && !(keep_old_outputs && theta_old == c->theta_old && tr_old == c->tr_old && tr_new == c->tr_new)) c->lw_old_ok = false;
  if ((((uintptr_t)theta_new) | ((uintptr_t)theta_old)) & 15) return fail(MJX_ERR_ARG, "parameter vectors must be 16-byte aligned");
  c->theta_new = theta_new; c->theta_old = theta_old; c->tr_new = tr_new; c->tr_old = tr_old;
  c->old_is_new = old_is_new ? 1 : 0;
  c->lw.invalidate();
  c->hcache_valid = false;
  return MJX_OK;
}

// ---------------------------------------------------------------------------- K6 baselines
namespace {
struct FeatTableCache { int kind = -1, n = -1, F = 0; FeatDesc* dev = nullptr; };
int get_feat_table(int kind, int n, FeatTableCache** out) {
  static thread_local FeatTableCache cache[3];
  if (kind < 0 || kind > 2 || n <= 0 || n > 4096) return fail(MJX_ERR_ARG, "bad feature kind / obs dim");
  FeatTableCache& c = cache[kind];
  if (c.n != n) {
    std::vector<FeatDesc> t = build_feat_table(kind, n);
    if (c.dev) hipFree(c.dev);
    HIPCHK(hipMalloc(&c.dev, t.size() * sizeof(FeatDesc)));
    HIPCHK(hipMemcpy(c.dev, t.data(), t.size() * sizeof(FeatDesc), hipMemcpyHostToDevice));
    c.kind = kind; c.n = n; c.F = (int)t.size();
  }
  *out = &c;
  return MJX_OK;
}
struct Scratch { void* p = nullptr; size_t cap = 0; };
int get_scratch(Scratch& s, size_t bytes) {
  if (bytes <= s.cap) return MJX_OK;
  if (s.p) hipFree(s.p);
  s.p = nullptr; s.cap = 0;
  HIPCHK(hipMalloc(&s.p, bytes));
  s.cap = bytes;
  return MJX_OK;
}
}  // namespace

int mjx_bl_num_features(int kind, int n) { return (kind < 0 || kind > 2 || n <= 0) ? -1 : bl_num_features(kind, n); }

int mjx_bl_features_f32(const double* obs, const int32_t* tpos, int64_t N, int n, float* out, void* stream) {
  if (!obs || !tpos || !out || N < 0 || n <= 0) return fail(MJX_ERR_ARG, "bad arguments");
  if (N == 0) return MJX_OK;
  hipLaunchKernelGGL(k_bl_features_f32, dim3(LayerwiseWS::ew_grid(N * (n + 4))), dim3(256), 0, (hipStream_t)stream, obs, tpos, N, n, out);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_bl_gram(int kind, const double* obs, const int32_t* tpos, const double* y, int64_t N, int n, double* G, void* stream) {
  if (!obs || !tpos || !y || !G || N <= 0) return fail(MJX_ERR_ARG, "bad arguments");
  FeatTableCache* ft;
  if (int rc = get_feat_table(kind, n, &ft)) return rc;
  const int F = ft->F, FA = F + 1, nbt = (FA + GT - 1) / GT, npairs = nbt * (nbt + 1) / 2;
  static thread_local Scratch part;
  const int T16 = (FA + 15) / 16;
  const char* no_mfma = getenv("MJX_GRAM_FMA");
  if (T16 <= GM_TMAX && n <= 24 && !(no_mfma && no_mfma[0] == '1')) {
    // fp64 matrix cores: one persistent workgroup per sample range, all features generated once per chunk
    int Z = (int)((N + 2047) / 2048);
    if (Z > 512) Z = 512;
    if (Z < 1) Z = 1;
    if (int rc = get_scratch(part, (size_t)Z * FA * FA * sizeof(double))) return rc;
    HIPCHK(hipMemsetAsync(part.p, 0, (size_t)Z * FA * FA * sizeof(double), (hipStream_t)stream));
    const size_t lds = ((size_t)32 * (n + 7) + (size_t)32 * 16 * (T16 | 1)) * sizeof(double);
    static thread_local bool attr_set = false;
    if (!attr_set) { HIPCHK(hipFuncSetAttribute((const void*)k_bl_gram_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); attr_set = true; }
    if (lds > 64 * 1024) return fail(MJX_ERR_UNSUPPORTED, "Gram kernel LDS staging");
    hipLaunchKernelGGL(k_bl_gram_mfma, dim3(Z), dim3(256), lds, (hipStream_t)stream, ft->dev, F, n, obs, tpos, y, N, (double*)part.p);
    hipLaunchKernelGGL(k_bl_gram_reduce, dim3(LayerwiseWS::ew_grid((int64_t)FA * FA)), dim3(256), 0, (hipStream_t)stream, (const double*)part.p, Z, FA, G, 16);
    HIPCHK(hipGetLastError());
    return MJX_OK;
  }
  if (!(no_mfma && no_mfma[0] == '1') && n <= 64) {
    // fp64 matrix cores, 128 x 128 feature blocks: one workgroup per (block pair, sample range)
    const int nb = (FA + GB_F - 1) / GB_F, nbp = nb * (nb + 1) / 2;
    int Z = (int)((N + 2047) / 2048);
    int maxz = (768 + nbp - 1) / nbp;                  // ~3 rounds of workgroups on the chip
    if (Z > maxz) Z = maxz;
    if (Z < 1) Z = 1;
    if (int rc = get_scratch(part, (size_t)Z * FA * FA * sizeof(double))) return rc;
    const size_t lds = (size_t)(32 * (n + 7) + 2 * 32 * GB_FS) * sizeof(double);
    static thread_local bool attr_blk = false;
    if (!attr_blk) { HIPCHK(hipFuncSetAttribute((const void*)k_bl_gram_mfma_blk, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr_blk = true; }
    hipLaunchKernelGGL(k_bl_gram_mfma_blk, dim3(nbp, Z), dim3(256), lds, (hipStream_t)stream, nb, ft->dev, F, n, obs, tpos, y, N, (double*)part.p);
    hipLaunchKernelGGL(k_bl_gram_reduce, dim3(LayerwiseWS::ew_grid((int64_t)FA * FA)), dim3(256), 0, (hipStream_t)stream, (const double*)part.p, Z, FA, G, GB_F);
    HIPCHK(hipGetLastError());
    return MJX_OK;
  }
  int Z = (int)((N + 4095) / 4096);
  int maxz = (2048 + npairs - 1) / npairs;
  if (Z > maxz) Z = maxz;
  if (Z < 1) Z = 1;
  if (int rc = get_scratch(part, (size_t)Z * FA * FA * sizeof(double))) return rc;
  HIPCHK(hipMemsetAsync(part.p, 0, (size_t)Z * FA * FA * sizeof(double), (hipStream_t)stream));
  size_t lds = ((size_t)GKS * n + 2 * (size_t)GKS * (GT + 1)) * sizeof(double);
  if (lds > 64 * 1024) return fail(MJX_ERR_UNSUPPORTED, "obs dim too large for the Gram kernel's LDS staging");
  hipLaunchKernelGGL(k_bl_gram, dim3(npairs, Z), dim3(256), lds, (hipStream_t)stream, nbt, ft->dev, F, n, obs, tpos, y, N, (double*)part.p);
  hipLaunchKernelGGL(k_bl_gram_reduce, dim3(LayerwiseWS::ew_grid((int64_t)FA * FA)), dim3(256), 0, (hipStream_t)stream, (const double*)part.p, Z, FA, G, GT);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_bl_predict(int kind, const double* obs, const int32_t* tpos, int64_t N, int n, const double* coef, double* out, void* stream) {
  if (!obs || !tpos || !coef || !out || N < 0) return fail(MJX_ERR_ARG, "bad arguments");
  if (N == 0) return MJX_OK;
  FeatTableCache* ft;
  if (int rc = get_feat_table(kind, n, &ft)) return rc;
  int nth = 256;                                          // one observation row per thread in LDS: fewer threads for wide observations
  while (nth > 64 && (size_t)nth * n * sizeof(double) > 64 * 1024) nth >>= 1;
  size_t lds = (size_t)nth * n * sizeof(double);
  if (lds > 64 * 1024) return fail(MJX_ERR_UNSUPPORTED, "obs dim too large for the predict kernel's LDS staging");
  int grid = (int)((N + nth - 1) / nth); if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_bl_predict, dim3(grid), dim3(nth), lds, (hipStream_t)stream, ft->dev, ft->F, n, obs, tpos, coef, N, out);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_mlp_predict(const float* feat, int64_t N, int d_in, const int* hidden, int n_hidden, const float* params, float* out, void* stream) {
  if (!feat || !params || !out || N < 0 || d_in <= 0 || n_hidden < 0 || (n_hidden && !hidden)) return fail(MJX_ERR_ARG, "bad arguments");
  if (N == 0) return MJX_OK;
  {
    // the reference's default value network (128 x 128 ReLU) in one launch: csrc/baseline.h k_mlp_predict128 (MJX_MLP_PREDICT_FUSED=0: A/B)
    const char* e = getenv("MJX_MLP_PREDICT_FUSED");
    if (!(e && e[0] == '0') && n_hidden == 2 && hidden[0] == 128 && hidden[1] == 128 && d_in <= 64 && (((uintptr_t)params) & 15) == 0 &&
        mlp_predict_lds_bytes(d_in) <= (size_t)160 * 1024) {
      int dev = 0, ncu = 256;
      HIPCHK(hipGetDevice(&dev));
      (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
      static thread_local std::vector<std::pair<int, size_t>> configured;      // (device, bytes): the dynamic-LDS limit is per device
      const size_t bytes = mlp_predict_lds_bytes(d_in);
      size_t* have = nullptr;
      for (auto& c : configured) if (c.first == dev) have = &c.second;
      if (!have) { configured.push_back({dev, 0}); have = &configured.back().second; }
      if (*have < bytes) { HIPCHK(hipFuncSetAttribute((const void*)k_mlp_predict128, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)); *have = bytes; }
      const int64_t ntiles = (N + 31) / 32;
      const int64_t want = (ntiles + 7) / 8;
      MlpPredictArgs pa{feat, params, out, N, d_in};
      hipLaunchKernelGGL(k_mlp_predict128, dim3((unsigned)(want < ncu ? want : ncu)), dim3(512), bytes, (hipStream_t)stream, pa);
      HIPCHK(hipGetLastError());
      return MJX_OK;
    }
  }
  MlpRegressor net; net.init(d_in, hidden, n_hidden);
  const int64_t CH = 1 << 17;
  size_t per_row = 0; for (int i = 0; i < n_hidden; ++i) per_row += hidden[i];
  // the activation scratch is kept PER STREAM: MLPBaseline.fit_async runs its before / after forwards on a side stream while the
  // caller's stream may be predicting with another baseline -- one shared block would be written by both at once
  static thread_local std::vector<std::pair<void*, Scratch>> by_stream;
  Scratch* scp = nullptr;
  for (auto& e : by_stream) if (e.first == stream) scp = &e.second;
  if (!scp) {
    if (by_stream.size() >= 8) {                 // (streams come and go: forget the oldest block once nothing can still be using it)
      HIPCHK(hipDeviceSynchronize());
      if (by_stream.front().second.p) (void)hipFree(by_stream.front().second.p);
      by_stream.erase(by_stream.begin());
    }
    by_stream.push_back({stream, Scratch{}});
    scp = &by_stream.back().second;
  }
  Scratch& sc = *scp;
  if (int rc = get_scratch(sc, (size_t)CH * (per_row ? per_row : 1) * sizeof(float))) return rc;
  std::vector<float*> acts(n_hidden);
  { float* q = (float*)sc.p; for (int i = 0; i < n_hidden; ++i) { acts[i] = q; q += (size_t)CH * hidden[i]; } }
  for (int64_t r0 = 0; r0 < N; r0 += CH) {
    int64_t rows = (N - r0 < CH) ? N - r0 : CH;
    net.forward(params, feat + r0 * d_in, rows, acts, out + r0, (hipStream_t)stream);
  }
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_mlp_fit_adam(const float* feat, const float* y, int64_t N, int d_in, const int* hidden, int n_hidden, float* params,
                     float* m, float* v, int64_t step0, const int32_t* perm, int epochs, int batch, float lr, float wd,
                     double* epoch_loss_out, void* stream) {
  if (!feat || !y || !params || !m || !v || !perm || !epoch_loss_out || N <= 0 || batch <= 0 || epochs < 0 || n_hidden < 0)
    return fail(MJX_ERR_ARG, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(epoch_loss_out, 0, sizeof(double) * (epochs > 0 ? epochs : 1), st));
  {
    // persistent single-workgroup trainer (csrc/mlp_fit.h) for the reference's default baseline shape
    const char* force = getenv("MJX_MLP_FIT_LAUNCHES");
    const int64_t steps_ = N / batch - 1;
    MlpFitLayout<128> L(d_in, d_in > 31);
    if (!(force && force[0] == '1') && n_hidden == 2 && hidden[0] == 128 && hidden[1] == 128 && batch == 64 && d_in <= 63 &&
        L.bytes() <= (size_t)160 * 1024 && steps_ > 0 && epochs > 0) {
      static thread_local Scratch mvws;
      const int64_t P = (int64_t)128 * d_in + 128 + 128 * 128 + 128 + 128 + 1;
      if (int rc = get_scratch(mvws, (size_t)P * 2 * sizeof(float))) return rc;
      MlpFitArgs a{feat, y, perm, N, d_in, epochs, steps_, params, m, v, (float*)mvws.p, step0, lr, wd, epoch_loss_out};
      static thread_local bool configured = false;
      if (!configured) {
        HIPCHK(hipFuncSetAttribute((const void*)k_mlp_fit<128, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        HIPCHK(hipFuncSetAttribute((const void*)k_mlp_fit<128, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        HIPCHK(hipFuncSetAttribute((const void*)k_mlp_fit<128, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        HIPCHK(hipFuncSetAttribute((const void*)k_mlp_fit<128, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        HIPCHK(hipFuncSetAttribute((const void*)k_mlp_fit1p<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        configured = true;
      }
      // r04: the Adam moments a thread owns stay in its registers for the whole run (REGMOM, mlp_fit.h); MJX_FIT_REGMOM=0: the r03
      // kernels, which stream them through L2 in every step (same arithmetic, same bits)
      const char* rm = getenv("MJX_FIT_REGMOM");            // (read per call: the tests compare the two in one process)
      const bool regmom = !(rm && rm[0] == '0');
      // ... and up to 23 inputs a step is ONE pass over its 64 rows (k_mlp_fit1p: two accumulator chains, every weight fragment
      // fetched once, 6 barriers instead of 11); MJX_FIT_ONEPASS=0: the two-halves kernel
      const char* op = getenv("MJX_FIT_ONEPASS");
      const MlpFit1pLayout<128> L1(d_in);
      if (d_in <= 23 && regmom && !(op && op[0] == '0') && L1.bytes() <= (size_t)160 * 1024) {
        hipLaunchKernelGGL((k_mlp_fit1p<128>), dim3(1), dim3(256), L1.bytes(), st, a);
      } else if (d_in <= 31) {
        if (regmom) hipLaunchKernelGGL((k_mlp_fit<128, 1, true>), dim3(1), dim3(256), L.bytes(), st, a);
        else hipLaunchKernelGGL((k_mlp_fit<128, 1, false>), dim3(1), dim3(256), L.bytes(), st, a);
      } else {                                                // (32 <= d_in <= 55: what 160 KB of LDS hold)
        if (regmom) hipLaunchKernelGGL((k_mlp_fit<128, 2, true>), dim3(1), dim3(256), L.bytes(), st, a);
        else hipLaunchKernelGGL((k_mlp_fit<128, 2, false>), dim3(1), dim3(256), L.bytes(), st, a);
      }
      HIPCHK(hipGetLastError());
      return MJX_OK;
    }
  }
  {
    // r05: wider inputs (Ant 111 + 4, Humanoid 376 + 4 = BASELINE configs[3]) on SEVERAL workgroups of the same persistent trainer:
    // feature slices of 48, one workgroup (= one CU) per slice, one grid barrier per 32-sample half (k_mlp_fit<.., MULTI>, mlp_fit.h)
    const char* force = getenv("MJX_MLP_FIT_LAUNCHES");
    const char* wide = getenv("MJX_FIT_WIDE");
    const int64_t steps_ = N / batch - 1;
    constexpr int FS = 48, GMAX = 16;
    const int G = (d_in + FS - 1) / FS;
    if (!(force && force[0] == '1') && !(wide && wide[0] == '0') && n_hidden == 2 && hidden[0] == 128 && hidden[1] == 128 && batch == 64 &&
        d_in > 48 && G <= GMAX && steps_ > 0 && epochs > 0) {
      MlpFitLayout<128> Lw(FS, true);
      static thread_local Scratch mvws;
      const int64_t P = (int64_t)128 * d_in + 128 + 128 * 128 + 128 + 128 + 1;
      if (int rc = get_scratch(mvws, (size_t)G * P * 2 * sizeof(float))) return rc;
      // the exchange block and the arrival counter: UNCACHED device memory (stores are acknowledged by memory, loads bypass the
      // per-XCD L2s -- the workgroups of one grid sit on different XCDs), allocated once per thread and device
      static thread_local std::vector<std::pair<int, char*>> xch_by_dev;
      int dev = 0;
      HIPCHK(hipGetDevice(&dev));
      char* xch = nullptr;
      for (auto& e : xch_by_dev) if (e.first == dev) xch = e.second;
      const size_t xbytes = (size_t)2 * GMAX * 128 * 32 * sizeof(float);
      if (!xch) {
        void* q = nullptr;
        HIPCHK(hipExtMallocWithFlags(&q, xbytes + 256, hipDeviceMallocUncached));
        xch = (char*)q;
        xch_by_dev.push_back({dev, xch});
      }
      HIPCHK(hipMemsetAsync(xch + xbytes, 0, 256, st));                       // the counter (the exchange block needs no clearing)
      MlpFitArgs a{feat, y, perm, N, d_in, epochs, steps_, params, m, v, (float*)mvws.p, step0, lr, wd, epoch_loss_out};
      a.xch = (float*)xch; a.bar = (unsigned*)(xch + xbytes); a.G = G; a.FS = FS;
      { const char* f = getenv("MJX_FIT_FAULT"); a.fault = (f && !strcmp(f, "straggler")) ? 1 : 0; }      // (tests/test_gpu_lifecycle.py)
      static thread_local bool configured = false;
      if (!configured) {
        HIPCHK(hipFuncSetAttribute((const void*)k_mlp_fit<128, 2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        HIPCHK(hipFuncSetAttribute((const void*)k_mlp_fit<128, 2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        configured = true;
      }
      const char* rm = getenv("MJX_FIT_REGMOM");
      if (!(rm && rm[0] == '0')) hipLaunchKernelGGL((k_mlp_fit<128, 2, true, true>), dim3(G), dim3(256), Lw.bytes(), st, a);
      else hipLaunchKernelGGL((k_mlp_fit<128, 2, false, true>), dim3(G), dim3(256), Lw.bytes(), st, a);
      // a barrier wait that timed out (a workgroup without a CU) is recorded in a word of its own; this turns it into NaN epoch
      // losses AFTER every workgroup has left -- the last writer of epoch_loss no longer decides
      hipLaunchKernelGGL(k_mlp_fit_verdict, dim3(1), dim3(64), 0, st, (const unsigned*)a.bar + 1, epoch_loss_out, epochs);
      HIPCHK(hipGetLastError());
      return MJX_OK;
    }
  }
  MlpRegressor net; net.init(d_in, hidden, n_hidden);
  const int L = net.nL(), bs = batch;
  size_t hsum = 0; for (int i = 0; i < n_hidden; ++i) hsum += hidden[i];
  // scratch: Xb, yb, yhat, d_out, acts (bs x h_l), deltas (bs x h_l), grads (P)
  static thread_local Scratch sc;
  size_t fl = (size_t)bs * d_in + 3 * (size_t)bs + 2 * (size_t)bs * hsum + (size_t)net.P;
  if (int rc = get_scratch(sc, fl * sizeof(float))) return rc;
  float* q = (float*)sc.p;
  float* Xb = q; q += (size_t)bs * d_in;
  float* yb = q; q += bs;
  float* yhat = q; q += bs;
  float* dout = q; q += bs;
  std::vector<float*> acts(n_hidden), dl(n_hidden);
  for (int i = 0; i < n_hidden; ++i) { acts[i] = q; q += (size_t)bs * hidden[i]; }
  for (int i = 0; i < n_hidden; ++i) { dl[i] = q; q += (size_t)bs * hidden[i]; }
  float* grads = q;
  const int64_t steps = N / bs - 1;                 // optimize_model.py:24
  int64_t t = step0;
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
  for (int ep = 0; ep < epochs; ++ep) {
    for (int64_t mb = 0; mb < steps; ++mb) {
      hipLaunchKernelGGL(k_gather_rows, dim3((bs * d_in + 255) / 256), dim3(256), 0, st, feat, y, perm + (int64_t)ep * N + mb * bs, bs, d_in, Xb, yb);
      net.forward(params, Xb, bs, acts, yhat, st);
      hipLaunchKernelGGL(k_mse_grad, dim3(1), dim3(256), 0, st, yhat, yb, bs, dout, epoch_loss_out + ep);
      const float* delta = dout;
      for (int l = L - 1; l >= 0; --l) {
        const int ho = net.sizes[l + 1], hi_ = net.sizes[l];
        const float* in = (l == 0) ? Xb : acts[l - 1];
        GemmArgs g{};
        g.M = ho; g.N = hi_; g.npairs = 1; g.K[0] = bs;
        g.A[0] = delta; g.a_rs[0] = 1; g.a_ks[0] = ho;
        g.B[0] = in; g.b_cs[0] = 1; g.b_ks[0] = hi_;
        g.C = grads + net.oW[l]; g.ldc = hi_; g.c_zs = 0; g.epi = EPI_STORE;
        LayerwiseWS::launch_gemm(g, 1, st);
        hipLaunchKernelGGL(k_colsum, dim3((ho + 63) / 64, 1), dim3(256), 0, st, delta, (int64_t)bs, ho, (int64_t)ho, grads + net.ob[l]);
        if (l > 0) {
          GemmArgs b{};
          b.M = bs; b.N = hi_; b.npairs = 1; b.K[0] = ho;
          b.A[0] = delta; b.a_rs[0] = ho; b.a_ks[0] = 1;
          b.B[0] = params + net.oW[l]; b.b_cs[0] = 1; b.b_ks[0] = hi_;
          b.C = dl[l - 1]; b.ldc = hi_; b.c_zs = 0;
          b.aux = acts[l - 1]; b.ld_aux = hi_; b.epi = EPI_BACK_RELU;
          LayerwiseWS::launch_gemm(b, 1, st);
          delta = dl[l - 1];
        }
      }
      ++t;
      const float bc1 = (float)(1.0 - std::pow((double)b1, (double)t));
      const float bc2s = (float)std::sqrt(1.0 - std::pow((double)b2, (double)t));
      hipLaunchKernelGGL(k_adam, dim3((unsigned)((net.P + 255) / 256)), dim3(256), 0, st, params, grads, m, v, net.P, lr, wd, b1, b2, eps, bc1, bc2s);
    }
  }
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_profile_enable(mjx_ctx* c, int on) {
  if (!c) return fail(MJX_ERR_ARG, "null context");
  HIPCHK(hipSetDevice(c->device));
  if (on && c->prof_ev.empty()) {
    c->prof_ev.resize(2 * 2048);
    for (auto& e : c->prof_ev) HIPCHK(hipEventCreate(&e));
  }
  c->prof_on = on != 0;
  c->prof_iter = on < 0;                       // on = -k: every k-th CG ITERATION (product, reduction / exchange, vector update) instead of the product alone
  if (on) { c->prof_used = 0; c->prof_seen = 0; c->prof_stride = on < 0 ? -on : on; }
  return MJX_OK;
}

int mjx_profile_read(mjx_ctx* c, double* out) {
  if (!c || !out) return fail(MJX_ERR_ARG, "bad arguments");
  HIPCHK(hipSetDevice(c->device));
  double tot = 0.0;
  for (size_t i = 0; i + 1 < c->prof_used; i += 2) {
    HIPCHK(hipEventSynchronize(c->prof_ev[i + 1]));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->prof_ev[i], c->prof_ev[i + 1]));
    tot += ms;
  }
  out[0] = tot; out[1] = (double)(c->prof_used / 2);
  return MJX_OK;
}

int mjx_profile_samples(mjx_ctx* c, double* ms_out_host, int cap, int* count_out) {
  if (!c || !count_out || (cap > 0 && !ms_out_host)) return fail(MJX_ERR_ARG, "bad arguments");
  HIPCHK(hipSetDevice(c->device));
  int n = 0;
  for (size_t i = 0; i + 1 < c->prof_used; i += 2, ++n) {
    if (n >= cap) continue;
    HIPCHK(hipEventSynchronize(c->prof_ev[i + 1]));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->prof_ev[i], c->prof_ev[i + 1]));
    ms_out_host[n] = (double)ms;
  }
  *count_out = n;
  return MJX_OK;
}

int mjx_set_debug_buffer(mjx_ctx* c, float* dbg, int64_t floats) {
  if (!c || (dbg && floats < 2048 * 8)) return fail(MJX_ERR_ARG, "debug buffer too small");
  c->dbg = dbg;
#ifdef MJX_PHASE_CLOCK
  // timing build: a buffer of at least LW_CLK_SLOTS slots also receives the per-workgroup stamps of the next k_gemm launches
  lw_clk_buf() = (dbg && floats >= 2 * (int64_t)LW_CLK_SLOT * LW_CLK_SLOTS) ? (long long*)dbg : nullptr;
  lw_clk_slot() = 0;
#endif
  return MJX_OK;
}

int mjx_set_clock_buffer(mjx_ctx* c, int64_t* clk) {
  if (!c) return fail(MJX_ERR_ARG, "null context");
  c->clk = (long long*)clk;
  return MJX_OK;
}

// pp (peer exchange): the gradient's reduction kernel writes this rank's slot (grad_out IS that slot) into every peer's buffer,
// K1's 4 sums travel with it at byte scal_off of the slot (scal_out: that place in the own slot), one arrival flag for both
static int surr_vpg_impl(mjx_ctx* c, float* grad_out, double* scal_out, void* stream, const PeerPush* pp, int scal_off) {
  if (int rc = check_bound(c, true)) return rc;
  if (!grad_out || !scal_out) return fail(MJX_ERR_ARG, "null output");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipSetDevice(c->device));
  if (c->N_local == 0) {                       // a rank without trajectories contributes zeros
    HIPCHK(hipMemsetAsync(grad_out, 0, c->d * sizeof(float), st));
    HIPCHK(hipMemsetAsync(scal_out, 0, 4 * sizeof(double), st));
    return MJX_OK;
  }
  if (!c->fused) {
    if (c->lw.surr_vpg(c->obs, c->act, c->adv, c->N_local, c->N_global, c->theta_new, c->theta_old,
                       c->tr_new ? c->tr_new : c->ident_tr, c->tr_old ? c->tr_old : c->ident_tr, c->old_is_new,
                       grad_out, scal_out, st)) return fail(MJX_ERR_STATE, "layer-wise surr_vpg failed");
    c->lw_old_ok = c->old_is_new != 0;          // its output block now holds the old policy's means for the bound rows
    return MJX_OK;
  }
  FusedArgs a = make_args(c, c->theta_old);
  c->hcache_valid = false;
  c->ximg_ok = false;
  c->fvp_seq = 0;
  if (c->use_hcache && c->old_is_new && c->hidden.size() == 2) {
    // keep h1 / h2 of every sample for the Fisher-vector products of this update (theta is fixed during CG)
    // per 32-sample tile: h1, h2 (sample-lane accumulator image) + the normalised observations (layer-1 operand image)
    const size_t need = (size_t)((c->N_local + 31) / 32) *
                        ((size_t)(c->hidden[0] / 32 + c->hidden[1] / 32) * 1024 + (size_t)(((c->n + 1 + 3) & ~3) / 4) * 128) * sizeof(float);
    if (need > c->hcache_bytes) {
      if (c->hcache) hipFree(c->hcache);
      c->hcache = nullptr; c->hcache_bytes = 0;
      if (hipMalloc(&c->hcache, need) == hipSuccess) c->hcache_bytes = need; else (void)hipGetLastError();
    }
    if (c->hcache) { a.hcache = c->hcache; c->hcache_valid = true; c->hcache_obs = c->obs; c->hcache_rows = c->N_local;
                     c->ximg_ok = true; c->ximg_rows = c->N_local; }
    // ... and the old policy's means / log-likelihoods for mjx_eval_surr_kl (old == new here), with a snapshot of
    // the parameters they belong to (the EVAL kernel compares before trusting them)
    const size_t oneed = (size_t)((c->N_local + 31) / 32) * 33 * 32 * sizeof(float);     // [tile][MP + 1][32] with MP <= 32 (largest fused variant)
    if (oneed > c->ocache_bytes) {
      if (c->ocache) hipFree(c->ocache);
      c->ocache = nullptr; c->ocache_bytes = 0;
      if (hipMalloc(&c->ocache, oneed) == hipSuccess) c->ocache_bytes = oneed; else (void)hipGetLastError();
    }
    if (!c->snap && hipMalloc(&c->snap, (size_t)(c->d + 2 * (c->n + c->m)) * sizeof(float)) != hipSuccess) { c->snap = nullptr; (void)hipGetLastError(); }
    c->ocache_valid = false;
    if (c->ocache && c->snap) {
      a.snap_out = c->snap;                      // written by the kernel itself (no separate copies on the stream)
      a.ocache = c->ocache; c->ocache_valid = true; c->ocache_rows = c->N_local;
    }
  }
  // (the one-call updates' evaluations may trust the snapshot without comparing: they run before anything outside the library can)
  c->lw_old_ok = c->old_is_new != 0 && a.ocache != nullptr;
  if (int rc = dispatch_fused(c, MODE_VPG, a, st)) return rc;
  if ((c->d & 3) == 0 || a.raw_dr > 0) {
    // the 4 sums are reduced by one extra workgroup of the vector reduction (r06: no launch of their own)
    const int cols = a.raw_dr > 0 ? a.raw_dr : (int)c->d;
    hipLaunchKernelGGL(k_reduce_partials4, dim3((cols + 31) / 32 + 1), dim3(256), 0, st, c->partials, c->grid, cols,
                       grad_out, (const float*)nullptr, (const float*)nullptr, c->oS, 0.f, pp ? *pp : PeerPush{},
                       ScalTail{c->spartials, c->grid, scal_out, pp ? scal_off : -1}, a.raw_dr > 0 ? (const int*)c->raw_perm : (const int*)nullptr);
  } else {
    hipLaunchKernelGGL(k_reduce_partials, dim3((c->d + 15) / 16), dim3(256), 0, st, c->partials, c->grid, (int)c->d,
                       grad_out, (const float*)nullptr, (const float*)nullptr, c->oS, 0.f);
    hipLaunchKernelGGL(k_reduce_scalars, dim3(1), dim3(256), 0, st, c->spartials, c->grid, scal_out, PeerPush{});
  }
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_surr_vpg(mjx_ctx* c, float* grad_out, double* scal_out, void* stream) { return surr_vpg_impl(c, grad_out, scal_out, stream, nullptr, -1); }

// pp: peer exchange folded into the reduction kernel (fused kernels with d % 4 == 0 only; `out` is then this rank's slot)
static int fvp_impl(mjx_ctx* c, const float* v, float* out, void* stream, const PeerPush* pp) {
  if (int rc = check_bound(c, false)) return rc;
  if (!v || !out) return fail(MJX_ERR_ARG, "null vector");
  if (c->fused && (((uintptr_t)v) & 15)) return fail(MJX_ERR_ARG, "v must be 16-byte aligned (fused path)");   // (the layer-wise path takes any 4-byte aligned v)
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipSetDevice(c->device));
  if (c->N_local == 0) { HIPCHK(hipMemsetAsync(out, 0, c->d * sizeof(float), st)); return MJX_OK; }
  if (!c->old_is_new) {
    // general position (e.g. input_normalization, npg_cg.py:101-107): exact Pearlmutter product on the layer-wise path
    if (c->lw.cap < c->N_local) { if (int rc = c->lw.reserve(c->N_local)) return fail(rc, "layer-wise workspace allocation failed"); }
    int rc = c->lw.hvp_general(c->obs, c->N_local, c->N_global, c->theta_new, c->theta_old, c->tr_new ? c->tr_new : c->ident_tr,
                               c->tr_old ? c->tr_old : c->ident_tr, v, out, st);
    return rc ? fail(MJX_ERR_STATE, "general Hessian-vector product failed (%d)", rc) : MJX_OK;
  }
  const float frac = (float)((double)c->N_local / (double)c->N_global);
  const bool prof = c->prof_on && !c->prof_iter && (c->prof_seen++ % (size_t)c->prof_stride == 0) && c->prof_used + 2 <= c->prof_ev.size();
  if (prof) HIPCHK(hipEventRecord(c->prof_ev[c->prof_used], st));
  if (!c->fused) {
    int rc = c->lw.fvp(c->obs, c->N_local, c->N_global, c->theta_new, c->tr_new ? c->tr_new : c->ident_tr, v, out, st);
    if (prof) { HIPCHK(hipEventRecord(c->prof_ev[c->prof_used + 1], st)); c->prof_used += 2; }
    return rc ? fail(MJX_ERR_STATE, "layer-wise fvp failed") : MJX_OK;
  }
  FusedArgs a = make_args(c, v);
  if (c->hcache_valid && c->N_local <= c->hcache_rows) a.hcache = c->hcache;
  // alternate sweep direction: K1 filled the cache front to back, so the first product of a solve starts at the back, the
  // next at the front, ... -- the lines the previous sweep touched last are the ones most likely still held by the
  // memory-side cache (the 592 MB image does not fit; a one-directional walk would evict every line before its reuse)
  const bool sweep_on = [] { const char* e = getenv("MJX_FVP_SWEEP"); return !(e && e[0] == '0'); }();
  a.reverse = (a.hcache && sweep_on) ? (int)((c->fvp_seq++ & 1u) ^ 1u) : 0;
  if (int rc = dispatch_fused(c, MODE_FVP, a, st)) return rc;
  if (prof) { HIPCHK(hipEventRecord(c->prof_ev[c->prof_used + 1], st)); c->prof_used += 2; }
  if ((c->d & 3) == 0 || a.raw_dr > 0) {
    const int cols = a.raw_dr > 0 ? a.raw_dr : (int)c->d;
    hipLaunchKernelGGL(k_reduce_partials4, dim3((cols + 31) / 32), dim3(256), 0, st, c->partials, c->grid, cols,
                       out, c->theta_new, v, c->oS, frac, pp ? *pp : PeerPush{}, ScalTail{nullptr, 0, nullptr, -1},
                       a.raw_dr > 0 ? (const int*)c->raw_perm : (const int*)nullptr);
  } else
    hipLaunchKernelGGL(k_reduce_partials, dim3((c->d + 15) / 16), dim3(256), 0, st, c->partials, c->grid, (int)c->d,
                       out, c->theta_new, v, c->oS, frac);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_fvp(mjx_ctx* c, const float* v, float* out, void* stream) { return fvp_impl(c, v, out, stream, nullptr); }

// pp (peer exchange): the reduction of K3's sums writes them into slot `rank` of every buffer (scal_out IS the own slot) and raises
// the flags; the caller launches the consumer (k_peer_sum<double>)
static int eval_impl(mjx_ctx* c, double* scal_out, void* stream, const PeerPush* pp, bool old_ok = false) {
  if (int rc = check_bound(c, true)) return rc;
  if (!scal_out) return fail(MJX_ERR_ARG, "null output");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipSetDevice(c->device));
  if (c->N_local == 0) { HIPCHK(hipMemsetAsync(scal_out, 0, 4 * sizeof(double), st)); return MJX_OK; }
  if (!c->fused)
    return c->lw.eval(c->obs, c->act, c->adv, c->N_local, c->theta_new, c->theta_old,
                      c->tr_new ? c->tr_new : c->ident_tr, c->tr_old ? c->tr_old : c->ident_tr, scal_out, st,
                      old_ok && c->lw_old_ok, c->old_is_new != 0)
               ? fail(MJX_ERR_STATE, "layer-wise eval failed") : MJX_OK;
  FusedArgs a = make_args(c, c->theta_old);
  if (c->ocache_valid && c->N_local <= c->ocache_rows) { a.ocache = c->ocache; a.snap = c->snap; a.snap_trusted = (old_ok && c->lw_old_ok) ? 1 : 0; }
  // K1's normalised-observation image of this batch (same rows, same observations; the kernel checks the input transform
  // against the snapshot before it trusts it) spares K3 the staging and normalisation of the raw observations
  const bool ximg_on = [] { const char* e = getenv("MJX_K3_XIMG"); return !(e && e[0] == '0'); }();
  if (ximg_on && a.ocache && c->ximg_ok && c->hcache && c->N_local <= c->ximg_rows && c->obs == c->hcache_obs) a.hcache = c->hcache;
  if (int rc = dispatch_fused(c, MODE_EVAL, a, st)) return rc;
  hipLaunchKernelGGL(k_reduce_scalars, dim3(1), dim3(256), 0, st, c->spartials, 2 * c->grid, scal_out, pp ? *pp : PeerPush{});
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_eval_surr_kl(mjx_ctx* c, double* scal_out, void* stream) { return eval_impl(c, scal_out, stream, nullptr); }

// ---------------------------------------------------------------------------- peer exchange (HIP IPC + stream memory operations)
namespace {
bool has_ranks(const mjx_ctx* c) { return c->comm || c->reduce_cb || c->peer.on; }
// slot `r` (the vector rank r contributed) of parity `par` in rank q's buffer; the arrival counter and the producers' ticket
char* peer_slot(const mjx_ctx* c, int q, int par, int r) { return c->peer.map[q] + (size_t)(par * c->peer.world + r) * c->peer.slot_bytes; }
// the flag block behind the slots of rank q's buffer: [0..15] one arrival flag per SOURCE rank (32-bit exchange numbers),
// [32] timed-out waits of this rank's consumers (mjx_peer_status); a slot of zeros follows at +256
unsigned* peer_flags(const mjx_ctx* c, int q) { return (unsigned*)(c->peer.map[q] + (size_t)2 * c->peer.world * c->peer.slot_bytes); }
PeerPush peer_push(const mjx_ctx* c, int par, uint32_t seq) {
  PeerPush pp{};
  pp.world = c->peer.world; pp.rank = c->peer.rank; pp.seq = seq;
  pp.ticket = c->ticket;
  for (int q = 0; q < c->peer.world; ++q) {
    pp.dst[q] = peer_slot(c, q, par, c->peer.rank);
    // the flag rank q polls for this rank; loop-back rehearsal (every peer is the own buffer): the flag this rank polls for "peer" q
    pp.flag[q] = c->peer.loopback ? peer_flags(c, c->peer.rank) + q : peer_flags(c, q) + c->peer.rank;
    // fault injection (MJX_PEER_FAULT, tests/test_gpu_multirank.py): what a mis-mapped buffer or a lost flag store would look like
    // to the known-answer sum the host runs before it trusts a transport (engine._transport_self_test)
    if (q != c->peer.rank && c->peer.fault == 1) pp.dst[q] = peer_slot(c, q, par, (c->peer.rank + 1) % c->peer.world);
    if (q != c->peer.rank && c->peer.fault == 2) pp.flag[q] = peer_flags(c, c->peer.rank) + 40;     // an unused word of the OWN flag block
  }
  return pp;
}
// (exchange numbers are 32 bits and compared as a signed difference: they may wrap)
// the consumer's view of exchange `seq`: its local slots in rank order (the surplus entries: a slot of zeros behind the flag
// block) and the flags the peers raise to `seq` once their vectors have landed
PeerSlots peer_slots(const mjx_ctx* c, int par, uint32_t seq) {
  PeerSlots ps{};
  ps.world = c->peer.world; ps.rank = c->peer.rank; ps.seq = seq;
  const char* zeros = (const char*)peer_flags(c, c->peer.rank) + 256;
  for (int r = 0; r < 16; ++r) ps.slot[r] = r < c->peer.world ? peer_slot(c, c->peer.rank, par, r) : zeros;
  ps.flags = peer_flags(c, c->peer.rank);
  ps.timeouts = peer_flags(c, c->peer.rank) + 32;
  ps.ticks = c->peer.timeout_ticks;
  return ps;
}
int peer_allreduce(mjx_ctx* c, void* buf, int64_t count, int dtype, hipStream_t st) {
  const size_t bytes = (size_t)count * (dtype ? 8 : 4);
  if (bytes > c->peer.slot_bytes) return fail(MJX_ERR_ARG, "peer all-reduce of %zu bytes exceeds the slot (%zu)", bytes, c->peer.slot_bytes);
  const uint32_t seq = ++c->peer.seq;
  const int par = (int)(seq & 1u);
  const unsigned grid = (unsigned)((count + 255) / 256);
  if (dtype) hipLaunchKernelGGL(k_peer_push<double>, dim3(grid), dim3(256), 0, st, (const double*)buf, peer_push(c, par, seq), count);
  else hipLaunchKernelGGL(k_peer_push<float>, dim3(grid), dim3(256), 0, st, (const float*)buf, peer_push(c, par, seq), count);
  HIPCHK(hipGetLastError());
  if (dtype) hipLaunchKernelGGL(k_peer_sum<double>, dim3(grid), dim3(256), 0, st, peer_slots(c, par, seq), (double*)buf, count);
  else hipLaunchKernelGGL(k_peer_sum<float>, dim3(grid), dim3(256), 0, st, peer_slots(c, par, seq), (float*)buf, count);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}
}  // namespace

int mjx_cg_init(mjx_ctx* c, const float* b, void* stream) {
  if (!c || !b) return fail(MJX_ERR_ARG, "bad arguments");
  c->fvp_seq = 0;                              // every solve walks the cache in the same sequence of directions (reproducible bits)
  hipLaunchKernelGGL(k_cg_init, dim3(1), dim3(1024), 0, (hipStream_t)stream, b, c->cg_x, c->cg_r, c->cg_p, c->cg_scal, (int)c->d);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}
const float* mjx_cg_p(mjx_ctx* c) { return c ? c->cg_p : nullptr; }
int mjx_cg_step(mjx_ctx* c, const float* Ap, float damping, double tol, void* stream) {
  if (!c || !Ap) return fail(MJX_ERR_ARG, "bad arguments");
  if (c->d <= 8 * 1024)
    hipLaunchKernelGGL(k_cg_step_reg<8>, dim3(1), dim3(1024), 0, (hipStream_t)stream, Ap, damping, tol, c->cg_x, c->cg_r, c->cg_p,
                       c->cg_scal, (int)c->d);
  else {
    static const bool multi = [] { const char* e = getenv("MJX_CG_MULTI"); return !(e && e[0] == '0'); }();
    hipStream_t st = (hipStream_t)stream;
    if (multi && c->d >= 4 * CGM_G) {                   // (the partials fit c->cg_z: d floats >= 2 x CGM_G doubles)
      // large d: three launches of CGM_G workgroups (vecops.h); c->cg_z serves as their 2 x CGM_G fp64 partials
      double* part = (double*)c->cg_z;
      hipLaunchKernelGGL(k_cgm_pz, dim3(CGM_G), dim3(CGM_T), 0, st, Ap, (const float*)c->cg_p, damping, (const double*)c->cg_scal, part, (int)c->d);
      hipLaunchKernelGGL(k_cgm_xr, dim3(CGM_G), dim3(CGM_T), 0, st, Ap, (const float*)c->cg_p, damping, c->cg_x, c->cg_r, c->cg_scal, part, (int)c->d);
      hipLaunchKernelGGL(k_cgm_p, dim3(CGM_G), dim3(CGM_T), 0, st, (const float*)c->cg_r, c->cg_p, tol, c->cg_scal, (const double*)part, (int)c->d);
    } else
      hipLaunchKernelGGL(k_cg_step, dim3(1), dim3(1024), 0, st, Ap, damping, tol, c->cg_x, c->cg_r, c->cg_p,
                         c->cg_z, c->cg_scal, (int)c->d);
  }
  HIPCHK(hipGetLastError());
  return MJX_OK;
}
int mjx_cg_finish(mjx_ctx* c, const float* b, float* x_out, double* bdotx_out, void* stream) {
  if (!c || !b) return fail(MJX_ERR_ARG, "bad arguments");
  hipLaunchKernelGGL(k_cg_finish, dim3(1), dim3(1024), 0, (hipStream_t)stream, b, c->cg_x, x_out, bdotx_out, (int)c->d);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

}  // extern "C"  (internal helpers with C++ types follow)
namespace {
// d-float vectors the peer exchange folds into the loop's own kernels (k_reduce_partials4 / k_cg_step_reg<8, W>)
bool peer_folded(const mjx_ctx* c) { return c->peer.on && c->fused && (c->d & 3) == 0 && c->d <= 8 * 1024; }
// where the 4 doubles that travel with a gradient sit in a slot (mjx_peer_export sizes the slots for it)
int peer_scal_off(const mjx_ctx* c) { return (int)(((size_t)c->d * sizeof(float) + 15) & ~(size_t)15); }

// The solve, with (r06) what precedes and follows it folded into its own launches when the shapes allow:
//  * b_slots: the right-hand side arrives as one slot per rank (the gradient's peer exchange, with K1's 4 sums riding along):
//    k_cg_init_w waits, sums, leaves the summed gradient in `b` and the sums in s4_out;
//  * fin: after the LAST vector update the same kernel forms b.x, x_out and (mode 2 / 3) the stepped parameters -- k_cg_finish and
//    k_apply_*_step are not launched (d <= 8192 and at least one iteration; otherwise they are, as before).
int cg_solve_impl(mjx_ctx* c, float* b, int iters, float damping, double tol, float* x_out, double* bdotx_out, mjx_allreduce_fn allreduce,
                  void* user, void* stream, const PeerSlots* b_slots, double* s4_out, CgFin fin) {
  hipStream_t st = (hipStream_t)stream;
  if (b_slots) {
    c->fvp_seq = 0;
    const int off = peer_scal_off(c);
#define MJX_INIT_W(W) hipLaunchKernelGGL((k_cg_init_w<W>), dim3(1), dim3(1024), 0, st, *b_slots, off, b, s4_out, c->cg_x, c->cg_r, c->cg_p, c->cg_scal, (int)c->d)
    if (c->peer.world <= 2) MJX_INIT_W(2); else if (c->peer.world <= 4) MJX_INIT_W(4); else if (c->peer.world <= 8) MJX_INIT_W(8); else MJX_INIT_W(16);
#undef MJX_INIT_W
    HIPCHK(hipGetLastError());
  } else if (int rc = mjx_cg_init(c, b, stream)) return rc;
  const bool reg = c->d <= 8 * 1024;
  bool fin_done = false;
  fin.b = b; fin.x_out = x_out; fin.bdotx = bdotx_out; fin.oS = c->oS;
  for (int i = 0; i < iters; ++i) {
    const CgFin f = (reg && i == iters - 1) ? fin : CgFin{};
    // (mjx_profile_enable(ctx, -k): every k-th iteration as a whole -- product, reduction / exchange, vector update -- between two events)
    const bool piter = c->prof_on && c->prof_iter && i + 1 < iters && (c->prof_seen++ % (size_t)c->prof_stride == 0) && c->prof_used + 2 <= c->prof_ev.size();
    if (piter) HIPCHK(hipEventRecord(c->prof_ev[c->prof_used], st));
    struct IterEnd { mjx_ctx* c; hipStream_t st; bool on; ~IterEnd() { if (on) { (void)hipEventRecord(c->prof_ev[c->prof_used + 1], st); c->prof_used += 2; } } } iter_end{c, st, piter};
    if (!allreduce && peer_folded(c) && c->old_is_new && c->N_local > 0) {
      // peer exchange, folded into the loop's own kernels: the product's reduction kernel writes this rank's vector into slot `rank` of
      // EVERY rank's buffer and raises its arrival flags, the vector-update kernel waits on the own flags and sums its local
      // slots (rank order): per iteration FVP -> reduction -> step -- exactly the launches of the one-rank loop, no host round trip.
      // (A rank on another route -- empty shard -- runs the same exchange through mjx_comm_allreduce.)
      const uint32_t seq = ++c->peer.seq;
      const int par = (int)(seq & 1u);
      const PeerPush pp = peer_push(c, par, seq);
      if (int rc = fvp_impl(c, c->cg_p, (float*)peer_slot(c, c->peer.rank, par, c->peer.rank), stream, &pp)) return rc;
      const PeerSlots ps = peer_slots(c, par, seq);
#define MJX_STEP_W(W) hipLaunchKernelGGL((k_cg_step_reg<8, W>), dim3(1), dim3(1024), 0, st, (const float*)nullptr, damping, tol, \
                                         c->cg_x, c->cg_r, c->cg_p, c->cg_scal, (int)c->d, ps, f)
      if (c->peer.world <= 2) MJX_STEP_W(2); else if (c->peer.world <= 4) MJX_STEP_W(4); else if (c->peer.world <= 8) MJX_STEP_W(8); else MJX_STEP_W(16);
#undef MJX_STEP_W
      HIPCHK(hipGetLastError());
      fin_done = f.mode != 0;
      continue;
    }
    if (int rc = mjx_fvp(c, c->cg_p, c->cg_Ap, stream)) return rc;
    if (allreduce) { if (int rc = allreduce(user, c->cg_Ap, c->d, stream)) return fail(rc, "allreduce callback failed (%d)", rc); }
    else if (has_ranks(c)) { if (int rc = mjx_comm_allreduce(c, c->cg_Ap, c->d, 0, stream)) return rc; }
    if (reg) {
      hipLaunchKernelGGL(k_cg_step_reg<8>, dim3(1), dim3(1024), 0, st, (const float*)c->cg_Ap, damping, tol, c->cg_x, c->cg_r, c->cg_p,
                         c->cg_scal, (int)c->d, PeerSlots{}, f);
      HIPCHK(hipGetLastError());
      fin_done = f.mode != 0;
    } else if (int rc = mjx_cg_step(c, c->cg_Ap, damping, tol, stream)) return rc;
  }
  if (fin_done) return MJX_OK;
  if (int rc = mjx_cg_finish(c, b, x_out, bdotx_out, stream)) return rc;
  if (fin.mode == 2) return mjx_apply_npg_step(c, fin.theta, x_out, bdotx_out, fin.step_size, fin.min_log_std, fin.theta_out, fin.alpha_out, stream);
  if (fin.mode == 3) return mjx_apply_step(c, fin.theta, x_out, fin.const_alpha, fin.min_log_std, fin.theta_out, stream);
  return MJX_OK;
}

CgFin fin_step(const float* theta, float* theta_out, double* alpha_out, double step_size, double const_alpha, float min_log_std) {
  CgFin f;
  f.mode = std::isnan(const_alpha) ? 2 : 3;
  f.theta = theta; f.theta_out = theta_out; f.alpha_out = alpha_out; f.step_size = step_size;
  f.const_alpha = std::isnan(const_alpha) ? 0.f : (float)const_alpha; f.min_log_std = min_log_std;
  return f;
}
CgFin fin_only() { CgFin f; f.mode = 1; return f; }

// K1 and the rank sums of its outputs, up to the point where the solve starts.  -> *slots_out set when the gradient's sum is
// left to the solve's first kernel (peer exchange, folded: ONE exchange carries the gradient and K1's sums -- every rank of the
// job takes this route or none does: the condition depends on the architecture and the transport only)
int vpg_and_rank_sums(mjx_ctx* c, float* grad_out, double* s4, bool need_s4_sum, void* stream, PeerSlots* slots_out, bool* folded) {
  *folded = false;
  hipStream_t st = (hipStream_t)stream;
  if (peer_folded(c) && need_s4_sum) {
    const uint32_t seq = ++c->peer.seq;
    const int par = (int)(seq & 1u);
    const PeerPush pp = peer_push(c, par, seq);
    const int off = peer_scal_off(c);
    char* own = peer_slot(c, c->peer.rank, par, c->peer.rank);
    if (c->N_local > 0) {
      if (int rc = surr_vpg_impl(c, (float*)own, (double*)(own + off), stream, &pp, off)) return rc;
    } else {                                     // a rank without samples: zeros, through the same exchange
      HIPCHK(hipMemsetAsync(grad_out, 0, c->d * sizeof(float), st));
      HIPCHK(hipMemsetAsync(s4, 0, 4 * sizeof(double), st));
      hipLaunchKernelGGL(k_peer_push_vs, dim3((unsigned)((c->d + 255) / 256)), dim3(256), 0, st, (const float*)grad_out, (const double*)s4, pp, (int)c->d, off);
      HIPCHK(hipGetLastError());
    }
    *slots_out = peer_slots(c, par, seq);
    *folded = true;
    return MJX_OK;
  }
  if (int rc = mjx_surr_vpg(c, grad_out, s4, stream)) return rc;
  if (c->comm && need_s4_sum) {
    RcclApi& r = rccl();
    (void)r.GroupStart();                       // one launch for the gradient and its scalars
    int rc = mjx_comm_allreduce(c, grad_out, c->d, 0, stream);
    if (!rc) rc = mjx_comm_allreduce(c, s4, 4, 1, stream);
    const int ge = r.GroupEnd();
    if (rc) return rc;
    if (ge) return fail(1000 + ge, "ncclGroupEnd: %s", r.GetErrorString(ge));
  } else if (has_ranks(c)) {
    if (int rc = mjx_comm_allreduce(c, grad_out, c->d, 0, stream)) return rc;
    if (need_s4_sum) if (int rc = mjx_comm_allreduce(c, s4, 4, 1, stream)) return rc;
  }
  return MJX_OK;
}

// K3 and the rank sum of its 4 doubles (peer exchange: the push rides on the reduction kernel)
// (one-call updates only: their evaluations may use the old policy's outputs K1 left -- nothing outside the library ran in between)
int eval_and_rank_sum(mjx_ctx* c, double* res4, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (c->peer.on && c->fused && c->N_local > 0) {
    const uint32_t seq = ++c->peer.seq;
    const int par = (int)(seq & 1u);
    const PeerPush pp = peer_push(c, par, seq);
    if (int rc = eval_impl(c, (double*)peer_slot(c, c->peer.rank, par, c->peer.rank), stream, &pp, true)) return rc;
    hipLaunchKernelGGL(k_peer_sum<double>, dim3(1), dim3(256), 0, st, peer_slots(c, par, seq), res4, (int64_t)4);
    HIPCHK(hipGetLastError());
    return MJX_OK;
  }
  if (int rc = eval_impl(c, res4, stream, nullptr, true)) return rc;
  if (has_ranks(c)) if (int rc = mjx_comm_allreduce(c, res4, 4, 1, stream)) return rc;
  return MJX_OK;
}
}  // namespace
extern "C" {

int mjx_cg_solve(mjx_ctx* c, const float* b, int iters, float damping, double tol, float* x_out, double* bdotx_out,
                 mjx_allreduce_fn allreduce, void* user, void* stream) {
  if (!c || !b || iters < 0) return fail(MJX_ERR_ARG, "bad arguments");
  return cg_solve_impl(c, const_cast<float*>(b), iters, damping, tol, x_out, bdotx_out, allreduce, user, stream, nullptr, nullptr, fin_only());
}

// ---------------------------------------------------------------------------- multi-rank (RCCL, bound at run time)
int mjx_comm_unique_id(char* id_out) {
  if (!id_out) return fail(MJX_ERR_ARG, "null id buffer");
  RcclApi& r = rccl();
  if (!r.load()) return fail(MJX_ERR_UNSUPPORTED, "%s", r.error.c_str());
  RcclApi::UniqueId id;
  if (int rc = r.GetUniqueId(&id)) return fail(1000 + rc, "ncclGetUniqueId: %s", r.GetErrorString(rc));
  memcpy(id_out, id.internal, MJX_COMM_ID_BYTES);
  return MJX_OK;
}

int mjx_comm_init(mjx_ctx* c, int rank, int world, const char* id_host) {
  if (!c || !id_host || world < 1 || rank < 0 || rank >= world) return fail(MJX_ERR_ARG, "bad arguments");
  if (c->comm) return fail(MJX_ERR_STATE, "a communicator is already attached");
  RcclApi& r = rccl();
  if (!r.load()) return fail(MJX_ERR_UNSUPPORTED, "%s", r.error.c_str());
  HIPCHK(hipSetDevice(c->device));
  RcclApi::UniqueId id;
  memcpy(id.internal, id_host, MJX_COMM_ID_BYTES);
  RcclApi::Comm comm = nullptr;
  if (int rc = r.CommInitRank(&comm, world, id, rank)) return fail(1000 + rc, "ncclCommInitRank: %s", r.GetErrorString(rc));
  c->comm = comm; c->comm_world = world; c->comm_rank = rank;
  return MJX_OK;
}

int mjx_comm_destroy(mjx_ctx* c) {
  if (!c) return fail(MJX_ERR_ARG, "null context");
  if (c->peer.buf) {
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < c->peer.world; ++r)
      if (r != c->peer.rank && c->peer.map[r] && c->peer.map[r] != c->peer.buf) (void)hipIpcCloseMemHandle(c->peer.map[r]);
    (void)hipFree(c->peer.buf);
    c->peer = mjx_ctx::Peer{};
    c->comm_world = 0;
  }
  if (c->comm) {
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    (void)rccl().CommDestroy(c->comm);
    c->comm = nullptr; c->comm_world = 0;
  }
  return MJX_OK;
}

int mjx_comm_world(const mjx_ctx* c) { return (c && (c->comm || c->reduce_cb || c->peer.on)) ? c->comm_world : 0; }

int mjx_peer_export(mjx_ctx* c, int rank, int world, char* handle_out) {
  if (!c || !handle_out || world < 2 || world > 16 || rank < 0 || rank >= world) return fail(MJX_ERR_ARG, "bad arguments (2 <= world <= 16)");
  if (c->comm || c->reduce_cb || c->peer.buf) return fail(MJX_ERR_STATE, "a transport is already attached");
  HIPCHK(hipSetDevice(c->device));
  size_t slot = (((size_t)c->d * sizeof(float) + 15) & ~(size_t)15) + 4 * sizeof(double);   // a d-float vector + the 4 doubles that may travel with it (peer_scal_off)
  if (slot < 64 * sizeof(double)) slot = 64 * sizeof(double);
  slot = (slot + 255) & ~(size_t)255;
  void* p = nullptr;
  const size_t total = (size_t)(2 * world + 1) * slot + 256;      // [2 parities][world slots] | flag block (256 B) | one slot of zeros
  HIPCHK(hipExtMallocWithFlags(&p, total, hipDeviceMallocUncached));
  HIPCHK(hipMemset(p, 0, total));
  HIPCHK(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  { hipError_t e = hipIpcGetMemHandle(&h, p); if (e != hipSuccess) { (void)hipFree(p); return fail((int)e, "hipIpcGetMemHandle: %s", hipGetErrorString(e)); } }
  static_assert(sizeof(hipIpcMemHandle_t) == MJX_PEER_HANDLE_BYTES, "handle size");
  memcpy(handle_out, &h, sizeof h);
  c->peer.buf = (char*)p; c->peer.slot_bytes = slot; c->peer.rank = rank; c->peer.world = world; c->peer.seq = 0;
  if (const char* f = getenv("MJX_PEER_FAULT")) {
    const char* only = getenv("MJX_PEER_FAULT_RANK");               // (default: every rank misbehaves)
    if (!only || atoi(only) == rank) c->peer.fault = !strcmp(f, "slot") ? 1 : !strcmp(f, "flag") ? 2 : 0;
  }
  if (const char* ms = getenv("MJX_PEER_TIMEOUT_MS")) {             // how long a consumer kernel waits for a peer's vector
    const double v = atof(ms);
    if (v > 0.0) c->peer.timeout_ticks = (unsigned long long)(v * 1e5);
  }
  return MJX_OK;
}

int mjx_peer_connect(mjx_ctx* c, const char* handles) {
  if (!c) return fail(MJX_ERR_ARG, "null context");
  if (!c->peer.buf || c->peer.on) return fail(MJX_ERR_STATE, "call mjx_peer_export first (once)");
  HIPCHK(hipSetDevice(c->device));
  for (int r = 0; r < c->peer.world; ++r) {
    if (r == c->peer.rank || !handles) { c->peer.map[r] = c->peer.buf; continue; }     // handles == NULL: loop-back rehearsal
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)r * MJX_PEER_HANDLE_BYTES, sizeof h);
    void* q = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail((int)e, "hipIpcOpenMemHandle (rank %d): %s", r, hipGetErrorString(e));
    c->peer.map[r] = (char*)q;
  }
  c->peer.on = true; c->peer.loopback = (handles == nullptr);
  c->comm_world = c->peer.world; c->comm_rank = c->peer.rank;
  return MJX_OK;
}

int mjx_peer_status(mjx_ctx* c, int* timeouts_out) {
  if (!c || !timeouts_out) return fail(MJX_ERR_ARG, "bad arguments");
  *timeouts_out = 0;
  if (!c->peer.on) return MJX_OK;
  HIPCHK(hipSetDevice(c->device));
  unsigned n = 0;
  HIPCHK(hipMemcpy(&n, peer_flags(c, c->peer.rank) + 32, sizeof n, hipMemcpyDeviceToHost));   // (synchronises: read after the update's own read-back)
  if (n) HIPCHK(hipMemset(peer_flags(c, c->peer.rank) + 32, 0, sizeof n));
  *timeouts_out = (int)n;
  return MJX_OK;
}

int mjx_comm_set_callback(mjx_ctx* c, mjx_reduce_fn fn, void* user, int world) {
  if (!c || (fn && world < 1)) return fail(MJX_ERR_ARG, "bad arguments");
  if (c->comm) return fail(MJX_ERR_STATE, "an RCCL communicator is attached");
  c->reduce_cb = fn; c->reduce_user = user; c->comm_world = fn ? world : 0;
  return MJX_OK;
}

int mjx_comm_allreduce(mjx_ctx* c, void* buf, int64_t count, int dtype, void* stream) {
  if (!c || !buf || count < 0 || (dtype != 0 && dtype != 1)) return fail(MJX_ERR_ARG, "bad arguments");
  if (count == 0) return MJX_OK;
  if (c->peer.on) return peer_allreduce(c, buf, count, dtype, (hipStream_t)stream);
  if (!c->comm) {
    if (!c->reduce_cb) return fail(MJX_ERR_STATE, "no communicator attached (mjx_comm_init)");
    if (int rc = c->reduce_cb(c->reduce_user, buf, count, dtype, stream)) return fail(rc, "transport hook failed (%d)", rc);
    return MJX_OK;
  }
  RcclApi& r = rccl();
  if (int rc = r.AllReduce(buf, buf, (size_t)count, dtype ? RcclApi::kFloat64 : RcclApi::kFloat32, RcclApi::kSum, c->comm, stream))
    return fail(1000 + rc, "ncclAllReduce: %s", r.GetErrorString(rc));
  return MJX_OK;
}

int mjx_npg_update(mjx_ctx* c, int iters, float damping, double tol, double step_size, double const_alpha, float min_log_std,
                   float* grad_out, float* x_out, float* theta_out, double* results, void* stream) {
  if (int rc = check_bound(c, true)) return rc;
  if (!grad_out || !x_out || !theta_out || !results || iters < 0) return fail(MJX_ERR_ARG, "bad arguments");
  if (!c->old_is_new) return fail(MJX_ERR_STATE, "mjx_npg_update starts from theta_new == theta_old (mjx_bind_policy with old_is_new)");
  if (theta_out == c->theta_old) return fail(MJX_ERR_ARG, "theta_out must not alias theta_old");
  PeerSlots gs{};
  bool folded = false;
  if (int rc = vpg_and_rank_sums(c, grad_out, results + 4, true, stream, &gs, &folded)) return rc;
  const float* base = c->theta_old;               // == theta_new in value; theta_out may be the theta_new buffer itself
  // (theta_out == theta_new: the stepped parameters are written by the solve's LAST kernel, after every product has read theta)
  if (int rc = cg_solve_impl(c, grad_out, iters, damping, tol, x_out, results + 8, nullptr, nullptr, stream, folded ? &gs : nullptr, results + 4,
                             fin_step(base, theta_out, results + 9, step_size, const_alpha, min_log_std))) return rc;
  if (int rc = bind_policy_impl(c, theta_out, c->theta_old, c->tr_new, c->tr_old, 0, true)) return rc;
  return eval_and_rank_sum(c, results, stream);
}

int mjx_trpo_update(mjx_ctx* c, int iters, float damping, double tol, double step_size, double kl_dist, int n_trials, int first,
                    float min_log_std, float* grad_out, float* x_out, float* theta_out, double* results, void* stream) {
  if (int rc = check_bound(c, true)) return rc;
  if (!grad_out || !x_out || !theta_out || !results || iters < 0 || n_trials < 1 || n_trials > 24) return fail(MJX_ERR_ARG, "bad arguments");
  if (theta_out == c->theta_old) return fail(MJX_ERR_ARG, "theta_out must not alias theta_old");
  hipStream_t st = (hipStream_t)stream;
  if (first) {
    if (!c->old_is_new) return fail(MJX_ERR_STATE, "mjx_trpo_update starts from theta_new == theta_old (mjx_bind_policy with old_is_new)");
    PeerSlots gs{};
    bool folded = false;
    if (int rc = vpg_and_rank_sums(c, grad_out, results + 4, true, stream, &gs, &folded)) return rc;
    if (int rc = cg_solve_impl(c, grad_out, iters, damping, tol, x_out, results + 8, nullptr, nullptr, stream, folded ? &gs : nullptr, results + 4,
                               fin_only())) return rc;
  }
  for (int t = 0; t < n_trials; ++t) {
    hipLaunchKernelGGL(k_trpo_try, dim3((c->d + 255) / 256), dim3(256), 0, st, c->theta_old, x_out, results, step_size,
                       (first && t == 0) ? 1 : 0, min_log_std, theta_out, (int)c->d, c->oS);
    HIPCHK(hipGetLastError());
    if (first && t == 0) { if (int rc = bind_policy_impl(c, theta_out, c->theta_old, c->tr_new, c->tr_old, 0, true)) return rc; }
    if (int rc = eval_and_rank_sum(c, results, stream)) return rc;
    hipLaunchKernelGGL(k_trpo_check, dim3(1), dim3(64), 0, st, results, kl_dist, (double)c->N_global);
    HIPCHK(hipGetLastError());
  }
  return MJX_OK;
}

int mjx_dapg_update(mjx_ctx* c, int iters, float damping, double tol, double step_size, float min_log_std, int64_t rows_on,
                    int64_t N_on_global, const float* adv_on, float* grad_out, float* x_out, float* theta_out, double* results,
                    void* stream) {
  if (int rc = check_bound(c, true)) return rc;
  if (!grad_out || !x_out || !theta_out || !results || (!adv_on && rows_on > 0) || iters < 0) return fail(MJX_ERR_ARG, "bad arguments");   // (a rank without on-policy rows has no advantages)
  if (!c->old_is_new) return fail(MJX_ERR_STATE, "mjx_dapg_update starts from theta_new == theta_old (mjx_bind_policy with old_is_new)");
  if (theta_out == c->theta_old) return fail(MJX_ERR_ARG, "theta_out must not alias theta_old");
  if (rows_on < 0 || rows_on > c->N_local || N_on_global <= 0 || N_on_global > c->N_global) return fail(MJX_ERR_ARG, "bad on-policy row counts");
  hipStream_t st = (hipStream_t)stream;
  const bool ranks = has_ranks(c);
  // the vanilla gradient over [on-policy ; demonstrations] (mean over N_all), then x N_all / N_on (dapg.py:97-98)
  const float coef = (float)((double)c->N_global / (double)N_on_global);
  if (int rc = mjx_surr_vpg(c, grad_out, results + 4, stream)) return rc;
  if (ranks) if (int rc = mjx_comm_allreduce(c, grad_out, c->d, 0, stream)) return rc;
  hipLaunchKernelGGL(k_scale_f32, dim3((c->d + 255) / 256), dim3(256), 0, st, grad_out, coef, (int)c->d);
  HIPCHK(hipGetLastError());
  // Fisher metric, surrogate and KL: the on-policy prefix with its own advantages, means over the on-policy count (:92, :103)
  if (int rc = mjx_bind_rows(c, rows_on, N_on_global, adv_on)) return rc;
  if (int rc = eval_and_rank_sum(c, results + 4, stream)) return rc;                   // surr_before (theta_new == theta_old)
  if (int rc = cg_solve_impl(c, grad_out, iters, damping, tol, x_out, results + 8, nullptr, nullptr, stream, nullptr, nullptr,
                             fin_step(c->theta_old, theta_out, results + 9, step_size, std::nan(""), min_log_std))) return rc;
  if (int rc = bind_policy_impl(c, theta_out, c->theta_old, c->tr_new, c->tr_old, 0, true)) return rc;
  return eval_and_rank_sum(c, results, stream);
}

int mjx_apply_step(mjx_ctx* c, const float* theta, const float* x, float alpha, float min_log_std, float* theta_out, void* stream) {
  if (!c || !theta || !x || !theta_out) return fail(MJX_ERR_ARG, "bad arguments");
  hipLaunchKernelGGL(k_apply_step, dim3((c->d + 255) / 256), dim3(256), 0, (hipStream_t)stream, theta, x, alpha, min_log_std,
                     theta_out, (int)c->d, c->oS);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_apply_npg_step(mjx_ctx* c, const float* theta, const float* x, const double* gdotx, double step_size, float min_log_std,
                       float* theta_out, double* alpha_out, void* stream) {
  if (!c || !theta || !x || !gdotx || !theta_out) return fail(MJX_ERR_ARG, "bad arguments");
  hipLaunchKernelGGL(k_apply_npg_step, dim3((c->d + 255) / 256), dim3(256), 0, (hipStream_t)stream, theta, x, gdotx, step_size,
                     min_log_std, theta_out, alpha_out, (int)c->d, c->oS);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_discount_scan(const double* x, const int64_t* offsets, int64_t n_traj, double gamma, double* y, void* stream) {
  if (n_traj == 0) return MJX_OK;
  if (!x || !offsets || !y || n_traj < 0) return fail(MJX_ERR_ARG, "bad arguments");
  hipLaunchKernelGGL(k_traj_scan<0>, dim3((unsigned)n_traj), dim3(256), 0, (hipStream_t)stream, x, (const double*)nullptr,
                     offsets, (const uint8_t*)nullptr, gamma, gamma, y);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_time_index(const int64_t* offsets, int64_t n_traj, int32_t* tpos, void* stream) {
  if (n_traj == 0) return MJX_OK;
  if (!offsets || !tpos || n_traj < 0) return fail(MJX_ERR_ARG, "bad arguments");
  hipLaunchKernelGGL(k_time_index, dim3((unsigned)n_traj), dim3(256), 0, (hipStream_t)stream, offsets, tpos);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_gae(const double* rewards, const double* baseline, const int64_t* offsets, const uint8_t* terminated,
            int64_t n_traj, double gamma, double lam, double* adv, void* stream) {
  if (n_traj == 0) return MJX_OK;
  if (!rewards || !baseline || !offsets || !adv || n_traj < 0) return fail(MJX_ERR_ARG, "bad arguments");
  if (lam < 0.0 || lam > 1.0 || std::isnan(lam))
    hipLaunchKernelGGL(k_traj_scan<2>, dim3((unsigned)n_traj), dim3(256), 0, (hipStream_t)stream, rewards, baseline, offsets,
                       terminated, gamma, 0.0, adv);
  else
    hipLaunchKernelGGL(k_traj_scan<1>, dim3((unsigned)n_traj), dim3(256), 0, (hipStream_t)stream, rewards, baseline, offsets,
                       terminated, gamma, gamma * lam, adv);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_sum_stats(const double* x, int64_t N, double shift, double* stats_out, void* stream) {
  if (!x || !stats_out || N < 0) return fail(MJX_ERR_ARG, "bad arguments");
  static thread_local double* part = nullptr;
  const int G = 512;
  if (!part) HIPCHK(hipMalloc(&part, G * 2 * sizeof(double)));
  hipLaunchKernelGGL(k_sum_stats_partial, dim3(G), dim3(256), 0, (hipStream_t)stream, x, N, shift, part);
  hipLaunchKernelGGL(k_sum_stats_final, dim3(1), dim3(256), 0, (hipStream_t)stream, part, G, N, stats_out);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_whiten_cast(const double* adv, int64_t N, double mean, double std, double eps, float* out32, void* stream) {
  if (!adv || !out32 || N < 0) return fail(MJX_ERR_ARG, "bad arguments");
  if (N == 0) return MJX_OK;
  int grid = (int)((N + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_whiten_cast, dim3(grid), dim3(256), 0, (hipStream_t)stream, adv, N, mean, std + eps, out32);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_policy_forward(mjx_ctx* c, const float* obs, int64_t N, const float* theta, const float* tr, float* mean_out, void* stream) {
  if (!c || N < 0 || (N > 0 && (!obs || !theta || !mean_out))) return fail(MJX_ERR_ARG, "bad arguments");
  if (N == 0) return MJX_OK;
  HIPCHK(hipSetDevice(c->device));
  if (c->lw.cap < N) { if (int rc = c->lw.reserve(N)) return fail(rc, "layer-wise workspace allocation failed"); }
  c->lw.invalidate();                           // the hidden activations of the bound policy are overwritten (scratch)
  c->lw.forward(theta, tr ? tr : c->ident_tr, obs, N, c->lw.T, mean_out, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

int mjx_policy_minibatch_adam(mjx_ctx* c, int loss, const float* obs, const float* act, const float* adv, const int32_t* idx,
                              int64_t steps, int B, float* theta, const float* tr, const float* theta_old, const float* tr_old,
                              int old_tracks_new, float* adam_m, float* adam_v, int64_t step0, float lr, float clip,
                              double* loss_trace, void* stream) {
  if (!c || loss < 0 || loss > 2 || steps < 0 || B <= 0 || step0 < 0) return fail(MJX_ERR_ARG, "bad arguments");
  if (steps == 0) return MJX_OK;
  if (!obs || !act || !idx || !theta || !adam_m || !adam_v) return fail(MJX_ERR_ARG, "null buffer");
  if (loss == 2 && (!adv || !theta_old)) return fail(MJX_ERR_ARG, "PPO needs advantages and the old parameters");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipSetDevice(c->device));
  const float* trn0 = tr ? tr : c->ident_tr;
  const float* tro0 = tr_old ? tr_old : c->ident_tr;
  // small two-layer nets, minibatches of up to 64 rows: the whole chain of steps in ONE launch (policy_fit.h)
  const char* no_fit_env = getenv("MJX_NO_POLICY_FIT");      // read per call: the tests run both paths in one process
  const bool no_fit = no_fit_env && no_fit_env[0] == '1';
  if (!no_fit && c->hidden.size() == 2 && c->hidden[0] == c->hidden[1] && (c->hidden[0] == 64 || c->hidden[0] == 32) &&
      B % 4 == 0 && B >= 8 && B <= 64 && c->n <= c->hidden[0] && c->n <= 63 && c->m <= 16) {
    const bool old_net = (loss == 2) && !old_tracks_new;
    const int H = c->hidden[0];
    const size_t bytes = 4 * (H == 64 ? PolicyFitLayout<64>(c->n, c->m).lds_floats(B, old_net) : PolicyFitLayout<32>(c->n, c->m).lds_floats(B, old_net));
    if (bytes <= 160 * 1024) {
      PolicyFitArgs a{obs, act, adv, idx, steps, B, c->n, c->m, theta, theta_old, trn0, tro0, loss, old_tracks_new, adam_m, adam_v,
                      step0, lr, clip, loss_trace, (int)(bytes / 4)};
      void (*k)(PolicyFitArgs) = (H == 64) ? k_policy_fit<64> : k_policy_fit<32>;
      HIPCHK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      hipLaunchKernelGGL(k, dim3(1), dim3(PFIT_THREADS), bytes, st, a);
      HIPCHK(hipGetLastError());
      return MJX_OK;
    }
  }
  LayerwiseWS& w = c->lwmb;
  if (w.cap < B) { if (int rc = w.reserve(B)) return fail(rc, "minibatch workspace allocation failed"); }
  if (c->mb_cap < B) {
    hipFree(c->mb_x); hipFree(c->mb_a); hipFree(c->mb_adv);
    HIPCHK(hipMalloc(&c->mb_x, (size_t)B * c->n * 4)); HIPCHK(hipMalloc(&c->mb_a, (size_t)B * c->m * 4)); HIPCHK(hipMalloc(&c->mb_adv, (size_t)B * 4));
    c->mb_cap = B;
  }
  if (!c->mb_grad) HIPCHK(hipMalloc(&c->mb_grad, (size_t)c->d * 4));
  const float* trn = tr ? tr : c->ident_tr;
  const float* tro = tr_old ? tr_old : c->ident_tr;
  const int64_t cnt = (loss == 0) ? (int64_t)c->oS : c->d;       // MSE: log_std has no gradient -> untouched (like torch's grad None)
  const int ggrid = (B * (c->n > c->m ? c->n : c->m) + 255) / 256;
  for (int64_t s = 0; s < steps; ++s) {
    hipLaunchKernelGGL(k_gather_minibatch, dim3(ggrid < 1 ? 1 : (ggrid > 1024 ? 1024 : ggrid)), dim3(256), 0, st, obs, act,
                       (loss == 2) ? adv : (const float*)nullptr, idx + s * B, B, c->n, c->m, c->mb_x, c->mb_a, c->mb_adv);
    const bool old_fwd = (loss == 2) && !old_tracks_new;
    if (old_fwd) w.forward(theta_old, tro, c->mb_x, B, w.T, w.mu2, st);          // old policy on the minibatch (activations are scratch)
    w.forward(theta, trn, c->mb_x, B, w.H, w.mu, st);
    hipLaunchKernelGGL(k_minibatch_head, dim3(1), dim3(256), 0, st, loss, w.mu, (loss == 2) ? (old_fwd ? w.mu2 : w.mu) : (const float*)nullptr, c->mb_a,
                       c->mb_adv, B, c->m, theta + c->oS, (loss == 2) ? theta_old + c->oS : (const float*)nullptr, trn + 2 * c->n + c->m,
                       clip, w.d3, c->mb_grad + c->oS, loss_trace ? loss_trace + s : (double*)nullptr);
    if (int rc = w.backward(theta, B, c->mb_grad, st)) return fail(MJX_ERR_STATE, "minibatch backward failed (%d)", rc);
    const double t = (double)(step0 + s + 1);
    const float bc1 = (float)(1.0 - std::pow(0.9, t)), bc2s = (float)std::sqrt(1.0 - std::pow(0.999, t));
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, theta, c->mb_grad, adam_m, adam_v, cnt, lr, 0.f,
                       0.9f, 0.999f, 1e-8f, bc1, bc2s);
  }
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

}  // extern "C"

// ---- host side of rollout ingestion: csrc/host_ingest.h (plain C++; also built under ASan / TSan by tests/c/host_san.cpp)
namespace {
inline bool mjx_hi_set_device(int idx) { return hipSetDevice(idx) == hipSuccess; }
inline int mjx_hi_h2d_async(void* dst, const void* src, size_t bytes, void* stream, const char** what) {
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
  if (e != hipSuccess) { *what = hipGetErrorString(e); return (int)e; }
  return 0;
}
}  // namespace
extern "C" int mjx_cast_f64_f32(const double* x, int64_t count, float* out32, void* stream);
namespace {
// reads a page-locked HOST block (device-mapped) and writes it to the device twice: as it is, and as fp32 (host_ingest.h: small raw blocks)
__global__ void k_pull_f64(const double* __restrict__ host, int64_t n, double* __restrict__ raw, float* __restrict__ o32) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = host[i];
    raw[i] = v; o32[i] = (float)v;
  }
}
inline int mjx_hi_pull_f64(const double* host, int64_t n, double* raw, float* o32, void* stream) {
  if (n <= 0) return MJX_OK;
  int grid = (int)((n + 255) / 256); if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(k_pull_f64, dim3(grid), dim3(256), 0, (hipStream_t)stream, host, n, raw, o32);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}
}  // namespace
#define MJX_HI_PULL_F64(x, n, r, o, st) mjx_hi_pull_f64(x, n, r, o, st)
#define MJX_HI_SET_DEVICE(i) mjx_hi_set_device(i)
#define MJX_HI_H2D_ASYNC(d, s_, b, st, w) mjx_hi_h2d_async(d, s_, b, st, w)
#define MJX_HI_CAST_F64_F32(x, n, o, st) mjx_cast_f64_f32(x, n, o, st)
#include "host_ingest.h"

extern "C" {

int mjx_cast_f64_f32(const double* x, int64_t count, float* out32, void* stream) {
  if (!x || !out32 || count < 0) return fail(MJX_ERR_ARG, "bad arguments");
  if (count == 0) return MJX_OK;
  int grid = (int)((count + 255) / 256); if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_cast_f64_f32, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, count, out32);
  HIPCHK(hipGetLastError());
  return MJX_OK;
}

}  // extern "C"
