// host_ingest.h -- the HOST side of rollout ingestion: persistent gather pools, the fp64 -> fp32 converting gather,
// per-path sums, and the asynchronous staging jobs (one native thread per block: gather group by group, queue each group's
// host-to-device copy behind it).  Plain C++17 + pthreads, no HIP types: mjx.hip includes it with the three device hooks below
// bound to the HIP runtime; tests/c/host_san.cpp includes it with the hooks bound to memcpy and builds it under
// -fsanitize=address,undefined and -fsanitize=thread (tests/test_host_sanitizers.py: the CPU lane runs both).
//
// The includer provides, before the #include:
//   int fail(int code, const char* fmt, ...)                 -> records the thread's error message, returns code
//   MJX_DEVICE_ENTRY()                                        -> fork guard + call counter (may `return` an error)
//   bool MJX_HI_SET_DEVICE(int index)
//   int  MJX_HI_H2D_ASYNC(void* dst_dev, const void* src_host, size_t bytes, void* stream, const char** what)   (0 = ok)
//   int  MJX_HI_CAST_F64_F32(const double* x_dev, int64_t count, float* out_dev, void* stream)                  (MJX_OK = ok)
//   int  MJX_HI_PULL_F64(const double* x_pinned_host, int64_t count, double* raw_dev, float* f32_dev, void* stream)   (MJX_OK = ok)
//        -- a kernel that READS the page-locked host block over the bus and writes the fp64 block and its fp32 image
#pragma once
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {
// Persistent host workers for mjx_host_gather: a training iteration issues ~12 gathers of 8-50 MB, and creating 7-15
// std::threads for each (20-50 us apiece) cost as much as the copies they performed.  Workers sleep on a condition variable;
// one job at a time (a second caller -- the prefetch thread and the trainer may overlap -- falls back to its own threads).
// The pool is leaked on purpose (no joins in static destructors; a forked child simply finds no workers and copies inline).
struct HostPool {
  static constexpr int MAXW = 31;
  std::mutex m, run_m;
  std::condition_variable cv, done_cv;
  const std::function<void(int)>* job = nullptr;
  int active = 0, pending = 0;
  uint64_t gen = 0;
  int nworkers = 0;
  pid_t owner = 0;
  void ensure(int n) {
    if (n > MAXW) n = MAXW;
    while (nworkers < n) {
      const int t = ++nworkers;
      std::thread([this, t] {
        uint64_t seen = 0;
        for (;;) {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [&] { return gen != seen; });
          seen = gen;
          const bool mine = t < active;
          const std::function<void(int)>* j = job;
          lk.unlock();
          if (mine) {
            (*j)(t);
            lk.lock();
            if (--pending == 0) done_cv.notify_one();
          }
        }
      }).detach();
    }
  }
  // run fn(0 .. nt-1), fn(0) on the calling thread; false if the pool is busy or unusable (the caller uses its own threads)
  bool run(int nt, const std::function<void(int)>& fn) {
    std::unique_lock<std::mutex> rl(run_m, std::try_to_lock);
    if (!rl.owns_lock()) return false;
    // (under the run lock: two first-time callers raced on `owner` -- found by the ThreadSanitizer build, tests/c/host_san.cpp)
    if (owner != getpid()) { if (owner != 0) return false; owner = getpid(); }
    {
      std::lock_guard<std::mutex> lk(m);
      ensure(nt - 1);
      job = &fn; active = nt; pending = nt - 1; ++gen;
    }
    cv.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(m);
    done_cv.wait(lk, [&] { return pending == 0; });
    job = nullptr; active = 0;
    return true;
  }
};
// three pools: the observation and the action block of a batch are staged by two helper threads at the same time (r04) while the
// training thread stages its 8 MB of advantages; a fourth concurrent caller falls back to its own threads
struct HostPools {
  HostPool a, b, c;                                // (c: the advantage / reward block the training thread stages meanwhile)
  bool run(int nt, const std::function<void(int)>& fn) { return a.run(nt, fn) || b.run(nt, fn) || c.run(nt, fn); }
};
HostPools& host_pool() { static HostPools* p = new HostPools(); return *p; }
}  // namespace

extern "C" {

int mjx_host_gather(void* dst, const void* const* src, const int64_t* offsets, int64_t first, int64_t count,
                    int64_t row_bytes, int n_threads) {
  if (!dst || !src || !offsets || first < 0 || count < 0 || row_bytes <= 0) return fail(MJX_ERR_ARG, "bad arguments");
  if (count == 0) return MJX_OK;
  const int64_t total = (offsets[first + count] - offsets[first]) * row_bytes;
  int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
  if (total < (int64_t)(1 << 20)) nt = 1;
  else if (total < (int64_t)(16 << 20) && nt > 8) nt = 8;      // a few MB: waking 31 workers costs more than they save
  // every thread takes a contiguous byte range of the destination (blocks are split where the range ends)
  auto work = [&](int t) {
    const int64_t lo = total * t / nt, hi = total * (t + 1) / nt;
    const int64_t base = offsets[first] * row_bytes;
    for (int64_t i = first; i < first + count; ++i) {
      const int64_t b0 = offsets[i] * row_bytes - base, b1 = offsets[i + 1] * row_bytes - base;
      const int64_t c0 = b0 > lo ? b0 : lo, c1 = b1 < hi ? b1 : hi;
      if (c1 > c0) memcpy((char*)dst + base + c0, (const char*)src[i] + (c0 - b0), (size_t)(c1 - c0));
      if (b0 >= hi) break;
    }
  };
  if (nt == 1) { work(0); return MJX_OK; }
  if (nt > HostPool::MAXW + 1) nt = HostPool::MAXW + 1;
  {
    const std::function<void(int)> fn = work;
    if (host_pool().run(nt, fn)) return MJX_OK;
  }
  std::vector<std::thread> th;
  th.reserve(nt - 1);
  for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  return MJX_OK;
}

// fp64 -> fp32 while gathering (round to nearest even, the bits of NumPy's astype and of mjx_cast_f64_f32): the blocks leave the
// host at half their size.  Every thread converts a contiguous element range of the destination.
namespace {
void cvt_range_generic(float* __restrict__ d, const double* __restrict__ s, int64_t n) {
  for (int64_t i = 0; i < n; ++i) d[i] = (float)s[i];
}
__attribute__((target("avx2"))) void cvt_range_avx2(float* __restrict__ d, const double* __restrict__ s, int64_t n) {
  for (int64_t i = 0; i < n; ++i) d[i] = (float)s[i];        // (vcvtpd2ps on 4-wide vectors: the loop vectorises under this target)
}
}  // namespace

int mjx_host_gather_f64_f32(float* dst, const double* const* src, const int64_t* offsets, int64_t first, int64_t count,
                            int64_t row_elems, int n_threads) {
  if (!dst || !src || !offsets || first < 0 || count < 0 || row_elems <= 0) return fail(MJX_ERR_ARG, "bad arguments");
  if (count == 0) return MJX_OK;
  const int64_t total = (offsets[first + count] - offsets[first]) * row_elems;      // elements
  int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
  if (total < (int64_t)(1 << 17)) nt = 1;
  else if (total < (int64_t)(2 << 20) && nt > 8) nt = 8;
  const char* no_avx2 = getenv("MJX_NO_AVX2");             // (tests: the portable loop on an AVX2 host)
  const bool avx2 = __builtin_cpu_supports("avx2") && !(no_avx2 && no_avx2[0] == '1');
  auto work = [&](int t) {
    const int64_t lo = total * t / nt, hi = total * (t + 1) / nt;
    const int64_t base = offsets[first] * row_elems;
    for (int64_t i = first; i < first + count; ++i) {
      const int64_t b0 = offsets[i] * row_elems - base, b1 = offsets[i + 1] * row_elems - base;
      const int64_t c0 = b0 > lo ? b0 : lo, c1 = b1 < hi ? b1 : hi;
      if (c1 > c0) {
        if (avx2) cvt_range_avx2(dst + base + c0, src[i] + (c0 - b0), c1 - c0);
        else cvt_range_generic(dst + base + c0, src[i] + (c0 - b0), c1 - c0);
      }
      if (b0 >= hi) break;
    }
  };
  if (nt == 1) { work(0); return MJX_OK; }
  if (nt > HostPool::MAXW + 1) nt = HostPool::MAXW + 1;
  {
    const std::function<void(int)> fn = work;
    if (host_pool().run(nt, fn)) return MJX_OK;
  }
  std::vector<std::thread> th;
  th.reserve(nt - 1);
  for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  return MJX_OK;
}

// per-trajectory sums of a 1-D fp64 quantity (path returns, batch_reinforce.py:187): out[i] = src[i][0] + src[i][1] + ... in that
// order -- the order of Python's sum() over the array, which is what the reference calls -- the trajectories spread over threads
int mjx_host_segment_sums(const double* const* src, const int64_t* lens, int64_t count, double* out, int n_threads) {
  if (!src || !lens || !out || count < 0) return fail(MJX_ERR_ARG, "bad arguments");
  if (count == 0) return MJX_OK;
  int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
  if (count < 64) nt = 1;
  auto work = [&](int t) {
    const int64_t lo = count * t / nt, hi = count * (t + 1) / nt;
    for (int64_t i = lo; i < hi; ++i) {
      const double* p = src[i];
      double a = 0.0;                                  // (sum() starts from int 0: 0 + x == x exactly)
      for (int64_t k = 0; k < lens[i]; ++k) a += p[k];
      out[i] = a;
    }
  };
  if (nt == 1) { work(0); return MJX_OK; }
  if (nt > HostPool::MAXW + 1) nt = HostPool::MAXW + 1;
  {
    const std::function<void(int)> fn = work;
    if (host_pool().run(nt, fn)) return MJX_OK;
  }
  std::vector<std::thread> th;
  th.reserve(nt - 1);
  for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  return MJX_OK;
}

// ---- NumPy's legacy permutation stream, natively (r05).  MLPBaseline.fit draws np.random.permutation(N) once per epoch
// (utils/optimize_model.py:22) and the draws must come from NumPy's GLOBAL MT19937 stream to keep the reference's minibatch order;
// at 1M timesteps the two draws were 15 ms of a 29 ms post-sampling iteration (the rest of the fit runs in the background).  This is
// RandomState.permutation(n) bit for bit -- arange(n), then for i = n - 1 .. 1: j = random_interval(i) (the smallest bit mask >= i,
// rejection on 32-bit outputs), swap(i, j) -- on the caller's copy of the generator state (np.random.get_state(): 624 key words +
// position), which is advanced in place and handed back with np.random.set_state.  int32 indices in one pass (NumPy shuffles int64
// and the caller converted).  n < 2^31.
namespace {
struct Mt19937 {
  uint32_t* key; int pos;
  void refill() {
    constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;
    int i = 0;
    for (; i < 624 - 397; ++i) { const uint32_t y = (key[i] & UP) | (key[i + 1] & LO); key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u); }
    for (; i < 623; ++i) { const uint32_t y = (key[i] & UP) | (key[i + 1] & LO); key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u); }
    const uint32_t y = (key[623] & UP) | (key[0] & LO);
    key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
    pos = 0;
  }
  inline uint32_t next() {
    if (pos == 624) refill();
    uint32_t y = key[pos++];
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
    return y;
  }
};
}  // namespace

extern "C" int mjx_host_mt19937_permutation(uint32_t* key624, int32_t* pos_io, int64_t n, int32_t* out) {
  if (!key624 || !pos_io || !out || n < 0 || n >= ((int64_t)1 << 31) || *pos_io < 0 || *pos_io > 624) return fail(MJX_ERR_ARG, "bad arguments");
  for (int64_t i = 0; i < n; ++i) out[i] = (int32_t)i;
  Mt19937 g{key624, *pos_io};
  // the swap partners j_i depend on the generator only, not on the array: draw them a block ahead and prefetch their cache lines
  // (the 4 MB index array of a 1M-row batch lives in L3: a swap per ~5 ns without, ~2 ns with)
  constexpr int B = 32;
  uint32_t js[B];
  int64_t i = n - 1;
  while (i >= 1) {
    const int cnt = (int)(i >= B ? B : i);
    {
      // (masked rejection without a data-dependent branch -- see mjx_host_mt19937_permutations below: the rejected third of the
      //  candidates cost ~5 ns per index in mispredictions)
      uint32_t ii = (uint32_t)i;
      uint32_t mask = ii;
      mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
      int b = 0;
      while (b < cnt) {
        const uint32_t v = g.next() & mask;
        const uint32_t acc = (uint32_t)(v <= ii);
        js[b] = v;
        b += (int)acc; ii -= acc;
        mask = (ii <= (mask >> 1)) ? (mask >> 1) : mask;
      }
      for (b = 0; b < cnt; ++b) __builtin_prefetch(out + js[b], 1, 1);
    }
    for (int b = 0; b < cnt; ++b) {
      const int64_t ii = i - b;
      const uint32_t j = js[b];
      const int32_t t = out[ii]; out[ii] = out[j]; out[j] = t;
    }
    i -= cnt;
  }
  *pos_io = g.pos;
  return MJX_OK;
}

// `epochs` consecutive np.random.permutation(n) of the same stream, out[e * n ..]: what MLPBaseline.fit draws per fit (one per epoch,
// optimize_model.py:22).  The draws are sequential in the GENERATOR only -- the swap partners j_i do not depend on the array -- so the
// work is cut where it can be: the calling thread runs the generator through all epochs (masked rejection, ~1.5 words per index) and
// publishes the partners in blocks, a second thread applies the swaps behind it (random accesses into the 4 MB index array, with the
// partners known far ahead to prefetch).  Same draws, same swaps, same order: the bits of `epochs` calls of the function above, in
// ~the time of the slower of the two halves instead of their sum (r06: 2 x 1M rows 11 -> ~6 ms on the GPU boxes' hosts -- the tail of
// these draws was what MLPBaseline.fit still cost on train_step's critical path).
extern "C" int mjx_host_mt19937_permutations(uint32_t* key624, int32_t* pos_io, int64_t n, int epochs, int32_t* out) {
  if (!key624 || !pos_io || !out || n < 0 || epochs < 0 || n >= ((int64_t)1 << 31) || *pos_io < 0 || *pos_io > 624) return fail(MJX_ERR_ARG, "bad arguments");
  if (n < 65536 || epochs == 0) {                     // small: the pipeline's hand-over costs more than it saves
    for (int e = 0; e < epochs; ++e)
      if (int rc = mjx_host_mt19937_permutation(key624, pos_io, n, out + (int64_t)e * n)) return rc;
    return MJX_OK;
  }
  const int64_t per = n - 1, total = per * epochs;    // swaps per epoch (i = n - 1 .. 1), over all epochs
  std::unique_ptr<uint32_t[]> js_own(new uint32_t[(size_t)total]);      // (not value-initialised: 8 MB per 2 x 1M rows, every word written before it is read)
  struct { uint32_t* p; uint32_t* data() const { return p; } } js{js_own.get()};
  std::atomic<int64_t> ready{0};                      // partners published so far
  std::thread applier([&] {
    constexpr int64_t AHEAD = 24;
    int64_t seen = 0;
    for (int e = 0; e < epochs; ++e) {
      int32_t* o = out + (int64_t)e * n;
      for (int64_t i = 0; i < n; ++i) o[i] = (int32_t)i;
      const uint32_t* je = js.data() + (int64_t)e * per;
      const int64_t base = (int64_t)e * per;
      for (int64_t k = 0; k < per; ++k) {
        if (base + k >= seen) {
          while ((seen = ready.load(std::memory_order_acquire)) <= base + k) std::this_thread::yield();
        }
        if (k + AHEAD < per && base + k + AHEAD < seen) __builtin_prefetch(o + je[k + AHEAD], 1, 1);
        const int64_t ii = n - 1 - k;
        const uint32_t j = je[k];
        const int32_t t = o[ii]; o[ii] = o[j]; o[j] = t;
      }
    }
  });
  Mt19937 g{key624, *pos_io};
  constexpr int64_t BLK = 8192;
  int64_t w = 0, published = 0;
  for (int e = 0; e < epochs; ++e) {
    // masked rejection (RandomState's random_interval: draw & mask until <= i) WITHOUT a data-dependent branch: a third of the
    // candidates is rejected, unpredictably -- as a branch that is ~5 ns per index in mispredictions, most of this function's time.
    // Every candidate is stored at the write position, the position (and the bound i) advance only when it was accepted; the mask
    // shrinks exactly when i reaches the next power of two minus one, as recomputing it from i would give.
    uint32_t ii = (uint32_t)(n - 1);
    uint32_t mask = ii;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    const int64_t w_end = w + per;
    uint32_t* jp = js.data();
    while (w < w_end) {
      const int64_t stop = (w_end - w < BLK) ? w_end : w + BLK;     // (publish a block at a time)
      while (w < stop) {
        const uint32_t v = g.next() & mask;
        const uint32_t acc = (uint32_t)(v <= ii);
        jp[w] = v;
        w += acc; ii -= acc;
        mask = (ii <= (mask >> 1)) ? (mask >> 1) : mask;
      }
      published = w;
      ready.store(published, std::memory_order_release);
    }
  }
  ready.store(w, std::memory_order_release);
  applier.join();
  *pos_io = g.pos;
  return MJX_OK;
}

// count draws of np.random.choice(n, size=...) / np.random.randint(0, n, size=...) of the same legacy stream (BC and PPO draw one
// minibatch of row indices per Adam step, behavior_cloning.py:113, ppo_clip.py:77: 156 k Python-level calls per 1M-timestep PPO
// iteration): value = next_uint32 & mask, rejected while > n - 1 -- RandomState's masked rejection for ranges below 2^32; calls
// of any sizes concatenate to the same stream, so steps x minibatch indices are one native loop.
extern "C" int mjx_host_mt19937_randint(uint32_t* key624, int32_t* pos_io, int64_t n, int64_t count, int32_t* out) {
  if (!key624 || !pos_io || !out || n < 1 || n >= ((int64_t)1 << 31) || count < 0 || *pos_io < 0 || *pos_io > 624) return fail(MJX_ERR_ARG, "bad arguments");
  const uint32_t rng = (uint32_t)(n - 1);
  if (rng == 0) { for (int64_t i = 0; i < count; ++i) out[i] = 0; return MJX_OK; }          // (NumPy draws nothing for a one-value range)
  uint32_t mask = rng;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  Mt19937 g{key624, *pos_io};
  for (int64_t i = 0; i < count; ++i) {
    uint32_t v;
    while ((v = (g.next() & mask)) > rng) {}
    out[i] = (int32_t)v;
  }
  *pos_io = g.pos;
  return MJX_OK;
}

// ---- asynchronous staging of one block of a rollout batch (r04): gather (+ fp64 -> fp32 conversion) group by group on the host
// pools and the group's host-to-device copy queued right behind it -- all on a native thread, so the caller (a Python training
// loop) gets control back at once and no interpreter lock is involved while 184 MB of rollouts move.  mjx_stage_wait joins.
namespace {
struct StageJob {
  std::thread th;
  int rc = MJX_OK;
  std::string err;
  std::vector<const void*> src;
  std::vector<int64_t> offs;
};
}  // namespace

int mjx_stage_async(void** job_out, const void* const* src, const int64_t* lens, int64_t count, int64_t row_elems, int src_itemsize,
                    int hostcast, void* pinned, void* device_raw, float* device_f32, int64_t group_rows, int n_threads,
                    int device_index, void* stream) {
  if (!job_out || !src || !lens || count < 0 || row_elems <= 0 || (src_itemsize != 4 && src_itemsize != 8) || !pinned || !device_raw ||
      group_rows <= 0 || (hostcast && src_itemsize != 8))
    return fail(MJX_ERR_ARG, "bad arguments");
  MJX_DEVICE_ENTRY();
  StageJob* job = new StageJob();
  job->src.assign(src, src + count);                       // (the caller's pointer / length arrays need not outlive the call)
  job->offs.resize(count + 1);
  job->offs[0] = 0;
  for (int64_t i = 0; i < count; ++i) {
    if (lens[i] < 0) { delete job; return fail(MJX_ERR_ARG, "negative length"); }
    job->offs[i + 1] = job->offs[i] + lens[i];
  }
  const int64_t dst_item = hostcast ? 4 : src_itemsize;
  const bool trace = [] { const char* e = getenv("MJX_STAGE_TRACE"); return e && e[0] == '1'; }();     // (diagnostic: where a staging job's time goes)
  job->th = std::thread([=] {
    const int64_t* offs = job->offs.data();
    const void* const* sp = job->src.data();
    const auto t_start = std::chrono::steady_clock::now();
    double ms_gather = 0, ms_h2d = 0, ms_cast = 0, ms_dev = 0;
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    if (!MJX_HI_SET_DEVICE(device_index)) { job->rc = MJX_ERR_STATE; job->err = "hipSetDevice failed in the staging thread"; return; }
    ms_dev = since(t_start);
    struct Report { bool on; const double *g, *h, *c, *d; std::chrono::steady_clock::time_point t0; int64_t rows, elems; int item;
                    ~Report() { if (on) fprintf(stderr, "[mjx stage job] %lld rows x %lld x %d B: set-device %.2f, gather %.2f, h2d enqueue %.2f, cast enqueue %.2f, total %.2f ms\n",
                                                (long long)rows, (long long)elems, item, *d, *g, *h, *c,
                                                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); } }
        report{trace, &ms_gather, &ms_h2d, &ms_cast, &ms_dev, t_start, offs[count], row_elems, src_itemsize};
    int64_t first = 0;
    while (first < count) {
      int64_t last = first;
      while (last < count && offs[last] - offs[first] < group_rows) ++last;     // whole trajectories, at least group_rows rows
      if (last == first) last = first + 1;
      int rc;
      auto t0 = std::chrono::steady_clock::now();
      if (hostcast) rc = mjx_host_gather_f64_f32((float*)pinned, (const double* const*)sp, offs, first, last - first, row_elems, n_threads);
      else rc = mjx_host_gather(pinned, sp, offs, first, last - first, row_elems * src_itemsize, n_threads);
      ms_gather += since(t0);
      if (rc != MJX_OK) { job->rc = rc; job->err = mjx_last_error(); return; }
      const int64_t lo = offs[first] * row_elems, hi = offs[last] * row_elems;   // elements
      // r06: a SMALL raw fp64 block (the 8 MB of advantages / rewards per 1M timesteps) is pulled by a kernel instead of a copy-engine
      // transfer: with the observation and action blocks' 90 MB in flight on two other streams, hipMemcpyAsync for a third stream
      // BLOCKED its caller for 6-7 ms in some calls (MJX_STAGE_TRACE=1: "h2d enqueue 7.21 ms" -- the runtime waiting for a copy engine),
      // which is what made calls 2-4 of a process take 15 ms instead of 8.  Page-locked host memory is mapped into the device's address
      // space; 8 MB over the bus by loads take 0.2 ms.  (MJX_STAGE_PULL=0: the copy engine for everything.)
      static const bool pull_on = [] { const char* e = getenv("MJX_STAGE_PULL"); return !(e && e[0] == '0'); }();
      if (hi > lo && pull_on && device_f32 && !hostcast && src_itemsize == 8 && (hi - lo) * 8 <= (int64_t)(32 << 20)) {
        t0 = std::chrono::steady_clock::now();
        rc = MJX_HI_PULL_F64((const double*)pinned + lo, hi - lo, (double*)device_raw + lo, device_f32 + lo, stream);
        ms_cast += since(t0);
        if (rc != MJX_OK) { job->rc = rc; job->err = mjx_last_error(); return; }
      } else if (hi > lo) {
        const char* what = nullptr;
        t0 = std::chrono::steady_clock::now();
        const int e = MJX_HI_H2D_ASYNC((char*)device_raw + lo * dst_item, (const char*)pinned + lo * dst_item, (size_t)((hi - lo) * dst_item), stream, &what);
        ms_h2d += since(t0);
        if (e != 0) { job->rc = e; job->err = std::string("hipMemcpyAsync: ") + (what ? what : "?"); return; }
        if (device_f32 && !hostcast && src_itemsize == 8) {                     // raw fp64 block + its fp32 image (cast on the device)
          t0 = std::chrono::steady_clock::now();
          rc = MJX_HI_CAST_F64_F32((const double*)device_raw + lo, hi - lo, device_f32 + lo, stream);
          ms_cast += since(t0);
          if (rc != MJX_OK) { job->rc = rc; job->err = mjx_last_error(); return; }
        }
      }
      first = last;
    }
  });
  *job_out = job;
  return MJX_OK;
}

int mjx_stage_wait(void* job_) {
  if (!job_) return fail(MJX_ERR_ARG, "null job");
  StageJob* job = (StageJob*)job_;
  if (job->th.joinable()) job->th.join();
  const int rc = job->rc;
  const std::string err = job->err;
  delete job;
  return rc == MJX_OK ? MJX_OK : fail(rc, "staging job failed: %s", err.c_str());
}

}  // extern "C"
