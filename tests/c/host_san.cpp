// host_san.cpp -- the host side of rollout ingestion (mjrl_amd/csrc/host_ingest.h: gather pools, converting gather, per-path sums,
// asynchronous staging jobs) built as plain C++ under -fsanitize=address,undefined or -fsanitize=thread, with the three device
// hooks bound to memcpy, and hammered from several caller threads at once -- what train_step does to it (two staging jobs + the
// training thread's own gathers in flight together).  tests/test_host_sanitizers.py builds and runs both variants.
#include "../../include/mjx.h"

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
#include <string>

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
}  // namespace
extern "C" const char* mjx_last_error(void) { return g_err.c_str(); }
#define MJX_DEVICE_ENTRY() do {} while (0)
#define MJX_HI_SET_DEVICE(i) true
static int h2d(void* d, const void* s, size_t b, void*, const char**) { memcpy(d, s, b); return 0; }
#define MJX_HI_H2D_ASYNC(d, s_, b, st, w) h2d(d, s_, b, st, w)
static int cast(const double* x, int64_t n, float* o, void*) { for (int64_t i = 0; i < n; ++i) o[i] = (float)x[i]; return MJX_OK; }
#define MJX_HI_CAST_F64_F32(x, n, o, st) cast(x, n, o, st)
static int pull(const double* x, int64_t n, double* r, float* o, void*) { for (int64_t i = 0; i < n; ++i) { r[i] = x[i]; o[i] = (float)x[i]; } return MJX_OK; }
#define MJX_HI_PULL_F64(x, n, r, o, st) pull(x, n, r, o, st)
#include "../../mjrl_amd/csrc/host_ingest.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, mjx_last_error()); exit(1); } } while (0)

struct Batch {
  int64_t count, row;
  std::vector<std::vector<double>> paths;
  std::vector<const void*> ptrs;
  std::vector<int64_t> lens, offs;
  int64_t rows() const { return offs.back(); }
};

static Batch make_batch(unsigned seed, int64_t count, int64_t row, int64_t max_len) {
  std::mt19937_64 g(seed);
  Batch b; b.count = count; b.row = row;
  b.offs.push_back(0);
  for (int64_t i = 0; i < count; ++i) {
    const int64_t T = 1 + (int64_t)(g() % (uint64_t)max_len);
    b.paths.emplace_back((size_t)(T * row));
    for (auto& x : b.paths.back()) x = (double)((int64_t)(g() % 2000001) - 1000000) / 1024.0;
    b.lens.push_back(T); b.offs.push_back(b.offs.back() + T);
  }
  for (auto& p : b.paths) b.ptrs.push_back(p.data());
  return b;
}

static void one_caller(unsigned seed, int rounds, std::atomic<int>* bad) {
  for (int r = 0; r < rounds; ++r) {
    Batch b = make_batch(seed + 31 * r, 40 + (seed % 7) * 30, 1 + (seed + r) % 9, 3000);
    const int64_t n = b.rows() * b.row;
    // ---- plain gather (bytes), every thread count that takes a different route: inline, pool, > pool size
    for (int nt : {1, 4, 16, 48}) {
      std::vector<double> dst((size_t)n, -1.0);
      CHECK(mjx_host_gather(dst.data(), b.ptrs.data(), b.offs.data(), 0, b.count, b.row * 8, nt) == MJX_OK);
      int64_t k = 0;
      for (auto& p : b.paths) for (double x : p) if (dst[(size_t)k++] != x) ++*bad;
    }
    // ---- converting gather (sub-range of the paths)
    {
      const int64_t first = b.count / 5, cnt = b.count - first - 1;
      std::vector<float> dst((size_t)n, -1.f);
      CHECK(mjx_host_gather_f64_f32(dst.data(), (const double* const*)b.ptrs.data(), b.offs.data(), first, cnt, b.row, 16) == MJX_OK);
      for (int64_t i = first; i < first + cnt; ++i)
        for (int64_t e = 0; e < b.lens[(size_t)i] * b.row; ++e)
          if (dst[(size_t)(b.offs[(size_t)i] * b.row + e)] != (float)b.paths[(size_t)i][(size_t)e]) ++*bad;
      if (first > 0 && dst[0] != -1.f) ++*bad;                              // nothing outside the range is touched
    }
    // ---- per-path sums, left to right
    if (b.row == 1) {
      std::vector<double> out((size_t)b.count, 0.0);
      CHECK(mjx_host_segment_sums((const double* const*)b.ptrs.data(), b.lens.data(), b.count, out.data(), 16) == MJX_OK);
      for (int64_t i = 0; i < b.count; ++i) {
        double a = 0.0;
        for (double x : b.paths[(size_t)i]) a += x;
        if (out[(size_t)i] != a) ++*bad;
      }
    }
    // ---- two staging jobs in flight together (raw block + its fp32 image; host-cast block), then joined
    {
      std::vector<double> pin1((size_t)n), raw((size_t)n, -1.0);
      std::vector<float> f32((size_t)n, -1.f), pin2((size_t)n), dev2((size_t)n, -1.f);
      void *j1 = nullptr, *j2 = nullptr;
      CHECK(mjx_stage_async(&j1, b.ptrs.data(), b.lens.data(), b.count, b.row, 8, 0, pin1.data(), raw.data(), f32.data(), 5000, 8, 0, nullptr) == MJX_OK);
      CHECK(mjx_stage_async(&j2, b.ptrs.data(), b.lens.data(), b.count, b.row, 8, 1, pin2.data(), dev2.data(), nullptr, 7000, 8, 0, nullptr) == MJX_OK);
      CHECK(mjx_stage_wait(j1) == MJX_OK);
      CHECK(mjx_stage_wait(j2) == MJX_OK);
      int64_t k = 0;
      for (auto& p : b.paths) for (double x : p) { if (raw[(size_t)k] != x || f32[(size_t)k] != (float)x || dev2[(size_t)k] != (float)x) ++*bad; ++k; }
    }
  }
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 3;
  std::atomic<int> bad{0};
  // bad arguments are refused, not dereferenced
  void* j = nullptr;
  CHECK(mjx_stage_async(&j, nullptr, nullptr, 1, 1, 8, 0, nullptr, nullptr, nullptr, 1, 1, 0, nullptr) == MJX_ERR_ARG);
  CHECK(mjx_host_gather(nullptr, nullptr, nullptr, 0, 1, 8, 1) == MJX_ERR_ARG);
  CHECK(mjx_stage_wait(nullptr) == MJX_ERR_ARG);
  std::vector<std::thread> callers;
  for (unsigned t = 0; t < 4; ++t) callers.emplace_back(one_caller, 100 + 17 * t, rounds, &bad);
  for (auto& t : callers) t.join();
  // r06: the two-thread permutation pipeline (generator on the caller, swaps on a second thread) against the one-thread function:
  // the same permutations and the same generator state -- and clean under ThreadSanitizer
  {
    const int64_t n = 70001; const int epochs = 3;
    std::vector<uint32_t> k1(624), k2(624);
    for (int i = 0; i < 624; ++i) k1[i] = k2[i] = 0x9e3779b9u * (uint32_t)(i + 1) + 12345u;
    int32_t p1 = 624, p2 = 624;
    std::vector<int32_t> a((size_t)(n * epochs)), c((size_t)(n * epochs));
    for (int e = 0; e < epochs; ++e) CHECK(mjx_host_mt19937_permutation(k1.data(), &p1, n, a.data() + (int64_t)e * n) == MJX_OK);
    CHECK(mjx_host_mt19937_permutations(k2.data(), &p2, n, epochs, c.data()) == MJX_OK);
    CHECK(a == c && k1 == k2 && p1 == p2);
    CHECK(mjx_host_mt19937_permutations(nullptr, &p2, n, epochs, c.data()) == MJX_ERR_ARG);
  }
  if (bad.load()) { fprintf(stderr, "FAILED: %d mismatches\n", bad.load()); return 1; }
  printf("host_san ok: 4 concurrent callers x %d rounds\n", rounds);
  return 0;
}
