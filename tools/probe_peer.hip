// Can two PROCESSES exchange device data with stream-ordered flags and no host synchronisation?  (feasibility probe for a peer
// all-reduce inside libmjx: RCCL refuses two ranks on one device, and an in-kernel spin would deadlock a shared test GPU against the
// 512-register Fisher kernel.)  Parent = rank 0, child (same binary, argv[1] = "child") = rank 1, both on device 0:
//   each allocates an uncached device buffer [data 1024 floats | flag], exports it (hipIpcGetMemHandle) through a file,
//   opens the peer's, then per round r = 1..R:  kernel writes r * (rank + 1) into its OWN data; hipStreamWriteValue32(own flag, r);
//   hipStreamWaitValue32(PEER's flag (IPC-mapped), r, GTE);  kernel sums own + peer data into `out` (peer data read through the mapping).
// Checks the sums, prints the time per round.  Variants of the flag memory are tried in turn: the uncached buffer itself,
// hipMallocSignalMemory (exported by IPC), pinned host memory in a POSIX shm segment registered by both.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe_peer tools/probe_peer.hip -lrt && tools/probe_peer
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%d] %s -> %s (line %d)\n", g_rank, #x, hipGetErrorString(e_), __LINE__); fflush(stdout); return 2; } } while (0)
static int g_rank = 0;

__global__ void k_fill(float* d, float v) { d[threadIdx.x + blockIdx.x * blockDim.x] = v; }
__global__ void k_sum(const float* a, const float* b, float* o) { int i = threadIdx.x + blockIdx.x * blockDim.x; o[i] = a[i] + b[i]; }

static void put(const char* path, const void* p, size_t n) { FILE* f = fopen(path, "wb"); fwrite(p, 1, n, f); fclose(f); char done[256]; snprintf(done, 256, "%s.ok", path); f = fopen(done, "w"); fclose(f); }
static bool get(const char* path, void* p, size_t n) {
  char done[256]; snprintf(done, 256, "%s.ok", path);
  for (int i = 0; i < 20000; ++i) { if (access(done, F_OK) == 0) { FILE* f = fopen(path, "rb"); size_t k = fread(p, 1, n, f); fclose(f); return k == n; } usleep(1000); }
  return false;
}

int run(int variant, const char* tag) {
  const int rank = g_rank, peer = 1 - rank;
  char f_me[128], f_peer[128], s_me[128], s_peer[128];
  snprintf(f_me, 128, "/tmp/probe_peer_v%d_%d", variant, rank); snprintf(f_peer, 128, "/tmp/probe_peer_v%d_%d", variant, peer);
  snprintf(s_me, 128, "/tmp/probe_peer_sig_v%d_%d", variant, rank); snprintf(s_peer, 128, "/tmp/probe_peer_sig_v%d_%d", variant, peer);
  CK(hipSetDevice(0));
  int can = 0; (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  float* buf = nullptr;
  CK(hipExtMallocWithFlags((void**)&buf, 8192, hipDeviceMallocUncached));
  CK(hipMemset(buf, 0, 8192));
  hipIpcMemHandle_t h_me, h_peer;
  CK(hipIpcGetMemHandle(&h_me, buf));
  put(f_me, &h_me, sizeof h_me);
  if (!get(f_peer, &h_peer, sizeof h_peer)) { printf("[%d] no peer handle\n", rank); return 3; }
  float* pbuf = nullptr;
  CK(hipIpcOpenMemHandle((void**)&pbuf, h_peer, hipIpcMemLazyEnablePeerAccess));
  uint32_t *flag_me = (uint32_t*)(buf + 1024), *flag_peer = (uint32_t*)(pbuf + 1024);
  if (variant == 1) {               // signal memory, exported by IPC
    uint64_t* sig = nullptr;
    CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
    CK(hipMemset(sig, 0, 8));
    hipIpcMemHandle_t s1, s2;
    CK(hipIpcGetMemHandle(&s1, sig));
    put(s_me, &s1, sizeof s1);
    if (!get(s_peer, &s2, sizeof s2)) return 3;
    void* psig = nullptr;
    CK(hipIpcOpenMemHandle(&psig, s2, hipIpcMemLazyEnablePeerAccess));
    flag_me = (uint32_t*)sig; flag_peer = (uint32_t*)psig;
  }
  if (variant == 2) {               // pinned host memory shared through POSIX shm
    int fd = shm_open("/probe_peer_shm", O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, 4096) != 0) { printf("[%d] shm failed\n", rank); return 3; }
    void* m = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    CK(hipHostRegister(m, 4096, hipHostRegisterMapped));
    void* dm = nullptr;
    CK(hipHostGetDevicePointer(&dm, m, 0));
    flag_me = (uint32_t*)dm + 16 * rank; flag_peer = (uint32_t*)dm + 16 * peer;
    if (rank == 0) memset(m, 0, 4096);
    usleep(200000);
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  float* out = nullptr; CK(hipMalloc((void**)&out, 4096));
  const int R = 200;
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 1; r <= R; ++r) {
    const int par = r & 1;
    hipLaunchKernelGGL(k_fill, dim3(2), dim3(256), 0, st, buf + 512 * par, (float)(r * (rank + 1)));
    CK(hipStreamWriteValue32(st, flag_me, (uint32_t)r, 0));
    CK(hipStreamWaitValue32(st, flag_peer, (uint32_t)r, hipStreamWaitValueGte, 0xFFFFFFFFu));
    hipLaunchKernelGGL(k_sum, dim3(2), dim3(256), 0, st, buf + 512 * par, pbuf + 512 * par, out);
  }
  CK(hipStreamSynchronize(st));
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
  float host[512]; CK(hipMemcpy(host, out, 2048, hipMemcpyDeviceToHost));
  bool ok = true; for (int i = 0; i < 512; ++i) ok &= (host[i] == (float)(R * 3));
  printf("[%d] %-28s can_wait_value=%d  %s  %.1f us per round (fill, write flag, wait peer flag, sum)\n", rank, tag, can, ok ? "sums OK" : "SUMS WRONG", us);
  fflush(stdout);
  return ok ? 0 : 4;
}

int main(int argc, char** argv) {
  const bool child = argc > 1 && !strcmp(argv[1], "child");
  g_rank = child ? 1 : 0;
  if (!child) { (void)system("rm -f /tmp/probe_peer_*"); shm_unlink("/probe_peer_shm"); }
  pid_t pid = 0;
  if (!child) { pid = fork(); if (pid == 0) { execl(argv[0], argv[0], "child", (char*)nullptr); _exit(127); } }
  int rc = 0;
  const char* tags[3] = {"flag in the uncached buffer", "hipMallocSignalMemory + IPC", "pinned host shm"};
  for (int v = 0; v < 3; ++v) { int r = run(v, tags[v]); if (r) { printf("[%d] variant %d failed (%d)\n", g_rank, v, r); fflush(stdout); rc |= 1 << v; } }
  if (!child) { int st = 0; waitpid(pid, &st, 0); }
  return rc ? 1 : 0;
}
