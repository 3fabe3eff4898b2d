"""Where the rollout ingestion's time goes: raw page-locked -> device copy rate for the 184 MB of a 1M-timestep fp64 batch
(one copy, chunks, two streams), the native gather alone by thread count, and the effect of running the gather threads on
the GPU's NUMA node (sched_setaffinity before the stager's threads are created)."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
N = 1_000_000
nbytes = N * 23 * 8
pin = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
pin.numpy()[:] = 1
dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
def t_copy(chunks, streams=1):
    ss = [torch.cuda.Stream() for _ in range(streams)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step = nbytes // chunks
    for i in range(chunks):
        with torch.cuda.stream(ss[i % streams]):
            dst[i * step:(i + 1) * step].copy_(pin[i * step:(i + 1) * step], non_blocking=True)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
for chunks, streams in ((1, 1), (4, 1), (8, 2), (16, 2), (16, 4)):
    t = min(t_copy(chunks, streams) for _ in range(4))
    print("H2D %d chunk(s) on %d stream(s): %.2f ms = %.1f GB/s" % (chunks, streams, 1e3 * t, nbytes / t / 1e9))
# gather alone
rng = np.random.RandomState(0)
paths = [rng.randn(1000, 17) for _ in range(1000)]
n = len(paths)
offs = np.zeros(n + 1, np.int64); np.cumsum([len(p) for p in paths], out=offs[1:])
arr = (ctypes.c_void_p * n)(*[p.ctypes.data for p in paths])
row_bytes = 17 * 8
def gather(nt):
    t0 = time.perf_counter()
    _lib.check(lib.mjx_host_gather(ctypes.c_void_p(pin.data_ptr()), arr, offs.ctypes.data_as(ctypes.c_void_p), 0, n, row_bytes, nt))
    return time.perf_counter() - t0
for nt in (1, 4, 8, 16, 32, 64):
    t = min(gather(nt) for _ in range(5))
    print("gather of 136 MB with %2d threads: %.2f ms = %.1f GB/s" % (nt, 1e3 * t, 136e6 / t / 1e9))
try:
    node = open("/sys/class/drm/card0/device/numa_node").read().strip()
    print("GPU numa node (card0):", node, "| cpus of this process:", len(os.sched_getaffinity(0)))
    for nd in sorted(os.listdir("/sys/devices/system/node")):
        if nd.startswith("node"):
            print(" ", nd, open("/sys/devices/system/node/%s/cpulist" % nd).read().strip())
except Exception as e:
    print("numa info unavailable:", e)
