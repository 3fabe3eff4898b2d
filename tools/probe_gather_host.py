"""Host side of the rollout staging alone: mjx_host_gather_f64_f32 (fp64 per-path arrays -> one fp32 block in page-locked memory)
for the observation block (1M x 17) and, concurrently, the action block (1M x 6) from cold sources.  No device work.
r04 ran it on an experimental build whose conversion loop had streaming stores (MJX_GATHER_NT=1: 32-byte non-temporal stores, no
read-for-ownership of the staging block) and a source prefetch (MJX_GATHER_PREFETCH=bytes): 0.60 ms (ordinary stores, 16 threads)
vs 0.62-0.80 ms for the whole observation block on the 2 x 64-core hosts -- no gain, the variant was not kept
(profiles/r04_e2e/gather_host_streaming_stores.log); on the library as committed the two environment variables do nothing and
all rows measure the ordinary loop.
python tools/probe_gather_host.py"""
import os, sys, time, threading, ctypes, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mjrl_amd import _lib
lib = _lib.load()
try:
    import torch
    pin = torch.cuda.is_available()
except Exception:
    pin = False
rng = np.random.RandomState(0)
NB = 6
batches = [[dict(observations=rng.randn(1000, 17), actions=rng.randn(1000, 6)) for _ in range(1000)] for _ in range(NB)]
def block(n, w):
    if pin:
        t = torch.empty((n, w), dtype=torch.float32, pin_memory=True)
        return t, t.numpy()
    a = np.empty((n, w), np.float32)
    return a, a
keep_o, dst_o = block(1_000_000, 17)
keep_a, dst_a = block(1_000_000, 6)
offs = np.arange(1001, dtype=np.int64) * 1000
def ptrs(paths, key):
    arr = (ctypes.c_void_p * len(paths))()
    for i, p in enumerate(paths):
        arr[i] = p[key].ctypes.data
    return arr
P = [(ptrs(b, "observations"), ptrs(b, "actions")) for b in batches]
def gather(dst, pp, w, nt, groups=4):
    for g in range(groups):                                  # group by group like mjx_stage_async (250 paths each)
        first, cnt = g * (1000 // groups), 1000 // groups
        rc = lib.mjx_host_gather_f64_f32(dst.ctypes.data, pp, offs.ctypes.data, first, cnt, w, nt)
        assert rc == 0
def run(nt, both):
    ts = []
    for rep in range(2 * NB):
        po, pa = P[rep % NB]
        t0 = time.perf_counter()
        if both:
            th = threading.Thread(target=gather, args=(dst_a, pa, 6, nt))
            th.start()
        gather(dst_o, po, 17, nt)
        t1 = time.perf_counter()
        if both:
            th.join()
        t2 = time.perf_counter()
        ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    ts = ts[NB:]
    return round(min(t[0] for t in ts), 3), round(sorted(t[0] for t in ts)[len(ts) // 2], 3), round(min(t[1] for t in ts), 3)
out = {}
for nt_mode, pf in ((0, 0), (1, 0), (1, 512), (1, 2048), (1, 4096)):
    os.environ["MJX_GATHER_NT"] = str(nt_mode)
    os.environ["MJX_GATHER_PREFETCH"] = str(pf)
    for nt in (16, 32):
        for both in (False, True):
            out["nt_stores=%d prefetch=%d threads=%d %s" % (nt_mode, pf, nt, "obs+act" if both else "obs alone")] = run(nt, both)
    want = np.concatenate([p["observations"] for p in batches[(2 * NB - 1) % NB]]).astype(np.float32)
    assert np.array_equal(dst_o, want), "conversion differs"
for k, v in out.items():
    print("%-56s obs min %.3f median %.3f ms, all landed min %.3f ms" % (k, *v))
