"""Where the 3.9 ms of staging 1M timesteps of fp64 rollouts (observations + actions -> fp32 device blocks) go:
pointer collection (Python), the native gather + conversion, the queued H2D copies, the final wait."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.engine import UpdateEngine
from mjrl_amd.utils import ingest
from mjrl_amd.utils.ingest import PathStager
eng = UpdateEngine(17, 6, (64, 64))
rng = np.random.RandomState(0)
base = [dict(observations=rng.randn(1000, 17), actions=rng.randn(1000, 6)) for _ in range(1000)]
def fresh():
    return [dict(observations=p["observations"].copy(), actions=p["actions"].copy()) for p in base]
for threads in (16, 32):
  for group_rows in (262144, 131072, 65536):
    st = {k: PathStager(eng.backend, threads=threads, group_rows=group_rows) for k in ("observations", "actions")}
    rows = []
    for it in range(6):
        paths = fresh()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in ("observations", "actions"):
            st[k].stage(paths, (k,), wait=False, hostcast=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    rows = rows[2:]
    print("threads %2d group_rows %6d: host side (pointers + gather + queueing) %.2f ms, until the copies have landed %.2f ms"
          % (threads, group_rows, min(r[0] for r in rows), min(r[1] for r in rows)))
# the pieces of one stage() call
paths = fresh()
s = st["observations"]
t0 = time.perf_counter(); n = len(paths); offs = np.zeros(n + 1, np.int64); np.cumsum([len(p["observations"]) for p in paths], out=offs[1:]); t1 = time.perf_counter()
arr = (ctypes.c_void_p * n)()
fb, ao = ctypes.c_char.from_buffer, ctypes.addressof
for i, p in enumerate(paths):
    arr[i] = ao(fb(p["observations"]))
t2 = time.perf_counter()
s.begin(("observations",), (17,), (np.float64,), 1000000, hostcast=True); t3 = time.perf_counter()
print("offsets %.3f ms, pointers %.3f ms, begin() %.3f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))

# ---- inside one stage(): the native gather calls and the copy submissions, group by group
import ctypes
from mjrl_amd._lib import check
lib = eng.lib
for key, width in (("observations", 17), ("actions", 6)):
    s = PathStager(eng.backend, threads=16)
    for rep in range(3):
        paths = fresh()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = ingest.collect_arrays(paths, key)
        offs = np.zeros(len(paths) + 1, np.int64); np.cumsum(got[1], out=offs[1:])
        t1 = time.perf_counter()
        s.begin((key,), (width,), (np.float64,), int(offs[-1]), hostcast=True)
        t2 = time.perf_counter()
        slot = s._slots[key]
        tg, ts = [], []
        first, n = 0, len(paths)
        while first < n:
            last = int(np.searchsorted(offs, offs[first] + s.group_rows, side="left")); last = min(max(last, first + 1), n)
            a = time.perf_counter()
            check(lib.mjx_host_gather_f64_f32(ctypes.c_void_p(slot["pin"].data_ptr()), ctypes.c_void_p(got[0].ctypes.data),
                                              offs.ctypes.data_as(ctypes.c_void_p), first, last - first, width, 16))
            b = time.perf_counter()
            s._send(int(offs[first]), int(offs[last]))
            c = time.perf_counter()
            tg.append(1e3 * (b - a)); ts.append(1e3 * (c - b)); first = last
        t3 = time.perf_counter()
        torch.cuda.synchronize(); t4 = time.perf_counter()
    print("%s: walk %.3f, begin %.3f, gathers %s, sends %s, total host %.3f, landed +%.3f ms"
          % (key, 1e3 * (t1 - t0), 1e3 * (t2 - t1), [round(x, 3) for x in tg], [round(x, 3) for x in ts], 1e3 * (t3 - t0), 1e3 * (t4 - t3)))
for nt in (8, 16, 32):
    paths = fresh()
    got = ingest.collect_arrays(paths, "observations")
    offs = np.zeros(len(paths) + 1, np.int64); np.cumsum(got[1], out=offs[1:])
    pin = torch.empty((1000000, 17), dtype=torch.float32, pin_memory=True)
    ts = []
    for rep in range(4):
        a = time.perf_counter()
        check(lib.mjx_host_gather_f64_f32(ctypes.c_void_p(pin.data_ptr()), ctypes.c_void_p(got[0].ctypes.data), offs.ctypes.data_as(ctypes.c_void_p), 0, 1000, 17, nt))
        ts.append(1e3 * (time.perf_counter() - a))
    print("convert-gather of 136 MB fp64 -> 68 MB fp32, %d threads: first (cold source) %.2f ms, then %.2f ms" % (nt, ts[0], min(ts[1:])))
