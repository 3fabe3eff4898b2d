"""Rollout ingestion: trajectories -> device-resident fp32 blocks.

The reference assembles every batch with ``np.concatenate`` over the path list and re-casts the
fp64 result to fp32 tensors at every likelihood call (mjrl/algos/batch_reinforce.py:178-182,
mjrl/policies/gaussian_mlp.py:102-109).  Here the per-path arrays are gathered straight into a
pre-allocated, page-locked staging block (mjx_host_gather: memcpy on a few native threads, no
GIL), and every group of paths is sent to the GPU on a side stream as soon as it is staged:
the host copy of group k+1 overlaps the PCIe transfer and the device-side fp64 -> fp32 cast of
group k.  No concatenated host array is ever built for observations / actions.

A sampler that wants to overlap ingestion with sampling calls ``begin / add_paths / finish``
as trajectories complete; ``stage`` does the three steps for a finished list.
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .._lib import check, ptr


class PathStager:
    """Page-locked staging + chunked asynchronous upload of per-path arrays.

    The returned tensors are views of buffers owned by the stager: they stay valid until the
    next ``begin`` / ``stage`` call (one batch at a time, like the update engine itself)."""

    def __init__(self, backend, threads=8, group_rows=262144, native=None):
        self.backend = backend
        self.torch = backend.torch
        self.device = backend.device
        self.on_gpu = self.device.type == "cuda"
        self.lib = getattr(backend, "lib", None)
        self.native = (self.lib is not None) if native is None else bool(native)   # host gather inside libmjx
        self.native_threads = int(threads)
        self.group_rows = int(group_rows)
        # (fallback without the library: NumPy copies; they hold the GIL for per-path sized arrays, so one thread)
        self.pool = ThreadPoolExecutor(max_workers=int(threads)) if (threads > 1 and not self.native) else None
        self.side = self.torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self._slots = {}            # key -> dict(pin, pin_np, dev_raw, dev_f32, width, dtype, cap)
        self._keys = ()
        self._rows = 0
        self._pending = []
        self._lock = threading.Lock()

    # ------------------------------------------------------------------ buffers
    def _slot(self, key, width, dtype, rows):
        """the page-locked staging block is kept across batches; the device tensors are allocated per batch (torch's
        caching allocator recycles the memory), so tensors handed out for an earlier batch stay valid while referenced"""
        torch = self.torch
        s = self._slots.get(key)
        tdt = torch.float64 if dtype == np.float64 else torch.float32
        if s is None or s["width"] != width or s["dtype"] != dtype or s["cap"] < rows:
            cap = max(rows, int(1.25 * (s["cap"] if s and s["width"] == width and s["dtype"] == dtype else 0)))
            pin = torch.empty((cap, width), dtype=tdt, pin_memory=self.on_gpu)
            s = dict(pin=pin, pin_np=pin.numpy(), width=width, dtype=dtype, cap=cap)
            self._slots[key] = s
        if self.on_gpu:
            s["dev_raw"] = torch.empty((rows, width), dtype=tdt, device=self.device)
            s["dev_f32"] = torch.empty((rows, width), dtype=torch.float32, device=self.device) if tdt == torch.float64 else s["dev_raw"]
        else:                                           # CPU stand-in (tests): same ownership rules, plain memory
            s["dev_raw"] = torch.empty((rows, width), dtype=tdt)
            s["dev_f32"] = torch.empty((rows, width), dtype=torch.float32) if tdt == torch.float64 else s["dev_raw"]
        return s

    # ------------------------------------------------------------------ incremental interface
    def begin(self, keys, widths, dtypes, capacity):
        """start a batch of at most `capacity` rows; keys e.g. ("observations", "actions")"""
        if self.on_gpu:
            self.side.synchronize()             # the previous batch's transfers have left the staging block
        self._keys = tuple(keys)
        for k, w, dt in zip(keys, widths, dtypes):
            self._slot(k, int(w), np.dtype(dt).type, int(capacity))
        self._rows = 0
        self._pending = []
        if self.on_gpu:
            self.side.wait_stream(self.torch.cuda.current_stream(self.device))   # kernels still reading the last batch

    def _copy_group(self, paths, row0):
        o = row0
        for p in paths:
            T = len(p[self._keys[0]])
            for k in self._keys:
                a = p[k]
                dst = self._slots[k]["pin_np"][o:o + T]
                np.copyto(dst, a if a.ndim == 2 else a.reshape(T, -1), casting="same_kind")
            o += T
        return row0, o

    def _send(self, lo, hi):
        """rows [lo, hi) are staged: transfer (and cast) them on the side stream"""
        if hi <= lo:
            return
        torch = self.torch
        for k in self._keys:
            s = self._slots[k]
            if not self.on_gpu:
                s["dev_raw"][lo:hi].copy_(s["pin"][lo:hi])
                if s["dev_f32"] is not s["dev_raw"]:
                    s["dev_f32"][lo:hi].copy_(s["pin"][lo:hi])       # fp64 -> fp32 (round to nearest even, like astype)
                continue
            with torch.cuda.stream(self.side):
                s["dev_raw"][lo:hi].copy_(s["pin"][lo:hi], non_blocking=True)
                if s["dev_f32"] is not s["dev_raw"]:
                    check(self.lib.mjx_cast_f64_f32(ptr(s["dev_raw"][lo:hi]), (hi - lo) * s["width"], ptr(s["dev_f32"][lo:hi]),
                                                    self.side.cuda_stream))

    def _add_paths_native(self, paths):
        """the gather runs in libmjx (mjx_host_gather: plain memcpy on a few threads, no GIL), group by group; each
        group's transfer is queued as soon as it is staged, so it overlaps the gather of the next group"""
        import ctypes
        n = len(paths)
        key0 = self._keys[0]
        offs = np.zeros(n + 1, np.int64)
        np.cumsum([len(p[key0]) for p in paths], out=offs[1:])
        total = int(offs[-1])
        row0 = self._rows
        cap = min(self._slots[k]["cap"] for k in self._keys)
        if row0 + total > cap:
            raise ValueError("PathStager: batch exceeds the capacity given to begin() (%d > %d rows)" % (row0 + total, cap))
        keep, srcs = [], {}
        for k in self._keys:
            s = self._slots[k]
            arr = (ctypes.c_void_p * n)()
            for i, p in enumerate(paths):
                a = p[k]
                if a.dtype != s["dtype"] or not a.flags.c_contiguous:
                    a = np.ascontiguousarray(a, dtype=s["dtype"])
                    keep.append(a)
                arr[i] = a.__array_interface__["data"][0]
            srcs[k] = arr
        offp = offs.ctypes.data_as(ctypes.c_void_p)
        first = 0
        while first < n:
            last = int(np.searchsorted(offs, offs[first] + self.group_rows, side="left"))
            last = min(max(last, first + 1), n)
            for k in self._keys:
                s = self._slots[k]
                row_bytes = s["width"] * s["pin"].element_size()
                dst = ctypes.c_void_p(s["pin"].data_ptr() + row0 * row_bytes)
                check(self.lib.mjx_host_gather(dst, srcs[k], offp, first, last - first, row_bytes, self.native_threads))
            self._send(row0 + int(offs[first]), row0 + int(offs[last]))
            first = last
        self._rows += total
        del keep

    def add_paths(self, paths):
        """stage `paths` (appended after what was added before) and queue their transfer"""
        if self.native:
            return self._add_paths_native(paths)
        self._drain(block=False)
        groups, cur, rows = [], [], 0
        for p in paths:
            cur.append(p)
            rows += len(p[self._keys[0]])
            if rows >= self.group_rows:
                groups.append((cur, rows)); cur, rows = [], 0
        if cur:
            groups.append((cur, rows))
        for g, r in groups:
            row0 = self._rows
            cap = min(self._slots[k]["cap"] for k in self._keys)
            if row0 + r > cap:
                raise ValueError("PathStager: batch exceeds the capacity given to begin() (%d > %d rows)" % (row0 + r, cap))
            self._rows += r
            if self.pool is not None:
                self._pending.append(self.pool.submit(self._copy_group, g, row0))
            else:
                self._send(*self._copy_group(g, row0))

    def _drain(self, block):
        while self._pending and (block or self._pending[0].done()):
            lo, hi = self._pending.pop(0).result()
            self._send(lo, hi)

    def finish(self):
        """-> dict key -> (rows, width) fp32 device tensor; the current stream is ordered after the transfers"""
        self._drain(block=True)
        if self.on_gpu:
            self.torch.cuda.current_stream(self.device).wait_stream(self.side)
        return {k: self._slots[k]["dev_f32"][:self._rows] for k in self._keys}

    def raw(self, key):
        """the batch as it was uploaded (fp64 when the paths are fp64): what the value baselines' feature kernels read"""
        return self._slots[key]["dev_raw"][:self._rows]

    # ------------------------------------------------------------------ one-shot
    def stage(self, paths, keys=("observations", "actions")):
        first = paths[0]
        widths = [first[k].shape[1] if first[k].ndim == 2 else 1 for k in keys]
        dtypes = [np.float64 if first[k].dtype == np.float64 else np.float32 for k in keys]
        rows = sum(len(p[keys[0]]) for p in paths)
        self.begin(keys, widths, dtypes, rows)
        self.add_paths(paths)
        return self.finish()

    def close(self):
        if self.pool is not None:
            self.pool.shutdown(wait=True)
            self.pool = None


# ---------------------------------------------------------------------- one upload per batch and process
# A training iteration touches the same trajectories three times -- baseline.predict (advantages), the policy
# update, baseline.fit (mjrl/algos/batch_reinforce.py:61-114) -- and the reference rebuilds / re-casts the
# concatenated arrays every time.  The registry below keeps the last staged batch per device: whoever asks first
# uploads, the others get the same device tensors.
_SHARED = {}
_SHARED_LOCK = threading.Lock()     # a prefetch thread and the main thread may ask for the same batch at the same time


def _probe(a):
    """a few values of one per-path array: catches in-place rewrites of arrays that are still the same objects"""
    f = a.reshape(-1)
    k = f.shape[0]
    return (float(f[0]), float(f[k // 3]), float(f[(2 * k) // 3]), float(f[-1])) if k else ()


def _same_batch(ent, paths, key):
    """is `paths` the very batch `ent` uploaded?  Identity of the list AND of every per-path array, against STRONG
    references the entry holds (an id() can be recycled once the objects are freed; a held object's cannot), plus a
    few probe values per array against in-place edits."""
    if ent["paths"] is not paths or len(ent["arrays"]) != len(paths):
        return False
    for a, p, pr in zip(ent["arrays"], paths, ent["probes"]):
        b = p[key]
        if a is not b or _probe(b) != pr:
            return False
    return True


def stage_shared(backend, paths, keys):
    """-> dict key -> dict(f32=(N, w) fp32 device tensor, raw=(N, w) tensor in the paths' dtype).  Re-uses the upload
    of the same `paths` list (the same list object holding the same array objects, see _same_batch) made earlier in
    this process on this device; train_step drops the entries when its iteration ends (drop_shared_batch)."""
    dev = backend.device
    out = {}
    with _SHARED_LOCK:
        reg = _SHARED.setdefault((dev.type, dev.index), {})
        for k in keys:
            ent = reg.get(k)
            if ent is None or ent.get("paths") is None or not _same_batch(ent, paths, k):
                st = ent["stager"] if ent is not None else PathStager(backend)
                f32 = st.stage(paths, (k,))[k]
                arrays = [p[k] for p in paths]
                ent = reg[k] = dict(stager=st, f32=f32, raw=st.raw(k), paths=paths, arrays=arrays,
                                    probes=[_probe(a) for a in arrays])
            out[k] = dict(f32=ent["f32"], raw=ent["raw"])
    return out


def drop_shared_batch():
    """forget WHICH batch is staged (the stagers and their page-locked blocks stay): the next stage_shared uploads
    again whatever it is given, and the host trajectories of the finished iteration are released."""
    with _SHARED_LOCK:
        for reg in _SHARED.values():
            for ent in reg.values():
                ent["paths"] = ent["arrays"] = ent["probes"] = None
                ent["f32"] = ent["raw"] = None


def drop_shared():
    """forget the staged batches (tests; releasing device memory)"""
    with _SHARED_LOCK:
        for reg in _SHARED.values():
            for ent in reg.values():
                ent["stager"].close()
        _SHARED.clear()
