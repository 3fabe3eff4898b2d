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
import contextlib
import os
import sys
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .._lib import check, ptr

try:                                        # C-speed walk over the path list (csrc/pathwalk.c, built by __graft_entry__.build());
    from .. import _pathwalk                # without it the same walk runs in Python (~0.5 us per array instead of ~0.05)
except Exception:                           # pragma: no cover
    _pathwalk = None


MALLOC_TUNED = None          # None: not asked for yet (importing this package changes nothing in the process); True / False: tune_malloc()'s outcome


def tune_malloc():
    """Host allocator settings for a TRAINING process that turns over hundreds of MB of rollout arrays per iteration.  Called by
    the training entry points -- BatchREINFORCE.train_step, mjrl_amd.dropin.install, bench.py -- never by importing the package
    (r06: a process-global allocator policy is not an import side effect of a drop-in library; a process that only unpickles a
    policy or calls get_action keeps glibc's defaults).  Idempotent; MJX_MALLOC_TUNE=0 leaves the allocator alone.

    Why: with glibc's defaults a rollout array of 136 KB (1 000 steps x 17 observations in fp64) sits right at the mmap threshold:
    batches end up either as 2 000 separate mappings (every free an munmap) or at the top of the heap (every free a trim, an sbrk
    with its page-table work): releasing ONE batch then costs 6-13 ms instead of 0.5 (tools/e2e_timeline.py with TL_ALTERNATE=1:
    the whole difference between an 8 ms and a 20 ms NPG.train_from_paths), wherever in the iteration the last reference dies.
    M_MMAP_THRESHOLD at its maximum (32 MB: rollout arrays come from the heap) and M_TRIM_THRESHOLD at 2 GB (the heap keeps what
    a batch needs) make that release a list insertion.
    Consequence for the resident set: freed heap pages are no longer handed back to the kernel, so the process keeps the high-water
    mark of its rollout batches (~190 MB per 1M fp64 HalfCheetah timesteps held at once; tools/soak.py: RSS flat after the first
    iterations of 300, profiles/r05_bench/soak.log) instead of oscillating by that amount every iteration."""
    global MALLOC_TUNED
    if MALLOC_TUNED is not None:
        return MALLOC_TUNED
    if os.environ.get("MJX_MALLOC_TUNE", "1") == "0" or not sys.platform.startswith("linux"):
        MALLOC_TUNED = False
        return False
    try:
        import ctypes
        libc = ctypes.CDLL(None)
        M_TRIM_THRESHOLD, M_MMAP_THRESHOLD = -1, -3
        ok = libc.mallopt(M_MMAP_THRESHOLD, 32 * 1024 * 1024) == 1
        ok = (libc.mallopt(M_TRIM_THRESHOLD, 2 ** 31 - 1) == 1) and ok
        MALLOC_TUNED = bool(ok)
    except Exception:                       # pragma: no cover  (another libc: nothing to tune)
        MALLOC_TUNED = False
    return MALLOC_TUNED


def minibatch_indices(lib, num_samples, steps, mb_size):
    """(steps, mb_size) int32 == np.stack([np.random.choice(num_samples, size=mb_size) for _ in range(steps)]) -- the reference's
    draws (behavior_cloning.py:113, ppo_clip.py:77), the same values and the same advance of NumPy's global stream -- in one native
    loop (mjx_host_mt19937_randint) instead of `steps` Python-level calls: 2.8 instead of 168 ms for the 15 625 steps of one epoch
    over 1M timesteps."""
    import ctypes
    st = np.random.get_state()
    if lib is None or st[0] != 'MT19937' or num_samples < 1 or num_samples >= 2 ** 31:
        return np.stack([np.random.choice(num_samples, size=mb_size) for _ in range(steps)]).astype(np.int32)
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = ctypes.c_int32(int(st[2]))
    out = np.empty((int(steps), int(mb_size)), np.int32)
    check(lib.mjx_host_mt19937_randint(ctypes.c_void_p(key.ctypes.data), ctypes.byref(pos), int(num_samples), int(steps) * int(mb_size),
                                       ctypes.c_void_p(out.ctypes.data)))
    np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))
    return out


def collect_arrays(paths, key):
    """addresses and first dimensions of paths[.][key] -> (ptrs: (n,) uint64 ndarray, lens: (n,) int64 ndarray, width, itemsize), or
    None when the arrays are not uniform C-contiguous float32 / float64 blocks (the caller converts / copies them itself).  The
    addresses are valid while the caller holds the arrays."""
    n = len(paths)
    if _pathwalk is None or n == 0 or type(paths) is not list:
        return None
    ptrs, lens = np.empty(n, np.uint64), np.empty(n, np.int64)
    r = _pathwalk.collect(paths, key, ptrs, lens)
    if r < 0:
        return None
    return ptrs, lens, r >> 4, r & 15


class PathStager:
    """Page-locked staging + chunked asynchronous upload of per-path arrays.

    The returned tensors are views of buffers owned by the stager: they stay valid until the
    next ``begin`` / ``stage`` call (one batch at a time, like the update engine itself)."""

    def __init__(self, backend, threads=None, group_rows=262144, native=None):
        # gather threads per staging job: the rollouts of a fresh batch sit in DRAM, not in cache -- converting 136 MB of fp64
        # observations alone takes 2.35 / 1.35 / 1.06 ms on 8 / 16 / 32 threads of the 2 x 64-core hosts (tools/stage_breakdown.py), but
        # a batch's blocks are staged by concurrent jobs (observations, actions, advantages): 16 each measured best end to end
        # (profiles/r04_e2e/sweep.log; MJX_STAGE_THREADS / MJX_STAGE_GROUP_ROWS override)
        if threads is None:
            import os
            threads = int(os.environ.get("MJX_STAGE_THREADS", "0")) or max(4, min(16, (os.cpu_count() or 8) // 2))
        self.backend = backend
        self.torch = backend.torch
        self.device = backend.device
        self.on_gpu = self.device.type == "cuda"
        self.lib = getattr(backend, "lib", None)
        self.native = (self.lib is not None) if native is None else bool(native)   # host gather inside libmjx
        self.native_threads = int(threads)
        import os as _os
        self.group_rows = int(_os.environ.get("MJX_STAGE_GROUP_ROWS", "0")) or int(group_rows)
        # (fallback without the library: NumPy copies; they hold the GIL for per-path sized arrays, so one thread)
        self.pool = ThreadPoolExecutor(max_workers=int(threads)) if (threads > 1 and not self.native) else None
        self.side = self.torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self._slots = {}            # key -> dict(pin, pin_np, dev_raw, dev_f32, width, dtype, cap)
        self._keys = ()
        self._rows = 0
        self._pending = []
        self._jobs = []             # native staging jobs in flight (mjx_stage_async): joined by join() / finish() / the next begin()
        self._lock = threading.Lock()

    # ------------------------------------------------------------------ buffers
    def _slot(self, key, width, dtype, rows, hostcast=False):
        """the page-locked staging block is kept across batches; the device tensors are allocated per batch (torch's
        caching allocator recycles the memory), so tensors handed out for an earlier batch stay valid while referenced.
        hostcast: fp64 paths are converted to fp32 by the gather itself (mjx_host_gather_f64_f32) -- the staging block and the
        upload are fp32, there is no raw fp64 device block"""
        torch = self.torch
        s = self._slots.get(key)
        hostcast = bool(hostcast and dtype == np.float64 and self.native)
        tdt = torch.float64 if (dtype == np.float64 and not hostcast) else torch.float32
        if s is None or s["width"] != width or s["dtype"] != dtype or s["cap"] < rows or s.get("hostcast", False) != hostcast:
            same = s and s["width"] == width and s["dtype"] == dtype and s.get("hostcast", False) == hostcast
            cap = max(rows, int(1.25 * (s["cap"] if same else 0)))
            pin = torch.empty((cap, width), dtype=tdt, pin_memory=self.on_gpu)
            s = dict(pin=pin, pin_np=pin.numpy(), width=width, dtype=dtype, cap=cap, hostcast=hostcast)
            self._slots[key] = s
        if self.on_gpu:
            s["dev_raw"] = torch.empty((rows, width), dtype=tdt, device=self.device)
            s["dev_f32"] = torch.empty((rows, width), dtype=torch.float32, device=self.device) if tdt == torch.float64 else s["dev_raw"]
        else:                                           # CPU stand-in (tests): same ownership rules, plain memory
            s["dev_raw"] = torch.empty((rows, width), dtype=tdt)
            s["dev_f32"] = torch.empty((rows, width), dtype=torch.float32) if tdt == torch.float64 else s["dev_raw"]
        return s

    def join(self):
        """wait until the native staging jobs of the last stage() have QUEUED all their copies on the side stream (the copies
        themselves are ordered by the stream: current_stream.wait_stream(self.side) / finish())"""
        with self._lock:
            jobs, self._jobs = self._jobs, []
        err = None
        for j in jobs:
            try:
                check(self.lib.mjx_stage_wait(j))
            except Exception as e:              # (join the rest first)
                err = err or e
        if err is not None:
            raise err

    # ------------------------------------------------------------------ incremental interface
    def begin(self, keys, widths, dtypes, capacity, hostcast=()):
        """start a batch of at most `capacity` rows; keys e.g. ("observations", "actions"); hostcast: the keys whose fp64
        arrays only have to reach the device as fp32 (no raw block)"""
        self.join()
        if self.on_gpu:
            self.side.synchronize()             # the previous batch's transfers have left the staging block
        self._keys = tuple(keys)
        for k, w, dt in zip(keys, widths, dtypes):
            self._slot(k, int(w), np.dtype(dt).type, int(capacity), hostcast=(hostcast is True or k in hostcast))
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
        keep, srcs = [], {}
        fast = {}
        pre = getattr(self, "_pre", None)
        pre = pre[1] if (pre is not None and pre[0] is paths) else {}
        self._pre = None
        for k in self._keys:                                 # the C walk: addresses + lengths of every array, ~50 ns apiece
            s = self._slots[k]
            got = pre[k] if k in pre else collect_arrays(paths, k)
            if got is not None and got[3] == np.dtype(s["dtype"]).itemsize and got[2] == s["width"]:
                fast[k] = got
        if key0 in fast:
            np.cumsum(fast[key0][1], out=offs[1:])
        else:
            np.cumsum([len(p[key0]) for p in paths], out=offs[1:])
        total = int(offs[-1])
        row0 = self._rows
        cap = min(self._slots[k]["cap"] for k in self._keys)
        if row0 + total > cap:
            raise ValueError("PathStager: batch exceeds the capacity given to begin() (%d > %d rows)" % (row0 + total, cap))
        from_buffer, addressof = ctypes.c_char.from_buffer, ctypes.addressof
        for k in self._keys:
            s = self._slots[k]
            if k in fast:
                keep.append(fast[k][0])
                srcs[k] = ctypes.c_void_p(fast[k][0].ctypes.data)
                continue
            arr = (ctypes.c_void_p * n)()
            for i, p in enumerate(paths):
                a = p[k]
                if a.dtype != s["dtype"] or not a.flags.c_contiguous:
                    a = np.ascontiguousarray(a, dtype=s["dtype"])
                    keep.append(a)
                try:                                    # (the buffer protocol: 0.26 us per array; __array_interface__ builds a
                    arr[i] = addressof(from_buffer(a))  #  dict per call, 1.1 us -- 3 ms of GIL time per 1 000-path iteration)
                except (TypeError, ValueError):         # read-only or empty arrays
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
                if s["hostcast"]:
                    check(self.lib.mjx_host_gather_f64_f32(dst, srcs[k], offp, first, last - first, s["width"], self.native_threads))
                else:
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

    def finish(self, wait=True):
        """-> dict key -> (rows, width) fp32 device tensor; the current stream is ordered after the transfers
        (wait=False: the caller orders its consumers itself, after an event it records on ``self.side``)"""
        self._drain(block=True)
        self.join()
        if self.on_gpu and wait:
            self.torch.cuda.current_stream(self.device).wait_stream(self.side)
        return {k: self._slots[k]["dev_f32"][:self._rows] for k in self._keys}

    def raw(self, key):
        """the batch as it was uploaded (fp64 when the paths are fp64): what the value baselines' feature kernels read.
        None for a key staged with hostcast (its fp64 values never reached the device)."""
        s = self._slots[key]
        return None if s["hostcast"] else s["dev_raw"][:self._rows]

    def _stage_async(self, paths, keys, pre):
        """the whole batch as one native job per key (mjx_stage_async): gather + conversion + queued copies run on libmjx's
        threads, this call returns at once.  -> False when some key's arrays are not uniform float blocks (general route)."""
        import ctypes
        if not (self.native and self.on_gpu and all(k in pre for k in keys)):
            return False
        for k in keys:
            s, (ptrs, lens, width, isz) = self._slots[k], pre[k]
            if isz != np.dtype(s["dtype"]).itemsize or width != s["width"]:
                return False
        rows = int(pre[keys[0]][1].sum())
        for k in keys:
            s, (ptrs, lens, width, isz) = self._slots[k], pre[k]
            if int(lens.sum()) != rows:
                raise ValueError("PathStager: the arrays of key %r do not have the row counts of key %r" % (k, keys[0]))
            job = ctypes.c_void_p()
            f32 = s["dev_f32"] if (s["dev_f32"] is not s["dev_raw"]) else None
            check(self.lib.mjx_stage_async(ctypes.byref(job), ctypes.c_void_p(ptrs.ctypes.data), ctypes.c_void_p(lens.ctypes.data), len(paths),
                                           width, isz, 1 if s["hostcast"] else 0, ptr(s["pin"]), ptr(s["dev_raw"]), ptr(f32),
                                           self.group_rows, self.native_threads, self.device.index or 0, ctypes.c_void_p(self.side.cuda_stream)))
            with self._lock:
                self._jobs.append(job)
        self._rows = rows
        return True

    # ------------------------------------------------------------------ one-shot
    def stage(self, paths, keys=("observations", "actions"), wait=True, hostcast=()):
        first = paths[0]
        widths = [first[k].shape[1] if first[k].ndim == 2 else 1 for k in keys]
        dtypes = [np.float64 if first[k].dtype == np.float64 else np.float32 for k in keys]
        pre = {k: g for k, g in ((k, collect_arrays(paths, k)) for k in keys) if g is not None} if self.native else {}
        self._pre = (paths, pre)                     # (one C walk per key serves the row count here and the gather below)
        rows = int(pre[keys[0]][1].sum()) if keys[0] in pre else sum(len(p[keys[0]]) for p in paths)
        self.begin(keys, widths, dtypes, rows, hostcast=hostcast)
        if self._stage_async(paths, keys, pre):
            self._pre = None
            if wait:
                return self.finish(True)
            return {k: self._slots[k]["dev_f32"][:self._rows] for k in keys}      # (the caller joins: join() / _order_after)
        self.add_paths(paths)
        return self.finish(wait)

    def close(self):
        try:
            self.join()                          # (a staging job must not outlive the buffers it writes)
        except Exception:                        # pragma: no cover
            pass
        if self.pool is not None:
            self.pool.shutdown(wait=True)
            self.pool = None


# ---------------------------------------------------------------------- host <-> device, never through pageable memory
# hipMemcpy between the device and PAGEABLE host memory pins the host range on the fly for transfers of about a megabyte
# and more (a userptr registration that ROCm keeps cached).  When the process later maps or unmaps memory over such a
# range -- NumPy allocating / freeing multi-megabyte temporaries is enough -- the driver's MMU notifier evicts the
# process's GPU queues and the NEXT submission, whatever it is, waits 10-35 ms for them to be restored: the
# "host-runtime stall" of round 1's iteration timings (tools/probe_stall2.py --pageable reproduces it, tools/
# probe_stall3.py localises it).  So every transfer of this package goes through page-locked memory: the PathStager's
# staging blocks for the rollouts, the bounce buffers below for everything else.
_BOUNCE = {}
_BOUNCE_LOCK = threading.Lock()
_BOUNCE_MIN = 1 << 16          # below 64 KB the runtime stages the copy itself (no on-the-fly pinning)


def _bounce_acquire(torch, dev, n):
    with _BOUNCE_LOCK:
        pool = _BOUNCE.setdefault((dev.type, dev.index), [])
        slot = None
        for s_ in pool:
            if s_["cap"] >= n and not s_["busy"] and s_["event"].query():
                slot = s_
                break
        if slot is None:
            cap = max(1 << 20, 1 << int(n - 1).bit_length())
            slot = dict(cap=cap, pin=torch.empty(cap, dtype=torch.uint8, pin_memory=True), event=torch.cuda.Event(), busy=False)
            slot["np"] = slot["pin"].numpy()
            slot["event"].record(torch.cuda.current_stream(dev))
            pool.append(slot)
            if len(pool) > 8:                           # keep the page-locked footprint bounded: drop idle buffers
                pool[:] = [q for q in pool if q is slot or q["busy"] or not q["event"].query()][-8:] or [slot]
        slot["busy"] = True
    return slot


def upload(backend, a, dtype=None):
    """host ndarray -> device tensor of the same shape (dtype: optional NumPy dtype to convert to on the host first).
    The copy is asynchronous on the current stream; the bounce buffer is recycled once its transfer has completed."""
    torch, dev = backend.torch, backend.device
    a = np.ascontiguousarray(a if dtype is None else np.asarray(a, dtype=dtype))
    if dev.type != "cuda" or a.nbytes < _BOUNCE_MIN:
        return torch.from_numpy(a).to(dev)
    n = a.nbytes
    slot = _bounce_acquire(torch, dev, n)
    try:
        slot["np"][:n] = a.reshape(-1).view(np.uint8)
        out = torch.empty(a.shape, dtype=torch.from_numpy(np.empty(0, a.dtype)).dtype, device=dev)
        out.view(torch.uint8).reshape(-1).copy_(slot["pin"][:n], non_blocking=True)
        slot["event"].record(torch.cuda.current_stream(dev))
    finally:
        slot["busy"] = False
    return out


def download(backend, t):
    """device tensor -> fresh host ndarray of the same shape / dtype (synchronises the current stream)"""
    torch, dev = backend.torch, backend.device
    n = t.numel() * t.element_size()
    if dev.type != "cuda" or not t.is_cuda or n < _BOUNCE_MIN:
        return t.cpu().numpy()
    t = t.contiguous()
    slot = _bounce_acquire(torch, dev, n)
    try:
        slot["pin"][:n].copy_(t.view(-1).view(torch.uint8))            # device -> page-locked host, blocking
        out = np.empty(tuple(t.shape), dtype=torch.empty(0, dtype=t.dtype).numpy().dtype)
        out.reshape(-1).view(np.uint8)[:] = slot["np"][:n]
        slot["event"].record(torch.cuda.current_stream(dev))
    finally:
        slot["busy"] = False
    return out


_OWNED = {}                     # (device) -> page-locked buffers whose memory is handed out AS the host block
_OWNED_MAX = int(os.environ.get("MJX_OWNED_BUFFERS", "12"))


def download_owned(backend, t):
    """device tensor -> host ndarray that IS the page-locked memory the copy engine wrote: no second copy and no fresh multi-
    megabyte allocation (page faults) per block.  For the blocks that are handed to the paths as views and live as long as the
    batch (returns, baseline, advantages: utils/process_samples.py).  A buffer goes back into circulation when the last view
    of it has died -- its NumPy reference count says so (every view's .base is the buffer's root array); at most
    MJX_OWNED_BUFFERS (12) buffers are kept, a caller that holds on to more batches than that gets ordinary copies."""
    torch, dev = backend.torch, backend.device
    n = t.numel() * t.element_size()
    if dev.type != "cuda" or not t.is_cuda or n < _BOUNCE_MIN or _OWNED_MAX <= 0:
        return download(backend, t)
    t = t.contiguous()
    out = None
    with _BOUNCE_LOCK:                                                        # (not re-entrant: the fall-back below runs outside it)
        pool = _OWNED.setdefault((dev.type, dev.index), [])
        ent = None
        for e in pool:
            if n <= e["cap"] <= max(4 * n, 1 << 20) and sys.getrefcount(e["np"]) <= 2:      # (2: the dict's reference + getrefcount's argument)
                ent = e
                break
        if ent is None and len(pool) >= _OWNED_MAX:
            idle = [e for e in pool if sys.getrefcount(e["np"]) <= 2]
            if idle:
                pool[:] = [e for e in pool if e is not idle[0]]               # (a wrong-sized idle buffer makes room)
        if ent is None and len(pool) < _OWNED_MAX:
            cap = max(1 << 20, 1 << int(n - 1).bit_length())
            ent = dict(cap=cap, pin=torch.empty(cap, dtype=torch.uint8, pin_memory=True))
            ent["np"] = ent["pin"].numpy()
            pool.append(ent)
        if ent is not None:                                                   # the view holds the buffer from here on
            out = ent["np"][:n].view(torch.empty(0, dtype=t.dtype).numpy().dtype).reshape(tuple(t.shape))
    if out is None:
        return download(backend, t)
    ent["pin"][:n].copy_(t.view(-1).view(torch.uint8))                        # device -> page-locked host, blocking
    return out


# ---------------------------------------------------------------------- one upload per batch and process
# A training iteration touches the same trajectories three times -- baseline.predict (advantages), the policy
# update, baseline.fit (mjrl/algos/batch_reinforce.py:61-114) -- and the reference rebuilds / re-casts the
# concatenated arrays every time.  The registry below keeps the last staged batch per device: whoever asks first
# uploads, the others get the same device tensors.
_SHARED = {}
_SHARED_LOCK = threading.Lock()     # guards the registry dicts (held briefly)
_KEY_LOCKS = {}                     # (device, key) -> lock held WHILE that key's block is being staged: a prefetch thread and
                                    # the main thread may ask for the same block at the same time; other keys are not held up


def _key_lock(dev, key):
    with _SHARED_LOCK:
        return _KEY_LOCKS.setdefault((dev.type, dev.index, key), threading.RLock())


def _order_after(backend, ent):
    """the caller's current stream waits for the block's transfers (queued on the stager's side stream -- by a native job that may
    still be running: joined first)"""
    with _DEFERRED_LOCK:                              # (a deferred hand-out is settled by whoever orders a consumer after it first)
        for i, e in enumerate(_DEFERRED):
            if e is ent:
                del _DEFERRED[i]
                break
    st = ent.get("active")
    if st is not None and st.on_gpu:
        st.join()
        backend.torch.cuda.current_stream(backend.device).wait_stream(st.side)
        return
    ev = ent.get("ready")
    if ev is not None:
        backend.torch.cuda.current_stream(backend.device).wait_event(ev)


def _probe(a):
    """a few values of one per-path array: catches in-place rewrites of arrays that are still the same objects"""
    f = a.reshape(-1)
    k = f.shape[0]
    return (float(f[0]), float(f[k // 3]), float(f[(2 * k) // 3]), float(f[-1])) if k else ()


def _probed(n):
    """indices of the per-path arrays whose values are probed (all of them up to 32 paths, 32 evenly spaced ones beyond:
    a lookup happens ~10 times per iteration and must stay far below a millisecond for 1 000 paths)"""
    return range(n) if n <= 32 else range(0, n, (n + 31) // 32)


def _probes(arrays):
    return [_probe(arrays[i]) for i in _probed(len(arrays))]


# Who may have touched the per-path arrays between two uses of a batch?  Inside BatchREINFORCE.train_step nothing but this
# package runs between compute_returns and baseline.fit -- identity of the objects plus a few probe values is enough there
# (`trusted_iteration`).  A caller that drives compute_returns / compute_advantages / train_from_paths itself may edit arrays in
# place in between (np.clip(..., out=p["rewards"]), per-path reweighting): outside the trusted scope a reuse must therefore be
# EXACT -- user-owned arrays (rewards, observations, actions: uploaded by stage_shared) are compared element by
# element with the page-locked host copy they were sent from, and blocks this package computed on the device (returns,
# baseline values, advantages: publish) are handed to the paths as READ-ONLY views, so an in-place edit raises instead of
# being silently ignored (assigning a new array to path[key] is always fine: the identity check sees it).
_TRUST = threading.local()


@contextlib.contextmanager
def trusted_iteration():
    """scope in which no foreign code can run between the uses of a batch (train_step)"""
    _TRUST.depth = getattr(_TRUST, "depth", 0) + 1
    try:
        yield
    finally:
        _TRUST.depth -= 1


def _trusted():
    return getattr(_TRUST, "depth", 0) > 0


def carried_trust():
    """the scope of the CALLING thread for a helper thread working on its behalf (the flag is thread-local): call it where the
    job is submitted, enter the returned context inside the job"""
    return trusted_iteration if _trusted() else contextlib.nullcontext


def _same_batch(ent, paths, key):
    """is `paths` the very batch `ent` uploaded?  Identity of the list AND of every per-path array, against STRONG
    references the entry holds (an id() can be recycled once the objects are freed; a held object's cannot), plus a
    few probe values of a sample of the arrays against in-place edits -- and, outside train_step, an exact comparison
    with the staged host copy (see above)."""
    arrays = ent["arrays"]
    if ent["paths"] is not paths or len(arrays) != len(paths):
        return False
    if _pathwalk is not None and type(paths) is list and type(arrays) is list:
        if not _pathwalk.identity(paths, key, arrays):
            return False
    else:
        for a, p in zip(arrays, paths):
            if a is not p[key]:
                return False
    for i, pr in zip(_probed(len(arrays)), ent["probes"]):
        if _probe(arrays[i]) != pr:
            return False
    if not _trusted() and ent.get("f32") is not None and ent.get("active") is not None:
        st = ent["active"]
        st.join()                                    # (the page-locked copy is complete once the staging job has finished)
        slot = st._slots.get(key)
        if slot is None:
            return False
        host, o = slot["pin_np"], 0
        for a in arrays:
            T = len(a)
            a2 = a if a.ndim == 2 else a.reshape(T, -1)
            if slot["hostcast"]:                       # staged as fp32: compare what the conversion of today's values gives
                a2 = a2.astype(np.float32)
            if not np.array_equal(a2, host[o:o + T]):
                return False
            o += T
    return True


_RAW_STICKY = set()          # (device type, index, key): some consumer asked for the raw block of this key before


_DEFERRED = []                # entries handed out by stage_shared(defer=True) whose consumers have not been ordered yet (settle())
_DEFERRED_LOCK = threading.Lock()


def _retire_deferred(match=None):
    """take deferred hand-outs off the list without ordering any stream after them (their consumer will never come: the batch
    is being replaced or dropped) -- the native staging job is joined so that nothing still writes into blocks about to be reused.
    match: only entries for which match(ent) holds (None: all).  (entries nobody settled kept a whole batch alive.)"""
    with _DEFERRED_LOCK:
        gone = [e for e in _DEFERRED if match is None or match(e)]
        _DEFERRED[:] = [e for e in _DEFERRED if not (match is None or match(e))]
    for e in gone:
        st = e.get("active")
        if st is not None and getattr(st, "on_gpu", False):
            try:
                st.join()
            except Exception:                     # pragma: no cover
                pass


def settle(backend):
    """order the caller's current stream after every block that stage_shared(defer=True) handed out: joins the native staging
    jobs (their copies are then queued) and makes the stream wait for the stagers' copy streams.  MUST run before the first
    kernel that reads such a block is enqueued."""
    with _DEFERRED_LOCK:
        ents, _DEFERRED[:] = list(_DEFERRED), []
    for ent in ents:
        _order_after(backend, ent)


def _release_other_batches(dev, paths, keys):
    """a NEW batch is about to be staged under `keys`: let go of the entries it replaces -- in one go and before the new batch's
    staging jobs start, instead of key by key in the middle of them.  (What a release costs is the host
    allocator's business -- 0.5 or 13 ms for the 2 000 arrays of a 1M-timestep batch with glibc's defaults, tools/e2e_timeline.py
    with TL_ALTERNATE=1; see tune_malloc above.)"""
    stale = []
    with _SHARED_LOCK:
        reg = _SHARED.get((dev.type, dev.index), {})
        for k in keys:                                     # (only the blocks about to be replaced: DAPG stages [on-policy ; demos] as a
            ent = reg.get(k)                               #  list of its own while the on-policy list's returns / advantages stay registered)
            if ent is not None and ent.get("paths") is not None and ent["paths"] is not paths:
                stale.append(ent)
    if not stale:
        return
    _retire_deferred(lambda e: any(e is x for x in stale))
    with _SHARED_LOCK:
        for ent in stale:
            if ent.get("paths") is not None and ent["paths"] is not paths:
                st = ent.get("active")
                if st is not None and getattr(st, "on_gpu", False):
                    try:
                        st.join()                          # (nothing may still be gathering from the arrays we are about to drop)
                    except Exception:                     # pragma: no cover
                        pass
                ent["paths"] = ent["arrays"] = ent["probes"] = None
                ent["f32"] = ent["raw"] = ent["active"] = None
                ent.pop("derived", None)


def _prefetch_inline(backend, paths, keys, raw):
    """the staging jobs of `keys` started from the calling thread (mjx_stage_async returns at once): no helper thread needed"""
    try:
        stage_shared(backend, paths, keys, raw=raw, defer=True)
        return True
    except Exception:                             # pragma: no cover  (left to the foreground request)
        return False


def stage_shared(backend, paths, keys, raw=None, defer=False):
    """-> dict key -> dict(f32=(N, w) fp32 device tensor, raw=(N, w) tensor in the paths' dtype).  Re-uses the upload
    of the same `paths` list (the same list object holding the same array objects, see _same_batch) made earlier in
    this process on this device; train_step drops the entries when its iteration ends (drop_shared_batch).
    raw: the keys whose block the caller needs in the paths' own dtype (None: all of them).  A key nobody has ever asked the
    raw block of is converted to fp32 by the host gather and uploaded at half its size (raw = None in the result); the first
    raw request for such a key stages it again in full -- and is remembered, so later batches go up raw at once and serve both.
    defer: return as soon as the staging jobs are STARTED (native threads gather and queue the copies); the caller does other work
    and calls settle() before it enqueues the first consumer."""
    dev = backend.device
    out = {}
    _release_other_batches(dev, paths, keys)
    for k in keys:
        need_raw = raw is None or k in raw
        tag = (dev.type, dev.index, k)
        with _key_lock(dev, k):
            with _SHARED_LOCK:
                reg = _SHARED.setdefault((dev.type, dev.index), {})
                ent = reg.get(k)
                if need_raw:
                    _RAW_STICKY.add(tag)
                as_raw = tag in _RAW_STICKY
            stale = ent is None or ent.get("paths") is None or not _same_batch(ent, paths, k)
            if stale or (need_raw and ent["raw"] is None):
                # one stager per mode and key: their page-locked blocks persist, a flow that alternates never re-allocates
                skey = "stager" if as_raw else "stager32"
                st = ent.get(skey) if ent is not None else None
                if st is None:
                    st = PathStager(backend)
                f32 = st.stage(paths, (k,), wait=False, hostcast=() if as_raw else True)[k]
                ready = None
                if st.on_gpu:
                    ready = backend.torch.cuda.Event()
                    ready.record(st.side)
                arrays = [p[k] for p in paths]
                new = dict(stager=ent.get("stager") if ent else None, stager32=ent.get("stager32") if ent else None, f32=f32,
                           raw=st.raw(k), paths=paths, arrays=arrays, probes=_probes(arrays), ready=ready)
                new[skey] = new["active"] = st
                old_ent, ent = ent, new
                if old_ent is not None:
                    _retire_deferred(lambda e: e is old_ent)        # a hand-out of the batch this one replaces: nobody will settle it
                with _SHARED_LOCK:
                    reg[k] = ent
            if defer:
                with _DEFERRED_LOCK:
                    _DEFERRED.append(ent)
            else:
                _order_after(backend, ent)
            out[k] = dict(f32=ent["f32"], raw=ent["raw"])
    return out


class StreamedBatch:
    """Rollout ingestion UNDER sampling (second half): the sampler hands over chunks of finished trajectories in
    episode order (mjrl_amd/samplers.py; any producer may: ``begin(total_episodes)``, ``add(chunk, T)`` ..., then
    ``finish(paths)`` with the complete list), every chunk's rewards / observations / actions are gathered into the page-locked
    staging blocks by libmjx's host threads and their transfers are queued at once -- when the last episode ends, the batch is
    resident.  ``finish`` REGISTERS the blocks exactly as ``stage_shared`` would have for that list (same stagers, same modes:
    raw fp64 where some consumer reads the rollouts in their own precision, fp32 converted by the gather otherwise), so everything
    downstream -- compute_returns, the baselines, train_from_paths -- finds its block by the identity of the list and its arrays
    and uploads nothing.  Rows are placed in episode order: the blocks are bit for bit the ones a stage-after-sampling builds
    (tests/test_gpu_operators.py::test_streamed_ingestion_equals_staging_after_sampling).

    Nothing here is load-bearing for correctness: whatever does not fit the scheme (ragged dtypes, a chunk beyond the capacity
    bound total_episodes x T, a retried request, a consumer that rebuilds the list) aborts the stream and the batch is staged
    after sampling as before."""
    KEYS = ("rewards", "observations", "actions")

    def __init__(self, backend, keys=None):
        self.backend, self.dev = backend, backend.device
        self.keys = tuple(keys) if keys is not None else self.KEYS
        self.ok = backend.device.type == "cuda" and getattr(backend, "lib", None) is not None
        self.why = None if self.ok else "no GPU / libmjx"
        self.total, self.st, self.mode, self.seen, self.rows, self.cap = None, None, {}, [], 0, 0
        self.chunks = 0

    @classmethod
    def for_current_device(cls, keys=None):
        """-> a StreamedBatch on torch's current CUDA device, or None (no GPU, MJX_STREAM_INGEST=0)"""
        if os.environ.get("MJX_STREAM_INGEST", "1") == "0":
            return None
        try:
            import torch
            from .. import _lib
            if not torch.cuda.is_available():
                return None
            return cls(DeviceHandle(torch, torch.device("cuda", torch.cuda.current_device()), _lib.load()), keys)
        except Exception:                         # pragma: no cover
            return None

    def begin(self, total_episodes):
        self.total = int(total_episodes)

    def abort(self, why):
        if self.ok:
            self.ok, self.why = False, str(why)
        if self.st:
            for st in self.st.values():          # (a staging job must not outlive this attempt; the blocks are reused by the next begin())
                try:
                    st.join()
                except Exception:                 # pragma: no cover
                    pass
        self.st, self.seen = None, []

    def _open(self, chunk, T):
        """first chunk: which keys stream (uniform float arrays), in which mode, with what capacity"""
        first = chunk[0]
        keys = []
        for k in self.keys:
            a = first.get(k)
            if not isinstance(a, np.ndarray) or a.ndim not in (1, 2) or a.dtype not in (np.float64, np.float32):
                continue
            if k == "rewards" and not (a.ndim == 1 and a.dtype == np.float64):     # (what process_samples._rewards_block stages; else its own route)
                continue
            keys.append(k)
        if not keys or self.total is None:
            return self.abort("nothing to stream")
        # capacity: total_episodes x horizon is the bound, but an env whose episodes end long before their horizon would make the
        # page-locked and device blocks many times the size of the data -- so 1.5 x (mean length of the first chunk) x episodes when
        # that is smaller; a later chunk that does not fit aborts the stream (the batch is then staged after sampling, as before)
        rows0 = sum(len(p["rewards"]) for p in chunk)
        est = int(1.5 * rows0 / max(len(chunk), 1) * int(self.total)) + 1
        self.cap = min(int(self.total) * int(max(1, min(int(T), 10 ** 7))), max(est, rows0))
        if self.cap > (1 << 31):
            return self.abort("capacity bound beyond 2^31 rows")
        dev = self.dev
        sentinel = object()
        _release_other_batches(dev, sentinel, keys)                # whatever batch is still registered under these keys is replaced
        self.st = {}
        with _SHARED_LOCK:
            reg = _SHARED.setdefault((dev.type, dev.index), {})
            for k in keys:
                as_raw = (k == "rewards") or ((dev.type, dev.index, k) in _RAW_STICKY)
                if k == "rewards":
                    _RAW_STICKY.add((dev.type, dev.index, k))
                ent = reg.get(k)
                skey = "stager" if as_raw else "stager32"
                st = ent.get(skey) if ent is not None else None
                self.st[k] = st if st is not None else PathStager(self.backend)
                self.mode[k] = skey
        for k in keys:
            a = first[k]
            self.st[k].begin((k,), [a.shape[1] if a.ndim == 2 else 1], [np.float64 if a.dtype == np.float64 else np.float32], self.cap,
                             hostcast=() if self.mode[k] == "stager" else True)

    def add(self, chunk, T=None):
        """the next episodes of the batch (in order); T: an upper bound of a trajectory's length (the env's horizon)"""
        if not self.ok or not chunk:
            return
        try:
            if self.st is None:
                self._open(chunk, T if T is not None else max(len(p["rewards"]) for p in chunk))
                if not self.ok:
                    return
            rows = sum(len(p["rewards"]) for p in chunk)
            if self.rows + rows > self.cap:
                return self.abort("a chunk beyond the capacity bound (trajectories longer than the horizon the sampler named, or much longer than the first chunk's)")
            for k, st in self.st.items():
                st.add_paths(chunk)                                 # native gather into the page-locked block + queued copies
            self.rows += rows
            self.seen.extend(chunk)
            self.chunks += 1
        except Exception as e:                                      # anything irregular: stage after sampling, as before
            self.abort("%s: %s" % (type(e).__name__, e))

    def finish(self, paths):
        """`paths`: the complete list the sampler returned.  Registers the resident blocks for it -> True; False when the stream
        was aborted or `paths` is not what was streamed (then nothing is registered and the batch is staged on first use)."""
        if not self.ok or self.st is None:
            return False
        if type(paths) is not list or len(paths) != len(self.seen) or any(p is not q for p, q in zip(paths, self.seen)):
            self.abort("the sampler's list is not the streamed episodes")
            return False
        dev, torch = self.dev, self.backend.torch
        try:
            for k, st in self.st.items():
                st._drain(block=True)
                st.join()
                if st._rows != self.rows:
                    raise ValueError("row count of %r" % k)
                arrays = [p[k] for p in paths]
                ready = torch.cuda.Event()
                ready.record(st.side)
                with _SHARED_LOCK:
                    reg = _SHARED.setdefault((dev.type, dev.index), {})
                    old = reg.get(k)
                    new = dict(stager=old.get("stager") if old else None, stager32=old.get("stager32") if old else None,
                               f32=st._slots[k]["dev_f32"][:self.rows], raw=st.raw(k), paths=paths, arrays=arrays, probes=_probes(arrays), ready=ready)
                    new[self.mode[k]] = new["active"] = st
                    reg[k] = new
        except Exception as e:
            self.abort("%s: %s" % (type(e).__name__, e))
            return False
        self.seen = []
        return True


_PREFETCH_POOL = None


def prefetch(backend, paths, keys, raw=()):
    """start staging paths[.][key] for `keys` on a helper thread (native gather threads + asynchronous copies on the
    stagers' side streams, no GIL for the bulk of it) and return at once: whoever asks for those blocks later in the
    iteration (stage_shared / lookup) finds them staged or waits for exactly the block it needs.  The helper adopts
    the CALLER's device and stream (both are thread-local in torch).  Failures are left to the foreground request."""
    global _PREFETCH_POOL
    torch, dev = backend.torch, backend.device
    keys = tuple(k for k in keys if k in paths[0] and isinstance(paths[0][k], np.ndarray))
    if dev.type != "cuda" or not keys:
        return None
    if getattr(backend, "lib", None) is not None and _pathwalk is not None and all(collect_arrays(paths, k) is not None for k in keys):
        # uniform float blocks: one native asynchronous staging job per key, started right here (r04) -- whoever asks for a block
        # later joins its job (_order_after); no Python thread, no interpreter lock while the rollouts move
        if _prefetch_inline(backend, paths, keys, raw):
            return None
    if _PREFETCH_POOL is None:
        _PREFETCH_POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mjx-prefetch")
    cur = torch.cuda.current_stream(dev)
    trust = carried_trust()

    def run():
        try:
            with torch.cuda.device(dev), torch.cuda.stream(cur), trust():
                stage_shared(backend, paths, keys, raw=raw)     # (keys with a raw consumer earlier in the process go up raw: _RAW_STICKY)
        except Exception:                     # pragma: no cover
            pass
    return _PREFETCH_POOL.submit(run)


class DeviceHandle:
    """what the registry needs from a backend: torch, the device, libmjx (the update engine's HipBackend has them too)"""

    def __init__(self, torch, device, lib):
        self.torch, self.device, self.lib = torch, device, lib


def publish(backend, paths, key, raw, arrays):
    """Register a per-timestep block that was COMPUTED on the device (returns, baseline values, advantages:
    utils/process_samples.py) under paths[i][key] = arrays[i] -- the host copies just handed to the paths.  Whoever
    needs that block on the device later in the iteration (the advantage whitening of process_paths, the baseline fit)
    finds it with lookup() instead of concatenating and uploading the host arrays again; the identity / probe check of
    _same_batch protects against paths whose arrays were replaced or edited in between."""
    dev = backend.device
    with _SHARED_LOCK:
        reg = _SHARED.setdefault((dev.type, dev.index), {})
        old = reg.get(key)
        reg[key] = dict(stager=old.get("stager") if old else None, stager32=old.get("stager32") if old else None, active=None,
                        f32=None, raw=raw, paths=paths, arrays=list(arrays), probes=_probes(arrays))


def lookup(backend, paths, key):
    """the device tensor registered for paths[.][key] (stage_shared upload or publish), or None"""
    dev = backend.device
    with _key_lock(dev, key):
        with _SHARED_LOCK:
            ent = _SHARED.get((dev.type, dev.index), {}).get(key)
        if ent is None or ent.get("paths") is None or not _same_batch(ent, paths, key):
            return None
        _order_after(backend, ent)
        return ent["raw"]


def host_block(backend, paths, key):
    """the page-locked HOST copy of a block staged by stage_shared -- the per-path arrays back to back, (N, w) in the
    paths' dtype -- or None when `paths` is not the staged batch.  Valid until the next batch is staged under `key`;
    lets host-side reductions run vectorised over one contiguous block instead of path by path."""
    dev = backend.device
    with _key_lock(dev, key):
        with _SHARED_LOCK:
            ent = _SHARED.get((dev.type, dev.index), {}).get(key)
        if ent is None or ent.get("paths") is None or ent.get("active") is None or ent.get("f32") is None or not _same_batch(ent, paths, key):
            return None
        st = ent["active"]
        st.join()
        if st._slots[key]["hostcast"]:                 # staged as fp32: no host copy in the paths' dtype
            return None
        return st._slots[key]["pin_np"][:st._rows]


def derived(backend, paths, anchor_key, name, build):
    """a block derived from the batch (e.g. the within-trajectory time index, the trajectory offsets) cached next to
    the staged / published block `anchor_key` of the same batch: build() runs once per batch"""
    dev = backend.device
    with _key_lock(dev, anchor_key):
        with _SHARED_LOCK:
            ent = _SHARED.get((dev.type, dev.index), {}).get(anchor_key)
        hit = ent is not None and ent.get("paths") is not None and _same_batch(ent, paths, anchor_key)
        if hit and name in ent.setdefault("derived", {}):
            return ent["derived"][name]
    val = build()
    if hit:
        with _SHARED_LOCK:
            ent.setdefault("derived", {})[name] = val
    return val


def drop_shared_batch():
    """forget WHICH batch is staged (the stagers and their page-locked blocks stay): the next stage_shared uploads
    again whatever it is given, and the host trajectories of the finished iteration are released."""
    _retire_deferred()
    with _SHARED_LOCK:
        for reg in _SHARED.values():
            for ent in reg.values():
                ent["paths"] = ent["arrays"] = ent["probes"] = None
                ent["f32"] = ent["raw"] = ent["active"] = None
                ent.pop("derived", None)


def drop_shared():
    """forget the staged batches (tests; releasing device memory)"""
    with _SHARED_LOCK:
        for reg in _SHARED.values():
            for ent in reg.values():
                for sk in ("stager", "stager32"):
                    if ent.get(sk) is not None:
                        ent[sk].close()
        _SHARED.clear()
        _RAW_STICKY.clear()
