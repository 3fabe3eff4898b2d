"""Device-side state of the policy-update hot path.

``UpdateEngine`` owns one ``mjx_ctx`` (include/mjx.h) plus the torch tensors whose device
pointers the C ABI reads and writes.  torch is plumbing only: allocation, the current HIP
stream and ``torch.distributed`` (backend "nccl" == RCCL) for the one all-reduce per
gradient / per CG iteration.  No torch.autograd, no torch math on the hot path.

Multi-GPU: one process per GPU; every rank binds its own trajectory shard and the
quantities that are sums over samples (gradient, Fisher-vector product, surrogate / KL
sums) are all-reduced; CG scalars are recomputed redundantly on every rank from the reduced
vectors, so ranks stay in lock-step without further communication.
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import check, ptr


def _dist():
    """torch.distributed when this process is one rank of several, else None.  MJX_COLLECTIVES_AT_WORLD1=1
    keeps the collectives in the path for a 1-rank group (the RCCL rehearsal of bench.py --rehearse-world)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (
            dist.get_world_size() > 1 or os.environ.get("MJX_COLLECTIVES_AT_WORLD1") == "1"):
        return dist
    return None


import contextlib


@contextlib.contextmanager
def _stdout_to_stderr():
    """file descriptor 1 -> 2 for the duration (native libraries that print banners to stdout; a caller's stdout may be a
    machine-readable stream, e.g. bench.py's single JSON line)"""
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


class HipBackend:
    """Local (this-rank) compute: thin ctypes shim over libmjx.so.  All arguments are torch tensors
    living on ``self.device``; nothing here communicates or synchronises."""

    def __init__(self, n, m, hidden_sizes, device=None):
        import torch
        self.torch = torch
        self.lib = _lib.load()
        if self.lib.mjx_device_count() < 1 or not torch.cuda.is_available():
            raise _lib.MjxError("mjrl_amd needs an AMD GPU (gfx950): no HIP device visible and there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        ctx = ctypes.c_void_p()
        hidden = tuple(int(h) for h in hidden_sizes)
        harr = (ctypes.c_int * max(1, len(hidden)))(*hidden)
        check(self.lib.mjx_create(ctypes.byref(ctx), self.device.index or 0, int(n), int(m), harr, len(hidden)))
        self.ctx = ctx
        self.d = int(self.lib.mjx_num_params(ctx))
        self.fused = bool(self.lib.mjx_uses_fused_path(ctx))

    def close(self):
        if getattr(self, "ctx", None) is not None:
            self.torch.cuda.synchronize(self.device)
            self.lib.mjx_destroy(self.ctx)
            self.ctx = None

    def stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def upload_f32(self, a):
        """host ndarray (fp64 or fp32) -> fp32 device tensor.  fp64 input is cast on the device
        (one PCIe pass of the raw fp64 rollouts; the reference re-casts on the CPU at every
        call, gaussian_mlp.py:102-109)."""
        torch = self.torch
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.float32).contiguous()
        from .utils.ingest import upload
        t = upload(self, a)                          # page-locked bounce buffer: pageable uploads crawl on this stack
        if t.dtype == torch.float64:
            out = torch.empty(t.shape, dtype=torch.float32, device=self.device)
            check(self.lib.mjx_cast_f64_f32(ptr(t), t.numel(), ptr(out), self.stream()))
            return out
        return t.to(torch.float32)

    def bind_policy(self, theta_new, theta_old, tr_new, tr_old, old_is_new):
        check(self.lib.mjx_bind_policy(self.ctx, ptr(theta_new), ptr(theta_old), ptr(tr_new), ptr(tr_old), int(old_is_new)))

    def bind_batch(self, obs, act, adv, rows, N_global):
        check(self.lib.mjx_bind_batch(self.ctx, ptr(obs), ptr(act), ptr(adv), int(rows), int(N_global)))

    def bind_rows(self, rows, N_global, adv=None):
        check(self.lib.mjx_bind_rows(self.ctx, int(rows), int(N_global), ptr(adv)))

    def surr_vpg(self, grad_out, scal_out):
        check(self.lib.mjx_surr_vpg(self.ctx, ptr(grad_out), ptr(scal_out), self.stream()))

    def fvp(self, v, out):
        check(self.lib.mjx_fvp(self.ctx, ptr(v), ptr(out), self.stream()))

    def eval_surr_kl(self, scal_out):
        check(self.lib.mjx_eval_surr_kl(self.ctx, ptr(scal_out), self.stream()))

    def cg_solve_local(self, b, iters, damping, tol, x_out, bdotx_out):
        check(self.lib.mjx_cg_solve(self.ctx, ptr(b), int(iters), float(damping), float(tol), ptr(x_out), ptr(bdotx_out),
                                    None, None, self.stream()))

    def cg_init(self, b):
        check(self.lib.mjx_cg_init(self.ctx, ptr(b), self.stream()))

    def fvp_of_cg_direction(self, out):
        check(self.lib.mjx_fvp(self.ctx, ctypes.c_void_p(self.lib.mjx_cg_p(self.ctx)), ptr(out), self.stream()))

    def cg_step(self, Ap, damping, tol):
        check(self.lib.mjx_cg_step(self.ctx, ptr(Ap), float(damping), float(tol), self.stream()))

    def cg_finish(self, b, x_out, bdotx_out):
        check(self.lib.mjx_cg_finish(self.ctx, ptr(b), ptr(x_out), ptr(bdotx_out), self.stream()))

    # ---- multi-rank (RCCL inside libmjx, include/mjx.h "multi-rank")
    def comm_unique_id(self):
        buf = ctypes.create_string_buffer(128)
        check(self.lib.mjx_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank, world, uid):
        check(self.lib.mjx_comm_init(self.ctx, int(rank), int(world), ctypes.create_string_buffer(bytes(uid), 128)))

    def comm_world(self):
        return int(self.lib.mjx_comm_world(self.ctx))

    def comm_set_callback(self, dist, world):
        """rank sums through `dist.all_reduce` (any torch.distributed backend) as libmjx's transport hook: the C loops
        of mjx_cg_solve / mjx_npg_update call back between their launches.  For process groups that are not RCCL (the
        tests put two gloo ranks on one GPU); every call costs a stream synchronisation."""
        torch, dev = self.torch, self.device

        class _View:                                  # zero-copy tensor over a device pointer
            def __init__(self, p, count, typestr):
                self.__cuda_array_interface__ = dict(shape=(int(count),), typestr=typestr, data=(int(p), False), version=2)

        def hook(user, buf, count, dtype, stream):
            try:
                torch.cuda.current_stream(dev).synchronize()
                t = torch.as_tensor(_View(buf, count, "<f8" if dtype else "<f4"), device=dev)
                dist.all_reduce(t)
                torch.cuda.current_stream(dev).synchronize()
                return 0
            except Exception:                         # pragma: no cover
                import traceback
                traceback.print_exc()
                return -5
        self._hook = _lib.REDUCE_FN(hook)             # keep the trampoline alive as long as the context
        check(self.lib.mjx_comm_set_callback(self.ctx, self._hook, None, int(world)))

    def peer_connect_all(self, dist):
        """the peer exchange of libmjx (include/mjx.h "Peer exchange": HIP IPC buffers + stream-ordered flags) as the rank-sum
        transport: every rank exports its buffer, the handles travel through the process group once"""
        rank, world = dist.get_rank(), dist.get_world_size()
        loop = int(os.environ.get("MJX_PEER_LOOPBACK_WORLD", "0"))
        if world == 1 and loop > 1:               # bench.py --rehearse-world: rank 0 of `loop`, every peer mapped onto its own buffer
            buf = ctypes.create_string_buffer(64)
            check(self.lib.mjx_peer_export(self.ctx, 0, loop, buf))
            check(self.lib.mjx_peer_connect(self.ctx, None))
            return
        buf = ctypes.create_string_buffer(64)
        rc = self.lib.mjx_peer_export(self.ctx, int(rank), int(world), buf)
        # every rank takes part in the gather whatever its own export did (a rank that skipped it would leave the others blocked
        # in the collective): a failed export travels as an empty handle and fails the connect on ALL ranks
        handles = [None] * world
        dist.all_gather_object(handles, buf.raw if rc == 0 else b"")
        check(rc)
        if any(len(h) != 64 for h in handles):
            raise _lib.MjxError("peer exchange: rank(s) %s could not export their buffer" % [r for r, h in enumerate(handles) if len(h) != 64])
        check(self.lib.mjx_peer_connect(self.ctx, ctypes.create_string_buffer(b"".join(handles), 64 * world)))

    def peer_timeouts(self):
        """number of peer-exchange waits that gave up since the last call (include/mjx.h mjx_peer_status)"""
        n = ctypes.c_int(0)
        check(self.lib.mjx_peer_status(self.ctx, ctypes.byref(n)))
        return int(n.value)

    def allreduce(self, t):
        """in-place sum over the ranks on the launch stream (fp32 / fp64 tensors)"""
        check(self.lib.mjx_comm_allreduce(self.ctx, ptr(t), t.numel(), 1 if t.dtype == self.torch.float64 else 0, self.stream()))

    def npg_update(self, iters, damping, tol, step_size, const_alpha, min_log_std, grad_out, x_out, theta_out, results):
        check(self.lib.mjx_npg_update(self.ctx, int(iters), float(damping), float(tol), float(step_size), float("nan") if const_alpha is None else float(const_alpha),
                                      float(min_log_std), ptr(grad_out), ptr(x_out), ptr(theta_out), ptr(results), self.stream()))

    def trpo_update(self, iters, damping, tol, step_size, kl_dist, n_trials, first, min_log_std, grad_out, x_out, theta_out, results):
        check(self.lib.mjx_trpo_update(self.ctx, int(iters), float(damping), float(tol), float(step_size), float(kl_dist), int(n_trials),
                                       1 if first else 0, float(min_log_std), ptr(grad_out), ptr(x_out), ptr(theta_out), ptr(results),
                                       self.stream()))

    def dapg_update(self, iters, damping, tol, step_size, min_log_std, rows_on, N_on_global, adv_on, grad_out, x_out, theta_out, results):
        check(self.lib.mjx_dapg_update(self.ctx, int(iters), float(damping), float(tol), float(step_size), float(min_log_std),
                                       int(rows_on), int(N_on_global), ptr(adv_on), ptr(grad_out), ptr(x_out), ptr(theta_out),
                                       ptr(results), self.stream()))

    def apply_step(self, base, x, alpha, min_log_std, out):
        check(self.lib.mjx_apply_step(self.ctx, ptr(base), ptr(x), float(alpha), float(min_log_std), ptr(out), self.stream()))

    def apply_npg_step(self, base, x, gdotx, step_size, min_log_std, out, alpha_out):
        check(self.lib.mjx_apply_npg_step(self.ctx, ptr(base), ptr(x), ptr(gdotx), float(step_size), float(min_log_std), ptr(out),
                                          ptr(alpha_out), self.stream()))


class UpdateEngine:
    """Orchestration of one policy update over (possibly) several ranks: owns the tensors, binds
    batch / policy, and places the all-reduces.  ``backend`` does this rank's arithmetic
    (``HipBackend`` in production; the multi-rank logic is exercised on CPU/gloo in
    tests/test_distributed_gloo.py with an oracle-backed stand-in)."""

    def __init__(self, n, m, hidden_sizes, device=None, backend=None):
        self.n, self.m, self.hidden = int(n), int(m), tuple(int(h) for h in hidden_sizes)
        self.backend = backend if backend is not None else HipBackend(n, m, hidden_sizes, device)
        self.torch = torch = self.backend.torch
        self.device = self.backend.device
        self.d = self.backend.d
        self.fused = getattr(self.backend, "fused", False)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.theta_new = torch.zeros(self.d, **f32)
        self.theta_old = torch.zeros(self.d, **f32)
        self.tr_new = torch.zeros(2 * self.n + 2 * self.m, **f32)
        self.tr_old = torch.zeros(2 * self.n + 2 * self.m, **f32)
        self.grad = torch.zeros(self.d, **f32)
        self.x = torch.zeros(self.d, **f32)
        self.Ap = torch.zeros(self.d, **f32)
        # every scalar an update produces sits in one device block, so that one read-back fetches them all:
        # K3's sums | K1's sums (kept apart from K3's) | b.x of the last solve | step length formed on the device
        self.results = torch.zeros(64, dtype=torch.float64, device=self.device)      # ([16:] the per-trial log of mjx_trpo_update)
        self.scal, self.scal_vpg = self.results[0:4], self.results[4:8]
        self.bdotx, self.alpha_dev = self.results[8:9], self.results[9:10]
        self._host_results = None                   # host copy of `results`, valid until the next launch that writes it
        self.obs = self.act = self.adv = None
        self.N_local = self.N_global = self.N_bound = 0
        self._block = self._prefix = None
        self.old_is_new = True
        self._dbg = None
        self.comm_kind = None                       # "rccl" | "peer" | "hook" once _native_comm() has attached a transport
        self._comm_state = None                     # None: not tried yet; True: libmjx holds an RCCL communicator; False: torch collectives

    # convenience handles used by bench / tests
    @property
    def lib(self):
        return self.backend.lib

    @property
    def ctx(self):
        return self.backend.ctx

    def stream(self):
        return self.backend.stream()

    def close(self):
        if getattr(self, "_stager", None) is not None:
            self._stager.close()
            self._stager = None
        if getattr(self, "backend", None) is not None:
            self.backend.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def to_device_f32(self, a):
        return self.backend.upload_f32(a)

    def to_host(self, t):
        """device tensor -> fresh ndarray through a page-locked bounce buffer (utils/ingest.download: no pageable hipMemcpy)"""
        from .utils.ingest import download
        return download(self.backend, t)

    # ------------------------------------------------------------------ binding
    def set_policy(self, theta_new, theta_old, tr_new, tr_old):
        """flat fp32 parameter vectors + packed transforms (host ndarrays)."""
        torch = self.torch
        tn, to = np.asarray(theta_new, np.float32), np.asarray(theta_old, np.float32)
        trn, tro = np.asarray(tr_new, np.float32), np.asarray(tr_old, np.float32)
        self.theta_new.copy_(torch.from_numpy(tn))
        self.theta_old.copy_(torch.from_numpy(to))
        self.tr_new.copy_(torch.from_numpy(trn))
        self.tr_old.copy_(torch.from_numpy(tro))
        self.old_is_new = bool(np.array_equal(tn, to) and np.array_equal(trn, tro))
        self._bind_policy()

    def _bind_policy(self):
        self.backend.bind_policy(self.theta_new, self.theta_old, self.tr_new, self.tr_old, self.old_is_new)

    def global_count(self, local_count):
        d = _dist()
        if d is None:
            return int(local_count)
        t = self.torch.tensor([float(local_count)], dtype=self.torch.float64, device=self.device)
        d.all_reduce(t)
        return int(round(t.item()))

    def set_batch(self, obs, act=None, adv=None, N_global=None):
        """obs (N,n), act (N,m), adv (N,) of THIS rank's shard; host arrays or device tensors."""
        self.obs = self.to_device_f32(obs)
        self.act = None if act is None else self.to_device_f32(act)
        self.adv = None if adv is None else self.to_device_f32(adv)
        self.N_local = int(self.obs.shape[0])
        self.N_global = self.global_count(self.N_local) if N_global is None else int(N_global)
        self.N_bound = self.N_local                  # rows the kernels currently run over (bind_rows narrows it)
        self._block = (self.obs, self.act, self.adv, self.N_local, int(self.N_global))
        self._prefix = None
        self.backend.bind_batch(self.obs, self.act, self.adv, self.N_local, int(self.N_global))

    def whitened_advantages(self, adv64):
        """(adv - mean) / (std + 1e-6) of process_paths (batch_reinforce.py:185) for an fp64 advantage block that is
        already on the device (utils/process_samples left it there): population mean / std over ALL ranks by two
        reduction passes (mean, then squared deviations about it -- NumPy's algorithm), cast to fp32 on the way out."""
        torch, lib = self.torch, self.lib
        N = int(adv64.numel())
        st = torch.zeros(3, dtype=torch.float64, device=self.device)
        check(lib.mjx_sum_stats(ptr(adv64), N, 0.0, ptr(st), self.stream()))
        self._rank_sum(st)
        s = st.cpu().numpy()
        mean = float(s[0] / s[2])
        check(lib.mjx_sum_stats(ptr(adv64), N, mean, ptr(st), self.stream()))
        self._rank_sum(st)
        s = st.cpu().numpy()
        std = float(np.sqrt(s[1] / s[2]))
        out = torch.empty(N, dtype=torch.float32, device=self.device)
        check(lib.mjx_whiten_cast(ptr(adv64), N, mean, std, 1e-6, ptr(out), self.stream()))
        return out

    def stage_paths(self, paths, keys=("observations", "actions"), defer=False):
        """per-path host arrays -> fp32 device blocks through the page-locked stager (utils/ingest.py): no
        concatenated host copy, chunked transfers overlapped with the staging copies.  One upload per batch and
        process: the value baselines (predict before, fit after the update) share it (utils/ingest.stage_shared)."""
        if getattr(self, "_stager", None) is not None:             # a caller-supplied stager (tools, tests)
            return self._stager.stage(paths, keys, hostcast=True)
        from .utils.ingest import stage_shared
        # defer: the blocks are handed out while libmjx's staging threads are still at work (mjx_stage_async); the caller runs
        # utils.ingest.settle() before it enqueues the first kernel that reads them
        return {k: v["f32"] for k, v in stage_shared(self.backend, paths, keys, raw=(), defer=defer).items()}

    def bind_rows(self, rows, adv=None, N_global=None):
        """(re)bind the first `rows` samples of the uploaded block (DAPG runs the Fisher on the
        on-policy prefix of the demo-augmented block, dapg.py:97-103)."""
        if adv is not None:
            self.adv = self.to_device_f32(adv)
        if N_global is not None:
            self.N_global = int(N_global)
        self.N_bound = int(rows)
        self._prefix = (int(rows), int(self.N_global), self.adv)
        self.backend.bind_rows(int(rows), int(self.N_global), self.adv)

    def rebind(self):
        """bind the uploaded block again, then the row prefix that was in force (after a caller bound something else on
        the backend, e.g. the row samples of hvp_sample_frac)."""
        obs, act, adv, rows, Ng = self._block
        self.backend.bind_batch(obs, act, adv, rows, Ng)
        if self._prefix is not None:
            self.backend.bind_rows(*self._prefix)

    # ------------------------------------------------------------------ collectives
    def _native_comm(self):
        """True when the rank sums run inside libmjx (include/mjx.h), decided once.  torch.distributed on the nccl (= RCCL)
        backend -- every rank owns its GPU: ncclAllReduce on the launch stream, the communicator id made by rank 0 and
        broadcast through the process group; with MJX_PEER_COMM=1 libmjx's peer exchange instead (HIP IPC buffers, the
        handles gathered through the process group).  Any other backend (gloo ranks sharing a GPU): the peer exchange, or
        with MJX_PEER_COMM=0 the transport hook over dist.all_reduce.  ``comm_kind`` names the choice.  Any failure
        leaves the torch.distributed collectives in place (MJX_NATIVE_COMM=0 forces that)."""
        if self._comm_state is not None:
            return self._comm_state
        d = _dist()
        ok = False
        if d is not None and hasattr(self.backend, "comm_init") and os.environ.get("MJX_NATIVE_COMM", "1") != "0":
            # MJX_PEER_COMM=1: libmjx's own peer exchange (HIP IPC + in-kernel flags) instead of RCCL; it is also what a process
            # group that is not RCCL gets (gloo ranks sharing one GPU in the tests; MJX_PEER_COMM=0: the host-side hook)
            peer = os.environ.get("MJX_PEER_COMM")
            multi = d.get_world_size() > 1
            if (multi or int(os.environ.get("MJX_PEER_LOOPBACK_WORLD", "0")) > 1) and (
                    peer == "1" or (d.get_backend() != "nccl" and peer != "0")):
                order = ["peer"] + ([] if d.get_backend() == "nccl" else ["hook"])
            elif d.get_backend() == "nccl":
                order = ["rccl", "hook"]         # (a communicator libmjx cannot create itself: the same loops over torch's own RCCL group)
            else:
                order = ["hook"]
            if os.environ.get("MJX_TRANSPORT_ORDER"):     # (tests: walk a given chain whatever the process group is)
                order = [k for k in os.environ["MJX_TRANSPORT_ORDER"].split(",") if k in ("rccl", "peer", "hook")]
            # every rank walks the same list and the ranks agree after each attempt (all or none): RCCL inside libmjx and the
            # one-call loops on some ranks, torch.distributed calls on others would issue different collectives and hang
            for kind in order:
                mine = self._attach_transport(kind, d)
                if self._all_ranks(d, mine):
                    # attached everywhere -- now a known-answer rank sum through it, before any update trusts it: a transport
                    # that delivers finite but wrong sums (a mis-mapped peer buffer, a flag that overtakes its data) would
                    # otherwise be silent
                    self.comm_kind = kind
                    if self._all_ranks(d, self._transport_self_test(d, kind)):
                        ok = True
                        break
                self._detach_transport()
        self._comm_state = ok
        return ok

    def _attach_transport(self, kind, d):
        try:
            if kind == "peer":
                self.backend.peer_connect_all(d)
            elif kind == "rccl":
                # rank 0 makes the communicator id.  If it cannot (librccl missing, MJX_RCCL_DISABLE=1) it still takes part in
                # the broadcast -- with None -- so that every rank leaves this attempt through the same collectives (ADVICE r04:
                # raising before the broadcast left the others blocked in it while rank 0 went on to the agreement round)
                box, err = [None], None
                if d.get_rank() == 0:
                    try:
                        box = [self.backend.comm_unique_id()]
                    except Exception as e:
                        err = e
                if d.get_world_size() > 1:
                    d.broadcast_object_list(box, src=0)
                if box[0] is None:
                    raise err if err is not None else _lib.MjxError("rank 0 could not create an RCCL communicator id")
                with _stdout_to_stderr():         # RCCL prints a version banner to stdout when a communicator is created
                    self.backend.comm_init(d.get_rank(), d.get_world_size(), box[0])
            else:                                # same C loops, transport hooked to dist.all_reduce (a host synchronisation per sum)
                self.backend.comm_set_callback(d, d.get_world_size())
            return True
        except Exception as e:                   # pragma: no cover - depends on the host's RCCL / IPC support
            import warnings
            warnings.warn("mjrl_amd: rank sums inside libmjx over '%s' unavailable (%s)" % (kind, e))
            return False

    def _transport_self_test(self, d, kind):
        """One known-answer rank sum of d floats + 4 doubles through the freshly attached transport, twice (the peer exchange
        alternates between two slot sets): rank r contributes (r + 1) x pattern, every term and every partial sum exactly
        representable, so the result must equal the closed form BIT FOR BIT whatever the order; then the ranks compare a
        digest of what they received through the side channel (torch.distributed).  -> this rank's verdict."""
        W, r = d.get_world_size(), d.get_rank()
        if W <= 1 or os.environ.get("MJX_TRANSPORT_SELF_TEST", "1") == "0":
            return True
        torch = self.torch
        try:
            base = ((torch.arange(self.d, dtype=torch.float64, device=self.device) % 251) - 125.0) / 8.0
            tri = W * (W + 1) / 2.0
            ok, digest = True, []
            for rep in range(2):
                v = ((r + 1) * (rep + 1) * base).to(torch.float32)
                s = torch.tensor([r + 1.0, (r + 1.0) ** 2, -0.5 * (r + 1), 1.0], dtype=torch.float64, device=self.device)
                self.backend.allreduce(v)
                self.backend.allreduce(s)
                want_s = torch.tensor([tri, W * (W + 1) * (2 * W + 1) / 6.0, -0.5 * tri, float(W)], dtype=torch.float64, device=self.device)
                ok = ok and bool(torch.equal(v, (tri * (rep + 1) * base).to(torch.float32))) and bool(torch.equal(s, want_s))
                digest.append(v.cpu().numpy().tobytes() + s.cpu().numpy().tobytes())
            import hashlib
            mine = hashlib.sha1(b"".join(digest)).hexdigest()
        except Exception as e:                   # pragma: no cover - a transport that errors out is simply not trusted
            ok, mine = False, "error: %s" % e
        every = [None] * W
        d.all_gather_object(every, mine)
        ok = ok and all(h == every[0] for h in every)
        if not ok:
            import warnings
            warnings.warn("mjrl_amd: the rank-sum transport '%s' failed its known-answer test on rank %d; trying the next one" % (kind, r))
        return ok

    def _all_ranks(self, d, flag):
        if d.get_world_size() <= 1:
            return bool(flag)
        t = self.torch.tensor([1 if flag else 0], dtype=self.torch.int32, device=self.device if d.get_backend() == "nccl" else "cpu")
        d.all_reduce(t, op=d.ReduceOp.MIN)
        return int(t.item()) == 1

    def _detach_transport(self):
        """undo a transport some rank could not attach (buffers, IPC maps, hook): the next candidate starts clean"""
        self.comm_kind = None
        try:
            check(self.lib.mjx_comm_destroy(self.ctx))
            check(self.lib.mjx_comm_set_callback(self.ctx, _lib.REDUCE_FN(0), None, 0))
        except Exception:                        # pragma: no cover
            pass

    def _rank_sum(self, *tensors):
        """sum device tensors over the ranks in place (no-op for a single process)"""
        d = _dist()
        if d is None:
            return
        if self._native_comm():
            for t in tensors:
                self.backend.allreduce(t)
        else:
            for t in tensors:
                d.all_reduce(t)

    # ------------------------------------------------------------------ kernels + collectives
    def surr_vpg(self, sync=True):
        """K1 -> (grad device tensor, surrogate float).  flat_vpg + CPI_surrogate
        (batch_reinforce.py:40-58).  sync=False: nothing is read back (see deferred())."""
        self._host_results = None
        self.backend.surr_vpg(self.grad, self.scal_vpg)
        self._rank_sum(self.grad, self.scal_vpg)
        if not sync:
            return self.grad, None                   # the surrogate stays on the device: read it with deferred()
        s = self.scal_vpg.cpu().numpy()
        return self.grad, float(s[0] / self.N_global)

    def fvp(self, v, out=None):
        """K2: (H v) without damping, reduced over ranks (npg_cg.py:62-81)."""
        out = self.Ap if out is None else out
        self.backend.fvp(v, out)
        self._rank_sum(out)
        return out

    def cg_solve(self, b, iters, damping, tol=1e-10, sync=True):
        """K4: x = CG(H + damping I, b), x0 = 0 (cg_solve.py:3-22) -> (x device tensor, b.x).
        One all-reduce of the d-float Fisher-vector product per iteration; the vector updates
        and dot products are recomputed identically on every rank."""
        d = _dist()
        be = self.backend
        self._host_results = None
        if d is None or self._native_comm():
            # one C call: iters x (K2, [ncclAllReduce on the launch stream,] vector update), nothing in between
            be.cg_solve_local(b, iters, damping, tol, self.x, self.bdotx)
        else:
            be.cg_init(b)
            for _ in range(int(iters)):
                be.fvp_of_cg_direction(self.Ap)
                d.all_reduce(self.Ap)
                be.cg_step(self.Ap, damping, tol)
            be.cg_finish(b, self.x, self.bdotx)
        return self.x, (float(self.bdotx.item()) if sync else None)

    def apply_step(self, alpha, min_log_std, base=None):
        """theta_new <- base + alpha * x with the log_std clamp (npg_cg.py:137-139)."""
        base = self.theta_old if base is None else base
        self.backend.apply_step(base, self.x, alpha, min_log_std, self.theta_new)
        self.old_is_new = False
        self._bind_policy()

    def apply_npg_step(self, step_size, min_log_std, base=None):
        """theta_new <- base + sqrt(|step_size / (g.x + 1e-20)|) x with the step length formed on the device from the last
        solve (npg_cg.py:133-139): no read-back between the solve and the step."""
        base = self.theta_old if base is None else base
        self._host_results = None
        self.backend.apply_npg_step(base, self.x, self.bdotx, step_size, min_log_std, self.theta_new, self.alpha_dev)
        self.old_is_new = False
        self._bind_policy()

    def npg_update(self, iters, damping, step_size, min_log_std, const_alpha=None, tol=1e-10, enqueue_only=False):
        """The whole NPG update (npg_cg.py:108-142: K1, CG, step length, step, K3) enqueued by ONE call into libmjx
        (mjx_npg_update), rank sums included -> (surr_after, kl); deferred() has surr_before / g.x / alpha.
        Needs theta_new == theta_old at entry; the torch.distributed fallback path (no RCCL inside libmjx) issues the same
        sequence call by call.  enqueue_only (r06): return as soon as the update is ENQUEUED -- the caller does host work that
        does not depend on it (path statistics, log entries) under the 3.8 ms of device time and then asks npg_update_result()."""
        assert self.old_is_new, "npg_update starts from theta_new == theta_old"
        d = _dist()
        if (d is None or self._native_comm()) and hasattr(self.backend, "npg_update"):
            self._host_results = None
            self.backend.npg_update(iters, damping, tol, step_size, const_alpha, min_log_std, self.grad, self.x, self.theta_new, self.results)
            self.old_is_new = False
            self._npg_pending = True
            return None if enqueue_only else self.npg_update_result()
        g, _ = self.surr_vpg(sync=False)
        self.cg_solve(g, iters, damping, tol, sync=const_alpha is not None)
        if const_alpha is not None:
            self.apply_step(const_alpha, min_log_std)
        else:
            self.apply_npg_step(step_size, min_log_std)
        self._npg_pending = self.eval_surr_kl()
        return None if enqueue_only else self.npg_update_result()

    def npg_update_result(self):
        """-> (surr_after, kl) of the update npg_update(..., enqueue_only=True) started: the one read-back"""
        pend, self._npg_pending = getattr(self, "_npg_pending", None), None
        if pend is None:
            raise _lib.MjxError("npg_update_result() without an update in flight")
        if pend is not True:                              # (the call-by-call fallback has read its results already)
            return pend
        s = self._host_results = self._checked(self.results.cpu().numpy())
        return float(s[0] / self.N_global), float(s[1] / self.N_global)

    def dapg_update(self, iters, damping, step_size, min_log_std, rows_on, adv_on, N_on_global=None, tol=1e-10):
        """The whole DAPG update (dapg.py:92-121) through ONE call into libmjx (mjx_dapg_update), rank sums included: K1
        over the bound [on-policy ; demonstrations] block, gradient x N_all / N_on, the on-policy prefix bound with its
        whitened advantages `adv_on`, K3 (surr_before), CG, step length from step_size = 2 kl_dist, step, K3 ->
        (surr_after, kl); deferred() has surr_before / g.x / alpha.  None when the one-call path is not available
        (torch.distributed fallback): the caller issues the sequence itself."""
        assert self.old_is_new, "dapg_update starts from theta_new == theta_old"
        d = _dist()
        if not ((d is None or self._native_comm()) and hasattr(self.backend, "dapg_update")):
            return None
        adv_dev = self.to_device_f32(adv_on)
        N_on_global = self.global_count(rows_on) if N_on_global is None else int(N_on_global)
        self._host_results = None
        self.backend.dapg_update(iters, damping, tol, step_size, min_log_std, rows_on, N_on_global, adv_dev, self.grad, self.x,
                                 self.theta_new, self.results)
        self.adv = adv_dev
        self.N_global, self.N_bound = N_on_global, int(rows_on)
        self._prefix = (int(rows_on), N_on_global, adv_dev)
        self.old_is_new = False
        s = self._host_results = self._checked(self.results.cpu().numpy())
        return float(s[0] / self.N_global), float(s[1] / self.N_global)

    def trpo_update(self, iters, damping, step_size, kl_dist, min_log_std, tol=1e-10, batch=3, max_trials=100):
        """The whole TRPO update (trpo.py:100-126: K1, CG, step length, backtracking line search on the KL) through libmjx's
        mjx_trpo_update: the line-search trials are enqueued `batch` at a time with the accept / shrink decision taken on the
        device, one read-back per batch (the call-by-call form reads back after every trial) -> dict(alpha, trials, accepted,
        surr_after, kl, history=[(surr, kl) per trial]); deferred() has surr_before / g.x.  The search runs to the reference's
        100 trials on the device (batches of 3, 6, 12, 24, 24, ... trials per read-back); when none is accepted the zero step
        is applied and evaluated like the reference does (accepted = False, alpha = 0).  None when the one-call path is not
        available (torch.distributed fallback): the caller runs the loop itself."""
        assert self.old_is_new, "trpo_update starts from theta_new == theta_old"
        d = _dist()
        if not ((d is None or self._native_comm()) and hasattr(self.backend, "trpo_update")):
            return None
        first, hist, done = True, [], 0
        while True:
            self._host_results = None
            nb = max(1, min(int(batch), 24, max_trials - done))          # the per-trial log is a ring of 24 entries
            self.backend.trpo_update(iters, damping, tol, step_size, kl_dist, nb, first, min_log_std, self.grad, self.x, self.theta_new,
                                     self.results)
            self.old_is_new = False
            first = False
            s = self._host_results = self._checked(self.results.cpu().numpy(), fields=(4, 8))
            trials, accepted = int(s[11]), s[10] != 0.0
            for k in range(len(hist), trials):
                hist.append((float(s[16 + 2 * (k % 24)] / self.N_global), float(s[17 + 2 * (k % 24)] / self.N_global)))
            done = trials
            if accepted or trials >= max_trials:
                break
            batch = 2 * nb                               # a long search reads back less and less often: 3, 6, 12, 24, 24, ...
        if accepted:
            self._checked(s)
        alpha, surr_after, kl = float(s[9]), float(s[0] / self.N_global), float(s[1] / self.N_global)
        if not accepted:
            # trpo.py:119-126: after 100 rejected step lengths alpha = 0 -- the parameters stay, KL and surrogate are evaluated there
            alpha = 0.0
            self.apply_step(0.0, min_log_std)
            surr_after, kl = self.eval_surr_kl()
        return dict(alpha=alpha, trials=trials, accepted=bool(accepted), surr_after=surr_after, kl=kl, history=hist)

    def _checked(self, s, fields=(0, 1, 4, 8, 9), timeouts_only=False):
        """the host copy of `results` after an update's read-back: a non-finite surrogate / KL / g.x / step length must not reach
        policy.set_param_values silently.  The peer exchange turns a wait that timed out (a lost or late rank) into NaN
        (csrc/vecops.h peer_arrived): say so; anything else non-finite is reported as what it is.  `fields`: the entries that
        must be finite at this point (a REJECTED line-search trial may legitimately overflow: trpo_update checks K1's sums and
        g.x per batch of trials and the accepted trial's surrogate / KL at the end; eval_surr_kl -- called once per trial by the
        call-by-call line searches, trpo.py:107-120 / batch_reinforce.py:153-160 -- raises for a timed-out exchange only and
        otherwise hands the values on as they are, like the reference's `kl < kl_dist` on a NaN)."""
        if not np.all(np.isfinite(s[list(fields)])):
            peer = self.comm_kind == "peer" and hasattr(self.backend, "peer_timeouts")
            timeouts = self.backend.peer_timeouts() if peer else 0
            if timeouts or (peer and not timeouts_only):
                # a rank whose wait timed out poisons its sums with NaN, and the poison reaches every other rank through the very
                # next exchange: all live ranks arrive here at the same update -- the ones that waited in vain with `timeouts`, the
                # others without -- and ALL of them drop the transport (their exchange sequences are out of step from here on)
                self._comm_state, self.comm_kind = False, None
                try:
                    check(self.lib.mjx_comm_destroy(self.ctx))
                except Exception:                    # pragma: no cover
                    pass
                if timeouts:
                    raise _lib.MjxError("peer exchange: %d wait(s) for another rank's vector timed out (MJX_PEER_TIMEOUT_MS, default 5000); "
                                        "the update is invalid and the transport was torn down" % timeouts)
                raise _lib.MjxError("non-finite update results under the peer exchange (%s): another rank's wait timed out and its NaN "
                                    "arrived through the rank sums, or the update itself diverged; the transport was torn down" % (s[:10],))
            if not timeouts_only and os.environ.get("MJX_ALLOW_NONFINITE") != "1":
                # (the reference carries NaN into set_param_values without a word; MJX_ALLOW_NONFINITE=1 restores that -- INTEGRATION.md)
                raise _lib.MjxError("the policy update produced non-finite results (surrogate / KL / g.x / step length: %s)" % (s[:10],))
        return s

    def deferred(self):
        """-> dict(surr_before, gdotx, alpha) of the calls made with sync=False / apply_npg_step (one read-back after the update)"""
        r = self._host_results                       # eval_surr_kl() already fetched the block: no further round trip
        if r is None:
            r = self._host_results = self.results.cpu().numpy()
        return dict(surr_before=float(r[4] / self.N_global), gdotx=float(r[8]), alpha=float(r[9]))

    def eval_surr_kl(self):
        """K3 -> (surrogate, mean KL) (batch_reinforce.py:40-52)."""
        self.backend.eval_surr_kl(self.scal)
        self._rank_sum(self.scal)
        s = self._host_results = self._checked(self.results.cpu().numpy(), fields=(0, 1), timeouts_only=True)   # the whole block: deferred() needs no second read-back
        return float(s[0] / self.N_global), float(s[1] / self.N_global)

    def enable_debug(self):
        self._dbg = self.torch.zeros(2048 * 8, dtype=self.torch.float32, device=self.device)
        check(self.lib.mjx_set_debug_buffer(self.ctx, ptr(self._dbg), self._dbg.numel()))
        return self._dbg
