"""MLP value baseline trained with minibatch Adam on the GPU.

Mirrors mjrl/baselines/mlp_baseline.py:10-105 (+ utils/optimize_model.py:7-36): same constructor,
``fit(paths, return_errors)``, ``predict(path)``; the ReLU network (n+4) -> hidden -> 1 is initialised
through torch's nn.Linear so a given seed reproduces the reference's initial weights, and every epoch
draws its row permutation from NumPy's global RNG like ``fit_data`` does.  State (weights, Adam
moments, step count) is plain NumPy, so the object pickles / deep-copies like the reference's.
"""
import ctypes

import numpy as np

from .._lib import check, ptr
from ..utils import ingest, ranks
from ._features import DeviceBlock, DeviceOnly


def _permutation_into(lib, out):
    """out[:] = np.random.permutation(len(out)) -- the same values, the same advance of NumPy's global stream -- drawn by libmjx
    (mjx_host_mt19937_permutation: RandomState's algorithm on a copy of the generator state, straight into the int32 block;
    2.5 instead of 7.5 ms per 1M rows, and these draws are what is left of the fit on the critical path)"""
    n = int(out.shape[0])
    st = np.random.get_state()
    if n < 2 or st[0] != 'MT19937' or out.dtype != np.int32 or not out.flags.c_contiguous:
        out[:] = np.random.permutation(n)
        return
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = ctypes.c_int32(int(st[2]))
    check(lib.mjx_host_mt19937_permutation(ctypes.c_void_p(key.ctypes.data), ctypes.byref(pos), n, ctypes.c_void_p(out.ctypes.data)))
    np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))


def _permutations_into(lib, out, n, epochs):
    """out[e * n : (e + 1) * n] = np.random.permutation(n) for e in range(epochs), consecutively from NumPy's global stream -- one native
    call (mjx_host_mt19937_permutations: the generator on this thread, the swaps on a second one)"""
    n, epochs = int(n), int(epochs)
    st = np.random.get_state()
    if n < 2 or epochs < 1 or st[0] != 'MT19937' or out.dtype != np.int32 or not out.flags.c_contiguous or out.shape[0] < n * epochs:
        for ep in range(epochs):
            _permutation_into(lib, out[ep * n:(ep + 1) * n])
        return
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = ctypes.c_int32(int(st[2]))
    check(lib.mjx_host_mt19937_permutations(ctypes.c_void_p(key.ctypes.data), ctypes.byref(pos), n, epochs, ctypes.c_void_p(out.ctypes.data)))
    np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))


_FIT_STREAMS = {}            # device -> the side stream the asynchronous fits run on


def _fit_stream(torch, dev):
    key = (dev.type, dev.index)
    if key not in _FIT_STREAMS:
        _FIT_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _FIT_STREAMS[key]


class PendingFit:
    """One fit in flight on the side stream (MLPBaseline.fit_async): the device tensors it reads and writes, the page-locked
    blocks its results land in, and what to do with them once it is over.  ``result()`` waits for it (once), writes parameters /
    moments / losses into the baseline, and -> (error_before, error_after) when errors were asked for, else None."""

    def __init__(self, baseline):
        self.baseline, self.done, self.value = baseline, False, None
        self.keep, self.start_ev, self.end_ev = [], None, None
        self.returns = self.pred_before = self.pred_after = self.host = None
        self.hooks = []                                   # callables run with (errors, device_seconds) at settlement (deferred log entries)
        self.device_ms = None

    def finished(self):
        """has the side stream got through it? (never waits)"""
        return self.done or self.end_ev is None or bool(self.end_ev.query())

    def result(self):
        if not self.done:
            self.baseline._settle()
        return self.value


class MLPBaseline:
    _STATE = ("params", "adam_m", "adam_v", "adam_steps", "epoch_losses")

    def __init__(self, env_spec, inp_dim=None, inp='obs', learn_rate=1e-3, reg_coef=0.0, batch_size=64, epochs=1,
                 use_gpu=False, hidden_sizes=(128, 128)):
        self.n = inp_dim if inp_dim is not None else env_spec.observation_dim
        self.batch_size = batch_size
        self.epochs = epochs
        self.reg_coef = reg_coef
        self.learn_rate = learn_rate
        self.use_gpu = use_gpu          # accepted for compatibility; the kernels always run on the GPU
        self.inp = inp
        self.hidden_sizes = tuple(int(h) for h in hidden_sizes)
        import torch
        sizes = (self.n + 4,) + self.hidden_sizes + (1,)
        flat = []
        for i in range(len(sizes) - 1):               # same construction order as mlp_baseline.py:21-28
            lin = torch.nn.Linear(sizes[i], sizes[i + 1])
            flat += [lin.weight.detach().numpy().ravel(), lin.bias.detach().numpy().ravel()]
        self.params = np.concatenate(flat).astype(np.float32)
        self.adam_m = np.zeros_like(self.params)
        self.adam_v = np.zeros_like(self.params)
        self.adam_steps = 0
        self.epoch_losses = []

    # ---- state access settles a fit in flight first: whoever reads (or replaces) parameters, moments, step count or losses --
    #      predict, the next fit, pickle.dump / copy.deepcopy (train_agent.py:83,102,129-131), a test -- sees the finished fit
    def __getattribute__(self, name):
        if name in MLPBaseline._STATE:
            d = object.__getattribute__(self, "__dict__")
            if d.get("_pending") is not None:
                object.__getattribute__(self, "_settle")()
        return object.__getattribute__(self, name)

    def __setattr__(self, name, value):
        if name in MLPBaseline._STATE and self.__dict__.get("_pending") is not None:
            self._settle()
        object.__setattr__(self, name, value)

    def __getstate__(self):
        self._settle()
        state = dict(self.__dict__)
        state.pop("_pending", None)
        state.pop("_pins", None)
        state.pop("predraw_stats", None)
        state.pop("_predraw_misses", None)
        state.pop("_predraw_warned", None)
        return state

    def _settle(self):
        """wait for the fit in flight (if any) and take its results over"""
        pend = self.__dict__.get("_pending")
        if pend is None:
            return
        d = self.__dict__
        d["_pending"] = None                              # (first: the assignments below must not recurse)
        pend.end_ev.synchronize()
        h = pend.host
        n = d["params"].size
        if self.epochs > 0 and np.isnan(h["losses"][:self.epochs]).any() and pend.num_samples > 0:
            # the several-workgroup trainer (csrc/mlp_fit.h, MULTI + k_mlp_fit_verdict) turns every loss into NaN when a workgroup
            # waited ~2 s for another.  The baseline keeps what it had BEFORE this fit (parameters, moments, step count, losses);
            # whoever waits on the pending entries (logger.PendingValue, PendingFit.result) is served -- with NaN -- before the
            # error is raised, so a later save_log() / pickle does not trip over a half-settled fit
            from .._lib import MjxError
            d["adam_steps"] = pend.steps_before
            pend.device_ms = float(pend.start_ev.elapsed_time(pend.end_ev))
            pend.value = (float("nan"), float("nan")) if pend.returns is not None else None
            pend.done, pend.keep = True, []
            hooks, pend.hooks = pend.hooks, []
            for hook in hooks:
                hook(pend.value, pend.device_ms)
            raise MjxError("MLPBaseline.fit: non-finite epoch losses (a workgroup of the persistent trainer gave up waiting for "
                           "the others, or the fit diverged); the baseline keeps the parameters it had before this fit")
        d["params"], d["adam_m"], d["adam_v"] = (h["pmv"][i * pend.seg:i * pend.seg + n].copy() for i in range(3))
        d["epoch_losses"] = list(h["losses"][:self.epochs].astype(np.float64) / max(pend.steps, 1))
        pend.device_ms = float(pend.start_ev.elapsed_time(pend.end_ev))
        if pend.returns is not None:
            r = pend.returns[:pend.num_samples]
            e0, e1 = r - pend.pred_before[:pend.num_samples], r - pend.pred_after[:pend.num_samples]
            den = np.sum(r ** 2) + 1e-8
            pend.value = (np.sum(e0 ** 2) / den, np.sum(e1 ** 2) / den)        # mlp_baseline.py:83,94
        pend.done = True
        pend.keep = []
        for hook in pend.hooks:
            hook(pend.value, pend.device_ms)
        pend.hooks = []

    def _hid(self):
        return (ctypes.c_int * max(1, len(self.hidden_sizes)))(*self.hidden_sizes)

    def _forward(self, blk, feat, params_t):
        N = int(feat.shape[0])
        out = blk.torch.empty(N, dtype=blk.torch.float32, device=blk.dev)
        check(blk.lib.mjx_mlp_predict(ptr(feat), N, self.n + 4, self._hid(), len(self.hidden_sizes), ptr(params_t), ptr(out), blk.st()))
        return out

    def _pinned(self, torch, name, count, dtype):
        """page-locked result blocks, kept per baseline and size (allocating them costs more than the copies they receive)"""
        pins = self.__dict__.setdefault("_pins", {})
        ent = pins.get(name)
        if ent is None or ent.numel() < count or ent.dtype != dtype:
            ent = pins[name] = torch.empty(max(int(count), 1), dtype=dtype, pin_memory=True)
        return ent

    def predraw(self, num_samples):
        """Start drawing the NEXT fit's epoch permutations on a helper thread, speculatively: the draws depend on NumPy's global
        generator state and on the row count only, both known before the policy update that train_step runs in between, and the
        native draw (mjx_host_mt19937_permutation) holds no interpreter lock -- it runs under the update's GPU time.  The draws work
        on a COPY of the generator state; fit_async takes them over only if the global state is still what it was when they started
        (nobody drew in between: then this is exactly the stream the reference would have consumed, optimize_model.py:22) and
        otherwise draws again.  -> a handle for fit_async(..., predrawn=handle), or None when there is nothing to gain."""
        import threading
        self._settle()                                           # (the previous fit's copy out of the permutation block is long done)
        st = np.random.get_state()
        num_samples = int(num_samples)
        if st[0] != 'MT19937' or self.epochs <= 0 or num_samples < 2 or ranks.group() is not None:
            return None
        try:
            blk = DeviceOnly()
        except Exception:
            return None
        pin = self._pinned(blk.torch, "perm", self.epochs * num_samples, blk.torch.int32)
        key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
        h = dict(n=num_samples, epochs=int(self.epochs), key0=np.array(st[1], dtype=np.uint32, copy=True), pos0=int(st[2]), gauss0=(st[3], st[4]),
                 key=key, pos=ctypes.c_int32(int(st[2])), pin=pin, error=None)

        def work():
            try:
                out = pin.numpy()                                # (all epochs in one native call: generator and swaps on two threads)
                check(blk.lib.mjx_host_mt19937_permutations(ctypes.c_void_p(key.ctypes.data), ctypes.byref(h["pos"]), num_samples, h["epochs"],
                                                            ctypes.c_void_p(out.ctypes.data)))
            except Exception as e:                               # pragma: no cover
                h["error"] = e
        h["thread"] = threading.Thread(target=work, name="mjx-predraw", daemon=True)
        h["thread"].start()
        return h

    def _take_predrawn(self, h, num_samples):
        """-> True when the speculative draws of predraw() ARE the draws this fit would make now (and NumPy's global state has been
        advanced past them)"""
        if h is None:
            return False
        h["thread"].join()
        st = np.random.get_state()
        same = (h["error"] is None and h["n"] == num_samples and h["epochs"] == int(self.epochs) and st[0] == 'MT19937' and int(st[2]) == h["pos0"]
                and (st[3], st[4]) == h["gauss0"] and np.array_equal(np.asarray(st[1], dtype=np.uint32), h["key0"])
                and self.__dict__.get("_pins", {}).get("perm") is h["pin"])
        if same:
            np.random.set_state(('MT19937', h["key"], int(h["pos"].value), st[3], st[4]))
        d = self.__dict__
        d["predraw_stats"] = (d.get("predraw_stats", (0, 0))[0] + int(same), d.get("predraw_stats", (0, 0))[1] + int(not same))    # (taken, discarded)
        # somebody else draws from np.random between compute_returns and the fit in EVERY iteration (a callback, an env wrapper): the
        # speculative draws are thrown away each time and ~11 ms per 1M rows x 2 epochs are back on the critical path.  Say so once.
        d["_predraw_misses"] = 0 if same else d.get("_predraw_misses", 0) + 1
        if d["_predraw_misses"] == 2 and not d.get("_predraw_warned"):
            d["_predraw_warned"] = True
            import warnings
            warnings.warn("mjrl_amd.MLPBaseline: the speculative epoch permutations were discarded twice in a row -- something draws from "
                          "NumPy's global generator between the returns and the baseline fit of an iteration, so the draws run on the "
                          "critical path again (correct results, ~11 ms per 1M timesteps x 2 epochs slower); predraw_stats = "
                          "(taken %d, discarded %d)" % d["predraw_stats"])
        return same

    def fit(self, paths, return_errors=False):
        """mlp_baseline.py:61-95 + optimize_model.py:7-36: fit_async + wait.  -> (error_before, error_after) when asked."""
        return self.fit_async(paths, return_errors).result()

    def fit_async(self, paths, return_errors=False, predrawn=None):
        """mlp_baseline.py:61-95 + optimize_model.py:7-36, OFF the caller's critical path: inputs are prepared on the caller's
        stream, the persistent trainer (one workgroup, 18 us per Adam step: 0.6 s per 1M timesteps x 2 epochs), the error
        evaluations and the read-backs run on a side stream, and the call returns a PendingFit at once.  The fitted baseline is
        not needed before the next iteration's compute_advantages (batch_reinforce.py:94-112: sampling comes first), so the fit
        hides under the next rollouts; any access to the baseline's state -- predict, fit, pickle, deepcopy -- waits for it.

        With torch.distributed initialised `paths` is this rank's trajectory
        shard; the reference fits ONE network on all paths of the iteration (batch_reinforce.py:94-110), and minibatch Adam is a
        sequential chain, not a sum over samples -- so the ranks' fp32 feature blocks and returns are concatenated in rank order
        (one all-gather of N x (n + 5) floats) and EVERY rank runs the identical persistent trainer on the whole block from the
        same permutations (the LAST rank's draw from NumPy's global stream -- it sampled the batch's last episodes, so under the
        samplers' per-episode seeding its stream stands where a single process's would; every rank draws, so the streams advance
        alike).  The
        trainer is one workgroup whatever the batch size: running it redundantly costs no wall time over running it once and
        broadcasting, and the ranks' baselines stay bit-identical -- equal to a one-rank fit on the rank-ordered paths."""
        self._settle()
        if paths:
            blk = DeviceBlock(paths, self.inp)
            torch = blk.torch
            feat = blk.mlp_features()
            y64 = blk.returns_dev()                            # fp64, left on the device by compute_returns when possible
            y = torch.empty(int(y64.shape[0]), dtype=torch.float32, device=blk.dev)
            check(blk.lib.mjx_cast_f64_f32(ptr(y64), int(y64.shape[0]), ptr(y), blk.st()))   # == astype('float32') (mlp_baseline.py:66)
        else:                                                  # a rank without trajectories still takes part in the gather
            blk = DeviceOnly()
            torch = blk.torch
            feat = torch.zeros((0, self.n + 4), dtype=torch.float32, device=blk.dev)
            y = torch.zeros(0, dtype=torch.float32, device=blk.dev)
        if ranks.group() is not None:
            feat, y = ranks.gather_rows(feat).contiguous(), ranks.gather_rows(y).contiguous()
        num_samples = int(y.shape[0])
        npar = self.params.size
        seg = (npar + 63) // 64 * 64                            # (256-byte aligned segments: the trainer reads them with wide loads)
        stacked = np.zeros(3 * seg, np.float32)
        for i, a in enumerate((self.params, self.adam_m, self.adam_v)):
            stacked[i * seg:i * seg + npar] = a
        pmv = torch.from_numpy(stacked).to(blk.dev)             # one upload: parameters | m | v
        p, m, v = pmv[:npar], pmv[seg:seg + npar], pmv[2 * seg:2 * seg + npar]
        # every epoch's row order from NumPy's global stream like fit_data (optimize_model.py:22), written straight into a page-
        # locked int32 block (one conversion pass per epoch instead of concatenate + astype over the lot)
        have = self._take_predrawn(predrawn, num_samples)      # (speculative draws of predraw(): taken only if they ARE this fit's draws)
        perm_pin = self._pinned(torch, "perm", max(self.epochs, 1) * max(num_samples, 1), torch.int32)
        perm = perm_pin.numpy()[:max(self.epochs * num_samples, 1)]
        if self.epochs * num_samples == 0:
            perm[:] = 0
        if not have:
            _permutations_into(blk.lib, perm, num_samples, self.epochs)
        if ranks.group() is not None:
            perm[:] = ranks.broadcast_host(perm, src=-1)        # the LAST rank's draw (see above)
        perm_t = perm_pin[:perm.shape[0]].to(blk.dev, non_blocking=True)
        losses = torch.zeros(max(self.epochs, 1), dtype=torch.float64, device=blk.dev)
        steps = max(int(num_samples / self.batch_size) - 1, 0)
        pend = PendingFit(self)
        pend.steps, pend.num_samples, pend.seg, pend.steps_before = steps, num_samples, seg, int(self.adam_steps)
        main, side = torch.cuda.current_stream(blk.dev), _fit_stream(torch, blk.dev)
        side.wait_stream(main)                                  # everything above is queued on the caller's stream
        host = dict(pmv=self._pinned(torch, "pmv", 3 * seg, torch.float32), losses=self._pinned(torch, "losses", max(self.epochs, 1), torch.float64))
        with torch.cuda.stream(side):
            pend.start_ev, pend.end_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            pend.start_ev.record(side)
            before = self._forward(blk, feat, p) if return_errors else None
            check(blk.lib.mjx_mlp_fit_adam(ptr(feat), ptr(y), num_samples, self.n + 4, self._hid(), len(self.hidden_sizes), ptr(p), ptr(m),
                                           ptr(v), self.adam_steps, ptr(perm_t), int(self.epochs), int(self.batch_size),
                                           float(self.learn_rate), float(self.reg_coef), ptr(losses), blk.st()))
            after = self._forward(blk, feat, p) if return_errors else None
            host["pmv"][:3 * seg].copy_(pmv, non_blocking=True)
            host["losses"][:max(self.epochs, 1)].copy_(losses, non_blocking=True)
            if return_errors:
                for name, t in (("returns", y), ("pred_before", before), ("pred_after", after)):
                    pin = self._pinned(torch, name, num_samples, torch.float32)
                    pin[:num_samples].copy_(t, non_blocking=True)
                    setattr(pend, name, pin.numpy())
            pend.end_ev.record(side)
        pend.host = dict(pmv=host["pmv"].numpy(), losses=host["losses"].numpy())
        pend.keep = [feat, y, pmv, perm_t, losses, before, after]          # alive until the side stream is through with them
        d = self.__dict__
        d["adam_steps"] = d["adam_steps"] + steps * self.epochs            # (known now; everything else at settlement)
        d["_pending"] = pend
        return pend

    def predict_batch_device(self, paths, shared=True):
        """fp32 predictions of all timesteps as a device block (the GAE chain of utils/process_samples stays there)"""
        blk = DeviceBlock(paths, self.inp, shared)
        p = blk.torch.from_numpy(self.params).to(blk.dev)            # (reading self.params waits for a fit in flight)
        return self._forward(blk, blk.mlp_features(), p)

    def predict_batch(self, paths, shared=True):
        out = self.predict_batch_device(paths, shared)
        import torch
        return ingest.download(ingest.DeviceHandle(torch, out.device, None), out)

    def predict(self, path):
        return self.predict_batch([path], shared=False)
