"""Ridge-regression value baselines on the GPU.

``QuadraticBaseline`` mirrors mjrl/baselines/quadratic_baseline.py:4-74 and ``LinearBaseline``
mirrors linear_baseline.py:5-65: same constructor, ``fit(paths, return_errors)``, ``predict(path)``,
``_coeffs``.  The feature matrix A (N x F, fp64; 52.7 GB at BASELINE cfg5) is never materialised:
``mjx_bl_gram`` accumulates A^T A and A^T y on the device from the raw observation block
(csrc/baseline.h k_bl_gram); the F x F solve stays on the host: the reference's ``np.linalg.lstsq`` call with its
regularisation escalation, preceded for F >= 128 by a Cholesky attempt on the same (SPD) system (``_solve_spd``).
"""
import copy

import numpy as np

from ..utils import ranks
from ._features import FEAT_LINEAR, FEAT_QUADRATIC, DeviceBlock, num_features


def _few_blas_threads(limit=8):
    """context manager capping the BLAS / LAPACK thread pools (threadpoolctl when importable, else a no-op)"""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=limit)
    except Exception:                                   # pragma: no cover
        import contextlib
        return contextlib.nullcontext()


class PendingRidgeFit:
    """One ridge fit in flight (_RidgeBaseline.fit_async): the normal equations are being accumulated on the GPU / the F x F system is
    being solved on a helper thread.  ``result()`` waits (once), installs the coefficients and -> (error_before, error_after) when
    errors were asked for, else None.  Same surface as baselines.mlp_baseline.PendingFit (train_step's pending log entries)."""

    def __init__(self, baseline):
        self.baseline, self.done, self.value, self.hooks, self.device_ms = baseline, False, None, [], None
        self.future = self.error = self.coeffs = None

    def finished(self):
        return self.done or self.future is None or self.future.done()

    def result(self):
        if not self.done:
            self.baseline._settle()
        return self.value


class _RidgeBaseline:
    _kind = None

    def __init__(self, env_spec, inp_dim=None, inp='obs', reg_coeff=1e-3):
        self.n = inp_dim if inp_dim is not None else env_spec.observation_dim
        self.inp = inp
        self._reg_coeff = reg_coeff
        self._coeffs = None

    # ---- whoever reads (or replaces) the coefficients -- predict, the next fit, pickle / deepcopy (train_agent.py:83,102,129-131) --
    #      sees the finished fit
    def __getattribute__(self, name):
        if name == "_coeffs":
            d = object.__getattribute__(self, "__dict__")
            if d.get("_pending") is not None:
                object.__getattribute__(self, "_settle")()
        return object.__getattribute__(self, name)

    def __setattr__(self, name, value):
        if name == "_coeffs" and self.__dict__.get("_pending") is not None:
            self._settle()
        object.__setattr__(self, name, value)

    def __getstate__(self):
        self._settle()
        state = dict(self.__dict__)
        state.pop("_pending", None)
        state.pop("_pins", None)
        return state

    def _settle(self):
        pend = self.__dict__.get("_pending")
        if pend is None:
            return
        self.__dict__["_pending"] = None
        pend.future.result()                              # (work() catches its own exceptions: pend.error)
        pend.done = True
        hooks, pend.hooks = pend.hooks, []
        if pend.error is not None:
            if pend.value is None and pend.want_errors:
                pend.value = (float("nan"), float("nan"))
            for hook in hooks:
                hook(pend.value, pend.device_ms or 0.0)
            raise pend.error
        self.__dict__["_coeffs"] = pend.coeffs
        for hook in hooks:
            hook(pend.value, pend.device_ms)

    def fit_async(self, paths, return_errors=False, predrawn=None):
        """quadratic_baseline.py:44-69 / linear_baseline.py:37-60 OFF the caller's critical path (r06): the fitted baseline is not
        read before the NEXT iteration's compute_advantages, and sampling comes first (batch_reinforce.py:78-112).  The Gram
        kernel (and, for the logged errors, the prediction with the old coefficients) is enqueued on the caller's stream, their
        results land in page-locked memory, and a helper thread waits for them, solves the F x F system (LAPACK holds no
        interpreter lock) and -- when errors were asked for -- evaluates the new coefficients on a side stream: the same calls, the
        same bits as fit().  -> PendingRidgeFit.  With torch.distributed initialised (the normal equations are summed over the
        ranks through the host) or MJX_ASYNC_FIT=0 the fit runs in place and the handle comes back settled."""
        import os
        self._settle()
        pend = PendingRidgeFit(self)
        pend.want_errors = bool(return_errors)
        if not paths or ranks.group() is not None or os.environ.get("MJX_ASYNC_FIT", "1") == "0":
            pend.value = self.fit(paths, return_errors)
            pend.done, pend.coeffs, pend.device_ms = True, self.__dict__["_coeffs"], 0.0
            return pend
        blk = DeviceBlock(paths, self.inp)
        torch, dev = blk.torch, blk.dev
        main = torch.cuda.current_stream(dev)
        coef_old = self.__dict__["_coeffs"]
        y = blk.returns_dev()
        F1 = num_features(self._kind, self.n) + 1
        start_ev, end_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start_ev.record(main)
        G = blk.gram_dev(self._kind, y)
        Gpin = self._pinned(torch, "gram", F1 * F1).view(F1, F1)
        Gpin.copy_(G, non_blocking=True)
        pred_pin = None
        if return_errors and coef_old is not None:
            pred_old = blk.predict_linear_dev(self._kind, coef_old)
            pred_pin = self._pinned(torch, "pred", blk.N)
            pred_pin.copy_(pred_old, non_blocking=True)
        end_ev.record(main)
        side = _side_stream(torch, dev)

        def work():
            try:
                end_ev.synchronize()
                pend.device_ms = float(start_ev.elapsed_time(end_ev))
                Gaug = Gpin.numpy()
                F = F1 - 1
                pend.coeffs = self._solve(Gaug[:F, :F].copy(), Gaug[:F, F].copy())
                if return_errors:
                    returns = np.concatenate([path["returns"] for path in paths])
                    den = np.sum(returns ** 2)
                    before = pred_pin.numpy() if pred_pin is not None else np.zeros(returns.shape)
                    with torch.cuda.device(dev), torch.cuda.stream(side):
                        after = blk.predict_linear(self._kind, pend.coeffs)          # (synchronises the side stream only)
                    pend.value = (np.sum((returns - before) ** 2) / den, np.sum((returns - after) ** 2) / den)
            except Exception as e:                     # delivered by _settle
                pend.error = e
            finally:
                pend.keep = None
        pend.keep = (blk, y, G, Gpin, pred_pin, paths)
        pend.future = _fit_worker().submit(work)          # ONE long-lived helper thread: libmjx keeps per-thread feature tables
        self.__dict__["_pending"] = pend
        return pend

    def _pinned(self, torch, name, count):
        """page-locked fp64 result blocks, kept per baseline (allocating one costs more than the copy it receives); a fit in
        flight is settled before the next one starts, so a block is never handed out twice at a time"""
        pins = self.__dict__.setdefault("_pins", {})
        ent = pins.get(name)
        if ent is None or ent.numel() < count:
            ent = pins[name] = torch.empty(max(int(count), 1), dtype=torch.float64, pin_memory=True)
        return ent[:count]

    def _solve(self, G, b):
        """quadratic_baseline.py:54-63 / linear_baseline.py:45-54.  The F x F solve (F <= a few hundred) is a job for a
        handful of cores: with the BLAS pool at its default of one thread per core, LAPACK's SVD spent 10-50 ms (erratic)
        on the 128-core hosts of the GPU boxes -- the unexplained stall of round 1's iteration timings."""
        reg_coeff = copy.deepcopy(self._reg_coeff)
        for _ in range(10):
            A = G + reg_coeff * np.identity(G.shape[0])
            coeffs = self._solve_spd(A, b) if G.shape[0] >= self._CHOLESKY_FROM else None
            if coeffs is None:
                with _few_blas_threads():
                    coeffs = np.linalg.lstsq(A, b, rcond=-1)[0]
            if not np.any(np.isnan(coeffs)):
                break
            reg_coeff *= 10
        return coeffs

    # From this many features on, the regularised normal matrix (symmetric positive definite: A^T A + reg I) is solved by
    # Cholesky factorisation first: LAPACK's SVD-based lstsq needs 65-130 ms for the 825 x 825 system of a 39-dimensional
    # observation (BASELINE configs[4]) and 2.2 ms (+ 0.8 ms for capping the BLAS pool) for the 176 x 176 one of a
    # 17-dimensional observation (configs[1]); the factorisation 6 ms / 0.3 ms, and the two solutions agree to ~1e-9 relative
    # (condition number ~1e9) -- seven orders below what the baseline's predictions are compared at.  Any failure (not positive
    # definite, non-finite result) falls back to the reference's lstsq call.  MJX_RIDGE_LSTSQ=1 forces lstsq everywhere.
    _CHOLESKY_FROM = 128

    @staticmethod
    def _solve_spd(A, b):
        import os
        if os.environ.get("MJX_RIDGE_LSTSQ") == "1":
            return None
        try:
            import scipy.linalg
            x = scipy.linalg.cho_solve(scipy.linalg.cho_factor(A, lower=True, check_finite=False), b, check_finite=False)
            return x if np.all(np.isfinite(x)) else None
        except Exception:
            return None

    def fit(self, paths, return_errors=False):
        """quadratic_baseline.py:44-69 / linear_baseline.py:37-60.  With torch.distributed initialised `paths` is this rank's
        trajectory shard and the fit is still ONE fit over all ranks' paths, as in the reference (batch_reinforce.py:94-110 hands
        every path of the iteration to baseline.fit): the augmented normal equations [A y]^T [A y] -- (F+1)^2 fp64, 5.4 MB at
        BASELINE configs[4] -- and the error sums are summed over the ranks before the host solve, so every rank ends with the
        same coefficients (the same bits: one all-reduce result, the same LAPACK call) and the same logged errors."""
        blk = DeviceBlock(paths, self.inp) if paths else None           # (a rank without trajectories contributes zeros)
        if return_errors:
            error_before = self._relative_error(blk, paths)
        if blk is not None:
            Gaug = blk.gram(self._kind, blk.returns_dev())
        else:
            F1 = num_features(self._kind, self.n) + 1
            Gaug = np.zeros((F1, F1))
        Gaug = ranks.sum_host(Gaug)                                      # (one process: itself)
        F = Gaug.shape[0] - 1
        self._coeffs = self._solve(Gaug[:F, :F], Gaug[:F, F])
        if return_errors:
            return error_before, self._relative_error(blk, paths)

    def _relative_error(self, blk, paths):
        """sum (returns - predictions)^2 / sum returns^2 over the paths of ALL ranks (quadratic_baseline.py:50-52, 66-68)"""
        if blk is not None:
            returns = np.concatenate([path["returns"] for path in paths])
            predictions = blk.predict_linear(self._kind, self._coeffs) if self._coeffs is not None else np.zeros(returns.shape)
            sums = np.array([np.sum((returns - predictions) ** 2), np.sum(returns ** 2)])
        else:
            sums = np.zeros(2)
        sums = ranks.sum_host(sums)
        return sums[0] / sums[1]

    def predict_batch_device(self, paths, shared=True):
        """concatenated predictions as an fp64 device block (utils/process_samples keeps the GAE chain on the device);
        None before the first fit (the caller falls back to predict_batch's zeros)"""
        if self._coeffs is None:
            return None
        return DeviceBlock(paths, self.inp, shared).predict_linear_dev(self._kind, self._coeffs)

    def predict_batch(self, paths, shared=True):
        """concatenated predictions for a list of paths in one device pass"""
        N = sum(len(p["rewards"]) for p in paths)
        if self._coeffs is None:
            return np.zeros(N)
        return DeviceBlock(paths, self.inp, shared).predict_linear(self._kind, self._coeffs)

    def predict(self, path):
        if self._coeffs is None:
            return np.zeros(len(path["rewards"]))
        return self.predict_batch([path], shared=False)


_SIDE = {}
_WORKER = []


def _fit_worker():
    if not _WORKER:
        from concurrent.futures import ThreadPoolExecutor
        _WORKER.append(ThreadPoolExecutor(max_workers=1, thread_name_prefix="mjx-ridge-fit"))
    return _WORKER[0]


def _side_stream(torch, dev):
    key = (dev.type, dev.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


class QuadraticBaseline(_RidgeBaseline):
    _kind = FEAT_QUADRATIC


class LinearBaseline(_RidgeBaseline):
    _kind = FEAT_LINEAR

    def __init__(self, env_spec, inp_dim=None, inp='obs', reg_coeff=1e-5):
        super().__init__(env_spec, inp_dim, inp, reg_coeff)
