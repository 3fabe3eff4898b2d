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


class _RidgeBaseline:
    _kind = None

    def __init__(self, env_spec, inp_dim=None, inp='obs', reg_coeff=1e-3):
        self.n = inp_dim if inp_dim is not None else env_spec.observation_dim
        self.inp = inp
        self._reg_coeff = reg_coeff
        self._coeffs = None

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


class QuadraticBaseline(_RidgeBaseline):
    _kind = FEAT_QUADRATIC


class LinearBaseline(_RidgeBaseline):
    _kind = FEAT_LINEAR

    def __init__(self, env_spec, inp_dim=None, inp='obs', reg_coeff=1e-5):
        super().__init__(env_spec, inp_dim, inp, reg_coeff)
