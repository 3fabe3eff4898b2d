"""The "no baseline" baseline: value estimate 0 everywhere, nothing to fit.

Same surface as the other baselines of this package (fit / predict / predict_batch), so the agents and
utils.process_samples can treat it uniformly; stands in for mjrl.baselines.zero_baseline.ZeroBaseline.
There is no device work here.
"""
import numpy as np

_UNFITTED_ERRORS = (1.0, 1.0)        # (error_before, error_after): a zero predictor explains none of the returns


def _zeros_like_rewards(path):
    return np.zeros(np.shape(path["rewards"])[0], dtype=np.float64)


class ZeroBaseline:
    _coeffs = None

    def __init__(self, env_spec=None, **unused):
        self.env_spec = env_spec

    def predict(self, path):
        return _zeros_like_rewards(path)

    def predict_batch(self, paths):
        return np.concatenate([_zeros_like_rewards(p) for p in paths]) if len(paths) else np.zeros(0)

    def fit(self, paths, return_errors=False):
        return _UNFITTED_ERRORS if return_errors else None
