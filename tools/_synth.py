"""Synthetic parameters for the measurement tools (so that nothing under tools/ imports oracle/, which is test
infrastructure): nn.Linear-style init + the 0.1 N(0,1) perturbation of SURVEY 8d."""
import numpy as np


def init_params(n, m, hidden, seed=1, init_log_std=-0.5):
    rng = np.random.RandomState(seed)
    sizes = (n,) + tuple(hidden) + (m,)
    flat = []
    for i in range(len(sizes) - 1):
        k = 1.0 / np.sqrt(sizes[i])
        W, b = rng.uniform(-k, k, (sizes[i + 1], sizes[i])), rng.uniform(-k, k, sizes[i + 1])
        if i == len(sizes) - 2:
            W, b = 1e-2 * W, 1e-2 * b
        flat += [W.ravel(), b]
    flat.append(np.full(m, init_log_std))
    return np.concatenate(flat).astype(np.float32)


def perturbed_params(theta0, seed=1, scale=0.1):
    return (theta0 + scale * np.random.RandomState(seed).randn(theta0.size)).astype(np.float32)
