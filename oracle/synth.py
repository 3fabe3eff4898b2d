"""Seeded synthetic rollouts.  TEST INFRASTRUCTURE ONLY.

HalfCheetah / Humanoid / Adroit cannot be stepped in this image (no mujoco), so
"identical seeded rollouts" means identical synthetic path sets fed to the
reference / oracle and to the HIP path.
"""
import numpy as np


def make_paths(n_traj, T, n, m, seed=0, ragged=False, act_scale=1.0):
    """paths in mjrl's dict format (mjrl/samplers/core.py:85-93)."""
    rng = np.random.RandomState(seed)
    paths = []
    for _ in range(n_traj):
        Ti = int(rng.randint(max(2, T // 20), T + 1)) if ragged else T
        paths.append(dict(
            observations=rng.randn(Ti, n),
            actions=act_scale * rng.randn(Ti, m),
            rewards=rng.randn(Ti),
            terminated=bool(ragged and Ti < T),
        ))
    return paths


def perturbed_params(theta0, seed=1, scale=0.1):
    """non-degenerate conditioning: theta0 + 0.1*RandomState(1).randn(d)."""
    return (theta0 + scale * np.random.RandomState(seed).randn(theta0.size)).astype(np.float32)


def init_params(n, m, hidden, seed=1, init_log_std=-0.5):
    """torch-free stand-in for MLP.__init__'s nn.Linear init (gaussian_mlp.py:31-37):
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) weights/biases, last layer x1e-2.  Used only
    where the reference policy is not importable (GPU box); parity tests that
    compare with golden fixtures load the reference's own get_param_values()."""
    rng = np.random.RandomState(seed)
    ls = (n,) + tuple(hidden) + (m,)
    out = []
    for i in range(len(ls) - 1):
        k = 1.0 / np.sqrt(ls[i])
        W = rng.uniform(-k, k, (ls[i + 1], ls[i]))
        b = rng.uniform(-k, k, ls[i + 1])
        if i == len(ls) - 2:
            W, b = 1e-2 * W, 1e-2 * b
        out += [W.ravel(), b]
    out.append(np.full(m, init_log_std))
    return np.concatenate(out).astype(np.float32)
