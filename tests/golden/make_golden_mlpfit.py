#!/usr/bin/env python
"""The UNMODIFIED reference's MLPBaseline.fit (mjrl/baselines/mlp_baseline.py:61-95 + utils/optimize_model.py:7-36) at
iteration scale: 300 000 timesteps (300 paths x 1000, obs 17), 2 epochs x 4 686 Adam steps -- statistical pin for the persistent
minibatch-Adam trainer beyond the short bit-level chains: after thousands of chaotic ReLU / Adam steps two
correct implementations no longer share parameters, but they reach the same fit quality on the same minibatch sequence.
Stored: the fit errors (before / after, :83,:94), the mean squared prediction error on held-out paths and summary statistics
of the predictions.  ~40 s of CPU:   python tests/golden/make_golden_mlpfit.py
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import _ref_import  # noqa: E402

_ref_import.install()
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mjrl.baselines.mlp_baseline import MLPBaseline  # noqa: E402
from mjrl.utils import process_samples  # noqa: E402
from mjrl.utils.gym_env import EnvSpec  # noqa: E402

torch.set_num_threads(8)


def make_paths(n_traj, T, n, seed):
    """returns with learnable structure: rewards are a smooth function of the observation plus noise"""
    rng = np.random.RandomState(seed)
    w = np.random.RandomState(1234).randn(n) / np.sqrt(n)
    paths = []
    for _ in range(n_traj):
        obs = np.cumsum(0.1 * rng.randn(T, n), axis=0) + rng.randn(n)
        rew = np.tanh(obs @ w) - 0.05 * np.sum(obs[:, :3] ** 2, axis=1) + 0.1 * rng.randn(T)
        paths.append(dict(observations=obs, rewards=rew, terminated=False))
    process_samples.compute_returns(paths, 0.995)
    return paths


def main(name="mlpfit_300k"):
    n, n_traj, T = 17, 300, 1000
    paths = make_paths(n_traj, T, n, seed=3)
    held = make_paths(20, T, n, seed=4)
    spec = EnvSpec(n, 6, T)
    torch.manual_seed(4); np.random.seed(4)
    bl = MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3)
    np.random.seed(11)
    t0 = time.time()
    e0, e1 = bl.fit(paths, return_errors=True)
    dt = time.time() - t0
    pred = np.concatenate([bl.predict(p) for p in held])
    y = np.concatenate([p["returns"] for p in held])
    out = dict(n=n, n_traj=n_traj, T=T, path_seed=3, held_seed=4, init_seed=4, fit_seed=11, err_before=e0, err_after=e1,
               held_mse=float(np.mean((pred - y) ** 2)), held_var=float(np.var(y)), pred_mean=float(pred.mean()), pred_std=float(pred.std()),
               ret_mean=float(y.mean()), reference_seconds=dt, steps=2 * (n_traj * T // 64 - 1))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: (float(v) if np.ndim(v) == 0 else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
