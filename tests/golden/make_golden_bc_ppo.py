#!/usr/bin/env python
"""Golden fixtures for the minibatch-Adam row: the UNMODIFIED reference's BC
(mjrl/algos/behavior_cloning.py) and PPO (mjrl/algos/ppo_clip.py) run a few minibatch steps on
seeded synthetic data; the minibatch indices are np.random.choice draws after np.random.seed,
exactly what the classes under test draw themselves.

Run in the build container only:   python tests/golden/make_golden_bc_ppo.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import _ref_import  # noqa: E402

_ref_import.install()
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mjrl.algos.behavior_cloning import BC  # noqa: E402
from mjrl.algos.ppo_clip import PPO  # noqa: E402
from mjrl.policies.gaussian_mlp import MLP  # noqa: E402
from mjrl.utils.gym_env import EnvSpec  # noqa: E402

from oracle import synth  # noqa: E402

torch.set_num_threads(4)


def bc_case(name, n, m, hidden, n_traj, T, loss_type, epochs, mb, lr, set_transforms):
    spec = EnvSpec(n, m, T)
    pol = MLP(spec, hidden_sizes=hidden, seed=1, init_log_std=-0.5)
    pol.set_param_values(synth.perturbed_params(pol.get_param_values(), scale=0.1))
    theta0 = pol.get_param_values()
    paths = synth.make_paths(n_traj, T, n, m, seed=21)
    bc = BC(paths, pol, epochs=epochs, batch_size=mb, lr=lr, loss_type=loss_type, save_logs=False, set_transforms=set_transforms)
    theta_start = pol.get_param_values()                 # (set_variance_with_data may have moved log_std)
    np.random.seed(77)
    bc.train(suppress_fit_tqdm=True)
    out = dict(n=n, m=m, hidden=np.array(hidden), n_traj=n_traj, T=T, epochs=epochs, mb=mb, lr=lr, seed_paths=21, seed_np=77,
               loss_type=loss_type, set_transforms=int(set_transforms), theta0=theta0, theta_start=theta_start,
               theta_final=pol.get_param_values(),
               in_shift=np.float32(pol.model.in_shift.numpy()), in_scale=np.float32(pol.model.in_scale.numpy()),
               out_shift=np.float32(pol.model.out_shift.numpy()), out_scale=np.float32(pol.model.out_scale.numpy()))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "moved", float(np.linalg.norm(out["theta_final"] - theta_start)))


def ppo_case(name, n, m, hidden, n_traj, T, epochs, mb, lr, clip):
    spec = EnvSpec(n, m, T)
    pol = MLP(spec, hidden_sizes=hidden, seed=1, init_log_std=-0.5)
    pol.set_param_values(synth.perturbed_params(pol.get_param_values(), scale=0.1))
    theta0 = pol.get_param_values()
    paths = synth.make_paths(n_traj, T, n, m, seed=31)
    rng = np.random.RandomState(5)
    for p in paths:
        p["advantages"] = rng.randn(len(p["rewards"])) * 2.0 + 0.3
    agent = PPO(None, pol, None, clip_coef=clip, epochs=epochs, mb_size=mb, learn_rate=lr, save_logs=False)
    finals = []
    np.random.seed(99)
    for it in range(2):                                     # two calls: the optimizer state carries over
        agent.train_from_paths(paths)
        finals.append(pol.get_param_values())
    out = dict(n=n, m=m, hidden=np.array(hidden), n_traj=n_traj, T=T, epochs=epochs, mb=mb, lr=lr, clip=clip, seed_paths=31,
               seed_adv=5, seed_np=99, theta0=theta0, theta_after_1=finals[0], theta_after_2=finals[1])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "moved", float(np.linalg.norm(finals[1] - theta0)))


if __name__ == "__main__":
    bc_case("bc_mse_32x32", 11, 3, (32, 32), 6, 50, "MSE", 3, 16, 1e-3, True)
    bc_case("bc_mle_64x64", 17, 6, (64, 64), 8, 40, "MLE", 2, 32, 1e-3, False)
    ppo_case("ppo_64x64", 17, 6, (64, 64), 10, 50, 2, 64, 3e-4, 0.2)
