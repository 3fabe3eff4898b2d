#!/usr/bin/env python
"""Golden fixture for the BASELINE configs[4] PIPELINE on ONE policy object, from the UNMODIFIED reference:

    BC pre-training on demonstrations (mjrl/algos/behavior_cloning.py:107-136; input / output transforms set from the
    demonstrations, :55-71)  ->  2 DAPG iterations (mjrl/algos/dapg.py:54-141) with a quadratic baseline: per iteration
    compute_returns, compute_advantages (GAE), DAPG.train_from_paths, baseline.fit  (= batch_reinforce.py:94-110 without the
    sampler: the on-policy paths are seeded synthetic ones, there is no Adroit environment in the reference tree).

Shapes: obs 39, act 28, 512 x 512 (door-v0 class), 25 demonstration paths x 200 steps, 250 on-policy paths x 200 = 50 000
timesteps per iteration.  Stored: parameters after BC and after each DAPG iteration (strided), step norms, alpha / kl per
iteration, the baseline's coefficients after each fit, the transforms BC set.  Run in the build container only (~2 min):

    python tests/golden/make_golden_pipeline.py
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
from mjrl.algos.behavior_cloning import BC  # noqa: E402
from mjrl.algos.dapg import DAPG  # noqa: E402
from mjrl.baselines.quadratic_baseline import QuadraticBaseline  # noqa: E402
from mjrl.policies.gaussian_mlp import MLP  # noqa: E402
from mjrl.utils import process_samples  # noqa: E402
from mjrl.utils.gym_env import EnvSpec  # noqa: E402
from mjrl.utils.logger import DataLog  # noqa: E402

from oracle import synth  # noqa: E402

torch.set_num_threads(8)
STRIDE = 16


def main(name="pipeline_cfg5"):
    n, m, hidden = 39, 28, (512, 512)
    cfg = dict(n=n, m=m, hidden=np.array(hidden), demo_n_traj=25, demo_T=200, demo_seed=7, n_traj=250, T=200, path_seeds=np.array([10, 11]),
               bc_epochs=2, bc_mb=64, bc_lr=1e-3, seed_np=123, gamma=0.995, gae_lambda=0.97, kl_dist=0.025, lam_0=1e-2, lam_1=0.95,
               cg_iters=10, damping=1e-4, stride=STRIDE, theta_scale=0.02)
    spec = EnvSpec(n, m, 200)
    pol = MLP(spec, hidden_sizes=hidden, seed=1, init_log_std=-0.5)
    pol.set_param_values(synth.perturbed_params(synth.init_params(n, m, hidden, seed=1, init_log_std=-0.5), scale=0.02))
    theta0 = pol.get_param_values()
    demos = synth.make_paths(25, 200, n, m, seed=7)
    out = dict(cfg)
    # ---- BC (MLE is the reference's default loss; set_transforms default True)
    t0 = time.time()
    bc = BC(demos, pol, epochs=2, batch_size=64, lr=1e-3, loss_type='MLE', save_logs=False, set_transforms=True)
    theta_start = pol.get_param_values()
    np.random.seed(123)
    bc.train(suppress_fit_tqdm=True)
    theta_bc = pol.get_param_values()
    print("BC %.1f s, moved %.4f" % (time.time() - t0, np.linalg.norm(theta_bc - theta_start)), flush=True)
    out.update(theta_start_sub=theta_start[::STRIDE], theta_bc=theta_bc, bc_moved=float(np.linalg.norm(theta_bc - theta_start)),
               in_shift=np.float32(pol.model.in_shift.numpy()), in_scale=np.float32(pol.model.in_scale.numpy()),
               out_shift=np.float32(pol.model.out_shift.numpy()), out_scale=np.float32(pol.model.out_scale.numpy()))
    # ---- DAPG iterations with a quadratic baseline
    bl = QuadraticBaseline(spec)
    agent = DAPG(None, pol, bl, demo_paths=demos, kl_dist=0.025, lam_0=1e-2, lam_1=0.95, FIM_invert_args={'iters': 10, 'damping': 1e-4},
                 save_logs=True)
    agent.logger = DataLog()
    prev = theta_bc.astype(np.float64)
    for it, seed in enumerate((10, 11)):
        paths = synth.make_paths(250, 200, n, m, seed=seed)
        t0 = time.time()
        process_samples.compute_returns(paths, 0.995)
        process_samples.compute_advantages(paths, bl, 0.995, 0.97)
        stats = agent.train_from_paths(paths)
        errs = bl.fit(paths, return_errors=True)
        th = pol.get_param_values()
        step = th.astype(np.float64) - prev
        out.update({"theta_it%d_sub" % it: th[::STRIDE], "theta_it%d_norm" % it: float(np.linalg.norm(th.astype(np.float64))),
                    "step_it%d_sub" % it: step[::STRIDE], "step_norm_it%d" % it: float(np.linalg.norm(step)),
                    "alpha_it%d" % it: agent.logger.log['alpha'][-1], "kl_it%d" % it: agent.logger.log['kl_dist'][-1],
                    "surr_improvement_it%d" % it: agent.logger.log['surr_improvement'][-1], "stats_it%d" % it: np.array(stats),
                    "bl_coeffs_it%d" % it: np.asarray(bl._coeffs, np.float64), "bl_errors_it%d" % it: np.array(errs, np.float64),
                    "adv0_it%d" % it: np.asarray(paths[0]["advantages"], np.float64), "ret0_it%d" % it: np.asarray(paths[0]["returns"], np.float64)})
        prev = th.astype(np.float64)
        print("DAPG iteration %d: %.1f s  alpha %r kl %r |step| %.5f  VF errors %r" % (it, time.time() - t0, out["alpha_it%d" % it],
              out["kl_it%d" % it], out["step_norm_it%d" % it], errs), flush=True)
    out["theta0_sub"] = theta0[::STRIDE]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("saved", name, os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024, "KB")


if __name__ == "__main__":
    main()
