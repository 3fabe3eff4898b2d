#!/usr/bin/env python
"""Generate golden fixtures by running the UNMODIFIED reference (/root/reference).

Run in the build container only:   python tests/golden/make_golden.py
Inputs are regenerated from seeds by oracle/synth.py (so fixtures stay small);
the policy / baseline parameters come from the reference's own constructors and
are stored.  Outputs stored: everything the parity tests compare against.
"""
import copy
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import _ref_import  # noqa: E402

_ref_import.install()
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mjrl.algos.dapg import DAPG  # noqa: E402
from mjrl.algos.npg_cg import NPG  # noqa: E402
from mjrl.algos.trpo import TRPO  # noqa: E402
from mjrl.baselines.linear_baseline import LinearBaseline  # noqa: E402
from mjrl.baselines.mlp_baseline import MLPBaseline  # noqa: E402
from mjrl.baselines.quadratic_baseline import QuadraticBaseline  # noqa: E402
from mjrl.policies.gaussian_linear import LinearPolicy  # noqa: E402
from mjrl.policies.gaussian_mlp import MLP  # noqa: E402
from mjrl.utils import process_samples  # noqa: E402
from mjrl.utils.cg_solve import cg_solve  # noqa: E402
from mjrl.utils.gym_env import EnvSpec  # noqa: E402

from oracle import synth  # noqa: E402

torch.set_num_threads(8)


BIG = 20000      # d above this: numpy-initialised params (regenerable) + subsampled outputs
STRIDE = 37


def make_policy(n, m, hidden, perturb=True, transforms=None):
    spec = EnvSpec(n, m, 1000)
    pol = MLP(spec, hidden_sizes=hidden, seed=1, init_log_std=-0.5) if len(hidden) else \
        LinearPolicy(spec, seed=1, init_log_std=-0.5)
    if pol.d > BIG:
        pol.set_param_values(synth.init_params(n, m, hidden, seed=1, init_log_std=-0.5))
    if perturb:
        pol.set_param_values(synth.perturbed_params(pol.get_param_values(), scale=0.1 if pol.d <= BIG else 0.02))
    if transforms is not None:
        pol.model.set_transformations(*transforms)
        pol.old_model.set_transformations(*transforms)
    return pol


def fake_advantages(paths, seed):
    rng = np.random.RandomState(seed)
    for p in paths:
        p["advantages"] = rng.randn(len(p["rewards"])) * 2.0 + 0.3


def cat(paths):
    return (np.concatenate([p["observations"] for p in paths]),
            np.concatenate([p["actions"] for p in paths]),
            np.concatenate([p["advantages"] for p in paths]))


def npg_case(name, n, m, hidden, n_traj, T, cg_iters, algo="npg", ragged=False, transforms=None,
             kl_dist=None, step=0.05, demo=None, input_normalization=None):
    pol = make_policy(n, m, hidden, transforms=transforms)
    theta0 = pol.get_param_values()
    paths = synth.make_paths(n_traj, T, n, m, seed=0, ragged=ragged)
    fake_advantages(paths, 5)
    obs, act, adv = cat(paths)
    adv_w = (adv - np.mean(adv)) / (np.std(adv) + 1e-6)
    kw = dict(FIM_invert_args={'iters': cg_iters, 'damping': 1e-4})
    if algo == "npg":
        agent = NPG(None, pol, None, normalized_step_size=step, input_normalization=input_normalization, **kw)
    elif algo == "trpo":
        agent = TRPO(None, pol, None, kl_dist=kl_dist, **kw)
    else:
        dpaths = synth.make_paths(demo[0], demo[1], n, m, seed=7)
        agent = DAPG(None, pol, None, demo_paths=dpaths, kl_dist=kl_dist, lam_0=1e-2, lam_1=0.95, **kw)
    out = dict(theta0=theta0, n=n, m=m, hidden=np.array(hidden, dtype=np.int64), n_traj=n_traj, T=T,
               cg_iters=cg_iters, ragged=ragged, path_seed=0, adv_seed=5, damping=1e-4, step=step,
               kl_dist=-1.0 if kl_dist is None else kl_dist, algo=algo, N=obs.shape[0])
    if transforms is not None:
        out.update(in_shift=transforms[0], in_scale=transforms[1], out_shift=transforms[2], out_scale=transforms[3])
    if demo is not None:
        out.update(demo_n_traj=demo[0], demo_T=demo[1], demo_seed=7, lam_0=1e-2, lam_1=0.95)
    # pieces (only meaningful for the plain NPG gradient; DAPG builds its own)
    out["input_normalization"] = -1.0 if input_normalization is None else input_normalization
    out["surr_before"] = agent.CPI_surrogate(obs, act, adv_w).data.numpy().ravel()[0]
    g = agent.flat_vpg(obs, act, adv_w)
    out["vpg"] = g
    out["hvp_of_vpg"] = agent.HVP(obs, act, g)
    out["cg_x"] = cg_solve(agent.build_Hvp_eval([obs, act], regu_coef=1e-4), g, x_0=g.copy(), cg_iters=cg_iters)
    # full update through the reference's own train_from_paths
    for p in paths:
        p_adv = p["advantages"]
    agent.save_logs = True
    from mjrl.utils.logger import DataLog
    agent.logger = DataLog()
    stats = agent.train_from_paths(paths)
    log = agent.logger.log
    out.update(final_in_shift=pol.model.in_shift.data.numpy(), final_in_scale=pol.model.in_scale.data.numpy())
    out.update(new_params=pol.get_param_values(), alpha=log['alpha'][-1], kl=log['kl_dist'][-1],
               surr_improvement=log['surr_improvement'][-1], base_stats=np.array(stats),
               running_score=agent.running_score)
    if theta0.size > BIG:      # keep the fixture small: strided samples + norms + a probe projection
        probe = np.random.RandomState(99).randn(theta0.size)
        for k in ("vpg", "hvp_of_vpg", "cg_x", "new_params"):
            v = out.pop(k)
            out[k + "_sub"] = v[::STRIDE].copy()
            out[k + "_norm"] = np.linalg.norm(v.astype(np.float64))
            out[k + "_probe"] = float(np.dot(v.astype(np.float64), probe))
        out.pop("theta0")
        out.update(stride=STRIDE, probe_seed=99, big=True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "N", obs.shape[0], "d", theta0.size, "alpha", out["alpha"], "kl", out["kl"],
          "surr_imp", out["surr_improvement"])


def hvp_general_case(name, n, m, hidden, n_traj, T):
    """old != new: the input_normalization situation (npg_cg.py:101-107) and a parameter offset."""
    pol = make_policy(n, m, hidden)
    theta_old = pol.get_param_values()
    rng = np.random.RandomState(3)
    theta_new = (theta_old + 0.05 * rng.randn(theta_old.size)).astype(np.float32)
    pol.set_param_values(theta_new, set_new=True, set_old=False)
    in_shift, in_scale = 0.1 * rng.randn(n), 1.0 + 0.1 * rng.rand(n)
    pol.model.set_transformations(in_shift, in_scale, None, None)      # only `model`, like npg_cg.py:107
    paths = synth.make_paths(n_traj, T, n, m, seed=0)
    fake_advantages(paths, 5)
    obs, act, adv = cat(paths)
    adv_w = (adv - np.mean(adv)) / (np.std(adv) + 1e-6)
    agent = NPG(None, pol, None)
    v = rng.randn(theta_old.size).astype(np.float32)
    out = dict(theta_old=theta_old, theta_new=pol.get_param_values(), in_shift=in_shift, in_scale=in_scale,
               n=n, m=m, hidden=np.array(hidden, dtype=np.int64), n_traj=n_traj, T=T, path_seed=0, adv_seed=5, v=v,
               hvp=agent.HVP(obs, act, v, regu_coef=1e-4), vpg=agent.flat_vpg(obs, act, adv_w),
               surr=agent.CPI_surrogate(obs, act, adv_w).data.numpy().ravel()[0],
               kl=agent.kl_old_new(obs, act).data.numpy().ravel()[0])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "kl", out["kl"], "surr", out["surr"])


def gae_case(name, n, n_traj, T, kind):
    m = 3
    spec = EnvSpec(n, m, T)
    paths = synth.make_paths(n_traj, T, n, m, seed=11, ragged=True)
    out = dict(n=n, m=m, n_traj=n_traj, T=T, path_seed=11, kind=kind, gamma=0.995, lam=0.97)
    torch.manual_seed(4); np.random.seed(4)
    if kind == "mlp":
        bl = MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3)
        out["bl_params"] = np.concatenate([p.data.numpy().ravel() for p in bl.model.parameters()])
    elif kind == "quadratic":
        bl = QuadraticBaseline(spec)
    else:
        bl = LinearBaseline(spec)
    process_samples.compute_returns(paths, 0.995)
    out["returns"] = np.concatenate([p["returns"] for p in paths])
    if kind != "mlp":
        e0, e1 = bl.fit(paths, return_errors=True)      # first fit from None coefficients
        out.update(coeffs=bl._coeffs, err_before=e0, err_after=e1)
    process_samples.compute_advantages(paths, bl, 0.995, 0.97)
    out["baseline_pred"] = np.concatenate([p["baseline"] for p in paths])
    out["advantages"] = np.concatenate([p["advantages"] for p in paths])
    paths2 = copy.deepcopy(paths)
    process_samples.compute_advantages(paths2, bl, 0.995, None)          # non-GAE branch
    out["advantages_nogae"] = np.concatenate([p["advantages"] for p in paths2])
    if kind == "mlp":
        np.random.seed(9)
        e0, e1 = bl.fit(paths, return_errors=True)
        out.update(fit_seed=9, err_before=e0, err_after=e1,
                   bl_params_after=np.concatenate([p.data.numpy().ravel() for p in bl.model.parameters()]))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "N", out["returns"].shape[0], "err", out.get("err_before"), out.get("err_after"))


if __name__ == "__main__":
    tr = (0.1 * np.arange(17) / 17.0, 1.0 + 0.05 * np.arange(17), 0.02 * np.arange(6), 1.0 + 0.1 * np.arange(6))
    npg_case("npg_cfg1_linear", 6, 2, (), 400, 25, 5)
    npg_case("npg_pointmass_32x32", 6, 2, (32, 32), 40, 25, 10)
    npg_case("npg_cfg2_small", 17, 6, (64, 64), 20, 500, 10)
    npg_case("npg_cfg2_ragged_tr", 17, 6, (64, 64), 37, 300, 10, ragged=True, transforms=tr)
    npg_case("npg_inputnorm_32x32", 11, 3, (32, 32), 30, 100, 10, input_normalization=0.7)
    npg_case("trpo_cfg3_small", 17, 6, (64, 64), 20, 500, 10, algo="trpo", kl_dist=0.01)
    npg_case("npg_cfg4_small", 376, 17, (256, 256), 80, 250, 25)
    npg_case("dapg_cfg5_small", 39, 28, (512, 512), 100, 100, 10, algo="dapg", kl_dist=0.025, demo=(5, 100))
    hvp_general_case("hvp_general_64x64", 17, 6, (64, 64), 20, 250)
    gae_case("gae_mlp", 17, 30, 400, "mlp")
    gae_case("gae_quadratic", 11, 30, 400, "quadratic")
    gae_case("gae_linear", 11, 30, 400, "linear")
