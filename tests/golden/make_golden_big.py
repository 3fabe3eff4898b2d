#!/usr/bin/env python
"""Golden fixtures for the two big BASELINE configs at WELL-CONDITIONED sizes (N >= d), from the UNMODIFIED
reference (/root/reference), plus the fp64-oracle numbers the parity tests fall back on.

Run in the build container only (minutes of CPU time):   python tests/golden/make_golden_big.py [case ...]

  npg_cfg4_wide     obs 376, act 17, 256x256, 800 x 250 = 200 000 timesteps (d = 166 690), NPG, 25 CG iterations
                    (mjrl/algos/npg_cg.py:108-142 at BASELINE configs[3] shapes)
  dapg_cfg5_wide    obs 39, act 28, 512x512, 1500 x 200 = 300 000 on-policy timesteps (d = 297 528) + 25 x 200
                    demonstration steps, DAPG, 10 CG iterations (mjrl/algos/dapg.py:92-121 at configs[4] shapes)
  bench_ref_1m      the same 1M-timestep batch through the UNMODIFIED reference: NPG.train_from_paths (configs[1]) and
                    TRPO.train_from_paths with kl_dist 0.025 (configs[2], 3 line-search trials); whole vectors stored
  npg_cfg4_shard    bench.py's configs[3] shard (500 000 x (376, 17), 256x256, 25 CG) through the reference's NPG.train_from_paths
  dapg_cfg5_shard   bench.py's configs[4] shard (1M x (39, 28), 512x512 + 5 000 demonstration rows) through DAPG.train_from_paths
  bench_cfg2_1m     the 1M-timestep batch bench.py runs (configs[1]): alpha / kl / surr_improvement of one NPG update
                    from the fp64 oracle (the reference needs ~20 s and agrees to 1e-6)

Inputs are regenerated from seeds (oracle/synth.py, bench.py); stored are strided samples of the reference's output
vectors (every STRIDE-th entry) with their norms and a random projection, the same for the fp64 oracle, and
rel-L2(reference, fp64 oracle) per vector -- the error level of the reference's own fp32 arithmetic on that problem.
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
from mjrl.algos.dapg import DAPG  # noqa: E402
from mjrl.algos.npg_cg import NPG  # noqa: E402
from mjrl.policies.gaussian_mlp import MLP  # noqa: E402
from mjrl.utils.cg_solve import cg_solve  # noqa: E402
from mjrl.utils.gym_env import EnvSpec  # noqa: E402
from mjrl.utils.logger import DataLog  # noqa: E402

from oracle import npg_oracle as O  # noqa: E402
from oracle import synth  # noqa: E402

torch.set_num_threads(8)
STRIDE = 8
PROBE_SEED = 99


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def fake_advantages(paths, seed):
    rng = np.random.RandomState(seed)
    for p in paths:
        p["advantages"] = rng.randn(len(p["rewards"])) * 2.0 + 0.3


def cat(paths, key):
    return np.concatenate([p[key] for p in paths])


def pack(out, key, ref, f64, probe):
    """strided samples + norm + projection of the reference vector and of the fp64-oracle vector"""
    for tag, v in (("", ref), ("_f64", f64)):
        v = np.asarray(v)
        out[key + tag + "_sub"] = v[::STRIDE].astype(np.float32 if tag == "" else np.float64)
        out[key + tag + "_norm"] = float(np.linalg.norm(v.astype(np.float64)))
        out[key + tag + "_probe"] = float(np.dot(v.astype(np.float64), probe))
    out["err_ref_vs_f64_" + key] = rel(ref, f64)


def big_case(name, n, m, hidden, n_traj, T, cg_iters, algo, kl_dist=None, step=0.05, demo=None):
    t_start = time.time()
    spec = EnvSpec(n, m, 1000)
    pol = MLP(spec, hidden_sizes=hidden, seed=1, init_log_std=-0.5)
    pol.set_param_values(synth.perturbed_params(synth.init_params(n, m, hidden, seed=1, init_log_std=-0.5), scale=0.02))
    theta0 = pol.get_param_values()
    paths = synth.make_paths(n_traj, T, n, m, seed=0)
    fake_advantages(paths, 5)
    obs, act, adv = cat(paths, "observations"), cat(paths, "actions"), cat(paths, "advantages")
    adv_w = (adv - np.mean(adv)) / (np.std(adv) + 1e-6)
    kw = dict(FIM_invert_args={'iters': cg_iters, 'damping': 1e-4})
    out = dict(n=n, m=m, hidden=np.array(hidden, dtype=np.int64), n_traj=n_traj, T=T, cg_iters=cg_iters, ragged=False,
               path_seed=0, adv_seed=5, damping=1e-4, step=step, kl_dist=-1.0 if kl_dist is None else kl_dist, algo=algo,
               N=obs.shape[0], stride=STRIDE, probe_seed=PROBE_SEED, big=True, wide=True, theta_scale=0.02)
    probe = np.random.RandomState(PROBE_SEED).randn(theta0.size)
    th64 = theta0.astype(np.float64)
    if algo == "npg":
        agent = NPG(None, pol, None, normalized_step_size=step, **kw)
        g_obs, g_act, g_adv, coef = obs, act, adv_w, 1.0
        delta = step
    else:
        dpaths = synth.make_paths(demo[0], demo[1], n, m, seed=7)
        agent = DAPG(None, pol, None, demo_paths=dpaths, kl_dist=kl_dist, lam_0=1e-2, lam_1=0.95, **kw)
        out.update(demo_n_traj=demo[0], demo_T=demo[1], demo_seed=7, lam_0=1e-2, lam_1=0.95)
        d_obs, d_act = cat(dpaths, "observations"), cat(dpaths, "actions")
        g_obs, g_act = np.concatenate([obs, d_obs]), np.concatenate([act, d_act])
        g_adv = 1e-2 * np.concatenate([adv_w / (np.std(adv_w) + 1e-8), 1e-2 * np.ones(d_obs.shape[0])])   # dapg.py:65-70, iter 0
        coef = g_adv.shape[0] / adv_w.shape[0]
        delta = 2.0 * kl_dist
    # ---- the reference's pieces (fp32 torch + numpy)
    t0 = time.time()
    out["surr_before"] = agent.CPI_surrogate(obs, act, adv_w).data.numpy().ravel()[0]
    g = agent.flat_vpg(g_obs, g_act, g_adv)
    if algo != "npg":
        g = coef * g                                            # the expression of dapg.py:98 (python float x fp32 array)
    hv = agent.HVP(obs, act, g)
    print(name, "reference pieces: vpg + hvp %.1f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    x = cg_solve(agent.build_Hvp_eval([obs, act], regu_coef=1e-4), g, x_0=g.copy(), cg_iters=cg_iters)
    print(name, "reference cg_solve %.1f s" % (time.time() - t0), flush=True)
    # ---- the reference's own train_from_paths
    agent.save_logs = True
    agent.logger = DataLog()
    t0 = time.time()
    stats = agent.train_from_paths(paths)
    out["reference_update_seconds"] = time.time() - t0
    print(name, "reference train_from_paths %.1f s" % out["reference_update_seconds"], flush=True)
    log = agent.logger.log
    new_params = pol.get_param_values()
    out.update(alpha=log['alpha'][-1], kl=log['kl_dist'][-1], surr_improvement=log['surr_improvement'][-1],
               base_stats=np.array(stats), running_score=agent.running_score)
    # ---- fp64 oracle on the same inputs
    t0 = time.time()
    if algo == "npg":
        ref64 = O.npg_update(th64, obs, act, adv_w, n, m, hidden, cg_iters=cg_iters, damping=1e-4, delta=step)
    else:
        ref64 = O.dapg_update(th64, obs, act, adv_w, d_obs, d_act, n, m, hidden, cg_iters=cg_iters, damping=1e-4,
                              kl_dist=kl_dist, lam_0=1e-2, lam_1=0.95, iter_count=0.0)
    hv64 = O.fvp(th64, obs, g.astype(np.float64), n, m, hidden, damping=1e-4)       # same input vector as the reference's HVP
    print(name, "fp64 oracle %.1f s" % (time.time() - t0), flush=True)
    step_ref = new_params.astype(np.float64) - th64
    step_64 = ref64["new_params"] - th64
    pack(out, "vpg", g, ref64["vpg"], probe)
    pack(out, "hvp_of_vpg", hv, hv64, probe)
    pack(out, "cg_x", x, ref64["npg"], probe)
    pack(out, "update_step", step_ref, step_64, probe)
    out.update(alpha_f64=ref64["alpha"], kl_f64=float(ref64["kl"]),
               surr_improvement_f64=float(ref64["surr_after"] - ref64["surr_before"]),
               surr_before_f64=float(ref64["surr_before"]), step_over_theta=float(np.linalg.norm(step_ref) / np.linalg.norm(th64)))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "N", obs.shape[0], "d", theta0.size, "alpha", out["alpha"], out["alpha_f64"], "kl", out["kl"], out["kl_f64"],
          "| reference vs fp64 oracle: vpg %.2e hvp %.2e cg_x %.2e step %.2e | total %.0f s"
          % (out["err_ref_vs_f64_vpg"], out["err_ref_vs_f64_hvp_of_vpg"], out["err_ref_vs_f64_cg_x"],
             out["err_ref_vs_f64_update_step"], time.time() - t_start), flush=True)


def bench_case(name):
    """fp64-oracle scalars of ONE NPG update on bench.py's own 1M-timestep synthetic batch (BASELINE configs[1])."""
    import bench
    theta0 = bench.initial_params()
    obs, act, adv = bench.synth_shard(0, 1)
    adv = (adv - adv.mean()) / (adv.std() + 1e-6)
    t0 = time.time()
    r = O.npg_update(theta0.astype(np.float64), obs.astype(np.float64), act.astype(np.float64), adv, bench.N_OBS, bench.N_ACT,
                     bench.HIDDEN, cg_iters=bench.CG_ITERS, damping=bench.DAMPING, delta=bench.STEP)
    step = r["new_params"] - theta0.astype(np.float64)
    out = dict(alpha=r["alpha"], kl=float(r["kl"]), surr_improvement=float(r["surr_after"] - r["surr_before"]),
               surr_before=float(r["surr_before"]), N=obs.shape[0], step_sub=step[::4], step_norm=float(np.linalg.norm(step)),
               stride=4, oracle_seconds=time.time() - t0)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v for k, v in out.items() if np.ndim(v) == 0}, flush=True)


def bench_reference_case(name):
    """The UNMODIFIED reference on bench.py's own 1M-timestep batch (BASELINE configs[1] and configs[2]):
    NPG.train_from_paths (mjrl/algos/npg_cg.py:91-163) and TRPO.train_from_paths with kl_dist = 0.025
    (mjrl/algos/trpo.py:56-146; the first two step lengths are rejected -> 3 trials) -- ~20 s + ~25 s of CPU.
    d = 5 708, so the whole vectors are stored (gradient, CG solution, both update steps)."""
    import contextlib
    import io
    import bench
    from mjrl.algos.trpo import TRPO
    n, m, hidden = bench.N_OBS, bench.N_ACT, bench.HIDDEN
    theta0 = bench.initial_params()
    obs, act, adv = bench.synth_shard(0, 1)
    T = bench.T
    paths = [dict(observations=obs[i * T:(i + 1) * T].astype(np.float64), actions=act[i * T:(i + 1) * T].astype(np.float64),
                  rewards=np.zeros(T), advantages=adv[i * T:(i + 1) * T].copy(), terminated=False) for i in range(bench.N_TRAJ)]
    kw = dict(FIM_invert_args={'iters': bench.CG_ITERS, 'damping': bench.DAMPING}, save_logs=True)
    spec = EnvSpec(n, m, T)
    out = dict(N=obs.shape[0], n=n, m=m, hidden=np.array(hidden, dtype=np.int64), cg_iters=bench.CG_ITERS, damping=bench.DAMPING,
               step=bench.STEP, trpo_kl_dist=0.025, theta0=theta0, torch_threads=torch.get_num_threads())

    def fresh():
        pol = MLP(spec, hidden_sizes=hidden, seed=1, init_log_std=-0.5)
        pol.set_param_values(theta0.copy())
        assert np.array_equal(pol.get_param_values(), theta0)
        return pol
    # ---- configs[1]: NPG
    pol = fresh()
    agent = NPG(None, pol, None, normalized_step_size=bench.STEP, **kw)
    agent.logger = DataLog()
    adv_w = (adv - np.mean(adv)) / (np.std(adv) + 1e-6)
    t0 = time.time()
    g = agent.flat_vpg(obs.astype(np.float64), act.astype(np.float64), adv_w)
    x = cg_solve(agent.build_Hvp_eval([obs.astype(np.float64), act.astype(np.float64)], regu_coef=bench.DAMPING), g, x_0=g.copy(),
                 cg_iters=bench.CG_ITERS)
    print(name, "reference vpg + cg_solve %.1f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    agent.train_from_paths(paths)
    out["npg_reference_seconds"] = time.time() - t0
    log = agent.logger.log
    out.update(npg_vpg=g, npg_cg_x=x, npg_alpha=log['alpha'][-1], npg_kl=log['kl_dist'][-1],
               npg_surr_improvement=log['surr_improvement'][-1], npg_new_params=pol.get_param_values())
    print(name, "reference NPG.train_from_paths %.1f s: alpha %r kl %r surr_improvement %r"
          % (out["npg_reference_seconds"], out["npg_alpha"], out["npg_kl"], out["npg_surr_improvement"]), flush=True)
    # ---- configs[2]: TRPO, kl_dist = 0.025
    pol = fresh()
    agent = TRPO(None, pol, None, kl_dist=0.025, **kw)
    agent.logger = DataLog()
    buf = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stdout(buf):
        agent.train_from_paths(paths)
    out["trpo_reference_seconds"] = time.time() - t0
    log = agent.logger.log
    rejected = buf.getvalue().count("Backtracking")
    out.update(trpo_alpha=log['alpha'][-1], trpo_kl=log['kl_dist'][-1], trpo_surr_improvement=log['surr_improvement'][-1],
               trpo_new_params=pol.get_param_values(), trpo_trials=rejected + 1)
    print(name, "reference TRPO.train_from_paths %.1f s: alpha %r kl %r surr_improvement %r trials %d"
          % (out["trpo_reference_seconds"], out["trpo_alpha"], out["trpo_kl"], out["trpo_surr_improvement"], rejected + 1), flush=True)
    assert rejected >= 2, "kl_dist must make the line search backtrack at least twice"
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def shard_case(key):
    """The UNMODIFIED reference at the per-GPU shard sizes bench.py quotes the layer-wise numbers on (bench.LW_SHARDS, seeded
    host inputs): NPG.train_from_paths on 500 k x (376, 17, 256^2, 25 CG) -- mjrl/algos/npg_cg.py:91-163 -- and
    DAPG.train_from_paths on 1M x (39, 28, 512^2) + 5 000 demonstration rows -- mjrl/algos/dapg.py:54-141.  Stored: strided
    samples of the gradient and of the update step, alpha / KL / surrogate improvement, a digest of the inputs; then (unless
    MJX_GOLDEN_NO_F64=1) the fp64 oracle's step on the same rows and the reference's distance from it."""
    import bench
    c = bench.LW_SHARDS[key]
    name = c["fixture"]
    n, m, hidden, T = c["n"], c["m"], c["hidden"], c["T"]
    inp = bench.lw_shard_inputs(key)
    theta0 = inp["theta"]
    N = inp["obs"].shape[0]
    spec = EnvSpec(n, m, T)
    pol = MLP(spec, hidden_sizes=hidden, seed=1, init_log_std=-0.5)
    pol.set_param_values(theta0.copy())
    assert np.array_equal(pol.get_param_values(), theta0)
    obs64, act64 = inp["obs"].astype(np.float64), inp["act"].astype(np.float64)
    paths = [dict(observations=obs64[i * T:(i + 1) * T], actions=act64[i * T:(i + 1) * T], rewards=np.zeros(T),
                  advantages=inp["adv"][i * T:(i + 1) * T].copy(), terminated=False) for i in range(c["n_traj"])]
    kw = dict(FIM_invert_args={'iters': c["cg_iters"], 'damping': 1e-4}, save_logs=True)
    adv_w = (inp["adv"] - np.mean(inp["adv"])) / (np.std(inp["adv"]) + 1e-6)
    out = dict(N=N, n=n, m=m, hidden=np.array(hidden, dtype=np.int64), cg_iters=c["cg_iters"], damping=1e-4, algo=c["algo"],
               stride=STRIDE, digest=bench.lw_inputs_digest(inp), torch_threads=torch.get_num_threads(), key=key)
    if c["algo"] == "npg":
        agent = NPG(None, pol, None, normalized_step_size=0.05, **kw)
        out["step"] = 0.05
        g_args = (obs64, act64, adv_w)
        coef = 1.0
    else:
        Td = 200
        dpaths = [dict(observations=inp["demo_obs"][i * Td:(i + 1) * Td].astype(np.float64),
                       actions=inp["demo_act"][i * Td:(i + 1) * Td].astype(np.float64)) for i in range(c["demo_rows"] // Td)]
        agent = DAPG(None, pol, None, demo_paths=dpaths, kl_dist=c["kl_dist"], lam_0=c["lam_0"], lam_1=0.95, **kw)
        out.update(kl_dist=c["kl_dist"], lam_0=c["lam_0"], lam_1=0.95, demo_rows=c["demo_rows"])
        all_adv = 1e-2 * np.concatenate([adv_w / (np.std(adv_w) + 1e-8), c["lam_0"] * np.ones(c["demo_rows"])])
        g_args = (np.concatenate([obs64, inp["demo_obs"].astype(np.float64)]), np.concatenate([act64, inp["demo_act"].astype(np.float64)]), all_adv)
        coef = all_adv.shape[0] / adv_w.shape[0]
    agent.logger = DataLog()
    t0 = time.time()
    g = coef * agent.flat_vpg(*g_args)
    print(name, "reference flat_vpg %.1f s" % (time.time() - t0), flush=True)
    del g_args
    t0 = time.time()
    agent.train_from_paths(paths)
    out["reference_update_seconds"] = time.time() - t0
    log = agent.logger.log
    step_ref = pol.get_param_values().astype(np.float64) - theta0.astype(np.float64)
    out.update(alpha=log['alpha'][-1], kl=log['kl_dist'][-1], surr_improvement=log['surr_improvement'][-1],
               vpg_sub=np.asarray(g)[::STRIDE].astype(np.float32), vpg_norm=float(np.linalg.norm(np.asarray(g, np.float64))),
               update_step_sub=step_ref[::STRIDE].astype(np.float32), update_step_norm=float(np.linalg.norm(step_ref)),
               update_step_probe=float(np.dot(step_ref, np.random.RandomState(PROBE_SEED).randn(theta0.size))))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "reference train_from_paths %.1f s: N %d d %d alpha %r kl %r surr_improvement %r"
          % (out["reference_update_seconds"], N, theta0.size, out["alpha"], out["kl"], out["surr_improvement"]), flush=True)
    if os.environ.get("MJX_GOLDEN_NO_F64") == "1":
        return
    t0 = time.time()
    th64 = theta0.astype(np.float64)
    if c["algo"] == "npg":
        r64 = O.npg_update(th64, obs64, act64, adv_w, n, m, hidden, cg_iters=c["cg_iters"], damping=1e-4, delta=0.05)
    else:
        r64 = O.dapg_update(th64, obs64, act64, adv_w, inp["demo_obs"].astype(np.float64), inp["demo_act"].astype(np.float64), n, m, hidden,
                            cg_iters=c["cg_iters"], damping=1e-4, kl_dist=c["kl_dist"], lam_0=c["lam_0"], lam_1=0.95, iter_count=0.0)
    step64 = r64["new_params"] - th64
    out.update(update_step_f64_sub=step64[::STRIDE], alpha_f64=r64["alpha"], kl_f64=float(r64["kl"]),
               err_ref_vs_f64_update_step=rel(step_ref, step64), err_ref_vs_f64_vpg=rel(g, r64["vpg"]), oracle_seconds=time.time() - t0)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "fp64 oracle %.0f s: reference vs fp64 step %.2e vpg %.2e" % (out["oracle_seconds"], out["err_ref_vs_f64_update_step"],
                                                                              out["err_ref_vs_f64_vpg"]), flush=True)


def inputnorm_truth(name):
    """fp64 truth for the input_normalization fixture of make_golden.py (npg_inputnorm_32x32: the update runs with
    theta_new == theta_old but an input transform on policy.model only, npg_cg.py:101-107, i.e. the GENERAL Hessian):
    the reference's op sequence (oracle/torch_port.py) in double on the same inputs, and the reference's distance
    from it."""
    from oracle.torch_port import TorchPolicy
    g = dict(np.load(os.path.join(HERE, "npg_inputnorm_32x32.npz")))
    n, m, hidden = int(g["n"]), int(g["m"]), tuple(int(h) for h in g["hidden"])
    paths = synth.make_paths(int(g["n_traj"]), int(g["T"]), n, m, seed=int(g["path_seed"]))
    fake_advantages(paths, int(g["adv_seed"]))
    obs, act, adv = cat(paths, "observations"), cat(paths, "actions"), cat(paths, "advantages")
    adv_w = (adv - np.mean(adv)) / (np.std(adv) + 1e-6)
    th = g["theta0"].astype(np.float64)
    trn = O.Transforms(n, m, g["final_in_shift"], g["final_in_scale"])
    pol = TorchPolicy(th, n, m, hidden, theta_old=th, tr_new=trn, tr_old=None, dtype=np.float64)
    surr_before = float(pol.surrogate(obs, act, adv_w))
    grad = pol.vpg(obs, act, adv_w)
    x = O.cg_solve(lambda p: pol.hvp(obs, act, p, 1e-4), grad, int(g["cg_iters"]))
    alpha = np.sqrt(np.abs(float(g["step"]) / (np.dot(grad, x) + 1e-20)))
    new = th + alpha * x
    new[-m:] = np.maximum(new[-m:], -3.0)
    pol.set_new(new)
    out = dict(update_step_f64=new - th, alpha_f64=float(alpha), kl_f64=float(pol.kl(obs, act)),
               surr_improvement_f64=float(pol.surrogate(obs, act, adv_w)) - surr_before,
               err_ref_vs_f64_update_step=rel(g["new_params"].astype(np.float64) - th, new - th))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "alpha", out["alpha_f64"], float(g["alpha"]), "kl", out["kl_f64"], float(g["kl"]),
          "reference vs fp64 step %.2e" % out["err_ref_vs_f64_update_step"], flush=True)


CASES = {
    "npg_inputnorm_32x32_f64": lambda: inputnorm_truth("npg_inputnorm_32x32_f64"),
    "npg_cfg4_wide": lambda: big_case("npg_cfg4_wide", 376, 17, (256, 256), 800, 250, 25, "npg"),
    "dapg_cfg5_wide": lambda: big_case("dapg_cfg5_wide", 39, 28, (512, 512), 1500, 200, 10, "dapg", kl_dist=0.025, demo=(25, 200)),
    "bench_cfg2_1m": lambda: bench_case("bench_cfg2_1m"),
    "bench_ref_1m": lambda: bench_reference_case("bench_ref_1m"),
    "npg_cfg4_shard": lambda: shard_case("configs3_humanoid_256x256"),
    "dapg_cfg5_shard": lambda: shard_case("configs4_adroit_512x512"),
}

if __name__ == "__main__":
    for c in (sys.argv[1:] or list(CASES)):
        CASES[c]()
