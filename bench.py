#!/usr/bin/env python
"""bench.py -- NPG updates/sec on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): HalfCheetah shapes obs=17 / act=6, 64x64 tanh MLP,
NPG with 10 CG iterations (damping 1e-4, normalized step 0.05) over a 1M-timestep batch
(1000 synthetic trajectories x 1000 steps), STRONG scaling: the 1M batch is sharded by
trajectory over the N ranks, one RCCL all-reduce of the flat gradient and one per CG
iteration.

One "step" = one device-resident NPG.train_from_paths (mjrl/algos/npg_cg.py:108-142) with the
whitened batch already in HBM:  K1 surrogate+VPG -> CG(10 x [K2 FVP + vector update]) ->
alpha from g.x (host scalar) -> theta += alpha x -> K3 surrogate+KL (host scalars) -> old := new.
Prints ONE JSON line (rank 0) with the `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

N_OBS, N_ACT, HIDDEN = 17, 6, (64, 64)
N_TRAJ, T = 1000, 1000
CG_ITERS, DAMPING, STEP = 10, 1e-4, 0.05
FP32_MFMA_PEAK_TF = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBS = 8000.0


def synth_shard(rank, world, n_traj=N_TRAJ, T_=T):
    """this rank's trajectories of the global synthetic batch (seeded per trajectory)."""
    lo, hi = rank * n_traj // world, (rank + 1) * n_traj // world
    obs = np.empty(((hi - lo) * T_, N_OBS), np.float32)
    act = np.empty(((hi - lo) * T_, N_ACT), np.float32)
    adv = np.empty((hi - lo) * T_, np.float64)
    for i, tr in enumerate(range(lo, hi)):
        rng = np.random.RandomState(1000 + tr)
        obs[i * T_:(i + 1) * T_] = rng.randn(T_, N_OBS)
        act[i * T_:(i + 1) * T_] = rng.randn(T_, N_ACT)
        adv[i * T_:(i + 1) * T_] = rng.randn(T_)
    return obs, act, adv


def initial_params():
    """random-init weights of the benchmark architecture (nn.Linear-style U(-1/sqrt(fan_in), 1/sqrt(fan_in)), last layer
    x 1e-2, log_std -0.5) + the 0.1 N(0,1) perturbation of SURVEY 8d.  (No checkpoints, no network: "data": "synthetic".)"""
    rng = np.random.RandomState(1)
    sizes = (N_OBS,) + tuple(HIDDEN) + (N_ACT,)
    flat = []
    for i in range(len(sizes) - 1):
        k = 1.0 / np.sqrt(sizes[i])
        W, b = rng.uniform(-k, k, (sizes[i + 1], sizes[i])), rng.uniform(-k, k, sizes[i + 1])
        if i == len(sizes) - 2:
            W, b = 1e-2 * W, 1e-2 * b
        flat += [W.ravel(), b]
    flat.append(np.full(N_ACT, -0.5))
    th = np.concatenate(flat).astype(np.float32)
    return (th + 0.1 * np.random.RandomState(1).randn(th.size)).astype(np.float32)


# the per-GPU shards of the 8-GPU configs (BASELINE configs[3] / [4]) the layer-wise numbers are quoted on: seeded host
# inputs, so that tests/golden/make_golden_big.py can run the UNMODIFIED reference on exactly these rows
LW_SHARDS = {
    "configs3_humanoid_256x256": dict(n=376, m=17, hidden=(256, 256), n_traj=500, T=1000, cg_iters=25, algo="npg", seed=31,
                                      fixture="npg_cfg4_shard"),
    "configs4_adroit_512x512": dict(n=39, m=28, hidden=(512, 512), n_traj=5000, T=200, cg_iters=10, algo="dapg", seed=41,
                                    demo_rows=5000, kl_dist=0.025, lam_0=1e-2, fixture="dapg_cfg5_shard"),
}


def lw_initial_params(n, m, hid):
    """nn.Linear-style random init (last layer x 1e-2, log_std -0.5) + 0.02 N(0,1), like initial_params()"""
    rng = np.random.RandomState(1)
    sizes = (n,) + tuple(hid) + (m,)
    flat = []
    for i in range(len(sizes) - 1):
        k = 1.0 / np.sqrt(sizes[i])
        flat += [rng.uniform(-k, k, (sizes[i + 1], sizes[i])).ravel() * (1e-2 if i == len(sizes) - 2 else 1.0), rng.uniform(-k, k, sizes[i + 1])]
    flat.append(np.full(m, -0.5))
    th = np.concatenate(flat).astype(np.float32)
    return (th + 0.02 * np.random.RandomState(1).randn(th.size)).astype(np.float32)


def lw_shard_inputs(name):
    """-> dict(theta, obs (N, n) f32, act (N, m) f32, adv (N,) f64 un-whitened[, demo_obs, demo_act]) of one LW_SHARDS entry,
    from a PCG64 stream (float32 ziggurat draws: 0.2 G samples in a second or two on one core)"""
    c = LW_SHARDS[name]
    N = c["n_traj"] * c["T"]
    g = np.random.Generator(np.random.PCG64(c["seed"]))
    out = dict(theta=lw_initial_params(c["n"], c["m"], c["hidden"]),
               obs=g.standard_normal((N, c["n"]), dtype=np.float32), act=g.standard_normal((N, c["m"]), dtype=np.float32),
               adv=g.standard_normal(N))
    if c.get("demo_rows"):
        out["demo_obs"] = g.standard_normal((c["demo_rows"], c["n"]), dtype=np.float32)
        out["demo_act"] = g.standard_normal((c["demo_rows"], c["m"]), dtype=np.float32)
    return out


def lw_inputs_digest(inp):
    """a few numbers that pin the generated rows (the fixtures carry them: a different numpy stream must not pass silently)"""
    return np.array([float(inp["obs"][::997].astype(np.float64).sum()), float(inp["act"][::997].astype(np.float64).sum()),
                     float(inp["adv"][::997].sum()), float(inp["theta"].astype(np.float64).sum())])


def cpu_baseline(theta0, sample_traj):
    """The CPU path timed on this box's host cores, rank 0, N = 1.

    kind "reference": the UNMODIFIED reference's NPG.train_from_paths (mjrl/algos/npg_cg.py:91-163) itself, on the FULL
    1M-timestep batch of the metric -- imported from /root/reference where that exists, else from the bytecode
    oracle/ref_stage.py compiled from it into oracle/_ref/ (what travels to the GPU box).  torch's intra-op thread count is
    calibrated first on a 200k-timestep slice -- a size that predicts the 1M run (r06; the 40k slice of r05 sat in cache) --
    over {8, 16, 32, 64} threads (many-core hosts are slower at their default of one thread per core on these skinny
    matrices), so the baseline is the CPU's best showing; the whole table is printed.
    kind "port" (only when the reference is not staged): oracle/torch_port.py, the same op sequence, on a slice scaled linearly."""
    import torch
    sys.path.insert(0, ROOT)
    from oracle import ref_loader
    ref_root = ref_loader.install()
    if ref_root is None:
        return cpu_baseline_port(theta0, sample_traj)
    import contextlib
    import io
    from mjrl.algos.npg_cg import NPG
    from mjrl.policies.gaussian_mlp import MLP
    from mjrl.utils.gym_env import EnvSpec
    obs, act, adv = synth_shard(0, 1)
    obs, act = obs.astype(np.float64), act.astype(np.float64)    # the reference holds fp64 rollouts
    spec = EnvSpec(N_OBS, N_ACT, T)

    def paths_of(n_traj):
        return [dict(observations=obs[i * T:(i + 1) * T], actions=act[i * T:(i + 1) * T], rewards=np.zeros(T),
                     advantages=adv[i * T:(i + 1) * T].copy(), terminated=False) for i in range(n_traj)]

    def one_update(paths):
        pol = MLP(spec, hidden_sizes=HIDDEN, seed=1, init_log_std=-0.5)
        pol.set_param_values(theta0.copy())
        agent = NPG(None, pol, None, normalized_step_size=STEP, FIM_invert_args={'iters': CG_ITERS, 'damping': DAMPING}, save_logs=False)
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            agent.train_from_paths(paths)
        return time.time() - t0, pol.get_param_values()

    default_threads = torch.get_num_threads()
    one_update(paths_of(10))                                      # warm-up
    cal, cal_paths = {}, paths_of(200)
    for k in sorted({k for k in (8, 16, 32, 64) if k <= default_threads} or {default_threads}):
        # (torch's default -- one thread per core: 128 on the GPU boxes -- took 27 s on this slice against 4.2 s at 16: not a candidate)
        torch.set_num_threads(k)
        cal[k] = one_update(cal_paths)[0]
    best = min(cal, key=cal.get)
    torch.set_num_threads(best)
    n_traj = N_TRAJ if sample_traj <= 0 or sample_traj >= N_TRAJ else sample_traj          # 0 (the default): the whole batch
    dt, theta_ref = one_update(paths_of(n_traj))
    torch.set_num_threads(default_threads)
    n = n_traj * T
    vs_fixture = None
    fr = os.path.join(ROOT, "tests", "golden", "bench_ref_1m.npz")
    if n_traj == N_TRAJ and os.path.exists(fr):
        # the run just timed against the fixture the GPU path is held to (the same reference on the same batch, made in the build
        # container on 8 threads): they differ by thread-count summation order only
        rs = np.load(fr)["npg_new_params"].astype(np.float64) - theta0
        vs_fixture = float(np.linalg.norm((theta_ref.astype(np.float64) - theta0) - rs) / np.linalg.norm(rs))
    return dict(value=(n / float(N_TRAJ * T)) / dt, unit="updates/s", cores=int(best), kind="reference", step_rel_l2_vs_fixture=vs_fixture,
                sample="%s: the unmodified reference's NPG.train_from_paths (mjrl/algos/npg_cg.py:91-163, imported from %s) on %d "
                       "timesteps (%d trajectories) of the metric's own batch in %.2f s, torch-CPU with %d intra-op threads (best of %s "
                       "seconds on a 200k-timestep calibration slice)%s"
                       % ("the FULL 1M-timestep batch" if n_traj == N_TRAJ else "a slice", "oracle/_ref (bytecode staged by oracle/ref_stage.py)"
                          if ref_root != "/root/reference" else ref_root, n, n_traj, dt, best, {k: round(v, 2) for k, v in cal.items()},
                          "" if n_traj == N_TRAJ else ", scaled linearly to 1M"),
                seconds=dt, nproc=os.cpu_count(), reference_new_params_norm=float(np.linalg.norm(theta_ref.astype(np.float64) - theta0)),
                calibration={"timesteps": 200 * T, "seconds_by_threads": {str(k): round(v, 3) for k, v in cal.items()},
                             "predicted_full_size_seconds": round(cal[best] * N_TRAJ / 200.0, 2)})


def cpu_baseline_port(theta0, sample_traj):
    """fallback when the reference is not staged: its CPU algorithm as a torch-autograd port (oracle/torch_port.py, validated
    against the reference's wall time and step: tests/golden/cpu_port_vs_reference.json) on a bounded slice."""
    import torch
    from oracle import torch_port
    sample_traj = 200 if sample_traj <= 0 else sample_traj       # (the port is a stand-in: a 200-trajectory slice unless told otherwise)
    obs, act, adv = synth_shard(0, N_TRAJ // sample_traj)        # first `sample_traj` trajectories
    obs, act = obs.astype(np.float64), act.astype(np.float64)    # the reference holds fp64 rollouts
    adv = (adv - adv.mean()) / (adv.std() + 1e-6)
    kw = dict(cg_iters=CG_ITERS, damping=DAMPING, delta=STEP)
    default_threads = torch.get_num_threads()
    cal = {}
    ncal = 40000
    for k in sorted({8, 16, 32, 64, default_threads}):
        if k > default_threads:
            continue
        torch.set_num_threads(k)
        torch_port.npg_update(theta0, obs[:10000], act[:10000], adv[:10000], N_OBS, N_ACT, HIDDEN, **kw)   # warm-up
        t0 = time.time()
        torch_port.npg_update(theta0, obs[:ncal], act[:ncal], adv[:ncal], N_OBS, N_ACT, HIDDEN, **kw)
        cal[k] = time.time() - t0
    best = min(cal, key=cal.get)
    torch.set_num_threads(best)
    t0 = time.time()
    torch_port.npg_update(theta0, obs, act, adv, N_OBS, N_ACT, HIDDEN, **kw)
    dt = time.time() - t0
    torch.set_num_threads(default_threads)
    n = obs.shape[0]
    ups = (n / float(N_TRAJ * T)) / dt                            # linear-in-N extrapolation to 1M
    kind, validation = "port", None
    vf = os.path.join(ROOT, "tests", "golden", "cpu_port_vs_reference.json")
    if os.path.exists(vf):
        validation = json.load(open(vf))
    return dict(value=ups, unit="updates/s", cores=int(best), kind=kind,
                validation=None if validation is None else dict(
                    what="wall time of this port / wall time of the unmodified reference's NPG.train_from_paths on the same "
                         "%d-timestep batch, best of 3 each, %d threads, build container (tests/golden/make_cpu_port_validation.py)"
                         % (validation["timesteps"], validation["threads"]),
                    port_over_reference=validation["port_over_reference"], step_rel_difference=validation["step_rel_difference"]),
                sample="%d-timestep slice (%d traj) of the 1M batch, one NPG update in %.2f s on torch-CPU "
                       "(autograd double-backward HVP, as the reference) with %d intra-op threads (best of %s on a "
                       "%d-sample calibration), scaled linearly to 1M; the reference itself is not staged here (oracle/_ref missing)"
                       % (n, sample_traj, dt, best, {k: round(v, 2) for k, v in cal.items()}, ncal),
                seconds=dt, nproc=os.cpu_count())


def secondary_measurements(eng, theta0, theta0_dev, ref=None, full_size=True):
    """Measured AFTER the primary timed region, on the same GPU (N = 1):
    * BASELINE configs[2]: one TRPO update (KL line search, mjrl/algos/trpo.py:100-126) on the same 1M batch -- K1, CG,
      then backtracking evaluations of K3 until KL < kl_dist (kl_dist = 0.025: the first two step lengths are rejected), the
      line search decided on the device (mjx_trpo_update);
    * the layer-wise path at the per-GPU shard sizes of configs[3] / [4] (the 8-GPU configs this 1-GPU run cannot time
      as a whole): HIP-event time of the Fisher-vector-product chain and its rate against the fp32-MFMA peak."""
    import torch
    from mjrl_amd._lib import check
    from mjrl_amd.engine import UpdateEngine
    out = {}
    # ---- TRPO
    kl_dist, trials_log, first = 0.025, [], {}

    def trpo_update():
        # what TRPO.train_from_paths runs: mjx_trpo_update (K1, CG, step length, line-search trials three at a time with the
        # accept / shrink decision on the device), one read-back per batch of trials
        res = eng.trpo_update(CG_ITERS, DAMPING, 2.0 * kl_dist, kl_dist, -3.0)
        assert res is not None and res["accepted"]
        trials_log.append(res["trials"])
        if not first:
            first.update(res, surr_before=eng.deferred()["surr_before"], step=eng.theta_new.cpu().numpy().astype(np.float64) - theta0)
        eng.theta_new.copy_(theta0_dev); eng.old_is_new = True; eng._bind_policy()
    trpo_update()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        trpo_update()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 5
    out["trpo_configs2"] = {"updates_per_s": 1e3 / ms, "ms_per_update": ms, "kl_dist": kl_dist, "line_search_trials": trials_log[-1],
                            "workload": "BASELINE configs[2]: the same 1M-timestep batch and 64x64 policy, TRPO with KL line search"}
    if ref is not None and abs(float(ref["trpo_kl_dist"]) - kl_dist) < 1e-12:
        rs = ref["trpo_new_params"].astype(np.float64) - theta0
        si = first["surr_after"] - first["surr_before"]
        drift = {"alpha": abs(first["alpha"] - float(ref["trpo_alpha"])) / float(ref["trpo_alpha"]),
                 "kl": abs(first["kl"] - float(ref["trpo_kl"])) / float(ref["trpo_kl"]),
                 "surr_improvement": abs(si - float(ref["trpo_surr_improvement"])) / float(ref["trpo_surr_improvement"]),
                 "step_rel_l2": float(np.linalg.norm(first["step"] - rs) / np.linalg.norm(rs))}
        same_trials = int(first["trials"]) == int(ref["trpo_trials"])
        out["trpo_configs2"]["check_vs_reference"] = {
            "rel_error": drift, "trials": int(first["trials"]), "reference_trials": int(ref["trpo_trials"]), "bars": {k: 1e-5 for k in drift},
            "fixture": "tests/golden/bench_ref_1m.npz",
            "what": "the unmodified reference's TRPO.train_from_paths (mjrl/algos/trpo.py:56-146) on this batch",
            "failed": bool(max(drift.values()) > 1e-5 or not same_trials)}
    # ---- layer-wise FVP at the shard sizes of the 8-GPU configs, on seeded host rows (lw_shard_inputs) the UNMODIFIED reference
    # was run on as well (tests/golden/{npg_cfg4_shard,dapg_cfg5_shard}.npz, make_golden_big.py): the whole update of the shard
    # is checked against it at the north-star's 1e-5 like the primary metric
    lw = {}
    for name, cfg in LW_SHARDS.items():
        n, m, hid = cfg["n"], cfg["m"], cfg["hidden"]
        inp = lw_shard_inputs(name)
        N = inp["obs"].shape[0]
        th = inp["theta"]
        adv_w = (inp["adv"] - inp["adv"].mean()) / (inp["adv"].std() + 1e-6)          # batch_reinforce.py:185 / dapg.py:60, fp64 on the host
        sizes = (n,) + tuple(hid) + (m,)
        e = UpdateEngine(n, m, hid)
        ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
        e.set_policy(th, th, ident, ident)
        e.set_batch(inp["obs"], inp["act"], adv_w)
        grad = e.surr_vpg()[0].clone()
        e.fvp(grad)
        torch.cuda.synchronize()
        prof = (ctypes.c_double * 2)()

        def timed_products(k):
            check(e.lib.mjx_profile_enable(e.ctx, 1))
            for _ in range(k):
                e.fvp(grad)
            check(e.lib.mjx_profile_read(e.ctx, prof))
            check(e.lib.mjx_profile_enable(e.ctx, 0))
            return prof[0] / prof[1]
        ms_first = timed_products(4)          # products 2-5 after K1 (r01-r05's figure: clocks / caches still settling)
        ms = timed_products(8)                # products 6-13: what the 10-25 products of a solve run at
        P = sum(sizes[i] * sizes[i + 1] for i in range(len(sizes) - 1))
        flop = 2 * (4 * P - 2 * n * hid[0]) * N
        ab = lw_switch_ab(e, grad, flop)
        # one whole NPG update of this shard (K1, the config's CG iterations, step, K3) through the one-call entry point
        cg_iters = cfg["cg_iters"]
        sa, kl = e.npg_update(cg_iters, 1e-4, 0.05, -3.0)
        late = e.deferred()
        got = dict(alpha=late["alpha"], kl=kl, surr_improvement=sa - late["surr_before"],
                   step=e.theta_new.cpu().numpy().astype(np.float64) - th)
        e.set_policy(th, th, ident, ident)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.npg_update(cg_iters, 1e-4, 0.05, -3.0)
        torch.cuda.synchronize()
        upd_ms = 1e3 * (time.perf_counter() - t0)
        lw[name] = {"rows": N, "fvp_ms": ms, "TFLOPs": flop / (ms * 1e-3) / 1e12, "frac_of_fp32_mfma_peak": flop / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
                    "flop_per_fvp": flop, "kernels": "k_gemm_p (persistent tangent / delta products, csrc/lw_gemm_p.h) + k_gemm<128,256> / <128,128> weight gradients + k_lw_head (one-pass output layer, csrc/lw_head.h)",
                    "timed": "8 products (the 6th to 13th after K1: a solve runs 10-25), HIP events around the whole chain of one product",
                    "fvp_ms_first_four_after_K1": ms_first, "frac_first_four_after_K1": flop / (ms_first * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
                    "npg_update_ms": upd_ms, "cg_iters": cg_iters, "inputs": "seeded host rows (bench.lw_shard_inputs, PCG64 seed %d)" % cfg["seed"],
                    "switch_ab_same_process": ab}
        if cfg["algo"] == "dapg":
            # ... and the algorithm configs[4] names: one DAPG update (mjrl/algos/dapg.py:92-121) of the same shard with 25 x 200
            # demonstration steps appended, through the one-call entry point mjx_dapg_update (K1 over [on-policy ; demos],
            # gradient x N_all / N_on, Fisher / surrogate / KL on the on-policy prefix, 10 CG iterations, step, K3)
            Nd = cfg["demo_rows"]
            obs_all = torch.cat([e.obs, torch.from_numpy(inp["demo_obs"]).to(e.device)])
            act_all = torch.cat([e.act, torch.from_numpy(inp["demo_act"]).to(e.device)])
            all_adv = 1e-2 * np.concatenate([adv_w / (np.std(adv_w) + 1e-8), cfg["lam_0"] * np.ones(Nd)])          # dapg.py:65-70, iteration 0
            dapg_ms = []
            for rep in range(2):
                e.set_policy(th, th, ident, ident)
                e.set_batch(obs_all, act_all, all_adv)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = e.dapg_update(cg_iters, 1e-4, 2.0 * cfg["kl_dist"], -3.0, N, adv_w, N_on_global=N)
                torch.cuda.synchronize()
                dapg_ms.append(1e3 * (time.perf_counter() - t0))
                if rep == 0:
                    late = e.deferred()
                    got = dict(alpha=late["alpha"], kl=res[1], surr_improvement=res[0] - late["surr_before"],
                               step=e.theta_new.cpu().numpy().astype(np.float64) - th)
            assert res is not None and np.isfinite(res[1]) and res[1] > 0
            lw[name].update(dapg_update_ms=dapg_ms[-1], dapg_kl=res[1], dapg_demo_rows=Nd,
                            dapg="one mjx_dapg_update call: [1M on-policy ; 5 000 demonstration] rows, kl_dist 0.025")
        fx = os.path.join(ROOT, "tests", "golden", cfg["fixture"] + ".npz")
        if os.path.exists(fx):
            g = np.load(fx)
            S = int(g["stride"])
            same_rows = bool(np.allclose(lw_inputs_digest(inp), g["digest"], rtol=1e-12, atol=1e-9))
            rs = g["update_step_sub"].astype(np.float64)
            drift = {"alpha": abs(got["alpha"] - float(g["alpha"])) / float(g["alpha"]),
                     "kl": abs(got["kl"] - float(g["kl"])) / float(g["kl"]),
                     "surr_improvement": abs(got["surr_improvement"] - float(g["surr_improvement"])) / abs(float(g["surr_improvement"])),
                     "step_rel_l2": float(np.linalg.norm(got["step"][::S] - rs) / np.linalg.norm(rs))}
            bars = {"step_rel_l2": 1e-5, "alpha": 1e-5, "kl": 1e-4, "surr_improvement": 1e-4}
            chk = {"rel_error": drift, "bars": bars, "fixture": "tests/golden/%s.npz" % cfg["fixture"], "same_rows_as_the_fixture": same_rows,
                   "bars_are": "step direction and step length carry the north-star's 1e-5; KL and surrogate improvement are fp32 sums over 0.5-1M "
                               "samples on BOTH sides (the reference's own value moves by ~1e-5 with its thread count): 1e-4, as in the tests",
                   "what": "the unmodified reference's %s.train_from_paths on these %d rows (%.0f s of CPU; every %d-th entry of the step)"
                           % ("DAPG" if cfg["algo"] == "dapg" else "NPG", N, float(g["reference_update_seconds"]), S),
                   # the scalars are fp32 sums over 0.5-1M samples on both sides: the step DIRECTION carries the north-star's bar,
                   # alpha with it; KL / surrogate improvement are reported (and asserted at 1e-4 in the tests)
                   "failed": bool(not same_rows or any(drift[k] > bars[k] for k in bars))}
            if "err_ref_vs_f64_update_step" in g.files:
                chk["reference_vs_fp64_oracle_step_rel_l2"] = float(g["err_ref_vs_f64_update_step"])
            lw[name]["check_vs_reference"] = chk
        e.close()
        del e, inp
        torch.cuda.empty_cache()
    out["roofline_lw"] = lw
    if full_size:
        # BASELINE configs[3] / [4] at their stated size on this one GPU (4M x 376-256^2-17, 25 CG; 8M x 39-512^2-28 + demos, DAPG)
        out["full_size"] = {"configs3": full_size_case("configs3_humanoid_256x256"), "configs4": full_size_case("configs4_adroit_512x512")}
    out.update(user_level_measurements())
    return out


def lw_switch_ab(e, v, flop, switches=("MJX_LW_HEAD8", "MJX_LW_HEAD_KTRIM", "MJX_LW_PERSIST"), rounds=3, per=3):
    """Same-process, same-box A/B of the layer-wise chain's switches (the boxes' spread, 0.649-0.692 at
    configs[3], is wider than the effects to adjudicate): for every switch, `rounds` alternations of [default ; switch = 0], `per`
    Fisher-vector products each between HIP events (libmjx re-reads its MJX_LW_* switches at every launch since r06).
    -> {switch: {on_ms, off_ms, off_over_on}} + the default chain's median over all its passes."""
    import torch
    from mjrl_amd._lib import check

    def timed():
        torch.cuda.synchronize()
        check(e.lib.mjx_profile_enable(e.ctx, 1))
        for _ in range(per):
            e.fvp(v)
        prof = (ctypes.c_double * 2)()
        check(e.lib.mjx_profile_read(e.ctx, prof))
        check(e.lib.mjx_profile_enable(e.ctx, 0))
        return prof[0] / prof[1]
    out, base_all = {}, []
    for sw in switches:
        on, off = [], []
        prev = os.environ.get(sw)
        try:
            for _ in range(rounds):
                os.environ.pop(sw, None)
                if prev is not None:
                    os.environ[sw] = prev
                e.fvp(v); on.append(timed())
                os.environ[sw] = "0"
                e.fvp(v); off.append(timed())
        finally:
            os.environ.pop(sw, None)
            if prev is not None:
                os.environ[sw] = prev
        base_all += on
        mon, moff = sorted(on)[len(on) // 2], sorted(off)[len(off) // 2]
        out[sw] = {"on_ms": mon, "off_ms": moff, "off_over_on": moff / mon, "on_ms_each": on, "off_ms_each": off,
                   "on_frac_of_fp32_mfma_peak": flop / (mon * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF}
    e.fvp(v)
    torch.cuda.synchronize()
    base_all.sort()
    out["default_chain_ms_median_of_all_passes"] = base_all[len(base_all) // 2]
    out["default_chain_frac_median"] = flop / (base_all[len(base_all) // 2] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF
    out["what"] = "alternating passes in ONE process on ONE box: [switch at its default ; switch = 0] x %d, %d products per pass, HIP events around the chain" % (rounds, per)
    return out


def full_size_case(name, rows=None, shards=8, seed_shift=0, time_it=True):
    """BASELINE configs[3] / configs[4] at their STATED size on ONE MI355X: `shards` x the per-GPU shard of
    LW_SHARDS[name] (4M x 376-256^2-17 with 25 CG iterations; 8M x 39-512^2-28 + 8 x 5 000 demonstration rows, DAPG), resident
    in HBM, rows generated on the device (torch.Generator; what matters here is that the full batch and its shards see the same
    rows) -- the N = 1 anchor of those configs' 1 -> 8 curve.  One engine holds the whole batch: K1, Fisher-vector products (HIP
    events around the chain), one whole update.  Then the SAME rows as `shards` engines of N / shards rows each with the global
    sample count (what the ranks of an 8-GPU job hold): gradient and product must be the sum of the shards' (product <= 1e-6, gradient <= 5e-6: see the bars), and the
    update composed from the shards' sums -- CG on the summed products with the library's own vector kernels, step length,
    K3 sums -- must give the full batch's alpha / KL / step.  Nothing in this function knows a 32-bit row or element index:
    tests/test_gpu_parity.py::test_layerwise_block_beyond_2_31_elements runs it at 4.3M x 512 (2.2e9 elements per activation block).
    rows: total on-policy rows (default: shards x the LW_SHARDS entry)."""
    import torch
    from mjrl_amd._lib import check, ptr
    from mjrl_amd.engine import UpdateEngine
    cfg = LW_SHARDS[name]
    n, m, hid = cfg["n"], cfg["m"], cfg["hidden"]
    S = (cfg["n_traj"] * cfg["T"]) if rows is None else int(rows) // shards
    N = S * shards
    dev = torch.device("cuda", torch.cuda.current_device())
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + cfg["seed"] + seed_shift)
    th = lw_initial_params(n, m, hid)
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    obs = torch.randn((N, n), generator=gen, device=dev, dtype=torch.float32)
    act = torch.randn((N, m), generator=gen, device=dev, dtype=torch.float32)
    adv64 = torch.randn(N, generator=gen, device=dev, dtype=torch.float64)
    adv_w64 = (adv64 - adv64.mean()) / (adv64.std(unbiased=False) + 1e-6)                       # batch_reinforce.py:185
    dapg = cfg["algo"] == "dapg"
    Nd_s = cfg.get("demo_rows", 0) if dapg else 0                                            # demonstration rows per shard
    cg_iters, damping = cfg["cg_iters"], 1e-4
    step_size = 2.0 * cfg["kl_dist"] if dapg else 0.05
    if dapg:
        dobs = torch.randn((Nd_s * shards, n), generator=gen, device=dev, dtype=torch.float32)
        dact = torch.randn((Nd_s * shards, m), generator=gen, device=dev, dtype=torch.float32)
        adv_on = adv_w64.to(torch.float32)
        adv_all_on = (1e-2 * adv_w64 / (adv_w64.std(unbiased=False) + 1e-8)).to(torch.float32)  # dapg.py:65-70, iteration 0
        adv_demo = torch.full((Nd_s * shards,), 1e-2 * cfg["lam_0"], device=dev, dtype=torch.float32)
    else:
        adv_on = adv_w64.to(torch.float32)
    del adv64, adv_w64
    N_all = N + Nd_s * shards
    sizes = (n,) + tuple(hid) + (m,)
    P = sum(sizes[i] * sizes[i + 1] for i in range(len(sizes) - 1))
    flop = 2 * (4 * P - 2 * n * hid[0]) * N

    def bind(e, lo, hi, dlo, dhi):
        """engine `e` <- on-policy rows [lo, hi) (+ demonstration rows [dlo, dhi)), global counts; K1 run -> gradient (a clone)"""
        e.set_policy(th, th, ident, ident)
        if dapg:
            e.set_batch(torch.cat([obs[lo:hi], dobs[dlo:dhi]]), torch.cat([act[lo:hi], dact[dlo:dhi]]),
                        torch.cat([adv_all_on[lo:hi], adv_demo[dlo:dhi]]), N_global=N_all)
        else:
            e.set_batch(obs[lo:hi], act[lo:hi], adv_on[lo:hi], N_global=N)
        g = e.surr_vpg(sync=False)[0].clone()
        if dapg:
            g *= np.float32(N_all / float(N))                                  # dapg.py:97-98 (fp32 product per element, like k_scale_f32)
            e.bind_rows(hi - lo, adv=adv_on[lo:hi], N_global=N)                   # Fisher / surrogate / KL: the on-policy prefix (:92, :103)
        return g

    def rel(a, b):
        a, b = a.double(), b.double()
        return float((a - b).norm() / b.norm())

    out = {"rows": N, "demo_rows": Nd_s * shards, "d": int(th.size), "cg_iters": cg_iters, "algo": cfg["algo"],
           "hbm_GB_inputs": (obs.numel() + act.numel() + adv_on.numel()) * 4 / 1e9}
    # ---- the whole batch on one engine
    e = UpdateEngine(n, m, hid)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g_full = bind(e, 0, N, 0, Nd_s * shards)
    torch.cuda.synchronize(); out["bind_plus_K1_ms"] = 1e3 * (time.perf_counter() - t0)
    v = g_full.clone()
    hv_full = e.fvp(v).clone()
    if time_it:
        torch.cuda.synchronize()
        check(e.lib.mjx_profile_enable(e.ctx, 1))
        for _ in range(3):
            e.fvp(v)
        prof = (ctypes.c_double * 2)()
        check(e.lib.mjx_profile_read(e.ctx, prof))
        check(e.lib.mjx_profile_enable(e.ctx, 0))
        ms = prof[0] / prof[1]
        out.update(fvp_ms=ms, fvp_TFLOPs=flop / (ms * 1e-3) / 1e12, fvp_frac_of_fp32_mfma_peak=flop / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
                   flop_per_fvp=flop)
        out["switch_ab_same_process"] = lw_switch_ab(e, v, flop, switches=("MJX_LW_HEAD8",), rounds=2, per=2)

    def update(eng):
        eng.set_policy(th, th, ident, ident)
        if dapg:                                                                 # back to [on-policy ; demonstrations] with K1's advantages
            eng.set_batch(eng.obs, eng.act, torch.cat([adv_all_on, adv_demo]), N_global=N_all)
            res = eng.dapg_update(cg_iters, damping, step_size, -3.0, N, adv_on, N_on_global=N)
        else:
            res = eng.npg_update(cg_iters, damping, step_size, -3.0)
        late = eng.deferred()
        return dict(alpha=late["alpha"], kl=res[1], surr_improvement=res[0] - late["surr_before"], gdotx=late["gdotx"],
                    step=(eng.theta_new - torch.from_numpy(th).to(dev)).clone())
    full = update(e)
    if time_it:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        update(e)
        torch.cuda.synchronize(); out["update_ms"] = 1e3 * (time.perf_counter() - t0)
        out["updates_per_s"] = 1e3 / out["update_ms"]
    out["hbm_GB_in_use_full_batch"] = torch.cuda.mem_get_info(dev)[1] / 1e9 - torch.cuda.mem_get_info(dev)[0] / 1e9
    e.close()
    del e
    torch.cuda.empty_cache()
    # ---- the same rows as `shards` engines (what the ranks of the 8-GPU job hold), composed by hand
    engs, g_sum = [], None
    for k in range(shards):
        ek = UpdateEngine(n, m, hid)
        gk = bind(ek, k * S, (k + 1) * S, k * Nd_s, (k + 1) * Nd_s)
        g_sum = gk if g_sum is None else g_sum + gk
        engs.append(ek)
    hv_sum = None
    for ek in engs:
        h = ek.fvp(v).clone()
        hv_sum = h if hv_sum is None else hv_sum + h
    out["shard_sum_vs_full"] = {"gradient_rel_l2": rel(g_sum, g_full), "fvp_rel_l2": rel(hv_sum, hv_full), "shards": shards,
                                "bars": {"gradient_rel_l2": 5e-6, "fvp_rel_l2": 1e-6},
                                "bars_are": "the product is a sum of positive semi-definite terms: two summation orders agree to ~1e-7 (bar 1e-6).  The gradient "
                                            "is an advantage-weighted sum of zero-mean terms over 4-8M rows in fp32 chains of 1 024 samples: its two orders differ by "
                                            "1-3e-6 (measured 1.3e-6 / 2.7e-6 / 1.8e-6 at 4M / 8M / 4.3M rows; bar 5e-6) -- an index that wrapped would show as O(1)"}
    # K3 at theta_old (DAPG's surr_before on the on-policy rows), CG on the summed products, step, K3
    e0 = engs[0]
    b = g_sum.contiguous()
    e0.backend.cg_init(b)
    Ap, tmp = torch.empty_like(b), torch.empty_like(b)
    p_ptr = ctypes.c_void_p(e0.lib.mjx_cg_p(e0.ctx))
    for _ in range(cg_iters):
        Ap.zero_()
        for ek in engs:
            check(ek.lib.mjx_fvp(ek.ctx, p_ptr, ptr(tmp), ek.stream()))
            Ap += tmp
        e0.backend.cg_step(Ap, damping, 1e-10)
    x, bdotx = torch.empty_like(b), torch.zeros(1, dtype=torch.float64, device=dev)
    e0.backend.cg_finish(b, x, bdotx)
    gdotx = float(bdotx.item())
    alpha = float(np.sqrt(abs(step_size / (gdotx + 1e-20))))
    th_new = (torch.from_numpy(th).to(dev) + np.float32(alpha) * x)
    th_new[-m:] = torch.clamp(th_new[-m:], min=-3.0)
    th_new_h = th_new.cpu().numpy()
    sa = kl = 0.0
    for ek in engs:
        ek.set_policy(th_new_h, th, ident, ident)
        s_, k_ = ek.eval_surr_kl()
        sa += s_; kl += k_
    comp = dict(alpha=alpha, kl=kl, step=th_new - torch.from_numpy(th).to(dev))
    out["update_vs_shard_composition"] = {
        "alpha_rel": abs(full["alpha"] - comp["alpha"]) / comp["alpha"], "kl_rel": abs(full["kl"] - comp["kl"]) / comp["kl"],
        "step_rel_l2": rel(full["step"], comp["step"]), "alpha": full["alpha"], "kl": full["kl"],
        "bars": {"alpha_rel": 1e-5, "kl_rel": 1e-4, "step_rel_l2": 3e-5},
        "bars_are": "two summation orders of the SAME fp32 arithmetic, CG-amplified (measured 6.2e-6 at 4M rows, 1.6e-5 at 8M; the reference's own "
                    "host-to-host spread at 1M rows is 4.9e-5, cpu_baseline.step_rel_l2_vs_fixture); against the reference the shards' steps are held to 1e-5",
        "what": "the full batch's one-call update against the update composed from %d shards' sums (gradient, every CG iteration's product, "
                "K3) with the library's own CG kernels: the arithmetic an 8-rank job performs, on one GPU" % shards}
    out["failed"] = bool(out["shard_sum_vs_full"]["gradient_rel_l2"] > 5e-6 or out["shard_sum_vs_full"]["fvp_rel_l2"] > 1e-6
                         or out["update_vs_shard_composition"]["alpha_rel"] > 1e-5 or out["update_vs_shard_composition"]["kl_rel"] > 1e-4
                         or out["update_vs_shard_composition"]["step_rel_l2"] > 3e-5)
    for ek in engs:
        ek.close()
    del engs, obs, act
    torch.cuda.empty_cache()
    return out


def _host_paths(rng, n_traj=N_TRAJ, T_=T, advantages=False):
    """fp64 rollouts as a sampler hands them over: a list of per-trajectory dicts (mjrl/samplers/core.py:85-93)"""
    paths = [dict(observations=rng.randn(T_, N_OBS), actions=rng.randn(T_, N_ACT), rewards=rng.randn(T_), terminated=False)
             for _ in range(n_traj)]
    if advantages:
        for p in paths:
            p["advantages"] = rng.randn(T_)
    return paths


def user_level_measurements():
    """What a caller of the mjrl-shaped classes sees at the metric's size (SURVEY 8d "One update" (ii)), all
    from fp64 HOST trajectories -- PCIe-inclusive, so none of this is `value`:
    * end_to_end: one NPG.train_from_paths (host path statistics || page-locked staging + upload, the update, parameter
      read-back + set_param_values), median of 6 fresh batches after 4 warm-ups;
    * iteration: everything train_step does after sampling (batch_reinforce.py:93-114: returns, baseline prediction + GAE, the
      update, the baseline fit) with the quadratic and with the MLP baseline (2 epochs, batch 64: policy_opt_job_script's setting);
    * mlp_fit_us_per_step: the persistent minibatch-Adam trainer alone (k_mlp_fit, 21 inputs, 8 000 steps, best of 3)."""
    import torch
    from mjrl_amd._lib import check, ptr
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.baselines.mlp_baseline import MLPBaseline
    from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
    from mjrl_amd.policies.gaussian_mlp import MLP
    from mjrl_amd.utils import ingest, process_samples
    spec = type("Spec", (), dict(observation_dim=N_OBS, action_dim=N_ACT, horizon=T))
    out = {}
    # ---- end to end
    rng = np.random.RandomState(0)
    pol = MLP(spec, hidden_sizes=HIDDEN, seed=1, init_log_std=-0.5)
    agent = NPG(None, pol, None, normalized_step_size=STEP, FIM_invert_args={'iters': CG_ITERS, 'damping': DAMPING})
    base = _host_paths(rng, advantages=True)

    def fresh():        # a new batch every time: the staged copy of an earlier list is never reused
        return [dict(observations=p["observations"].copy(), actions=p["actions"].copy(), rewards=p["rewards"], advantages=p["advantages"],
                     terminated=False) for p in base]
    ts = []
    for it in range(10):
        # (r05: a call used to run 12-20 ms instead of 8 whenever the previous batch's 2 000 host arrays were released inside it --
        # glibc unmapping / trimming them one by one; the training process's allocator is tuned, utils/ingest.tune_malloc -- by main() here, by train_step / dropin.install in a job)
        b = fresh()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        agent.train_from_paths(b)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    in_order = [round(x, 3) for x in ts]
    ts = sorted(ts[1:])                       # every call after the first (which allocates the page-locked blocks and the device cache)
    out["end_to_end"] = {"npg_train_from_paths_ms_median": ts[len(ts) // 2], "ms_min": ts[0], "ms_max": ts[-1], "ms_each_sorted": ts, "ms_all_calls_in_order": in_order,
                         "updates_per_s": 1e3 / ts[len(ts) // 2],
                         "what": "NPG.train_from_paths on 1000 x 1000-step fp64 host trajectories (184 MB): path statistics || page-locked "
                                 "staging + upload (fp64 -> fp32 on the gather threads), mjx_npg_update, read-back, policy.set_param_values; "
                                 "median over ALL calls after the first (9 fresh batches); ms_all_calls_in_order has every call"}
    agent.engine.close()
    del agent, base
    # ---- a whole post-sampling iteration
    def stand_in_sampling(stream):
        """the stand-in for sampling: 20 chunks of 50 trajectories, generated one after the other (~12 ms each: ~240 ms in all).  stream:
        every finished chunk goes to utils/ingest.StreamedBatch the way mjrl_amd.samplers hands over its workers' results (SURVEY 8f
        N2: rewards / observations / actions are resident when "sampling" ends); else the batch is staged after sampling (r05)."""
        sb = ingest.StreamedBatch.for_current_device() if stream else None
        paths = []
        if sb is not None:
            sb.begin(N_TRAJ)
        for lo in range(0, N_TRAJ, 50):
            chunk = _host_paths(rng, n_traj=50)
            paths += chunk
            if sb is not None:
                sb.add(chunk, T)
        return paths, (bool(sb.finish(paths)) if sb is not None else False)

    it_out = {}
    for name, reps, stream in (("quadratic", 7, True), ("quadratic", 5, False), ("mlp", 5, True), ("mlp", 4, False)):
        pol = MLP(spec, hidden_sizes=HIDDEN, seed=1, init_log_std=-0.5)
        bl = QuadraticBaseline(spec) if name == "quadratic" else MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3)
        agent = NPG(None, pol, bl, normalized_step_size=STEP, FIM_invert_args={'iters': CG_ITERS, 'damping': DAMPING})
        rows, overl, streamed_all = [], [], True
        pend = None
        main = torch.cuda.current_stream()
        for it in range(reps):
            ts = time.perf_counter()
            paths, streamed = stand_in_sampling(stream)  # (the previous iteration's MLP fit runs under it)
            streamed_all = streamed_all and (streamed or not stream)
            stand_in_ms = 1e3 * (time.perf_counter() - ts)
            wait_ms = 0.0
            if pend is not None:
                # what is left of the background fit when "sampling" is over counts as waiting -- reported, and part of
                # the critical path whenever real sampling is not longer than the fit
                tw = time.perf_counter()
                pend.result()
                wait_ms = 1e3 * (time.perf_counter() - tw)
                overl.append(dict(baseline_fit_device_ms=pend.device_ms, wait_after_stand_in_sampling_ms=wait_ms, stand_in_sampling_ms=stand_in_ms))
            main.synchronize(); t0 = time.perf_counter()
            with ingest.trusted_iteration():
                process_samples.compute_returns(paths, 0.995); t1 = time.perf_counter()
                pre = bl.predraw(N_TRAJ * T) if name == "mlp" else None       # (train_step: the epoch permutations drawn under advantages + update)
                process_samples.compute_advantages(paths, bl, 0.995, 0.97); t2 = time.perf_counter()
                agent.train_from_paths(paths); main.synchronize(); t3 = time.perf_counter()
                # what BatchREINFORCE.train_step does: the fit is started, not waited for (the fitted baseline is first read by the
                # NEXT iteration's compute_advantages, after sampling)
                pend = bl.fit_async(paths, predrawn=pre) if name == "mlp" else bl.fit_async(paths)
                t4 = time.perf_counter()
            ingest.drop_shared_batch()
            rows.append([1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)] + [stand_in_ms])
        if pend is not None:
            pend.result()
        rows = rows[1:]                                  # the first iteration allocates
        med = sorted(rows, key=lambda r: r[4])[len(rows) // 2]
        res = dict(zip(["returns_ms", "advantages_ms", "update_ms", "baseline_fit_enqueue_ms", "total_ms", "stand_in_sampling_ms"], med))
        last = overl[-1] if overl else {}
        res["baseline_fit_ms"] = last.get("baseline_fit_device_ms")
        res["baseline_fit"] = ("overlapped: MLPBaseline.fit_async runs the Adam chain on a side stream under the next iteration's sampling" if name == "mlp" else
                               "overlapped: the ridge baselines' fit_async enqueues the fp64 Gram kernel behind the update and a helper thread solves the "
                               "F x F system under the next iteration's sampling") + "; total_ms is the critical path (fit enqueued, not waited for)"
        res["overlap_each"] = overl
        res["total_ms_each"] = [r[4] for r in rows]
        res["iterations_timed"] = len(rows)
        res["ingestion"] = ("streamed under the stand-in sampler (utils/ingest.StreamedBatch: every chunk of 50 trajectories staged and sent as it "
                            "is finished; resident when sampling ends): %s" % streamed_all) if stream else "staged after sampling (one upload per block on first use)"
        if stream:
            it_out[name] = res
        else:
            it_out[name]["staged_after_sampling"] = res
        agent.engine.close()
        del agent
    it_out["what"] = ("returns, baseline prediction + GAE, NPG.train_from_paths, baseline.fit on fresh 1M-timestep fp64 host batches under "
                      "ingest.trusted_iteration() like BatchREINFORCE.train_step; the median iteration's stage times.  The batch comes from a stand-in "
                      "sampler that produces 20 chunks over ~240 ms; total_ms is everything AFTER sampling (what the reference's "
                      "batch_reinforce.py:93-114 runs), with ingestion streamed under the sampler; `staged_after_sampling`: the same without")
    out["iteration"] = it_out
    ingest.drop_shared()
    # ---- the MLP-baseline trainer alone: HalfCheetah's 17 + 4 inputs (one workgroup) and Humanoid's 376 + 4 (BASELINE configs[3]:
    # eight workgroups of the same persistent kernel, one grid barrier per 32-sample half -- csrc/mlp_fit.h MULTI)
    dev = torch.device("cuda", torch.cuda.current_device())
    from mjrl_amd import _lib
    lib = _lib.load()
    fit_out = {}
    for d_in, steps in ((N_OBS + 4, 8000), (376 + 4, 4000)):
        Nf = 64 * (steps + 1)
        r2 = np.random.RandomState(0)
        feat = torch.from_numpy(r2.randn(Nf, d_in).astype(np.float32)).to(dev)
        y = torch.from_numpy(r2.randn(Nf).astype(np.float32)).to(dev)
        Pn = 128 * d_in + 128 + 128 * 128 + 128 + 128 + 1
        params = torch.from_numpy((0.1 * r2.randn(Pn)).astype(np.float32)).to(dev)
        m_, v_ = torch.zeros(Pn, device=dev), torch.zeros(Pn, device=dev)
        perm = torch.from_numpy(r2.permutation(Nf).astype(np.int32)).to(dev)
        loss = torch.zeros(32, dtype=torch.float64, device=dev)
        hid = (ctypes.c_int * 2)(128, 128)
        tt = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            check(lib.mjx_mlp_fit_adam(ptr(feat), ptr(y), Nf, d_in, hid, 2, ptr(params), ptr(m_), ptr(v_), 0, ptr(perm), 1, 64, 1e-3, 0.0,
                                       ptr(loss), None))
            torch.cuda.synchronize(); tt.append(time.perf_counter() - t0)
        fit_out[d_in] = 1e6 * min(tt) / steps
        del feat, y, perm
    d21 = N_OBS + 4
    out["mlp_fit_us_per_step"] = {"value": fit_out[d21], "steps": 8000, "inputs": d21, "hidden": [128, 128], "batch": 64,
                                  "kernel": "k_mlp_fit1p<128> (csrc/mlp_fit.h): one persistent workgroup, the whole Adam chain in one launch, one pass per 64-row "
                                            "step, Adam moments register-resident, pinned LDS addressing (r03: k_mlp_fit<128,1>, 23.3 us)",
                                  "per_1M_timesteps_2_epochs_s": 1e-6 * fit_out[d21] * 2 * (N_TRAJ * T // 64 - 1),
                                  "at_380_inputs": {"value": fit_out[380], "steps": 4000, "workgroups": 8,
                                                    "kernel": "k_mlp_fit<128,2,REGMOM,MULTI>: 8 workgroups (48-feature slices of the first layer), partial pre-activations "
                                                              "exchanged through an uncached block, one grid barrier per 32-sample half, the rest of the step replicated "
                                                              "(r04: ~14 launches per step, 115-155 us)"}}
    return out


def rehearsal_world8(one_rank_updates_per_s):
    """secondary.rehearsal_world8: rank 0's 125 k-row share of the 1M batch as one rank of EIGHT, on this one
    GPU, in a process of its own per transport -- `peer`: libmjx's peer exchange in loop-back (every exchange: the vector stored into
    8 slots, 7 flags raised, the bounded wait, the 8-slot sum -- all onto this rank's own buffer, so the link latency is NOT in it);
    `rccl`: the in-library RCCL path on a 1-rank communicator (which launches nothing: the loop with a free all-reduce).  NOT a
    measurement of 8 GPUs: the ceiling the per-rank work puts on strong scaling, i.e. (updates/s of the share) / (updates/s of one
    rank on the whole batch, same process lineage, same box)."""
    import subprocess
    out = {}
    for transport in ("peer", "rccl"):
        cmd = [sys.executable, os.path.abspath(__file__), "--rehearse-world", "8", "--rehearse-transport", transport, "--no-cpu-baseline",
               "--no-secondary", "--steps", "40", "--warmup", "5", "--repeats", "3"]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29533 + (os.getpid() % 400) + (1 if transport == "rccl" else 0)))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                out[transport] = {"error": (r.stderr or r.stdout)[-600:]}
                continue
            j = json.loads(line[-1])
            det = j.get("rehearsal_detail", {})
            det["transport"] = j.get("rehearsal")
            det["ms_per_update_each_repeat_sorted"] = j["timing"]["ms_per_step_each_repeat_sorted"]
            det["implied_ceiling_x"] = det["updates_per_s_of_the_share"] / one_rank_updates_per_s
            det["check"] = j.get("check")
            out[transport] = det
        except Exception as e:                                   # pragma: no cover
            out[transport] = {"error": "%s: %s" % (type(e).__name__, e)}
    out["one_rank_updates_per_s"] = one_rank_updates_per_s
    out["north_star"] = {"strong_scaling_at_8": 6.0, "share_ms_needed": 1e3 / (6.0 * one_rank_updates_per_s)}
    out["what"] = ("rank 0's 1/8 share of BASELINE configs[1] on ONE GPU with the 8-rank transports' device work in the loop; implied_ceiling_x = "
                   "share's updates/s / one-rank updates/s of this run.  Unmeasured on multi-GPU hardware: link latency, 8 real peers.")
    return out


def self_launch(n):
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n and os.environ.get("MJX_BENCH_SHARE_GPU") != "1":
        sys.exit("bench.py: --gpus %d but %d GPU(s) visible (MJX_BENCH_SHARE_GPU=1 + MJX_BENCH_BACKEND=gloo put all ranks on "
                 "GPU 0: a test mode, not a measurement)" % (n, have))
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL and the peer exchange need it on these hosts
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fvp-event-stride", type=int, default=11,
                    help="bracket every k-th Fisher-vector-product launch with HIP events (k coprime to the CG iteration count: every CG position is sampled equally); 0: none")
    ap.add_argument("--cpu-sample-traj", type=int, default=0,
                    help="cpu_baseline: 0 (default) = the whole 1M-timestep batch through the staged reference (oracle/_ref, ~20 s) -- "
                         "or, without it, a 200-trajectory slice through the torch port; k > 0: exactly k trajectories, scaled linearly")
    ap.add_argument("--repeats", type=int, default=3,
                    help="the timed region (barrier + synchronize, EXACTLY --steps updates, barrier + synchronize) is run this many "
                         "times; `value` is the median repeat, all repeats are reported")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurements (TRPO line-search update = BASELINE configs[2]; layer-wise FVP at the "
                         "per-GPU shard sizes of configs[3] / [4]); they run after the primary timed region, N = 1 only")
    ap.add_argument("--no-full-size", action="store_true", help="skip secondary.full_size (configs[3] / [4] at their stated 4M / 8M rows on this GPU)")
    ap.add_argument("--no-rehearsal", action="store_true", help="skip secondary.rehearsal_world8 (two sub-runs of this script as rank 0 of 8)")
    ap.add_argument("--rehearse-world", type=int, default=0,
                    help="diagnostic, 1 GPU: run rank 0's share of an R-rank job (1/R of the batch, global N, "
                         "RCCL collectives on a 1-rank group); the line is tagged 'rehearsal' and is not the metric")
    ap.add_argument("--rehearse-transport", choices=["rccl", "peer"], default="rccl",
                    help="--rehearse-world: the rank sums on a 1-rank RCCL group, or on libmjx's peer exchange in loop-back (the stores "
                         "to R buffers, counter updates, stream wait and R-slot sums of an R-rank exchange, all onto this rank's buffer)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher -- one rank per GPU under torch.distributed.run (what the
        # driver's multi-GPU command does itself); rank 0's JSON line passes through on stdout, the exit code is the job's
        return self_launch(args.gpus)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or run "
                 "`python bench.py --gpus %d` with WORLD_SIZE unset: it launches the ranks itself)" % (args.gpus, world, args.gpus, args.gpus))
    # (test hooks: MJX_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and MJX_BENCH_BACKEND=gloo replaces RCCL, which refuses
    #  two ranks on one device -- tests/test_gpu_parity.py runs the N = 2 path of this script on a 1-GPU box that way)
    dev_index = 0 if os.environ.get("MJX_BENCH_SHARE_GPU") == "1" else local_rank
    backend = os.environ.get("MJX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    shards = world
    if args.rehearse_world > 1:
        assert world == 1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ["MJX_COLLECTIVES_AT_WORLD1"] = "1"
        if args.rehearse_transport == "peer":
            os.environ["MJX_PEER_COMM"] = "1"
            os.environ["MJX_PEER_LOOPBACK_WORLD"] = str(args.rehearse_world)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        shards = args.rehearse_world

    from mjrl_amd._lib import check
    from mjrl_amd.engine import UpdateEngine
    from mjrl_amd.utils import ingest as _ingest
    _ingest.tune_malloc()            # a training process (what BatchREINFORCE.train_step / dropin.install do; MJX_MALLOC_TUNE=0 opts out)

    theta0 = initial_params()
    obs, act, adv = synth_shard(rank, shards)
    # advantage whitening over the global batch (batch_reinforce.py:185); identical on all ranks
    s = torch.tensor([adv.sum(), float(adv.size)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(s)
    mean = float(s[0] / s[1])
    q = torch.tensor([((adv - mean) ** 2).sum()], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(q)
    std = float(np.sqrt(q.item() / s[1].item()))
    adv = (adv - mean) / (std + 1e-6)

    eng = UpdateEngine(N_OBS, N_ACT, HIDDEN)
    ident = np.concatenate([np.zeros(N_OBS), np.ones(N_OBS), np.zeros(N_ACT), np.ones(N_ACT)]).astype(np.float32)
    eng.set_policy(theta0, theta0, ident, ident)
    eng.set_batch(obs, act, adv, N_global=N_TRAJ * T if args.rehearse_world > 1 else None)   # resident in HBM from here on
    assert eng.N_global == N_TRAJ * T, eng.N_global
    theta0_dev = torch.from_numpy(theta0).to(eng.device)
    last, last_vec = {}, {}

    def one_update():
        # the call sequence of NPG.train_from_paths (mjrl_amd/algos/npg_cg.py): everything is enqueued, one read-back
        # ONE call into libmjx (mjx_npg_update): K1, rank sum, CG (10 x [K2, rank sum, vector update]), step length on the
        # device, step, K3, rank sum -- then one read-back
        surr_after, kl = eng.npg_update(CG_ITERS, DAMPING, STEP, -3.0)
        late = eng.deferred()
        last.update(alpha=late["alpha"], kl=kl, surr_improvement=surr_after - late["surr_before"])
        if "step" not in last_vec:                   # (first warm-up update only: the step every later update repeats)
            last_vec["step"] = eng.theta_new.cpu().numpy().astype(np.float64) - theta0
        # old := new happens here in training; the bench restores theta0 so every step does identical work
        eng.theta_new.copy_(theta0_dev)
        eng.old_is_new = True
        eng._bind_policy()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_update()
    fence()
    check(eng.lib.mjx_profile_enable(eng.ctx, args.fvp_event_stride))
    dts = []
    for _ in range(max(1, args.repeats)):
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_update()
        fence()
        dts.append(time.perf_counter() - t0)
    prof = (ctypes.c_double * 2)()
    check(eng.lib.mjx_profile_read(eng.ctx, prof))
    samples_buf, samples_n = (ctypes.c_double * 4096)(), ctypes.c_int(0)
    check(eng.lib.mjx_profile_samples(eng.ctx, samples_buf, 4096, ctypes.byref(samples_n)))
    fvp_samples = [float(samples_buf[i]) for i in range(min(samples_n.value, 4096))]
    check(eng.lib.mjx_profile_enable(eng.ctx, 0))
    tmax = torch.tensor(dts, dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)          # per repeat: the slowest rank
    dts = sorted(float(x) for x in tmax.cpu())
    dt = dts[len(dts) // 2] if len(dts) % 2 else 0.5 * (dts[len(dts) // 2 - 1] + dts[len(dts) // 2])     # the median repeat

    failed = False
    if rank == 0:
        fvp_ms = prof[0] / prof[1] if prof[1] > 0 else float("nan")      # (--fvp-event-stride 0: no roofline figures)
        P = N_OBS * 64 + 64 * 64 + 64 * N_ACT
        # The CG loop runs the cached-forward FVP instance: K1 stores h1 / h2 and the normalised observation image once
        # per update (theta is fixed during CG), each product then costs the tangent + backward passes:
        # 2(4P - 2 n h1) FLOP and 4(h1 + h2 + NP) B per sample, NP = n + 1 padded to 4 (SURVEY 8d "cached-activation
        # variant").  The recompute instance (2(5P - 2 n h1) = 51 328 FLOP, 68 B per sample) is what mjx_fvp runs
        # without a preceding K1.
        flop_per_sample = 2 * (4 * P - 2 * N_OBS * 64)               # 40 192 @cfg2
        bytes_per_sample = 4 * (64 + 64 + ((N_OBS + 1 + 3) & ~3))    # 592 @cfg2
        flop_recompute = 2 * (5 * P - 2 * N_OBS * 64)                # 51 328 @cfg2
        n_loc = eng.N_local
        achieved_tf = flop_per_sample * n_loc / (fvp_ms * 1e-3) / 1e12
        traffic, traffic_source = None, None
        import glob
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fvp_pmc.json")))
        if pmcs and world == 1:
            try:
                traffic = json.load(open(pmcs[-1])).get("hbm_bytes_per_launch")
                traffic_source = ("%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over an EARLIER run of this command "
                                  "(tools/profile_bench.sh); a constant read from the file, not measured by this run"
                                  % os.path.relpath(pmcs[-1], ROOT))
            except Exception:
                traffic = None
        out = {
            "metric": "NPG updates/sec (1M-timestep batch, 64x64 MLP)",
            "value": args.steps / dt,
            "unit": "updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: HalfCheetah-v2 shapes (obs=17, act=6), 64x64 tanh MLP, "
                                   "NPG 10 CG iters, 1M timesteps/batch (1000 traj x 1000), device-resident update",
                       "global_batch": N_TRAJ * T, "parallelism": "dp%d (trajectory shards, one rank sum per CG iteration: %s)"
                                      % (world, {"rccl": "RCCL all-reduce inside libmjx", "peer": "libmjx peer exchange over HIP IPC",
                                                 "hook": "transport hook", None: "none at one rank"}.get(eng.comm_kind, str(eng.comm_kind))),
                       "cg_iters": CG_ITERS, "damping": DAMPING},
            "roofline": {"bound": "mfma", "kernel": "k_fused<64,64,1,8,MODE_FVP,NP=20,CACHED>",
                         "achieved": achieved_tf, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP32_MFMA_PEAK_TF, "traffic": traffic, "traffic_source": traffic_source,
                         "avg_launch_ms": fvp_ms, "launches": int(prof[1]),
                         "launch_ms_samples": {"sorted": sorted(round(x, 5) for x in fvp_samples),
                                               "min": min(fvp_samples) if fvp_samples else None,
                                               "median": sorted(fvp_samples)[len(fvp_samples) // 2] if fvp_samples else None,
                                               "max": max(fvp_samples) if fvp_samples else None,
                                               "what": "every bracketed launch's HIP-event time (mjx_profile_samples); avg_launch_ms is their mean"},
                         "launches_timed": "every %d-th of %d (HIP events on the launch stream, inside the timed regions)" % (max(args.fvp_event_stride, 1), args.steps * CG_ITERS * len(dts)),
                         "flop_per_launch": flop_per_sample * n_loc,
                         "algorithmic_bytes_per_launch": bytes_per_sample * n_loc,
                         "hbm_GBps_algorithmic": bytes_per_sample * n_loc / (fvp_ms * 1e-3) / 1e9,
                         "hbm_frac_algorithmic": bytes_per_sample * n_loc / (fvp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "fvp_equivalent_TFLOPs_if_recomputed": flop_recompute * n_loc / (fvp_ms * 1e-3) / 1e12},
            "timing": {"repeats": len(dts), "ms_per_step_each_repeat_sorted": [1e3 * x / args.steps for x in dts],
                       "ms_per_step_min": 1e3 * dts[0] / args.steps, "ms_per_step_median": 1e3 * dt / args.steps,
                       "value_is": "steps / median repeat (each repeat: barrier + synchronize, exactly `steps` updates, barrier + synchronize; max over ranks)"},
            "check": last,
        }
        # the update's scalars against the fp64 oracle on this very batch (tests/golden/bench_cfg2_1m.npz, 125 s of CPU to
        # regenerate: tests/golden/make_golden_big.py): a drift beyond the 1e-5 parity bar fails the run loudly
        fx = os.path.join(ROOT, "tests", "golden", "bench_cfg2_1m.npz")
        if os.path.exists(fx) and args.rehearse_world <= 1:
            g = np.load(fx)
            drift = {k: abs(last[k] - float(g[k])) / abs(float(g[k])) for k in ("alpha", "kl", "surr_improvement")}
            out["check_vs_fp64_oracle"] = {"rel_error": drift, "bars": {k: 1e-5 for k in drift}, "fixture": "tests/golden/bench_cfg2_1m.npz"}
            if max(drift.values()) > 1e-5:
                print(json.dumps({"error": "update drifted from the fp64 oracle beyond 1e-5", "drift": drift, "check": last}), file=sys.stderr, flush=True)
                failed = True
        # ... and against the UNMODIFIED REFERENCE's NPG.train_from_paths on this very batch (tests/golden/bench_ref_1m.npz,
        # tests/golden/make_golden_big.py bench_ref_1m: 25 s + 45 s of CPU for configs[1] / configs[2]): scalars and the step
        # direction at the north-star's 1e-5
        fr = os.path.join(ROOT, "tests", "golden", "bench_ref_1m.npz")
        ref = np.load(fr) if os.path.exists(fr) else None
        if ref is not None and args.rehearse_world <= 1 and "step" in last_vec:
            rs = ref["npg_new_params"].astype(np.float64) - theta0
            drift = {"alpha": abs(last["alpha"] - float(ref["npg_alpha"])) / float(ref["npg_alpha"]),
                     "kl": abs(last["kl"] - float(ref["npg_kl"])) / float(ref["npg_kl"]),
                     "surr_improvement": abs(last["surr_improvement"] - float(ref["npg_surr_improvement"])) / float(ref["npg_surr_improvement"]),
                     "step_rel_l2": float(np.linalg.norm(last_vec["step"] - rs) / np.linalg.norm(rs))}
            out["check_vs_reference"] = {"rel_error": drift, "bars": {k: 1e-5 for k in drift}, "fixture": "tests/golden/bench_ref_1m.npz",
                                         "what": "the unmodified reference's NPG.train_from_paths (mjrl/algos/npg_cg.py:91-163) on this batch"}
            if max(drift.values()) > 1e-5:
                print(json.dumps({"error": "update differs from the reference beyond 1e-5", "drift": drift, "check": last}), file=sys.stderr, flush=True)
                failed = True
        if args.rehearse_world > 1:
            out["rehearsal"] = "rank 0 of %d on one GPU, %s: NOT the metric" % (
                args.rehearse_world, "1-rank RCCL group" if eng.comm_kind == "rccl" else "peer exchange in loop-back" if eng.comm_kind == "peer" else eng.comm_kind)
            out["roofline"]["traffic"] = None
            # one more short pass with whole CG ITERATIONS between the events (mjx_profile_enable(ctx, -k)): product + reduction /
            # exchange + vector update; what is left of an update after 10 of those is K1, K3, their sums and the read-back
            check(eng.lib.mjx_profile_enable(eng.ctx, -7))
            for _ in range(8):
                one_update()
            torch.cuda.synchronize()
            pit = (ctypes.c_double * 2)()
            check(eng.lib.mjx_profile_read(eng.ctx, pit))
            check(eng.lib.mjx_profile_enable(eng.ctx, 0))
            it_us = 1e3 * pit[0] / pit[1] if pit[1] > 0 else float("nan")
            out["rehearsal_detail"] = {
                "share_rows": int(eng.N_local), "updates_per_s_of_the_share": args.steps / dt, "ms_per_update": 1e3 * dt / args.steps,
                "fvp_us": 1e3 * fvp_ms, "cg_iteration_us": it_us, "reduce_exchange_and_vector_update_us": it_us - 1e3 * fvp_ms,
                "outside_the_cg_loop_us": 1e6 * dt / args.steps - CG_ITERS * it_us,
                "iterations_timed": int(pit[1]),
                "what": "HIP events on the launch stream: fvp_us around the product kernel alone (every %d-th launch, in the timed region), "
                        "cg_iteration_us around a whole iteration (every 7-th, in 8 extra updates); outside = ms_per_update - %d x iteration "
                        "(K1, K3, their reductions and rank sums, the read-back, the host's turn-around)" % (max(args.fvp_event_stride, 1), CG_ITERS)}
        if world == 1 and not args.no_secondary and args.rehearse_world <= 1:
            out["secondary"] = secondary_measurements(eng, theta0, theta0_dev, ref, full_size=not args.no_full_size)
            if not args.no_rehearsal:
                eng.close()                                       # (the rehearsals run in processes of their own, on the same GPU)
                out["secondary"]["rehearsal_world8"] = rehearsal_world8(out["value"])
            if out["secondary"]["trpo_configs2"].get("check_vs_reference", {}).get("failed"):
                print(json.dumps({"error": "TRPO update differs from the reference beyond 1e-5",
                                  "check": out["secondary"]["trpo_configs2"]["check_vs_reference"]}), file=sys.stderr, flush=True)
                failed = True
            for fs_name, fs_res in out["secondary"].get("full_size", {}).items():
                if fs_res.get("failed"):
                    print(json.dumps({"error": "%s at full size: the batch's gradient / product / update is not the sum over its shards" % fs_name,
                                      "check": {k: fs_res[k] for k in ("shard_sum_vs_full", "update_vs_shard_composition")}}), file=sys.stderr, flush=True)
                    failed = True
            for lw_name, lw_res in out["secondary"]["roofline_lw"].items():
                if lw_res.get("check_vs_reference", {}).get("failed"):
                    print(json.dumps({"error": "%s: the shard's update differs from the reference beyond its bars" % lw_name,
                                      "check": lw_res["check_vs_reference"]}), file=sys.stderr, flush=True)
                    failed = True
        if world == 1 and not args.no_cpu_baseline and args.rehearse_world <= 1:
            out["cpu_baseline"] = cpu_baseline(theta0, args.cpu_sample_traj)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        if not failed:
            print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
