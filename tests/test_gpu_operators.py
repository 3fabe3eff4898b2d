"""GPU tests that EXECUTE the public operator methods the reference exposes on its agents
(mjrl/algos/batch_reinforce.py:40-58, npg_cg.py:62-88, utils/cg_solve.py) and the less-travelled keyword paths
(const_learn_rate npg_cg.py:128-130, compute_advantages(normalize=True) process_samples.py:14-19,
sample_mode='samples' batch_reinforce.py:83-86, DAPG with hvp_sample_frac), against the golden fixtures of the
unmodified reference and the fp64 oracle; plus the whole-update assertion at the BASELINE size (1M timesteps)."""
import numpy as np
import pytest

from oracle import npg_oracle as O
from oracle import synth
from tests._cases import NpgCase, load

pytestmark = pytest.mark.gpu

TOL_VPG = 3e-6
TOL_FVP = 3e-6
TOL_STEP = 1e-5          # the north-star bar


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def make_agent(c, cls=None, **kw):
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.policies.gaussian_mlp import MLP, LinearPolicy
    spec = type("Spec", (), dict(observation_dim=c.n, action_dim=c.m, horizon=1000))
    pol = MLP(spec, hidden_sizes=c.hidden, seed=1, init_log_std=-0.5) if c.hidden else LinearPolicy(spec, seed=1, init_log_std=-0.5)
    pol.set_param_values(c.theta0)
    if c.tr is not None:
        pol.model.set_transformations(*c.tr)
        pol.old_model.set_transformations(*c.tr)
    kw.setdefault("FIM_invert_args", {'iters': c.cg_iters, 'damping': 1e-4})
    return (cls or NPG)(None, pol, None, **kw), pol


@pytest.mark.parametrize("name", ["npg_cfg2_small", "npg_cfg2_ragged_tr", "npg_cfg1_linear"])
def test_agent_operator_methods_vs_reference(name):
    """CPI_surrogate / flat_vpg / HVP / build_Hvp_eval + cg_solve called the way the reference's own code and its
    users call them (host ndarrays in; tensors / ndarrays out) == the reference's outputs for the same calls."""
    import torch
    from mjrl_amd.utils.cg_solve import cg_solve
    c = NpgCase(name)
    agent, pol = make_agent(c)
    surr = agent.CPI_surrogate(c.obs, c.act, c.adv_w)
    assert isinstance(surr, torch.Tensor)                                   # the reference returns torch scalars ...
    assert abs(surr.data.numpy().ravel()[0] - float(c.g["surr_before"])) < 1e-6      # ... unwrapped like batch_reinforce.py:139
    g = agent.flat_vpg(c.obs, c.act, c.adv_w)
    assert isinstance(g, np.ndarray) and g.dtype == np.float32 and g.shape == c.g["vpg"].shape
    assert rel(g, c.g["vpg"]) < TOL_VPG
    hv = agent.HVP(c.obs, c.act, c.g["vpg"])                                # regu_coef defaults to FIM_invert_args['damping']
    assert isinstance(hv, np.ndarray) and rel(hv, c.g["hvp_of_vpg"]) < TOL_FVP
    hv0 = agent.HVP(c.obs, c.act, c.g["vpg"], regu_coef=0.0)
    assert rel(hv0 + np.float32(1e-4) * c.g["vpg"], c.g["hvp_of_vpg"]) < TOL_FVP
    hvp = agent.build_Hvp_eval([c.obs, c.act], regu_coef=1e-4)              # npg_cg.py:83-88
    assert rel(hvp(c.g["vpg"]), c.g["hvp_of_vpg"]) < TOL_FVP
    x = cg_solve(hvp, c.g["vpg"], x_0=c.g["vpg"].copy(), cg_iters=c.cg_iters)    # the call of npg_cg.py:122-123
    assert isinstance(x, np.ndarray) and rel(x, c.g["cg_x"]) < TOL_STEP
    # a plain host callable goes through the reference's host loop (x_0 ignored, residual_tol break)
    A = np.diag(np.arange(1.0, 9.0)); b = np.arange(8.0)
    xs = cg_solve(lambda v: A.dot(v), b, x_0=np.ones(8), cg_iters=20)
    np.testing.assert_allclose(xs, b / np.arange(1.0, 9.0), rtol=1e-8, atol=1e-10)
    # kl_old_new at theta_new == theta_old: exactly the reference's value (0 up to the 1e-8 of Dr)
    kl = agent.kl_old_new(c.obs, c.act)
    assert isinstance(kl, torch.Tensor) and abs(float(kl)) < 1e-6
    agent.engine.close()


def test_agent_operators_with_old_neq_new():
    """the same public methods in general position (theta_new != theta_old, input transform only on policy.model --
    the state input_normalization leaves behind, npg_cg.py:101-107) against the reference's values."""
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.policies.gaussian_mlp import MLP
    g = load("hvp_general_64x64")
    n, m, hidden = int(g["n"]), int(g["m"]), tuple(int(h) for h in g["hidden"])
    paths = synth.make_paths(int(g["n_traj"]), int(g["T"]), n, m, seed=int(g["path_seed"]))
    rng = np.random.RandomState(int(g["adv_seed"]))
    for p in paths:
        p["advantages"] = rng.randn(len(p["rewards"])) * 2.0 + 0.3
    obs = np.concatenate([p["observations"] for p in paths]); act = np.concatenate([p["actions"] for p in paths])
    adv_w = O.whiten(np.concatenate([p["advantages"] for p in paths]))
    spec = type("Spec", (), dict(observation_dim=n, action_dim=m, horizon=1000))
    pol = MLP(spec, hidden_sizes=hidden, seed=1, init_log_std=-0.5)
    pol.set_param_values(g["theta_old"], set_new=True, set_old=True)
    pol.set_param_values(g["theta_new"], set_new=True, set_old=False)
    pol.model.set_transformations(g["in_shift"], g["in_scale"], None, None)
    agent = NPG(None, pol, None)
    assert abs(float(agent.CPI_surrogate(obs, act, adv_w)) - float(g["surr"])) < 2e-6
    assert abs(float(agent.kl_old_new(obs, act)) - float(g["kl"])) < 1e-5 * float(g["kl"]) + 1e-8
    assert rel(agent.flat_vpg(obs, act, adv_w), g["vpg"]) < TOL_STEP        # (likelihood ratios span 7e-5 .. 1.3e3: the reference itself sits 3.3e-6 from fp64)
    assert rel(agent.HVP(obs, act, g["v"], regu_coef=1e-4), g["hvp"]) < TOL_STEP
    agent.engine.close()


def test_CG_solve_standalone_equals_cg_solve_and_reference():
    """the north-star's public name: agent.CG_solve(b) on the bound batch, device tensor or host vector in."""
    import torch
    c = NpgCase("npg_cfg2_small")
    agent, pol = make_agent(c)
    agent._bind(c.obs, c.act, c.adv_w)
    x, bx = agent.CG_solve(c.g["vpg"])                                      # host vector
    assert rel(x.cpu().numpy(), c.g["cg_x"]) < TOL_STEP
    assert abs(bx - float(np.dot(c.g["vpg"].astype(np.float64), x.cpu().numpy().astype(np.float64)))) < 1e-6 * abs(bx)
    gdev = torch.from_numpy(c.g["vpg"]).to(agent.engine.device)
    x2, bx2 = agent.CG_solve(gdev, iters=c.cg_iters, damping=1e-4)          # device tensor, explicit arguments
    assert np.array_equal(x2.cpu().numpy(), x.cpu().numpy()) and bx2 == bx  # bit-reproducible
    x3, _ = agent.CG_solve(gdev, iters=3)
    ref3 = O.cg_solve(lambda p: O.fvp(c.theta0.astype(np.float64), c.obs, p, c.n, c.m, c.hidden, damping=1e-4),
                      c.g["vpg"].astype(np.float64), 3)
    assert rel(x3.cpu().numpy(), ref3) < TOL_STEP
    agent.engine.close()


@pytest.mark.parametrize("name", ["npg_cfg2_small", "npg_pointmass_32x32"])
def test_const_learn_rate_branch(name):
    """NPG(const_learn_rate=a): new = theta + a * npg_grad, delta logged as a^2 g.x (npg_cg.py:128-130): the step is
    a times the reference's CG solution of the fixture."""
    c = NpgCase(name)
    a = 0.2
    agent, pol = make_agent(c, const_learn_rate=a, save_logs=True)
    agent.train_from_paths(c.paths)
    step = pol.get_param_values().astype(np.float64) - c.theta0
    assert rel(step, a * c.g["cg_x"].astype(np.float64)) < TOL_STEP
    lg = agent.logger.get_current_log()
    gx = float(np.dot(c.g["vpg"].astype(np.float64), c.g["cg_x"].astype(np.float64)))
    assert lg["alpha"] == a and abs(lg["delta"] - a * a * gx) < 1e-4 * abs(a * a * gx)
    r = O.npg_update(c.theta0.astype(np.float64), c.obs, c.act, c.adv_w, c.n, c.m, c.hidden, c.transforms(), cg_iters=c.cg_iters,
                     const_alpha=a)
    assert abs(lg["kl_dist"] - r["kl"]) < 1e-4 * r["kl"] + 1e-9
    assert pol.old_equals_new()
    agent.engine.close()


@pytest.mark.parametrize("kind", ["quadratic", "linear"])
def test_compute_advantages_normalize(kind):
    """compute_advantages(..., normalize=True): (adv - mean) / (std + 1e-8) over the whole batch, both branches
    (process_samples.py:14-19, 30-35)."""
    from mjrl_amd.utils import process_samples
    g = load("gae_" + kind)
    n, m = int(g["n"]), int(g["m"])
    paths = synth.make_paths(int(g["n_traj"]), int(g["T"]), n, m, seed=int(g["path_seed"]), ragged=True)
    gamma, lam = float(g["gamma"]), float(g["lam"])
    process_samples.compute_returns(paths, gamma)

    class Frozen:
        def __init__(self):
            self.k = 0
        def predict(self, path):
            T = len(path["rewards"]); out = g["baseline_pred"][self.k:self.k + T]; self.k += T
            return np.asarray(out, np.float64)
    for lam_, key in ((lam, "advantages"), (None, "advantages_nogae")):
        process_samples.compute_advantages(paths, Frozen(), gamma, lam_, normalize=True)
        a = g[key]
        want = (a - a.mean()) / (a.std() + 1e-8)
        got = np.concatenate([p["advantages"] for p in paths])
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)
        assert abs(got.mean()) < 1e-12 and abs(got.std() - 1.0) < 1e-6


def test_train_step_sample_mode_samples():
    """train_step(sample_mode='samples') draws whole trajectories until N timesteps are in (batch_reinforce.py:83-86)
    and rejects other modes (:73-75)."""
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
    from mjrl_amd.policies.gaussian_mlp import MLP

    class Walk:                            # 1-D random walk to the origin, horizon 20
        horizon = 20
        def __init__(self):
            self.rng = np.random.RandomState(0)
        def set_seed(self, s):
            self.rng = np.random.RandomState(s)
        def reset(self):
            self.x = self.rng.uniform(-1, 1, 3); return self.x.copy()
        def step(self, a):
            self.x = self.x + 0.1 * np.clip(a, -1, 1)[:3]
            return self.x.copy(), -float(np.abs(self.x).sum()), False, {}

    spec = type("Spec", (), dict(observation_dim=3, action_dim=3, horizon=20))
    pol = MLP(spec, hidden_sizes=(32, 32), seed=3, init_log_std=-0.5)
    agent = NPG(Walk(), pol, QuadraticBaseline(spec), normalized_step_size=0.05, seed=5, save_logs=True)
    stats = agent.train_step(N=330, sample_mode='samples', gamma=0.95, gae_lambda=0.97, num_cpu=1)
    lg = agent.logger.get_current_log()
    assert 330 <= lg["num_samples"] < 330 + 20 and stats[-1] == 330 and agent.seed == 5 + 330
    assert 0 < lg["kl_dist"] < 0.1
    with pytest.raises(ValueError):
        agent.train_step(N=10, sample_mode='timesteps')
    agent.engine.close()


def test_dapg_with_hvp_sample_frac_draws_from_the_on_policy_rows():
    """DAPG + hvp_sample_frac < 0.99: the Fisher rows of every product are drawn with replacement from the
    ON-POLICY block only (dapg.py:103 hands HVP the on-policy arrays; npg_cg.py:65-69 samples from what it is handed),
    from NumPy's global RNG, and the [on-policy ; demonstrations] binding is restored afterwards."""
    from mjrl_amd.algos.dapg import DAPG
    c = NpgCase("npg_cfg2_small")
    demos = synth.make_paths(3, 200, c.n, c.m, seed=7)
    frac, iters = 0.8, 6
    agent, pol = make_agent(c, cls=DAPG, demo_paths=demos, kl_dist=0.025, lam_0=1e-2, lam_1=0.95, hvp_sample_frac=frac,
                            FIM_invert_args={'iters': iters, 'damping': 1e-4})
    np.random.seed(11)
    agent.train_from_paths(c.paths)
    th = c.theta0.astype(np.float64)
    a = (c.n, c.m, c.hidden)
    d_obs = np.concatenate([p["observations"] for p in demos]); d_act = np.concatenate([p["actions"] for p in demos])
    N = c.obs.shape[0]
    all_adv = 1e-2 * np.concatenate([c.adv_w / (np.std(c.adv_w) + 1e-8), 1e-2 * np.ones(d_obs.shape[0])])
    g = (all_adv.shape[0] / N) * O.vpg(th, th, np.concatenate([c.obs, d_obs]), np.concatenate([c.act, d_act]), all_adv, *a)
    np.random.seed(11)
    def hv(p):
        idx = np.random.choice(N, size=int(frac * N))
        return O.fvp(th, c.obs[idx], p, *a, damping=1e-4)
    x = O.cg_solve(hv, g, iters)
    alpha = np.sqrt(abs(2 * 0.025 / (g.dot(x) + 1e-20)))
    step = pol.get_param_values().astype(np.float64) - c.theta0
    assert rel(step, alpha * x) < TOL_STEP, rel(step, alpha * x)
    new = th + alpha * x
    assert abs(agent.last_update["kl_dist"] - O.mean_kl(new, th, c.obs, *a)) < 1e-4 * agent.last_update["kl_dist"]
    eng = agent.engine
    assert eng.N_bound == N and eng.N_local == N + d_obs.shape[0]      # prefix binding restored over the whole block
    agent.engine.close()


@pytest.mark.parametrize("layerwise", [False, True])
def test_one_call_dapg_update_equals_call_sequence(layerwise, monkeypatch):
    """mjx_dapg_update (K1 over [on-policy ; demonstrations], gradient x N_all / N_on, on-policy prefix, K3, CG, step, K3 from
    ONE C call) == the same steps issued call by call from Python (dapg.py:92-121), bit for bit; two iterations, so the
    decaying demonstration weight (lam_1^iter) and the re-binding of the next batch are covered too."""
    from mjrl_amd.algos.dapg import DAPG
    from mjrl_amd.engine import UpdateEngine
    if layerwise:
        monkeypatch.setenv("MJX_FORCE_LAYERWISE", "1")
    c = NpgCase("npg_cfg2_small")
    demos = synth.make_paths(3, 200, c.n, c.m, seed=7)
    outs = []
    for one_call in (True, False):
        if not one_call:
            monkeypatch.setattr(UpdateEngine, "dapg_update", lambda self, *a, **k: None)
        agent, pol = make_agent(c, cls=DAPG, demo_paths=demos, kl_dist=0.025, lam_0=1e-2, lam_1=0.95)
        assert agent.engine.fused == (not layerwise)
        log = []
        for _ in range(2):
            agent.train_from_paths(c.paths)
            u = agent.last_update
            log.append((pol.get_param_values().copy(), u["alpha"], u["kl_dist"], u["surr_before"], u["surr_after"], u["gdotx"]))
        eng = agent.engine
        assert eng.N_bound == c.obs.shape[0] and eng.N_local > eng.N_bound        # left bound to the on-policy prefix
        outs.append(log)
        agent.engine.close()
    for a, b in zip(*outs):
        assert np.array_equal(a[0], b[0])
        assert a[1:] == b[1:], (a[1:], b[1:])
    assert not np.array_equal(outs[0][0][0], outs[0][1][0])


def _bench_engine():
    import bench
    from mjrl_amd.engine import UpdateEngine
    theta0 = bench.initial_params()
    obs, act, adv = bench.synth_shard(0, 1)
    adv = (adv - adv.mean()) / (adv.std() + 1e-6)
    eng = UpdateEngine(bench.N_OBS, bench.N_ACT, bench.HIDDEN)
    ident = np.concatenate([np.zeros(bench.N_OBS), np.ones(bench.N_OBS), np.zeros(bench.N_ACT), np.ones(bench.N_ACT)]).astype(np.float32)
    eng.set_policy(theta0, theta0, ident, ident)
    eng.set_batch(obs, act, adv)
    return bench, eng, theta0


def test_whole_update_at_the_baseline_size():
    """ONE NPG update on bench.py's own 1M-timestep batch (BASELINE configs[1]) against the UNMODIFIED REFERENCE's
    NPG.train_from_paths on that batch (tests/golden/bench_ref_1m.npz, made by make_golden_big.py bench_ref_1m: 25 s of
    CPU; npg_cg.py:108-142): step direction at TOL_STEP, alpha / kl / surr_improvement at 1e-5 -- and against the fp64
    oracle's values (bench_cfg2_1m.npz), call by call and through the one-call entry point."""
    g, r = load("bench_cfg2_1m"), load("bench_ref_1m")
    bench, eng, theta0 = _bench_engine()
    assert eng.N_global == int(g["N"]) == int(r["N"]) and np.array_equal(theta0, r["theta0"])
    ref_step = r["npg_new_params"].astype(np.float64) - theta0
    for one_call in (False, True):
        if one_call:
            eng.set_policy(theta0, theta0, eng.tr_new.cpu().numpy(), eng.tr_old.cpu().numpy())
            surr_after, kl = eng.npg_update(bench.CG_ITERS, bench.DAMPING, bench.STEP, -3.0)
        else:
            grad, _ = eng.surr_vpg(sync=False)
            assert rel(grad.cpu().numpy(), r["npg_vpg"]) < 1e-5
            eng.cg_solve(grad, bench.CG_ITERS, bench.DAMPING, sync=False)
            assert rel(eng.x.cpu().numpy(), r["npg_cg_x"]) < TOL_STEP
            eng.apply_npg_step(bench.STEP, -3.0)
            surr_after, kl = eng.eval_surr_kl()
        late = eng.deferred()
        step = eng.theta_new.cpu().numpy().astype(np.float64) - theta0
        # ---- the reference itself
        assert rel(step, ref_step) < TOL_STEP, rel(step, ref_step)
        assert abs(late["alpha"] - float(r["npg_alpha"])) < 1e-5 * float(r["npg_alpha"])
        assert abs(kl - float(r["npg_kl"])) < 1e-5 * float(r["npg_kl"])
        assert abs((surr_after - late["surr_before"]) - float(r["npg_surr_improvement"])) < 1e-5 * float(r["npg_surr_improvement"])
        # ---- fp64 truth
        assert abs(late["alpha"] - float(g["alpha"])) < 1e-5 * float(g["alpha"])
        assert abs(kl - float(g["kl"])) < 1e-5 * float(g["kl"])
        assert abs((surr_after - late["surr_before"]) - float(g["surr_improvement"])) < 1e-5 * float(g["surr_improvement"])
        s = int(g["stride"])
        assert rel(step[::s], g["step_sub"]) < TOL_STEP
        assert abs(np.linalg.norm(step) - float(g["step_norm"])) < 1e-5 * float(g["step_norm"])
    eng.close()


def test_trpo_update_at_the_baseline_size():
    """BASELINE configs[2]: ONE TRPO update (kl_dist 0.025: the first two step lengths are rejected) on the same
    1M-timestep batch against the UNMODIFIED REFERENCE's TRPO.train_from_paths (bench_ref_1m.npz; trpo.py:100-126, 45 s
    of CPU): the same number of line-search trials, alpha / kl / surr_improvement at 1e-5, step at TOL_STEP -- device-side
    line search (mjx_trpo_update) and the agent's call-by-call loop."""
    r = load("bench_ref_1m")
    bench, eng, theta0 = _bench_engine()
    kl_dist = float(r["trpo_kl_dist"])
    ref_step = r["trpo_new_params"].astype(np.float64) - theta0
    res = eng.trpo_update(bench.CG_ITERS, bench.DAMPING, 2.0 * kl_dist, kl_dist, -3.0)
    late = eng.deferred()
    assert res["accepted"] and res["trials"] == int(r["trpo_trials"]) == 3
    assert abs(res["alpha"] - float(r["trpo_alpha"])) < 1e-5 * float(r["trpo_alpha"])
    assert abs(res["kl"] - float(r["trpo_kl"])) < 1e-5 * float(r["trpo_kl"]) and res["kl"] < kl_dist
    assert abs((res["surr_after"] - late["surr_before"]) - float(r["trpo_surr_improvement"])) < 1e-5 * float(r["trpo_surr_improvement"])
    step = eng.theta_new.cpu().numpy().astype(np.float64) - theta0
    assert rel(step, ref_step) < TOL_STEP, rel(step, ref_step)
    # the rejected trials' KL values exceed the bound, in the reference's order (alpha, 0.9 alpha, 0.81 alpha)
    assert [h[1] >= kl_dist for h in res["history"]] == [True, True, False]
    eng.close()


@pytest.mark.parametrize("name,layerwise", [("npg_cfg2_small", False), ("npg_cfg2_ragged_tr", False), ("npg_pointmass_32x32", True)])
def test_one_call_update_equals_call_sequence(name, layerwise, monkeypatch):
    """mjx_npg_update (K1, CG, device-side step length, step, K3 enqueued by one C call) == the same kernels driven call
    by call from Python, bit for bit; and == the reference's update."""
    from mjrl_amd.engine import UpdateEngine
    c = NpgCase(name)
    if layerwise:
        monkeypatch.setenv("MJX_FORCE_LAYERWISE", "1")
    tr = np.concatenate([np.float32(x).ravel() for x in c.tr]) if c.tr is not None else \
        np.concatenate([np.zeros(c.n), np.ones(c.n), np.zeros(c.m), np.ones(c.m)]).astype(np.float32)
    step = float(c.g["step"])
    out = []
    for one_call in (False, True):
        eng = UpdateEngine(c.n, c.m, c.hidden)
        eng.set_policy(c.theta0, c.theta0, tr, tr)
        eng.set_batch(c.obs, c.act, c.adv_w)
        if one_call:
            sa, kl = eng.npg_update(c.cg_iters, 1e-4, step, -3.0)
        else:
            g, _ = eng.surr_vpg(sync=False)
            eng.cg_solve(g, c.cg_iters, 1e-4, sync=False)
            eng.apply_npg_step(step, -3.0)
            sa, kl = eng.eval_surr_kl()
        out.append((eng.theta_new.cpu().numpy().copy(), eng.x.cpu().numpy().copy(), eng.grad.cpu().numpy().copy(), sa, kl, eng.deferred()))
        assert not eng.old_is_new
        eng.close()
    a, b = out
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3:] == b[3:]
    stp = b[0].astype(np.float64) - c.theta0
    assert rel(stp, c.g["new_params"].astype(np.float64) - c.theta0) < TOL_STEP
    assert abs(b[5]["alpha"] - float(c.g["alpha"])) < 1e-5 * float(c.g["alpha"])
    # the constant-step-size branch of the same entry point (npg_cg.py:128-130)
    eng = UpdateEngine(c.n, c.m, c.hidden)
    eng.set_policy(c.theta0, c.theta0, tr, tr)
    eng.set_batch(c.obs, c.act, c.adv_w)
    eng.npg_update(c.cg_iters, 1e-4, step, -3.0, const_alpha=0.2)
    assert rel(eng.theta_new.cpu().numpy().astype(np.float64) - c.theta0, 0.2 * c.g["cg_x"].astype(np.float64)) < TOL_STEP
    eng.close()


def test_rccl_inside_libmjx_one_rank_group():
    """the RCCL binding of libmjx (dlopen, ncclCommInitRank, ncclAllReduce on the launch stream) on a 1-rank
    communicator: sums are identities, the C loops with the collectives in place give the single-process bits."""
    import ctypes
    import torch
    from mjrl_amd._lib import check
    from mjrl_amd.engine import UpdateEngine
    c = NpgCase("npg_cfg2_small")
    tr = np.concatenate([np.zeros(c.n), np.ones(c.n), np.zeros(c.m), np.ones(c.m)]).astype(np.float32)
    res = []
    for with_comm in (False, True):
        eng = UpdateEngine(c.n, c.m, c.hidden)
        if with_comm:
            eng.backend.comm_init(0, 1, eng.backend.comm_unique_id())
            assert eng.backend.comm_world() == 1
            t = torch.arange(7, dtype=torch.float32, device=eng.device) + 0.5
            t64 = torch.arange(5, dtype=torch.float64, device=eng.device) - 2.25
            eng.backend.allreduce(t); eng.backend.allreduce(t64)
            torch.cuda.synchronize()
            assert torch.equal(t.cpu(), torch.arange(7, dtype=torch.float32) + 0.5) and torch.equal(t64.cpu(), torch.arange(5, dtype=torch.float64) - 2.25)
        eng.set_policy(c.theta0, c.theta0, tr, tr)
        eng.set_batch(c.obs, c.act, c.adv_w)
        sa, kl = eng.npg_update(c.cg_iters, 1e-4, float(c.g["step"]), -3.0)
        res.append((eng.theta_new.cpu().numpy().copy(), sa, kl))
        eng.close()
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1:] == res[1][1:]


@pytest.mark.parametrize("world", [2, 8, 16])
def test_peer_exchange_in_loop_back_gives_the_single_rank_bits(world):
    """libmjx's peer exchange with every "peer" mapped onto this rank's own buffer (mjx_peer_connect(NULL)): the stores into
    `world` buffers, the arrival counters, the bounded waits and the `world`-slot sums (2 / 8 / 16-slot instances of the folded
    vector update) all execute, the sums see this rank's vector + zeros -> NPG, TRPO and plain all-reduces give the bits of a
    context without a transport.  Also the call-order rules of mjx_peer_export / mjx_peer_connect."""
    import ctypes
    import torch
    from mjrl_amd import _lib
    from mjrl_amd.engine import UpdateEngine
    c = NpgCase("npg_cfg2_small")
    tr = np.concatenate([np.zeros(c.n), np.ones(c.n), np.zeros(c.m), np.ones(c.m)]).astype(np.float32)
    res = []
    for with_peer in (False, True):
        eng = UpdateEngine(c.n, c.m, c.hidden)
        lib, ctx = eng.backend.lib, eng.backend.ctx
        if with_peer:
            h = ctypes.create_string_buffer(64)
            assert lib.mjx_peer_connect(ctx, None) != 0                       # connect before export
            assert lib.mjx_peer_export(ctx, 0, 1, h) != 0 and lib.mjx_peer_export(ctx, 0, 17, h) != 0 and lib.mjx_peer_export(ctx, 3, 2, h) != 0
            _lib.check(lib.mjx_peer_export(ctx, 0, world, h))
            assert lib.mjx_peer_export(ctx, 0, world, h) != 0                 # a transport is already attached
            _lib.check(lib.mjx_peer_connect(ctx, None))
            assert lib.mjx_peer_connect(ctx, None) != 0                       # once
            assert eng.backend.comm_world() == world
            t = torch.arange(7, dtype=torch.float32, device=eng.device) + 0.5
            t64 = torch.arange(5, dtype=torch.float64, device=eng.device) - 2.25
            for _ in range(3):                                                # (both parities of the slot pairs)
                eng.backend.allreduce(t); eng.backend.allreduce(t64)
            torch.cuda.synchronize()
            assert torch.equal(t.cpu(), torch.arange(7, dtype=torch.float32) + 0.5) and torch.equal(t64.cpu(), torch.arange(5, dtype=torch.float64) - 2.25)
        eng.set_policy(c.theta0, c.theta0, tr, tr)
        eng.set_batch(c.obs, c.act, c.adv_w)
        sa, kl = eng.npg_update(c.cg_iters, 1e-4, float(c.g["step"]), -3.0)
        th_npg = eng.theta_new.cpu().numpy().copy()
        eng.set_policy(c.theta0, c.theta0, tr, tr)
        t2 = eng.trpo_update(c.cg_iters, 1e-4, 0.02, 0.002, -3.0)
        res.append((th_npg, sa, kl, eng.theta_new.cpu().numpy().copy(), t2["trials"], t2["alpha"], t2["kl"]))
        eng.close()
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1:3] == res[1][1:3]
    assert np.array_equal(res[0][3], res[1][3]) and res[0][4:] == res[1][4:]


def test_upload_download_never_touch_pageable_memory_and_round_trip():
    """utils/ingest.upload / download: host <-> device through page-locked bounce buffers (pageable
    hipMemcpy of ~1 MB and more leaves userptr registrations behind that later stall the GPU queues), bit-exact for the
    dtypes the package moves, any shape, non-contiguous inputs, sizes around the bounce threshold."""
    import torch
    from mjrl_amd import _lib
    from mjrl_amd.utils import ingest
    h = ingest.DeviceHandle(torch, torch.device("cuda", torch.cuda.current_device()), _lib.load())
    rng = np.random.RandomState(0)
    for shape, dt in (((1_000_003,), np.float64), ((70_001, 17), np.float32), ((5,), np.float64), ((16384,), np.int32),
                      ((16385,), np.int32), ((300_000,), np.int64), ((3, 1000, 7), np.float32), ((2_000_000,), np.uint8)):
        a = (rng.randn(*shape) * 100).astype(dt)
        t = ingest.upload(h, a)
        assert t.is_cuda and tuple(t.shape) == shape
        back = ingest.download(h, t)
        assert back.dtype == a.dtype and np.array_equal(back, a)
        assert np.array_equal(t.cpu().numpy(), a)
    nc = rng.randn(4000, 60)[:, ::3]                                       # non-contiguous view
    assert np.array_equal(ingest.download(h, ingest.upload(h, nc)), nc)
    assert np.array_equal(ingest.download(h, ingest.upload(h, nc, np.float32)), nc.astype(np.float32))
    big = torch.arange(3_000_000, dtype=torch.float32, device=h.device).reshape(1000, 3000)
    assert np.array_equal(ingest.download(h, big[:, ::2]), big[:, ::2].cpu().numpy())     # non-contiguous device tensor
    many = [ingest.upload(h, rng.randn(200_000)) for _ in range(12)]       # more uploads in flight than bounce buffers kept
    torch.cuda.synchronize()
    assert all(m.shape == (200_000,) for m in many)


def test_owned_downloads_are_zero_copy_exact_and_recycled_only_when_unreferenced():
    """utils/ingest.download_owned: the host block the paths get views of IS the page-locked memory the copy wrote; a buffer is
    handed out again only after every view of it has died, and a caller that keeps more blocks alive than the pool holds still
    gets correct (copied) arrays."""
    import torch
    from mjrl_amd import _lib
    from mjrl_amd.utils import ingest
    h = ingest.DeviceHandle(torch, torch.device("cuda", torch.cuda.current_device()), _lib.load())
    ts = [torch.randn(300_000, dtype=torch.float64, device=h.device) + i for i in range(ingest._OWNED_MAX + 4)]
    ref = [t.cpu().numpy() for t in ts]
    held = [ingest.download_owned(h, t) for t in ts]                       # all alive at once: the last ones fall back to copies
    views = [a[1000:2000] for a in held]
    for a, r in zip(held, ref):
        assert a.dtype == r.dtype and np.array_equal(a, r)
    addr = {a.ctypes.data for a in held}
    assert len(addr) == len(held)                                          # no buffer handed out twice while referenced
    first = held[0].ctypes.data
    keep = views[0]
    del held
    again = ingest.download_owned(h, ts[1])                                # buffer 0 is still referenced through `keep`
    assert again.ctypes.data != first and np.array_equal(keep, ref[0][1000:2000])
    del keep, views
    recycled = ingest.download_owned(h, ts[2])
    pool_addrs = {e["np"].ctypes.data for e in ingest._OWNED[(h.device.type, h.device.index)]}
    assert recycled.ctypes.data in pool_addrs and np.array_equal(recycled, ref[2])
    small = ingest.download_owned(h, ts[0][:10])                           # below the bounce threshold: an ordinary copy
    assert np.array_equal(small, ref[0][:10])
    del recycled, again
    wide = torch.randn(5_000_000, dtype=torch.float64, device=h.device)    # no buffer of the pool fits: an idle one makes room
    got = ingest.download_owned(h, wide)
    assert np.array_equal(got, wide.cpu().numpy())
    assert len(ingest._OWNED[(h.device.type, h.device.index)]) <= ingest._OWNED_MAX
    assert got.ctypes.data in {e["np"].ctypes.data for e in ingest._OWNED[(h.device.type, h.device.index)]}
    del got
    # blocks far smaller than the smallest buffer (a 100 kB batch) circulate through the SAME buffers: no allocation per call
    tiny = torch.randn(12_000, dtype=torch.float64, device=h.device)
    a1 = ingest.download_owned(h, tiny)
    a1 = ingest.download_owned(h, tiny)                                    # (the previous block is alive during the call: two buffers alternate)
    ids = {id(e) for e in ingest._OWNED[(h.device.type, h.device.index)]}
    for _ in range(20):
        a1 = ingest.download_owned(h, tiny)
        assert np.array_equal(a1, tiny.cpu().numpy())
    assert {id(e) for e in ingest._OWNED[(h.device.type, h.device.index)]} == ids


def test_time_index_kernel_equals_the_per_path_arange():
    """mjx_time_index (the baselines' time feature index, quadratic_baseline.py:28 / mlp_baseline.py:47) on ragged trajectories,
    including empty and one-step ones"""
    import torch
    from mjrl_amd import _lib
    from mjrl_amd._lib import check, ptr
    lib = _lib.load()
    rng = np.random.RandomState(3)
    lens = np.concatenate([[0, 1, 255, 256, 257, 1000, 0, 3], rng.randint(1, 2000, 200)]).astype(np.int64)
    off = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    offd = torch.from_numpy(off).cuda()
    out = torch.full((int(off[-1]),), -7, dtype=torch.int32, device="cuda")
    check(lib.mjx_time_index(ptr(offd), len(lens), ptr(out), 0))
    want = np.concatenate([np.arange(l, dtype=np.int32) for l in lens])
    assert np.array_equal(out.cpu().numpy(), want)


def test_device_resident_chain_equals_host_chain():
    """returns -> baseline values -> GAE -> whitening -> NPG update -> baseline fit with the blocks left on the device
    (utils/process_samples + the registry of utils/ingest) == the same iteration with every hand-over through the host
    arrays of the paths (registry dropped after each step: the reference's data flow)."""
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
    from mjrl_amd.policies.gaussian_mlp import MLP
    from mjrl_amd.utils import ingest, process_samples
    n, m = 11, 3
    spec = type("Spec", (), dict(observation_dim=n, action_dim=m, horizon=300))

    def run(device_chain):
        ingest.drop_shared()
        paths = synth.make_paths(40, 300, n, m, seed=3, ragged=True)
        pol = MLP(spec, hidden_sizes=(32, 32), seed=1, init_log_std=-0.5)
        bl = QuadraticBaseline(spec)
        agent = NPG(None, pol, bl, normalized_step_size=0.05)
        drop = (lambda: None) if device_chain else ingest.drop_shared_batch
        process_samples.compute_returns(paths, 0.99); drop()
        bl.fit(paths); drop()                                              # a fitted baseline, so that predictions are not zeros
        process_samples.compute_advantages(paths, bl, 0.99, 0.95); drop()
        hit = ingest.lookup(agent.engine.backend, paths, "advantages") is not None
        stats = agent.train_from_paths(paths); drop()
        errs = bl.fit(paths, return_errors=True)
        out = dict(theta=pol.get_param_values().copy(), coeffs=bl._coeffs.copy(), stats=np.array(stats), errs=np.array(errs), hit=hit,
                   adv=np.concatenate([p["advantages"] for p in paths]), ret=np.concatenate([p["returns"] for p in paths]),
                   bas=np.concatenate([p["baseline"] for p in paths]), kl=agent.last_update["kl_dist"])
        agent.engine.close()
        return out
    dev, host = run(True), run(False)
    assert dev["hit"] and not host["hit"]                                  # the two runs really took the two routes
    assert np.array_equal(dev["ret"], host["ret"]) and np.array_equal(dev["bas"], host["bas"]) and np.array_equal(dev["adv"], host["adv"])
    np.testing.assert_allclose(dev["stats"], host["stats"], rtol=1e-13)
    np.testing.assert_allclose(dev["coeffs"], host["coeffs"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(dev["errs"], host["errs"], rtol=1e-9)
    # whitening statistics formed on the device (two fp64 reduction passes) vs NumPy's: the fp32 advantages may differ in
    # the last bit, the update by fp32 round-off
    assert rel(dev["theta"], host["theta"]) < 1e-6 and abs(dev["kl"] - host["kl"]) < 1e-5 * host["kl"]
    ingest.drop_shared()


@pytest.mark.parametrize("n", [39, 46])
def test_wide_quadratic_baseline_block_gram_and_cholesky_route(monkeypatch, n):
    """obs 39 -> 824 quadratic features (BASELINE configs[4]; 46 = the widest hand_dapg observation): normal equations from the 128 x 128-block fp64-MFMA kernel,
    solved by Cholesky (>= 256 features) == the same fit through the reference's lstsq call, == the fp64 oracle's
    ridge fit on the explicit feature matrix."""
    from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
    spec = type("Spec", (), dict(observation_dim=n, action_dim=3, horizon=100))
    paths = synth.make_paths(60, 100, n, 3, seed=4, ragged=True)
    rng = np.random.RandomState(1)
    for p in paths:
        p["returns"] = rng.randn(len(p["rewards"])) + p["observations"][:, 0] * p["observations"][:, 1]
    preds = {}
    for route in ("cholesky", "lstsq"):
        if route == "lstsq":
            monkeypatch.setenv("MJX_RIDGE_LSTSQ", "1")
        bl = QuadraticBaseline(spec)
        e0, e1 = bl.fit(paths, return_errors=True)
        assert e0 == 1.0 and 0.0 < e1 < 1.0
        preds[route] = (bl.predict_batch(paths), bl._coeffs.copy(), e1)
    a, b = preds["cholesky"], preds["lstsq"]
    assert a[1].shape == (n + n * (n + 1) // 2 + 5,)
    np.testing.assert_allclose(a[0], b[0], rtol=1e-7, atol=1e-8)
    assert abs(a[2] - b[2]) < 1e-9
    F = O.quadratic_baseline_features([p["observations"] for p in paths])
    ret = np.concatenate([p["returns"] for p in paths])
    coef = O.ridge_fit(F, ret, 1e-3)
    np.testing.assert_allclose(a[0], F.dot(coef), rtol=1e-6, atol=1e-7)


def test_background_prefetch_hands_out_complete_blocks():
    """compute_returns starts the upload of observations / actions on a helper thread (utils/ingest.prefetch); whoever asks
    next -- on whatever stream position -- gets the complete block (consumer ordered after the transfer event), the very
    upload the helper made (no second one), and an edited array is noticed."""
    import torch
    from mjrl_amd.utils import ingest, process_samples
    ingest.drop_shared()
    n, m = 17, 6
    paths = synth.make_paths(300, 400, n, m, seed=11, ragged=True)
    process_samples.compute_returns(paths, 0.99)
    h = process_samples._handle()
    got = ingest.stage_shared(h, paths, ("observations", "actions"))
    obs = np.concatenate([p["observations"] for p in paths])
    act = np.concatenate([p["actions"] for p in paths])
    assert torch.equal(got["observations"]["raw"].cpu(), torch.from_numpy(obs))
    assert torch.equal(got["actions"]["f32"].cpu(), torch.from_numpy(act.astype(np.float32)))
    again = ingest.stage_shared(h, paths, ("observations",))
    assert again["observations"]["raw"].data_ptr() == got["observations"]["raw"].data_ptr()      # the same upload
    paths[0]["observations"][0, 0] += 1.0                                                          # an in-place edit of a probed array
    fresh = ingest.stage_shared(h, paths, ("observations",))
    assert torch.equal(fresh["observations"]["raw"].cpu(), torch.from_numpy(np.concatenate([p["observations"] for p in paths])))
    ingest.drop_shared()


def test_streamed_ingestion_equals_staging_after_sampling():
    """SURVEY 8f N2, second half (r06): chunks of finished trajectories handed to utils/ingest.StreamedBatch WHILE the rest is still
    being "sampled" end up in exactly the device blocks a stage-after-sampling builds -- rewards / observations raw fp64, actions
    fp32 by the converting gather, rows in episode order, bit for bit -- registered for the final list, so that compute_returns,
    the baseline, the update and the fit upload nothing; the whole iteration gives the same bits as without streaming.  Irregular
    producers (a list that is not the streamed episodes, a chunk beyond the capacity bound) fall back to staging after."""
    import torch
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
    from mjrl_amd.policies.gaussian_mlp import MLP
    from mjrl_amd.utils import ingest, process_samples
    n, m, T = 17, 6, 300
    spec = type("Spec", (), dict(observation_dim=n, action_dim=m, horizon=T))

    def iteration(streamed, warm):
        ingest.drop_shared()
        h = process_samples._handle()
        if warm:                  # a consumer has asked for the raw observations before (the quadratic baseline does, every iteration)
            wp = synth.make_paths(3, 20, n, m, seed=99)
            ingest.stage_shared(h, wp, ("observations",)); ingest.drop_shared_batch()
        paths = synth.make_paths(64, T, n, m, seed=5, ragged=True)
        info = None
        if streamed:
            sb = ingest.StreamedBatch(h)
            sb.begin(len(paths))
            for lo in range(0, len(paths), 10):                   # 7 chunks, the last one short
                sb.add(paths[lo:lo + 10], T)
            assert sb.finish(paths), sb.why
            info = dict(chunks=sb.chunks, rows=sb.rows)
            reg = ingest._SHARED[(h.device.type, h.device.index)]
            assert all(reg[k]["paths"] is paths for k in ("rewards", "observations", "actions"))
        blocks = ingest.stage_shared(h, paths, ("rewards", "observations", "actions"), raw=("rewards",) + (("observations",) if warm else ()))
        if streamed:              # ... served from the stream: no second staging pass
            assert blocks["rewards"]["raw"].data_ptr() == reg["rewards"]["raw"].data_ptr()
            assert blocks["actions"]["f32"].data_ptr() == reg["actions"]["f32"].data_ptr()
        snap = {k: (None if v["raw"] is None else v["raw"].clone(), v["f32"].clone()) for k, v in blocks.items()}
        pol = MLP(spec, hidden_sizes=(64, 64), seed=1, init_log_std=-0.5)
        bl = QuadraticBaseline(spec)
        agent = NPG(None, pol, bl, normalized_step_size=0.05)
        with ingest.trusted_iteration():
            process_samples.compute_returns(paths, 0.99)
            bl.fit(paths)
            process_samples.compute_advantages(paths, bl, 0.99, 0.95)
            stats = agent.train_from_paths(paths)
            errs = bl.fit(paths, return_errors=True)
        out = dict(snap=snap, theta=pol.get_param_values().copy(), coeffs=bl._coeffs.copy(), stats=np.array(stats), errs=np.array(errs),
                   adv=np.concatenate([p["advantages"] for p in paths]), ret=np.concatenate([p["returns"] for p in paths]), info=info, paths=paths)
        agent.engine.close()
        return out
    for warm in (True, False):
        a, b = iteration(True, warm), iteration(False, warm)
        assert a["info"]["chunks"] == 7 and a["info"]["rows"] == sum(len(p["rewards"]) for p in a["paths"])
        for k in ("rewards", "observations", "actions"):
            (ra, fa), (rb, fb) = a["snap"][k], b["snap"][k]
            assert (ra is None) == (rb is None) and (ra is None or torch.equal(ra, rb)) and torch.equal(fa, fb), k
        obs = np.concatenate([p["observations"] for p in a["paths"]])
        assert torch.equal(a["snap"]["observations"][1].cpu(), torch.from_numpy(obs.astype(np.float32)))
        assert np.array_equal(a["theta"], b["theta"]) and np.array_equal(a["coeffs"], b["coeffs"]) and np.array_equal(a["adv"], b["adv"])
        assert np.array_equal(a["ret"], b["ret"]) and np.array_equal(a["stats"], b["stats"]) and np.array_equal(a["errs"], b["errs"])
    # fall-backs: nothing registered, the ordinary staging serves the batch
    h = process_samples._handle()
    ingest.drop_shared()
    paths = synth.make_paths(8, 50, n, m, seed=6)
    sb = ingest.StreamedBatch(h); sb.begin(8); sb.add(paths[:4], 50); sb.add(paths[4:], 50)
    assert not sb.finish(list(reversed(paths))) and "not the streamed episodes" in sb.why
    sb = ingest.StreamedBatch(h); sb.begin(8); sb.add(paths[:4], 10)                          # a horizon that was a lie
    assert not sb.ok and "capacity" in sb.why and not sb.finish(paths)
    got = ingest.stage_shared(h, paths, ("observations",))["observations"]["f32"]
    assert torch.equal(got.cpu(), torch.from_numpy(np.concatenate([p["observations"] for p in paths]).astype(np.float32)))
    ingest.drop_shared()


def test_train_step_streams_its_rollouts_and_changes_nothing(monkeypatch):
    """BatchREINFORCE.train_step hands its own sampler a StreamedBatch (pool and in-process): three iterations end on the bits of the
    runs that stage after sampling (MJX_STREAM_INGEST=0), and the log of the stream says every batch was resident when sampling ended"""
    from mjrl_amd import samplers
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
    from mjrl_amd.policies.gaussian_mlp import MLP
    from mjrl_amd.utils import ingest
    from tests import _driver_env as DE
    spec = type("Spec", (), dict(observation_dim=6, action_dim=2, horizon=25))
    finals = {}
    for mode, num_cpu in (("1", 2), ("1", 1), ("0", 2)):
        monkeypatch.setenv("MJX_STREAM_INGEST", mode)
        ingest.drop_shared()
        pol = MLP(spec, hidden_sizes=(32, 32), seed=4, init_log_std=-0.5)
        agent = NPG(DE.make_point_mass, pol, QuadraticBaseline(spec), normalized_step_size=0.05, seed=11, save_logs=True)
        seen = []
        for _ in range(3):
            agent.train_step(N=32, sample_mode='trajectories', gamma=0.95, gae_lambda=0.97, num_cpu=num_cpu)
            seen.append(agent.last_ingest)
        if mode == "1":
            assert all(s is not None and s["streamed"] and s["chunks"] >= 4 for s in seen), seen
        else:
            assert all(s is None for s in seen)
        finals[(mode, num_cpu)] = (pol.get_param_values().copy(), agent.baseline._coeffs.copy())
        agent.engine.close()
    for k in (("1", 1), ("0", 2)):
        assert np.array_equal(finals[("1", 2)][0], finals[k][0]) and np.array_equal(finals[("1", 2)][1], finals[k][1]), k
    samplers.close_pools()
    ingest.drop_shared()


@pytest.mark.parametrize("d_in", [5, 8, 9, 21, 24, 25, 43, 63, 64])
def test_fused_mlp_predict_equals_the_layerwise_forward_and_numpy(monkeypatch, d_in):
    """r06: mjx_mlp_predict for the reference's default value network ((n + 4) -> 128 -> 128 -> 1, ReLU: mlp_baseline.py:21-28) is ONE
    launch of k_mlp_predict128 (chained MFMA accumulators, weights in LDS) instead of three GEMM launches per 131 072-row chunk.  Against
    the layer-by-layer route (MJX_MLP_PREDICT_FUSED=0) and an fp64 NumPy forward, row counts around the 32-sample tile and the grid edges."""
    import ctypes
    import torch
    from mjrl_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(100 + d_in)
    P = 128 * d_in + 128 + 128 * 128 + 128 + 128 + 1
    params = (rng.randn(P) / np.sqrt(np.concatenate([np.full(128 * d_in + 128, d_in), np.full(128 * 128 + 128, 128.0), np.full(129, 128.0)]))).astype(np.float32)
    W1 = params[:128 * d_in].reshape(128, d_in).astype(np.float64); b1 = params[128 * d_in:128 * d_in + 128].astype(np.float64)
    o = 128 * d_in + 128
    W2 = params[o:o + 128 * 128].reshape(128, 128).astype(np.float64); b2 = params[o + 128 * 128:o + 128 * 128 + 128].astype(np.float64)
    o += 128 * 128 + 128
    W3 = params[o:o + 128].astype(np.float64); b3 = float(params[o + 128])
    pt = torch.from_numpy(params).cuda()
    hid = (ctypes.c_int * 2)(128, 128)
    for N in (1, 31, 32, 33, 257, 8191, 100003, 400000):
        X = rng.randn(N, d_in).astype(np.float32)
        xt = torch.from_numpy(X).cuda()
        outs = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("MJX_MLP_PREDICT_FUSED", mode)
            out = torch.full((N,), float("nan"), dtype=torch.float32, device="cuda")
            _lib.check(lib.mjx_mlp_predict(_lib.ptr(xt), N, d_in, hid, 2, _lib.ptr(pt), _lib.ptr(out), None))
            torch.cuda.synchronize()
            outs[mode] = out.cpu().numpy().astype(np.float64)
        ref = np.maximum(np.maximum(X.astype(np.float64) @ W1.T + b1, 0) @ W2.T + b2, 0) @ W3 + b3
        scale = max(np.abs(ref).max(), 0.25)           # (a lone sample's output can be a small difference of O(0.3) terms)
        assert np.all(np.isfinite(outs["1"]))
        assert np.abs(outs["1"] - ref).max() < 3e-6 * scale, (N, np.abs(outs["1"] - ref).max() / scale)
        assert np.abs(outs["1"] - outs["0"]).max() < 3e-6 * scale, (N, np.abs(outs["1"] - outs["0"]).max() / scale)


@pytest.mark.parametrize("d_in", [9, 21, 23, 27, 35, 43, 50, 55, 56, 64, 96, 97, 115, 380, 768, 769])
def test_persistent_mlp_trainer_equals_per_step_launches(monkeypatch, d_in):
    """The persistent single-workgroup trainer of the MLP baseline (csrc/mlp_fit.h; two 32-feature blocks of the input
    layer beyond 31 inputs, as far as 160 KB of LDS reach: 55; the Adroit observations + 4 time features are 43..50)
    against the same minibatch-Adam chain issued as ~14 launches per step: same permutation, same arithmetic up to summation
    order -> parameters within 1e-5 of the movement after 2 x 40 steps.  (Kept short on purpose: on random regression data
    the ReLU chain is chaotic -- a last-bit difference that flips one unit's sign decides whether two correct implementations
    agree to 3e-7 or to per cent after a few hundred steps; tools/probe_fit_wide.py shows both, seed by seed.)"""
    import ctypes
    import torch
    from mjrl_amd import _lib
    from mjrl_amd._lib import check, ptr
    lib = _lib.load()
    N = 64 * 41
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(d_in)
    feat = torch.from_numpy(rng.randn(N, d_in).astype(np.float32)).to(dev)
    y = torch.from_numpy(rng.randn(N).astype(np.float32)).to(dev)
    P = 128 * d_in + 128 + 128 * 128 + 128 + 128 + 1
    p0 = (0.1 * rng.randn(P)).astype(np.float32)
    perm = torch.from_numpy(np.concatenate([rng.permutation(N), rng.permutation(N)]).astype(np.int32)).to(dev)
    hid = (ctypes.c_int * 2)(128, 128)
    out = {}
    for mode in ("persistent", "launches", "two_halves", "persistent_r03"):
        monkeypatch.setenv("MJX_MLP_FIT_LAUNCHES", "1" if mode == "launches" else "0")
        monkeypatch.setenv("MJX_FIT_REGMOM", "0" if mode == "persistent_r03" else "1")
        monkeypatch.setenv("MJX_FIT_ONEPASS", "1" if mode == "persistent" else "0")
        params = torch.from_numpy(p0.copy()).to(dev)
        m, v = torch.zeros(P, device=dev), torch.zeros(P, device=dev)
        loss = torch.zeros(32, dtype=torch.float64, device=dev)
        check(lib.mjx_mlp_fit_adam(ptr(feat), ptr(y), N, d_in, hid, 2, ptr(params), ptr(m), ptr(v), 0, ptr(perm), 2, 64, 1e-3, 1e-3, ptr(loss), None))
        torch.cuda.synchronize()
        out[mode] = (params.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy(), loss[:2].cpu().numpy())
    a, b = out["persistent"], out["launches"]
    move = np.linalg.norm(b[0] - p0)
    assert move > 0.1 * np.linalg.norm(p0) * 0.01
    assert np.linalg.norm(a[0] - b[0]) < 1e-5 * move
    assert np.linalg.norm(a[1] - b[1]) < 1e-4 * np.linalg.norm(b[1]) and np.linalg.norm(a[2] - b[2]) < 1e-4 * np.linalg.norm(b[2])
    np.testing.assert_allclose(a[3], b[3], rtol=1e-6)
    # r04: Adam moments resident in registers for the whole run vs streamed through L2 every step (MJX_FIT_REGMOM=0): the same
    # arithmetic on the same values in the same order -- parameters, both moments and the epoch losses bit for bit
    c, h = out["persistent_r03"], out["two_halves"]
    for k in range(4):
        np.testing.assert_array_equal(h[k], c[k])
    # ... and the one-pass kernel (up to 23 inputs; beyond that "persistent" IS the two-halves kernel) against both
    assert np.linalg.norm(h[0] - b[0]) < 1e-5 * move and np.linalg.norm(a[0] - h[0]) < 1e-5 * move


@pytest.mark.parametrize("hid", [(64, 64), (128, 128)])
def test_device_line_search_equals_call_by_call_trpo(hid):
    """mjx_trpo_update (K1, CG, step length, backtracking trials with the accept / shrink decision on the device, one read-back
    per batch of trials) against the same update issued call by call with a read-back after every trial (trpo.py:100-126):
    same number of trials, same step length, the same parameters bit for bit -- on the fused and on the layer-wise path,
    with a KL bound tight enough to force several shrinks (more than one batch of three trials), one that needs MORE THAN 24
    trials (the per-trial log of the device loop is a ring of 24) and one that is never met: the reference gives up after 100
    trials with alpha = 0 (trpo.py:119-126)."""
    import torch
    from mjrl_amd.engine import UpdateEngine
    n, m, N = 17, 6, 20000 + 3
    rng = np.random.RandomState(8)
    th = synth.perturbed_params(synth.init_params(n, m, hid))
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    obs, act, adv = rng.randn(N, n), rng.randn(N, m), rng.randn(N)
    if hid == (64, 64):
        cases = ((0.01, 0.02), (0.002, 0.02), (4e-5, 0.02), (-1.0, 0.02), (0.01, 1e60))
    else:
        cases = ((0.01, 0.02), (0.002, 0.02))
    # (the second: the step is sized for 5 x the KL it has to meet; third: ~27 trials; fourth: never; fifth: a step so long that
    #  every trial overflows fp32 -- surrogate / KL of the trials are not finite, the comparison `kl < kl_dist` rejects them like
    #  the reference's, nothing raises, the search gives up with alpha = 0)
    for kl_dist, step_size in cases:
        overflow = step_size > 1e30
        eng = UpdateEngine(n, m, hid)
        eng.set_policy(th, th, ident, ident)
        eng.set_batch(obs, act, adv)
        res = eng.trpo_update(10, 1e-4, step_size, kl_dist, -3.0)
        assert res is not None and res["accepted"] == (kl_dist > 0 and not overflow)
        one = dict(theta=eng.theta_new.clone(), **res)
        late = eng.deferred()
        # call by call
        eng.set_policy(th, th, ident, ident)
        g, surr_before = eng.surr_vpg()
        _, gdotx = eng.cg_solve(g, 10, 1e-4)
        alpha = np.sqrt(np.abs(step_size / (gdotx + 1e-20)))
        trials = 0
        for k in range(100):
            eng.apply_step(alpha, -3.0)
            surr_after, kl = eng.eval_surr_kl()
            trials += 1
            hist_last = (surr_after, kl)
            if kl < kl_dist:
                break
            alpha = 0.9 * alpha
            if k == 99:
                alpha = 0.0
        eng.apply_step(alpha, -3.0)                    # the reference's closing re-evaluation (trpo.py:122-125)
        surr_after, kl = eng.eval_surr_kl()
        assert trials == one["trials"] and (trials > 3 or kl_dist == 0.01)
        if kl_dist == 4e-5:
            assert 24 < trials < 100, trials
        if kl_dist < 0 or overflow:
            assert trials == 100 and one["alpha"] == 0.0 and torch.equal(eng.theta_new, eng.theta_old)
            assert np.isfinite(kl) and np.isfinite(surr_after)
        if overflow:
            assert not np.all(np.isfinite(one["history"][0]))
        assert float(alpha) == one["alpha"] and kl == one["kl"] and surr_after == one["surr_after"]
        assert torch.equal(eng.theta_new, one["theta"])
        assert late["surr_before"] == surr_before and late["gdotx"] == gdotx
        assert len(one["history"]) == trials and np.array_equal(one["history"][-1], hist_last, equal_nan=True)
        eng.close()
