"""train_agent's lifecycle on a LIVE GPU agent (mjrl/utils/train_agent.py:97-131): every iteration the reference deep-copies
agent.policy, every save_freq iterations it pickles policy and baseline, and a resumed job unpickles both into a fresh agent
(:42-45).  Here an agent that owns an mjx_ctx with a bound batch goes through exactly that while it trains, and a second
agent rebuilt from the pickles continues bit-identically."""
import copy
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class PointMass:                       # obs = [pos(2), vel(2), target(2)], act = force(2), horizon 25
    horizon = 25

    def __init__(self):
        self.rng = np.random.RandomState(0)

    def set_seed(self, s):
        self.rng = np.random.RandomState(s)

    def reset(self):
        self.p, self.v, self.g, self.t = self.rng.uniform(-1, 1, 2), np.zeros(2), self.rng.uniform(-1, 1, 2), 0
        return np.concatenate([self.p, self.v, self.g])

    def step(self, a):
        self.v = 0.9 * self.v + 0.1 * np.clip(a, -1, 1); self.p = self.p + 0.1 * self.v; self.t += 1
        return np.concatenate([self.p, self.v, self.g]), -float(np.linalg.norm(self.p - self.g)), False, {}


SPEC = type("Spec", (), dict(observation_dim=6, action_dim=2, horizon=25))
STEP = dict(N=60, sample_mode='trajectories', gamma=0.95, gae_lambda=0.97, num_cpu=1)


def make(kind, algo, policy=None, baseline=None, seed=3):
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.algos.trpo import TRPO
    from mjrl_amd.baselines.mlp_baseline import MLPBaseline
    from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
    from mjrl_amd.policies.gaussian_mlp import MLP
    if policy is None:
        policy = MLP(SPEC, hidden_sizes=(32, 32), seed=2, init_log_std=-0.5)
    if baseline is None:
        baseline = MLPBaseline(SPEC, reg_coef=1e-3, batch_size=64, epochs=1, learn_rate=1e-3) if kind == "mlp" else QuadraticBaseline(SPEC)
    if algo == "trpo":
        return TRPO(PointMass(), policy, baseline, kl_dist=0.02, seed=seed, save_logs=True)
    return NPG(PointMass(), policy, baseline, normalized_step_size=0.05, seed=seed, save_logs=True)


@pytest.mark.parametrize("kind,algo", [("quadratic", "npg"), ("mlp", "npg"), ("quadratic", "trpo")])
def test_deepcopy_pickle_resume_mid_training(kind, algo):
    # ---- the uninterrupted job: 4 iterations, deep copy of the policy every iteration, pickles after the second
    np.random.seed(0)
    agent = make(kind, algo)
    best, saved = None, None
    for i in range(4):
        best = copy.deepcopy(agent.policy)                         # train_agent.py:102 -- the agent's engine holds a bound batch from i >= 1
        before = agent.policy.get_param_values().copy()
        assert np.array_equal(best.get_param_values(), before)
        agent.train_step(**STEP)
        assert np.array_equal(best.get_param_values(), before)     # the copy does not follow the live policy ...
        assert not np.array_equal(agent.policy.get_param_values(), before)
        a, info = best.get_action(np.zeros(6))                     # ... and is a working policy
        assert a.shape == (2,) and np.all(np.isfinite(info["mean"]))
        if i == 1:
            saved = dict(policy=pickle.dumps(agent.policy), baseline=pickle.dumps(agent.baseline), best=pickle.dumps(best),
                         rng=np.random.get_state(), seed=agent.seed, params=agent.policy.get_param_values().copy())
    final = agent.policy.get_param_values().copy()
    probe = dict(observations=np.random.RandomState(9).randn(25, 6), rewards=np.zeros(25))
    final_bl = np.asarray(agent.baseline.predict(probe)).copy()
    agent.engine.close()

    # ---- the resumed job: fresh agent around the unpickled policy / baseline (train_agent.py:42-45), same RNG position
    pol, bl = pickle.loads(saved["policy"]), pickle.loads(saved["baseline"])
    assert np.array_equal(pol.get_param_values(), saved["params"]) and np.array_equal(pol.get_old_param_values(), saved["params"])
    assert np.array_equal(pickle.loads(saved["best"]).get_param_values(), pickle.loads(saved["best"]).get_old_param_values())
    resumed = make(kind, algo, policy=pol, baseline=bl, seed=saved["seed"])
    np.random.set_state(saved["rng"])
    for i in range(2):
        resumed.train_step(**STEP)
    assert np.array_equal(resumed.policy.get_param_values(), final)                  # bit-identical continuation
    np.testing.assert_array_equal(np.asarray(resumed.baseline.predict(probe)), final_bl)
    resumed.engine.close()


def test_bc_dapg_quadratic_baseline_pipeline_vs_reference():
    """BASELINE configs[4] as a PIPELINE on one policy object (obs 39, act 28, 512 x 512): BC pre-training on demonstrations
    (behavior_cloning.py:107-136, transforms from the demonstrations) -> 2 DAPG iterations (dapg.py:54-141), each = returns,
    GAE advantages against the quadratic baseline, the update, the baseline fit (batch_reinforce.py:94-110 without the sampler)
    -- against the UNMODIFIED reference run of tests/golden/make_golden_pipeline.py on the same seeded data.
    (i) the chain as a whole: our BC, then our DAPG on the policy our BC produced; (ii) stage parity with the reference's own
    intermediate parameters as the starting point of each stage, so that the stages' tolerances do not compound."""
    from mjrl_amd.algos.behavior_cloning import BC
    from mjrl_amd.algos.dapg import DAPG
    from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
    from mjrl_amd.policies.gaussian_mlp import MLP
    from mjrl_amd.utils import process_samples
    from oracle import synth
    from tests._cases import load
    g = load("pipeline_cfg5")
    n, m, hid = int(g["n"]), int(g["m"]), tuple(int(h) for h in g["hidden"])
    S = int(g["stride"])
    spec = type("Spec", (), dict(observation_dim=n, action_dim=m, horizon=200))

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.linalg.norm(a - b) / np.linalg.norm(b))

    demos = synth.make_paths(int(g["demo_n_traj"]), int(g["demo_T"]), n, m, seed=int(g["demo_seed"]))
    pol = MLP(spec, hidden_sizes=hid, seed=1, init_log_std=-0.5)
    pol.set_param_values(synth.perturbed_params(synth.init_params(n, m, hid, seed=1, init_log_std=-0.5), scale=float(g["theta_scale"])))
    assert np.array_equal(pol.get_param_values()[::S], g["theta0_sub"])
    # ---- stage 1: BC
    bc = BC(demos, pol, epochs=int(g["bc_epochs"]), batch_size=int(g["bc_mb"]), lr=float(g["bc_lr"]), loss_type='MLE', save_logs=False,
            set_transforms=True)
    assert rel(pol.get_param_values()[::S], g["theta_start_sub"]) < 1e-6
    np.testing.assert_allclose(pol.model.in_shift, g["in_shift"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pol.model.in_scale, g["in_scale"], rtol=1e-5)
    np.testing.assert_allclose(pol.model.out_scale, g["out_scale"], rtol=1e-5)
    np.random.seed(int(g["seed_np"]))
    bc.train()
    bc_err = np.linalg.norm(pol.get_param_values() - g["theta_bc"]) / float(g["bc_moved"])
    assert bc_err < 5e-3, bc_err                                   # minibatch-Adam chain: statistical parity (156 steps)

    def dapg_iterations(policy, tag):
        bl = QuadraticBaseline(spec)
        agent = DAPG(None, policy, bl, demo_paths=demos, kl_dist=float(g["kl_dist"]), lam_0=float(g["lam_0"]), lam_1=float(g["lam_1"]),
                     FIM_invert_args={'iters': int(g["cg_iters"]), 'damping': float(g["damping"])}, save_logs=True)
        assert not agent.engine.fused
        res = []
        for it, seed in enumerate(int(s) for s in g["path_seeds"]):
            paths = synth.make_paths(int(g["n_traj"]), int(g["T"]), n, m, seed=seed)
            before = policy.get_param_values().astype(np.float64)
            process_samples.compute_returns(paths, float(g["gamma"]))
            process_samples.compute_advantages(paths, bl, float(g["gamma"]), float(g["gae_lambda"]))
            stats = agent.train_from_paths(paths)
            errs = bl.fit(paths, return_errors=True)
            res.append(dict(step=policy.get_param_values().astype(np.float64) - before, alpha=agent.last_update["alpha"],
                            kl=agent.last_update["kl_dist"], stats=stats, coeffs=np.asarray(bl._coeffs, np.float64).copy(), errs=errs,
                            adv0=np.asarray(paths[0]["advantages"]).copy(), ret0=np.asarray(paths[0]["returns"]).copy()))
            print(tag, "iteration", it, "alpha", res[-1]["alpha"], float(g["alpha_it%d" % it]), "kl", res[-1]["kl"], float(g["kl_it%d" % it]),
                  "step rel", rel(res[-1]["step"][::S], g["step_it%d_sub" % it]))
        agent.engine.close()
        return res

    # ---- (i) the chain on the policy object BC just trained: finite, close to the reference's trajectory
    chain = dapg_iterations(pol, "chain")
    for it, r in enumerate(chain):
        assert np.all(np.isfinite(r["step"])) and 0 < r["kl"] < 4 * float(g["kl_dist"])
        assert abs(r["alpha"] - float(g["alpha_it%d" % it])) < 1e-5 * float(g["alpha_it%d" % it])
        assert rel(r["step"][::S], g["step_it%d_sub" % it]) < 2e-5                 # (measured 8.0e-6 / 8.9e-6: BC's 156 Adam steps included)
    assert rel(pol.get_param_values()[::S], g["theta_it1_sub"]) < 1e-5
    # ---- (ii) stage parity from the reference's post-BC parameters
    pol2 = MLP(spec, hidden_sizes=hid, seed=1, init_log_std=-0.5)
    pol2.set_param_values(g["theta_bc"])
    for mdl in (pol2.model, pol2.old_model):
        mdl.set_transformations(g["in_shift"], g["in_scale"], g["out_shift"], g["out_scale"])
    sync = dapg_iterations(pol2, "resynced")
    r0 = sync[0]
    np.testing.assert_allclose(r0["ret0"], g["ret0_it0"], rtol=1e-12)
    np.testing.assert_allclose(r0["adv0"], g["adv0_it0"], rtol=1e-9, atol=1e-12)
    assert abs(r0["alpha"] - float(g["alpha_it0"])) < 1e-5 * float(g["alpha_it0"])
    assert abs(r0["kl"] - float(g["kl_it0"])) < 1e-4 * float(g["kl_it0"])
    assert rel(r0["step"][::S], g["step_it0_sub"]) < 1e-5, rel(r0["step"][::S], g["step_it0_sub"])    # the north-star bar (measured 5.8e-6)
    np.testing.assert_allclose(r0["stats"], g["stats_it0"], rtol=1e-12)
    np.testing.assert_allclose(r0["errs"], g["bl_errors_it0"], rtol=1e-6)
    assert rel(r0["coeffs"], g["bl_coeffs_it0"]) < 1e-5
    assert rel(sync[1]["step"][::S], g["step_it1_sub"]) < 1e-5                    # (measured 6.7e-6)
    assert rel(pol2.get_param_values()[::S], g["theta_it1_sub"]) < 1e-5


def test_background_fit_equals_the_blocking_fit_and_fills_the_log(tmp_path, monkeypatch):
    """BatchREINFORCE.train_step starts MLPBaseline's fit on a side stream and returns (the fitted baseline is not read before the
    next iteration's compute_advantages, batch_reinforce.py:94-112).  Same training run as with the blocking fit, bit for bit;
    VF_error_* / time_VF enter the log as pending entries and are plain floats once anything settles the fit -- the next
    iteration, a deep copy / pickle of the baseline (train_agent.py:102,129-131), save_log."""
    from mjrl_amd.utils.logger import PendingValue
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MJX_ASYNC_FIT", mode)
        np.random.seed(0)
        agent = make("mlp", "npg")
        pending_seen = 0
        for i in range(3):
            agent.train_step(**STEP)
            last = agent.logger.log["VF_error_after"][-1]
            pending_seen += isinstance(last, PendingValue)
            if i == 1:
                snap = copy.deepcopy(agent.baseline)                    # waits for the fit in flight; the copy carries its result
                assert snap.__dict__.get("_pending") is None and np.array_equal(snap.params, agent.baseline.params)
                assert not isinstance(agent.logger.log["VF_error_after"][-1], PendingValue)      # ... and the log entry was delivered
        agent.logger.save_log(str(tmp_path))                            # settles whatever is left
        log = agent.logger.log
        assert all(isinstance(v, float) or np.isscalar(v) for k in ("time_VF", "VF_error_before", "VF_error_after") for v in log[k])
        assert all(0.0 < v < 10.0 for v in log["VF_error_after"]) and all(v > 0 for v in log["time_VF"])
        if mode == "1":                                                  # the epoch permutations were drawn under the update and taken over
            assert agent.baseline.__dict__.get("predraw_stats") == (3, 0), agent.baseline.__dict__.get("predraw_stats")
            assert log["VF_predraw_taken"] == [1, 2, 3] and log["VF_predraw_discarded"] == [0, 0, 0]      # ... and the log says so (r06)
        runs[mode] = dict(theta=agent.policy.get_param_values().copy(), bl=agent.baseline.params.copy(), m=agent.baseline.adam_m.copy(),
                          steps=agent.baseline.adam_steps, err=[float(v) for v in log["VF_error_after"]], pending=pending_seen,
                          pickled=pickle.loads(pickle.dumps(agent.baseline)).params.copy())
        agent.engine.close()
    a, b = runs["1"], runs["0"]
    assert a["pending"] >= 1 and b["pending"] == 0                      # the background mode really deferred something
    assert np.array_equal(a["theta"], b["theta"]) and np.array_equal(a["bl"], b["bl"]) and np.array_equal(a["m"], b["m"])
    assert a["steps"] == b["steps"] and a["err"] == b["err"] and np.array_equal(a["pickled"], a["bl"])


def test_speculative_permutation_draws_are_discarded_when_the_stream_moved():
    """MLPBaseline.predraw draws the next fit's epoch permutations from a COPY of NumPy's global generator state while the update
    runs; fit_async takes them only if the global state is still the one they started from.  Somebody drawing in between (here: the
    test) makes the fit draw again -- the result is the fit a plain `fit` produces from that stream position, bit for bit."""
    from mjrl_amd.baselines.mlp_baseline import MLPBaseline
    from mjrl_amd.utils import process_samples
    spec = type("Spec", (), dict(observation_dim=6, action_dim=2, horizon=40))
    rng = np.random.RandomState(1)
    paths = [dict(observations=rng.randn(40, 6), rewards=rng.randn(40), terminated=False) for _ in range(30)]
    process_samples.compute_returns(paths, 0.99)
    out = {}
    for mode in ("taken", "discarded", "plain"):
        import torch
        torch.manual_seed(3); np.random.seed(3)
        bl = MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3)
        np.random.seed(21)
        h = bl.predraw(30 * 40) if mode != "plain" else None
        if mode != "taken":
            np.random.randn(7)                                           # somebody uses the global stream in between
        e = bl.fit_async(paths, return_errors=True, predrawn=h).result()
        out[mode] = (bl.params.copy(), e, np.random.randn(3), bl.__dict__.get("predraw_stats"))
    assert out["taken"][3] == (1, 0) and out["discarded"][3] == (0, 1)
    assert np.array_equal(out["discarded"][0], out["plain"][0]) and out["discarded"][1] == out["plain"][1]
    assert np.array_equal(out["discarded"][2], out["plain"][2])          # ... and the stream stands where the plain fit leaves it
    # the taken draws are the plain fit's draws from the UNDISTURBED stream position
    torch.manual_seed(3); np.random.seed(3)
    ref = MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3)
    np.random.seed(21)
    ref.fit(paths)
    assert np.array_equal(out["taken"][0], ref.params) and np.array_equal(out["taken"][2], np.random.randn(3))
    # a consumer of np.random that draws in EVERY iteration forfeits the speculative draws every time: one warning, at the second miss
    import warnings
    bl = MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=1, learn_rate=1e-3)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for _ in range(3):
            h = bl.predraw(30 * 40)
            np.random.randn(1)
            bl.fit_async(paths, predrawn=h).result()
    assert bl.__dict__["predraw_stats"] == (0, 3)
    assert len([w for w in caught if "discarded twice in a row" in str(w.message)]) == 1


def test_a_straggling_workgroup_cannot_launder_a_timed_out_fit(monkeypatch):
    """in the several-workgroup trainer (csrc/mlp_fit.h MULTI) a workgroup that waited ~2 s on a grid barrier gives up.
    If workgroup 0 was the straggler it later passes every abandoned barrier at once and writes finite losses and parameters LAST.
    The give-up is recorded in a word of its own and k_mlp_fit_verdict turns it into NaN losses after the kernel; the baseline then
    keeps the state it had before the fit, serves the pending log entries (NaN) and raises.  MJX_FIT_FAULT=straggler delays
    workgroup 0 by 2.6 s."""
    import torch
    from mjrl_amd._lib import MjxError
    from mjrl_amd.baselines.mlp_baseline import MLPBaseline
    from mjrl_amd.utils import process_samples
    spec = type("Spec", (), dict(observation_dim=60, action_dim=2, horizon=40))       # 64 inputs: two workgroups
    rng = np.random.RandomState(2)
    paths = [dict(observations=rng.randn(40, 60), rewards=rng.randn(40), terminated=False) for _ in range(20)]
    process_samples.compute_returns(paths, 0.99)
    torch.manual_seed(5); np.random.seed(5)
    bl = MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3)
    bl.fit(paths)
    before = (bl.params.copy(), bl.adam_m.copy(), bl.adam_v.copy(), bl.adam_steps, list(bl.epoch_losses))
    assert before[3] == 2 * (800 // 64 - 1) and np.all(np.isfinite(before[4]))
    monkeypatch.setenv("MJX_FIT_FAULT", "straggler")
    pend = bl.fit_async(paths, return_errors=True)
    delivered = []
    pend.hooks.append(lambda errors, ms: delivered.append((errors, ms)))
    with pytest.raises(MjxError, match="keeps the parameters it had before"):
        pend.result()
    assert pend.done and len(delivered) == 1 and np.isnan(delivered[0][0][0]) and np.isnan(delivered[0][0][1]) and delivered[0][1] > 2000.0
    assert pend.result() == pend.value                     # a second look no longer raises (nor returns None into a float())
    assert np.array_equal(bl.params, before[0]) and np.array_equal(bl.adam_m, before[1]) and np.array_equal(bl.adam_v, before[2])
    assert bl.adam_steps == before[3] and list(bl.epoch_losses) == before[4]
    monkeypatch.delenv("MJX_FIT_FAULT")
    e0, e1 = bl.fit(paths, return_errors=True)             # the baseline is still usable
    assert np.isfinite(e0) and np.isfinite(e1) and bl.adam_steps == 2 * before[3] and not np.array_equal(bl.params, before[0])


@pytest.mark.parametrize("kind", ["quadratic", "linear"])
def test_ridge_fit_async_equals_fit_and_settles_on_access(kind, monkeypatch):
    """r06: the ridge baselines' fit_async -- Gram kernel enqueued on the caller's stream, the F x F solve (and, for the logged errors,
    the evaluation of the new coefficients) on a helper thread -- gives the coefficients and errors of fit() bit for bit; reading
    `_coeffs`, predict, pickle and deepcopy wait for it; a second fit settles the first; MJX_ASYNC_FIT=0 runs in place."""
    from mjrl_amd.baselines.quadratic_baseline import LinearBaseline, QuadraticBaseline, PendingRidgeFit
    from mjrl_amd.utils import ingest, process_samples
    cls = QuadraticBaseline if kind == "quadratic" else LinearBaseline
    spec = type("Spec", (), dict(observation_dim=11, action_dim=3, horizon=120))
    rng = np.random.RandomState(3)

    def batch(seed):
        r = np.random.RandomState(seed)
        ps = [dict(observations=r.randn(T, 11), rewards=r.randn(T), terminated=False) for T in r.randint(20, 121, size=50)]
        process_samples.compute_returns(ps, 0.99)
        return ps
    a, b = batch(1), batch(2)
    ref = cls(spec)
    e_ref = [ref.fit(a, return_errors=True)]
    c1 = ref._coeffs.copy()
    e_ref.append(ref.fit(b, return_errors=True))
    ingest.drop_shared_batch()
    bl = cls(spec)
    p1 = bl.fit_async(a, return_errors=True)
    assert isinstance(p1, PendingRidgeFit)
    delivered = []
    p1.hooks.append(lambda errs, ms: delivered.append((errs, ms)))
    snap = copy.deepcopy(bl)                                  # waits; the copy carries the finished fit and no pending state
    assert p1.done and "_pending" not in snap.__dict__ and np.array_equal(snap._coeffs, c1)
    assert delivered and delivered[0][0] == p1.value == e_ref[0] and delivered[0][1] > 0.0
    assert np.array_equal(pickle.loads(pickle.dumps(bl))._coeffs, c1)
    p2 = bl.fit_async(b, return_errors=True)                  # error_before uses the coefficients of the first fit
    pred = bl.predict(b[0])                                   # settles p2
    assert p2.done and p2.result() == e_ref[1] and np.array_equal(bl._coeffs, ref._coeffs)
    np.testing.assert_array_equal(pred, ref.predict(b[0]))
    p3 = bl.fit_async(a)                                      # no errors asked for: value None, coefficients those of a fit from here
    assert p3.result() is None and p3.done
    ref.fit(a)
    assert np.array_equal(bl._coeffs, ref._coeffs)
    monkeypatch.setenv("MJX_ASYNC_FIT", "0")
    p4 = bl.fit_async(b, return_errors=True)
    assert p4.done and p4.future is None and p4.value == ref.fit(b, return_errors=True) and np.array_equal(bl._coeffs, ref._coeffs)
    ingest.drop_shared()
