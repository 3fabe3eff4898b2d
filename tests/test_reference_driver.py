"""The reference's OWN driver over this package's agent (north star: "drops into examples/policy_opt_job_script.py").

``mjrl.utils.train_agent.train_agent`` (mjrl/utils/train_agent.py:62-155) and ``mjrl.samplers.core.sample_paths``
(mjrl/samplers/core.py:99-148, mp.Pool of FORKED workers :189-210, the policy pickled into them :196) are imported UNMODIFIED
through oracle/ref_loader (sources in the build container, oracle/_ref bytecode on the GPU box) and run
``NPG(GymEnv(id), MLP, baseline)`` of mjrl_amd the way examples/policy_opt_job_script.py:60-101 does: deep copies of the policy every
iteration, pickles every save_freq, evaluation rollouts, logs, a second call that resumes from the job folder -- with a worker
pool (num_cpu = 2) forked from a parent that holds an mjx_ctx, page-locked staging blocks and libmjx's gather threads.  The env
(tests/_driver_env.py, a NumPy point-mass behind a stand-in gym.make) reports from inside every step which process ran it and
what that process did with libmjx.

CPU lane: the harness itself (reference agent through the reference driver), and mjrl_amd.samplers' own worker pool.
"""
import os
import pickle
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import _driver_env as DE  # noqa: E402

SEED, NITER, NTRAJ = 7, 4, 64
JOB = dict(seed=SEED, gamma=0.95, gae_lambda=0.97, sample_mode='trajectories', num_traj=NTRAJ, save_freq=1, evaluation_rollouts=4)


def _need_reference():
    ge = DE.install_fake_gym()
    if ge is None:
        pytest.skip("the reference is neither at /root/reference nor staged under oracle/_ref")
    return ge


# ------------------------------------------------------------------------------------------------------------ CPU lane
def _policy(seed=2):
    from mjrl_amd.policies.gaussian_mlp import MLP
    spec = type("Spec", (), dict(observation_dim=6, action_dim=2, horizon=25))
    return MLP(spec, hidden_sizes=(32, 32), seed=seed, init_log_std=-0.5)


def _same_paths(a, b):
    assert len(a) == len(b)
    for p, q in zip(a, b):
        for k in ("observations", "actions", "rewards"):
            assert np.array_equal(p[k], q[k]), k
        assert bool(p["terminated"]) == bool(q["terminated"])
        assert np.array_equal(p["agent_infos"]["mean"], q["agent_infos"]["mean"])


def test_native_sampler_pool_matches_the_serial_sampler():
    """mjrl_amd.samplers with num_cpu = 2: spawned workers, worker i seeded base_seed + i * paths_per_cpu (core.py:126-133) ==
    the episodes of the one-process call; ceil split when num_cpu does not divide num_traj (:124); the workers are other processes
    and never loaded libmjx"""
    from mjrl_amd import samplers
    pol = _policy()
    serial = samplers.sample_paths(8, DE.make_point_mass, pol, base_seed=100, num_cpu=1)
    pooled = samplers.sample_paths(8, DE.make_point_mass, pol, base_seed=100, num_cpu=2, suppress_print=True)
    _same_paths(serial, pooled)
    ev = DE.worker_evidence(pooled)
    assert len(ev["pids"]) == 2 and os.getpid() not in ev["pids"] and ev["loaded"] == 0, ev
    assert DE.worker_evidence(serial)["pids"] == [os.getpid()]
    assert len(samplers.sample_paths(5, DE.make_point_mass, pol, base_seed=3, num_cpu=2, suppress_print=True)) == 6
    # evaluation mode acts with the mean (core.py:71-72); env objects are accepted like factories
    ev_paths = samplers.sample_paths(2, DE.PointMassGym(), pol, eval_mode=True, base_seed=5, num_cpu=1)
    assert np.array_equal(ev_paths[0]["actions"], ev_paths[0]["agent_infos"]["evaluation"])
    # sample_data_batch (core.py:151-186): rounds of paths_per_call * num_cpu episodes, base_seed += 12345 per round
    a = samplers.sample_data_batch(120, DE.make_point_mass, pol, base_seed=9, num_cpu=2, paths_per_call=2)
    b = samplers.sample_data_batch(120, DE.make_point_mass, pol, base_seed=9, num_cpu=1, paths_per_call=4)
    assert sum(len(p["rewards"]) for p in a) >= 120
    _same_paths(a, b)
    with pytest.raises(RuntimeError):
        samplers.sample_paths(2, "Hopper-v2", pol, base_seed=1)        # an env ID needs mjrl + gym (and, with them, a registered env)
    samplers.close_pools()


class _RecordingSink:
    """what mjrl_amd.samplers asks of a sink (utils/ingest.StreamedBatch): begin(total), add(chunk, T) in episode order, abort(why)"""
    def __init__(self):
        self.total, self.chunks, self.T, self.aborted = None, [], [], None

    def begin(self, total):
        self.total = total

    def add(self, chunk, T=None):
        self.chunks.append(chunk); self.T.append(T)

    def abort(self, why):
        self.aborted = why


def test_native_sampler_streams_chunks_in_episode_order():
    """r06: with a sink the request is cut into more jobs than workers and every chunk is handed on, in EPISODE
    order, the moment it arrives -- the same episodes, the same path list as without one (pool and in-process)"""
    from mjrl_amd import samplers
    pol = _policy()
    plain = samplers.sample_paths(12, DE.make_point_mass, pol, base_seed=40, num_cpu=1)
    for num_cpu in (2, 1):
        sink = _RecordingSink()
        got = samplers.sample_paths(12, DE.make_point_mass, pol, base_seed=40, num_cpu=num_cpu, suppress_print=True, sink=sink)
        _same_paths(plain, got)
        assert sink.total == 12 and sink.aborted is None and len(sink.chunks) > num_cpu and set(sink.T) == {25}
        flat = [p for c in sink.chunks for p in c]
        assert len(flat) == len(got) and all(a is b for a, b in zip(flat, got))       # the very dicts of the returned list, in its order
    # ceil split (core.py:124): 5 episodes on 2 workers are 6, streamed or not
    sink = _RecordingSink()
    assert len(samplers.sample_paths(5, DE.make_point_mass, pol, base_seed=3, num_cpu=2, suppress_print=True, sink=sink)) == 6 and sink.total == 6
    samplers.close_pools()


class EnvClassWithKwargs(DE.PointMassGym):
    def __init__(self, shift=0.0):
        super().__init__()
        self.shift = shift


def test_native_sampler_accepts_env_classes_and_survives_unpicklable_payloads(monkeypatch):
    """(i) an env CLASS is a factory (core.py:36-39 instantiates any callable; a class also has `step`); (ii) a payload the
    workers cannot be sent -- here a lambda -- is served in this process with one warning instead of a lost task and a timeout;
    (iii) num_cpu='max' is capped"""
    import warnings
    from mjrl_amd import samplers
    pol = _policy()
    a = samplers.sample_paths(3, EnvClassWithKwargs, pol, base_seed=7, num_cpu=1, env_kwargs=dict(shift=1.0))
    b = samplers.sample_paths(3, DE.make_point_mass, pol, base_seed=7, num_cpu=1)
    _same_paths(a, b)
    monkeypatch.setattr(samplers, "_SERIAL_ONLY", {})
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        c = samplers.sample_paths(4, lambda: DE.PointMassGym(), pol, base_seed=7, num_cpu=2, suppress_print=True)
        d = samplers.sample_paths(4, lambda: DE.PointMassGym(), pol, base_seed=7, num_cpu=2, suppress_print=True)
    _same_paths(c, samplers.sample_paths(4, DE.make_point_mass, pol, base_seed=7, num_cpu=1))
    _same_paths(c, d)
    assert len([w for w in caught if "served in the training process itself" in str(w.message)]) == 1
    assert DE.worker_evidence(c)["pids"] == [os.getpid()]
    monkeypatch.setenv("MJX_SAMPLER_MAX_WORKERS", "3")
    assert samplers._resolve_num_cpu('max') == min(3, os.cpu_count()) and samplers._resolve_num_cpu(None) == 1
    samplers.close_pools()


def test_a_main_script_without_a_guard_is_not_handed_to_spawned_workers(tmp_path):
    """a spawned worker imports the main module again: an unguarded training script would re-run itself in every worker.  Such a
    main module is detected, the request is served in-process (one warning), the script's results are the serial ones."""
    import subprocess
    import sys
    script = tmp_path / "unguarded_job.py"
    script.write_text(
        "import sys, warnings\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import _driver_env as DE\n"
        "from mjrl_amd import samplers\n"
        "from mjrl_amd.policies.gaussian_mlp import MLP\n"
        "spec = type('Spec', (), dict(observation_dim=6, action_dim=2, horizon=25))\n"
        "pol = MLP(spec, hidden_sizes=(32, 32), seed=1, init_log_std=-0.5)\n"
        "with warnings.catch_warnings(record=True) as w:\n"
        "    warnings.simplefilter('always')\n"
        "    paths = samplers.sample_paths(4, DE.make_point_mass, pol, base_seed=1, num_cpu=2, suppress_print=True)\n"
        "print('RESULT', len(paths), len(DE.worker_evidence(paths)['pids']), sum('without a __main__ guard' in str(x.message) for x in w))\n"
        % (ROOT, os.path.join(ROOT, "tests")))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")] == ["RESULT 4 1 1"], r.stdout      # ran once, in one process


class _SlowEnv(DE.PointMassGym):
    def step(self, a):
        import time
        time.sleep(0.2)
        return super().step(a)


def make_slow_env():
    return _SlowEnv()


def test_native_sampler_timeouts_retry_then_fail_loudly():
    """core.py:189-203: a worker set that does not answer within max_process_time is torn down and the request retried,
    max_timeouts times; then no rollouts -- an error here, not a None the caller trips over"""
    from mjrl_amd import samplers
    with pytest.raises(RuntimeError, match="worker timeouts"):
        samplers.sample_paths(2, make_slow_env, _policy(), base_seed=1, num_cpu=2, max_process_time=0.5, max_timeouts=2, suppress_print=True)
    samplers.close_pools()


def _reference_job(tmp_path, name, num_cpu, niter=NITER, eps=0.0, damping=1e-4):
    """the reference's own NPG + MLP + QuadraticBaseline through the reference's train_agent (the yardstick); eps: relative
    perturbation of the initial parameters (how far does the REFERENCE land from itself?)"""
    ge = _need_reference()
    from mjrl.algos.npg_cg import NPG
    from mjrl.baselines.quadratic_baseline import QuadraticBaseline
    from mjrl.policies.gaussian_mlp import MLP
    from mjrl.utils.train_agent import train_agent
    e = ge.GymEnv(DE.ENV_ID)
    policy = MLP(e.spec, hidden_sizes=(32, 32), seed=SEED, init_log_std=-0.5)
    if eps:
        th = policy.get_param_values()
        policy.set_param_values((th * (1.0 + eps * np.random.RandomState(0).randn(th.size))).astype(np.float32))
    agent = NPG(e, policy, QuadraticBaseline(e.spec), normalized_step_size=0.05, seed=SEED, save_logs=True,
                FIM_invert_args={'iters': 10, 'damping': damping})
    train_agent(job_name=str(tmp_path / name), agent=agent, niter=niter, num_cpu=num_cpu, **JOB)
    return agent


def test_reference_driver_harness_on_cpu(tmp_path):
    """the harness alone: the reference's agent through the reference's driver, 2 forked workers == 1 process"""
    a2 = _reference_job(tmp_path, "ref2", 2, niter=2)
    a1 = _reference_job(tmp_path, "ref1", 1, niter=2)
    assert np.array_equal(a2.policy.get_param_values(), a1.policy.get_param_values())
    assert os.path.exists(tmp_path / "ref2" / "logs" / "log.csv") and os.path.exists(tmp_path / "ref2" / "iterations" / "policy_1.pickle")


# ------------------------------------------------------------------------------------------------------------ GPU lane
def _our_agent(ge, kind, seed=SEED, evidence=None, damping=1e-4):
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.baselines.mlp_baseline import MLPBaseline
    from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
    from mjrl_amd.policies.gaussian_mlp import MLP

    class Recording(NPG):
        def train_from_paths(self, paths):
            if evidence is not None:
                evidence.append(DE.worker_evidence(paths))
            return super().train_from_paths(paths)
    e = ge.GymEnv(DE.ENV_ID)                                           # policy_opt_job_script.py:60
    policy = MLP(e.spec, hidden_sizes=(32, 32), seed=SEED, init_log_std=-0.5)
    baseline = (MLPBaseline(e.spec, reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3) if kind == "mlp"
                else QuadraticBaseline(e.spec))
    return Recording(e, policy, baseline, normalized_step_size=0.05, seed=seed, save_logs=True, FIM_invert_args={'iters': 10, 'damping': damping})


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["quadratic", "mlp"])
def test_reference_train_agent_and_fork_pool_drive_our_agent(kind, tmp_path):
    ge = _need_reference()
    from mjrl.utils.train_agent import train_agent
    from mjrl_amd._lib import load
    import ctypes
    # ---- (A) 4 iterations, 2 forked workers, checkpoints every iteration, evaluation rollouts
    ev = []
    agent = _our_agent(ge, kind, evidence=ev)
    agent.engine                                                       # the parent holds an mjx_ctx BEFORE the first fork
    st = (ctypes.c_int64 * 2)()
    load().mjx_process_state(st)
    assert st[0] > 0 and st[1] == 0
    train_agent(job_name=str(tmp_path / "A"), agent=agent, niter=NITER, num_cpu=2, **JOB)
    final = agent.policy.get_param_values().copy()
    assert np.all(np.isfinite(final)) and len(ev) == NITER
    for e in ev:
        # every env step of every iteration ran in one of two OTHER processes, forked from this one after it had created device
        # state (flag 1), and none of them made a single device-touching libmjx call
        assert len(e["pids"]) == 2 and os.getpid() not in e["pids"], e
        assert e["forked"] == 1 and e["device_calls"] == 0 and e["loaded"] == 1, e
    jobA = tmp_path / "A"
    for f in ("logs/log.csv", "logs/log.pickle", "logs/stoc_pol_mean.png", "results.txt", "iterations/best_policy.pickle",
              "iterations/policy_3.pickle", "iterations/baseline_3.pickle"):
        assert os.path.exists(jobA / f), f
    with open(jobA / "iterations" / "policy_3.pickle", "rb") as fp:
        assert np.array_equal(pickle.load(fp).get_param_values(), final)
    with open(jobA / "iterations" / "best_policy.pickle", "rb") as fp:
        best = pickle.load(fp)
    assert best.get_action(np.zeros(6))[0].shape == (2,)
    # ---- the reference's own agent through the same driver: same log keys, same training curve
    ref = _reference_job(tmp_path, "R", 2)
    ours_log, ref_log = agent.logger.log, ref.logger.log
    assert set(ref_log) <= set(ours_log), sorted(set(ref_log) - set(ours_log))
    assert all(len(ours_log[k]) == NITER for k in ref_log)
    if kind == "quadratic":
        th0 = _our_agent(ge, kind).policy.get_param_values().astype(np.float64)

        def dist(a, b):
            return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64) - th0))
        # (i) the job script's defaults (damping 1e-4, 10 CG iterations; 1 600 samples for 1 346 parameters): this job is
        #     ILL-POSED in fp32 -- the reference lands percent away from ITSELF when its initial parameters move by 1e-7 relative
        #     (an unconverged Krylov solve on a nearly singular Fisher matrix) -- so the yardstick for "the same training run" is
        #     the reference's own sensitivity, measured here
        sens = dist(_reference_job(tmp_path, "Rp", 2, eps=1e-7).policy.get_param_values(), ref.policy.get_param_values())
        d4 = dist(final, ref.policy.get_param_values())
        assert ours_log["stoc_pol_mean"][0] == pytest.approx(ref_log["stoc_pol_mean"][0], rel=1e-6)     # the same first batch
        assert d4 < 5 * sens + 1e-3, (d4, sens)
        # (ii) the same job made well-posed (damping 1.0: the reference moves 1e-6 .. 3e-6 under that perturbation, over 1 and over
        #      4 iterations): sampling, returns, GAE, four NPG updates and four baseline fits under the reference's driver end at
        #      the reference's policy
        ours_w = _our_agent(ge, kind, damping=1.0)
        train_agent(job_name=str(tmp_path / "Aw"), agent=ours_w, niter=NITER, num_cpu=2, **JOB)
        ref_w = _reference_job(tmp_path, "Rw", 2, damping=1.0)
        sens_w = dist(_reference_job(tmp_path, "Rwp", 2, eps=1e-7, damping=1.0).policy.get_param_values(), ref_w.policy.get_param_values())
        dw = dist(ours_w.policy.get_param_values(), ref_w.policy.get_param_values())
        curve = np.max(np.abs(np.array(ours_w.logger.log["stoc_pol_mean"]) - np.array(ref_w.logger.log["stoc_pol_mean"])))
        print("under the reference driver, 4 iterations: default damping %.2e from the reference's policy (the reference from itself "
              "under a 1e-7 perturbation: %.2e); damping 1.0: %.2e (the reference from itself: %.2e), training-curve difference %.2e"
              % (d4, sens, dw, sens_w, curve))
        assert dw < 5e-5, (dw, sens_w)
        assert curve < 1e-5 * max(1.0, np.max(np.abs(ref_w.logger.log["stoc_pol_mean"])))
        assert abs(ours_w.logger.log["eval_score"][-1] - ref_w.logger.log["eval_score"][-1]) < 1e-5 * abs(ref_w.logger.log["eval_score"][-1])
        ours_w.engine.close()
    # ---- (B) the same job in one process: per-episode seeding (core.py:52-57) makes the batches identical
    if kind == "quadratic":                                            # (the MLP baseline's minibatch order follows the PARENT's RNG,
        agent_b = _our_agent(ge, kind)                                 #  which in-process sampling re-seeds -- in the reference too)
        train_agent(job_name=str(tmp_path / "B"), agent=agent_b, niter=NITER, num_cpu=1, **JOB)
        assert np.array_equal(agent_b.policy.get_param_values(), final)
    # ---- (C) 2 iterations, then a second call with a FRESH agent that resumes from the job folder (train_agent.py:15-60,88-93)
    agent_c = _our_agent(ge, kind)
    train_agent(job_name=str(tmp_path / "C"), agent=agent_c, niter=2, num_cpu=2, **JOB)
    fresh = _our_agent(ge, kind, seed=SEED + 2 * NTRAJ)               # (the reference does not checkpoint agent.seed: batch_reinforce.py:91)
    fresh.policy.set_param_values(np.zeros_like(final))                # whatever it held is replaced by the pickles
    train_agent(job_name=str(tmp_path / "C"), agent=fresh, niter=NITER, num_cpu=2, **JOB)
    assert fresh.logger.max_len == NITER
    if kind == "quadratic":
        assert np.array_equal(fresh.policy.get_param_values(), final)  # bit-identical continuation
    else:
        assert np.all(np.isfinite(fresh.policy.get_param_values()))
    for a in (agent, agent_c, fresh):
        a.engine.close()


@pytest.mark.gpu
def test_native_sampler_pool_under_a_live_context():
    """mjrl_amd.samplers' spawned pool from a training process that holds device state: train_step(num_cpu = 2) == train_step(num_cpu = 1)"""
    from mjrl_amd import samplers
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
    finals = []
    for num_cpu in (2, 1):
        ev = []

        class Recording(NPG):
            def train_from_paths(self, paths):
                ev.append(DE.worker_evidence(paths))
                return super().train_from_paths(paths)
        pol = _policy(seed=4)
        spec = type("Spec", (), dict(observation_dim=6, action_dim=2, horizon=25))
        agent = Recording(DE.make_point_mass, pol, QuadraticBaseline(spec), normalized_step_size=0.05, seed=11, save_logs=True)
        agent.engine
        for _ in range(3):
            agent.train_step(N=32, sample_mode='trajectories', gamma=0.95, gae_lambda=0.97, num_cpu=num_cpu)
        if num_cpu == 2:
            assert all(len(e["pids"]) == 2 and os.getpid() not in e["pids"] and e["loaded"] == 0 for e in ev), ev
        finals.append(pol.get_param_values().copy())
        agent.engine.close()
    assert np.array_equal(finals[0], finals[1])
    samplers.close_pools()


@pytest.mark.gpu
def test_a_forked_child_is_refused_device_work():
    """include/mjx.h mjx_process_state: after a fork from a process that holds device state, mjx_create / mjx_malloc fail with
    MJX_ERR_STATE (a message, not a hang inside the runtime) and mjx_device_count reports no device"""
    import ctypes
    from mjrl_amd._lib import load
    from mjrl_amd.engine import UpdateEngine
    eng = UpdateEngine(6, 2, (32, 32))
    lib = load()
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:                                                       # child: report through the pipe, leave without cleanup
        try:
            st = (ctypes.c_int64 * 2)()
            lib.mjx_process_state(st)
            ctx, p = ctypes.c_void_p(), ctypes.c_void_p()
            hid = (ctypes.c_int * 2)(32, 32)
            rc_create = lib.mjx_create(ctypes.byref(ctx), 0, 6, 2, hid, 2)
            msg = lib.mjx_last_error()
            rc_malloc = lib.mjx_malloc(ctypes.byref(p), 1024)
            os.write(w, repr((int(st[0]), int(st[1]), rc_create, rc_malloc, lib.mjx_device_count(), b"forked" in msg)).encode())
        finally:
            os._exit(0)
    os.close(w)
    os.waitpid(pid, 0)
    got = eval(os.read(r, 4096).decode())
    assert got == (0, 1, -2, -2, 0, True), got
    st = (ctypes.c_int64 * 2)()
    lib.mjx_process_state(st)
    assert st[0] > 0 and st[1] == 0                                   # the parent is unaffected
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("algorithm", ["NPG", "PPO"])
def test_the_unmodified_job_script_runs_on_the_gpu_classes(tmp_path, algorithm):
    """examples/policy_opt_job_script.py, byte for byte (source in the build container, staged bytecode on the GPU box), through
    `python -m mjrl_amd.dropin`-style module aliasing: its own import block (:8-15) now yields this package's MLP / MLPBaseline /
    NPG / PPO, its own GymEnv and train_agent drive them with num_cpu = 2 forked workers, checkpoints and evaluation rollouts, from
    a config file in the format of examples/example_configs/*.txt."""
    import json
    import subprocess
    _need_reference()
    cfg = tmp_path / "cfg.txt"
    cfg.write_text(repr({
        'env': DE.ENV_ID, 'algorithm': algorithm, 'seed': 123, 'sample_mode': 'trajectories', 'rl_num_traj': 32, 'rl_num_iter': 3,
        'num_cpu': 2, 'save_freq': 1, 'eval_rollouts': 2, 'exp_notes': 'point mass behind a stand-in gym.make',
        'policy_size': (32, 32), 'init_log_std': -0.5, 'vf_hidden_size': (128, 128), 'vf_batch_size': 64, 'vf_epochs': 2,
        'vf_learn_rate': 1e-3, 'rl_step_size': 0.05, 'rl_gamma': 0.995, 'rl_gae': 0.97,
        'alg_hyper_params': dict() if algorithm == "NPG" else dict(epochs=2, mb_size=64, learn_rate=3e-4)}))
    job, summary = tmp_path / "job", tmp_path / "summary.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_job_script_runner.py"), str(job), str(cfg), str(summary)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    s = json.load(open(summary))
    assert s["agent"] == ("mjrl_amd.algos.npg_cg.NPG" if algorithm == "NPG" else "mjrl_amd.algos.ppo_clip.PPO"), s
    assert s["policy"] == "mjrl_amd.policies.gaussian_mlp" and s["baseline"] == "mjrl_amd.baselines.mlp_baseline"
    assert s["train_agent"] == "mjrl.utils.train_agent" and s["gymenv"] == "mjrl.utils.gym_env"       # the reference's own driver
    assert s["native_fused"] and len(s["stoc_pol_mean"]) == 3 and all(np.isfinite(s["stoc_pol_mean"])) and all(0 < v < 10 for v in s["vf_after"])
    for f in ("job_config.json", "logs/log.csv", "iterations/policy_2.pickle", "iterations/baseline_2.pickle", "iterations/best_policy.pickle", "results.txt"):
        assert os.path.exists(job / f), f
    assert "Starting policy learning" in r.stdout and "ITERATION : 2" in r.stdout
