"""Soak: 300 NPG / TRPO / PPO iterations with batches of changing size and raggedness; checks finiteness, KL behaviour and that
device memory does not grow."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.algos.npg_cg import NPG
from mjrl_amd.utils import ingest as _ingest; _ingest.tune_malloc()   # a training process (what train_step / dropin.install do)
from mjrl_amd.algos.trpo import TRPO
from mjrl_amd.algos.ppo_clip import PPO
from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
from mjrl_amd.policies.gaussian_mlp import MLP
from mjrl_amd.utils import process_samples
spec = type("Spec", (), dict(observation_dim=17, action_dim=6, horizon=1000))
rng = np.random.RandomState(0)
out = {}
for name, cls, kw in (("npg", NPG, dict(normalized_step_size=0.05)), ("trpo", TRPO, dict(kl_dist=0.01)), ("ppo", PPO, dict(epochs=1, mb_size=256))):
    pol = MLP(spec, hidden_sizes=(64, 64), seed=1, init_log_std=-0.5)
    bl = QuadraticBaseline(spec)
    agent = cls(None, pol, bl, **kw)
    mem = []
    t0 = time.perf_counter()
    for it in range(100):
        n_traj = int(rng.randint(20, 200))
        paths = []
        for _ in range(n_traj):
            T = int(rng.randint(1, 1000))
            obs = rng.randn(T, 17)
            act = pol.model.forward(np.float32(obs)) + np.exp(pol.log_std_val) * rng.randn(T, 6)
            paths.append(dict(observations=obs, actions=act, rewards=-np.sum(act ** 2, axis=1) + rng.randn(T) * 0.1, terminated=bool(T < 999)))
        process_samples.compute_returns(paths, 0.995)
        process_samples.compute_advantages(paths, bl, 0.995, 0.97)
        stats = agent.train_from_paths(paths)
        bl.fit(paths)
        th = pol.get_param_values()
        assert np.all(np.isfinite(th)) and np.all(np.isfinite(stats)), (name, it)
        if name != "ppo":
            assert 0.0 <= agent.last_update["kl_dist"] < 0.2, (name, it, agent.last_update)
        mem.append(torch.cuda.memory_allocated())
    out[name] = dict(seconds=round(time.perf_counter() - t0, 2), mem_first_MB=round(mem[10] / 2**20, 1), mem_last_MB=round(mem[-1] / 2**20, 1),
                     mem_max_MB=round(max(mem) / 2**20, 1), final_log_std=float(np.mean(pol.log_std_val)))
# the same loop with the MLP baseline (persistent Adam trainer, all-gather-free single rank) and the host's resident set watched:
# the page-locked hand-out buffers (ingest.download_owned) and the staging blocks must not grow with the iterations
import psutil
from mjrl_amd.baselines.mlp_baseline import MLPBaseline
pol = MLP(spec, hidden_sizes=(64, 64), seed=1, init_log_std=-0.5)
bl = MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=1, learn_rate=1e-3)
agent = NPG(None, pol, bl, normalized_step_size=0.05)
proc = psutil.Process()
rss, mem = [], []
t0 = time.perf_counter()
for it in range(60):
    n_traj = int(rng.randint(20, 200))
    paths = []
    for _ in range(n_traj):
        T = int(rng.randint(1, 1000))
        obs = rng.randn(T, 17)
        act = pol.model.forward(np.float32(obs)) + np.exp(pol.log_std_val) * rng.randn(T, 6)
        paths.append(dict(observations=obs, actions=act, rewards=-np.sum(act ** 2, axis=1) + rng.randn(T) * 0.1, terminated=bool(T < 999)))
    process_samples.compute_returns(paths, 0.995)
    process_samples.compute_advantages(paths, bl, 0.995, 0.97)
    stats = agent.train_from_paths(paths)
    bl.fit(paths)
    assert np.all(np.isfinite(pol.get_param_values())) and np.all(np.isfinite(stats)), ("npg+mlp", it)
    rss.append(proc.memory_info().rss); mem.append(torch.cuda.memory_allocated())
out["npg_mlp_baseline"] = dict(seconds=round(time.perf_counter() - t0, 2), host_rss_MB_at_10=round(rss[10] / 2**20, 1), host_rss_MB_last=round(rss[-1] / 2**20, 1),
                               host_rss_MB_max=round(max(rss) / 2**20, 1), dev_mem_MB_at_10=round(mem[10] / 2**20, 1), dev_mem_MB_last=round(mem[-1] / 2**20, 1))
# r06: the layer-wise path under the same loop (a 128 x 128 policy, d > 8 192: the multi-workgroup CG update, the one-pass likelihood head,
# the old-output reuse of the one-call updates) -- NPG and TRPO with batches of changing size
for name, cls, kw in (("npg_layerwise_128x128", NPG, dict(normalized_step_size=0.05)), ("trpo_layerwise_128x128", TRPO, dict(kl_dist=0.01))):
    pol = MLP(spec, hidden_sizes=(128, 128), seed=1, init_log_std=-0.5)
    bl = QuadraticBaseline(spec)
    agent = cls(None, pol, bl, **kw)
    assert not agent.engine.fused
    mem = []
    t0 = time.perf_counter()
    for it in range(40):
        n_traj = int(rng.randint(20, 120))
        paths = []
        for _ in range(n_traj):
            T = int(rng.randint(1, 1000))
            obs = rng.randn(T, 17)
            act = pol.model.forward(np.float32(obs)) + np.exp(pol.log_std_val) * rng.randn(T, 6)
            paths.append(dict(observations=obs, actions=act, rewards=-np.sum(act ** 2, axis=1) + rng.randn(T) * 0.1, terminated=bool(T < 999)))
        process_samples.compute_returns(paths, 0.995)
        process_samples.compute_advantages(paths, bl, 0.995, 0.97)
        stats = agent.train_from_paths(paths)
        bl.fit(paths)
        assert np.all(np.isfinite(pol.get_param_values())) and np.all(np.isfinite(stats)), (name, it)
        assert 0.0 <= agent.last_update["kl_dist"] < 0.2, (name, it, agent.last_update)
        mem.append(torch.cuda.memory_allocated())
    out[name] = dict(seconds=round(time.perf_counter() - t0, 2), mem_at_10_MB=round(mem[10] / 2**20, 1), mem_last_MB=round(mem[-1] / 2**20, 1),
                     mem_max_MB=round(max(mem) / 2**20, 1), final_log_std=float(np.mean(pol.log_std_val)))
print(json.dumps(out))
