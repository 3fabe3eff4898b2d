"""cProfile of NPG.train_from_paths on fresh fp64 host batches (1M timesteps): the host side of tools/bench_e2e.py."""
import os, sys, time, json, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.algos.npg_cg import NPG
from mjrl_amd.utils import ingest as _ingest; _ingest.tune_malloc()   # a training process (what train_step / dropin.install do)
from mjrl_amd.policies.gaussian_mlp import MLP
spec = type("Spec", (), dict(observation_dim=17, action_dim=6, horizon=1000))
rng = np.random.RandomState(0)
paths = [dict(observations=rng.randn(1000, 17), actions=rng.randn(1000, 6), rewards=rng.randn(1000), advantages=rng.randn(1000), terminated=False) for _ in range(1000)]
pol = MLP(spec, hidden_sizes=(64, 64), seed=1, init_log_std=-0.5)
agent = NPG(None, pol, None, normalized_step_size=0.05)
def fresh():
    return [dict(observations=p["observations"].copy(), actions=p["actions"].copy(), rewards=p["rewards"], advantages=p["advantages"], terminated=False) for p in paths]
batches = [fresh() for _ in range(9)]
for b in batches[:3]:
    agent.train_from_paths(b)
torch.cuda.synchronize()
pr = cProfile.Profile(); ts = []
pr.enable()
for b in batches[3:9]:
    t0 = time.perf_counter(); agent.train_from_paths(b); torch.cuda.synchronize(); ts.append(round(1e3 * (time.perf_counter() - t0), 2))
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(38); print(s.getvalue()[:7000]); print(ts)
