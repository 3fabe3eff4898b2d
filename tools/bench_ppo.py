"""PPO minibatch epochs on device: steps/s at HalfCheetah shapes (mb 64, as the reference default) and at mb 4096."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.algos.ppo_clip import PPO
from mjrl_amd.policies.gaussian_mlp import MLP
spec = type("Spec", (), dict(observation_dim=17, action_dim=6, horizon=1000))
rng = np.random.RandomState(0)
paths = [dict(observations=rng.randn(1000, 17), actions=rng.randn(1000, 6), rewards=rng.randn(1000), advantages=rng.randn(1000)) for _ in range(100)]
out = {}
for mb in (64, 4096):
    pol = MLP(spec, hidden_sizes=(64, 64), seed=1, init_log_std=-0.5)
    agent = PPO(None, pol, None, epochs=1, mb_size=mb)
    agent.train_from_paths(paths); torch.cuda.synchronize()
    t0 = time.perf_counter(); agent.train_from_paths(paths); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    steps = 100000 // mb
    out["mb%d" % mb] = dict(steps=steps, seconds=dt, us_per_step=1e6 * dt / steps, samples_per_s=steps * mb / dt)
print(json.dumps(out))
