"""Ridge-baseline normal equations at the BASELINE configs[4] shard (obs 39 -> 824 quadratic features, 1M timesteps per GPU)
and at configs[1] (obs 17 -> 175 features): time of mjx_bl_gram and of a whole fit."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.baselines._features import DeviceBlock, FEAT_QUADRATIC
from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
out = {}
rng = np.random.RandomState(0)
for n, ntraj, T in ((17, 1000, 1000), (39, 5000, 200)):
    paths = [dict(observations=rng.randn(T, n), rewards=rng.randn(T), returns=rng.randn(T)) for _ in range(ntraj)]
    blk = DeviceBlock(paths, 'obs')
    y = blk.returns_dev()
    def gram():
        G = blk.gram(FEAT_QUADRATIC, y); torch.cuda.synchronize(); return G
    gram()
    t0 = time.perf_counter(); G = gram(); dt = time.perf_counter() - t0
    F = G.shape[0] - 1
    spec = type("Spec", (), dict(observation_dim=n, action_dim=3, horizon=T))
    bl = QuadraticBaseline(spec)
    bl.fit(paths)
    t0 = time.perf_counter(); bl.fit(paths); torch.cuda.synchronize(); dfit = time.perf_counter() - t0
    out["obs%d" % n] = dict(features=F, rows=ntraj * T, gram_ms=1e3 * dt, fp64_TFLOPs=ntraj * T * (F + 1) * (F + 2) / dt / 1e12, fit_ms=1e3 * dfit)
print(json.dumps(out))
