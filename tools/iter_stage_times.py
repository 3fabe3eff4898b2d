"""Where one post-sampling iteration spends its time, with the staging calls (thread, keys, start, duration) of every thread."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.algos.npg_cg import NPG
from mjrl_amd.utils import ingest as _ingest; _ingest.tune_malloc()   # a training process (what train_step / dropin.install do)
from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
from mjrl_amd.policies.gaussian_mlp import MLP
from mjrl_amd.utils import process_samples, ingest
spec = type("Spec", (), dict(observation_dim=17, action_dim=6, horizon=1000))
rng = np.random.RandomState(0)
pol = MLP(spec, hidden_sizes=(64, 64), seed=1, init_log_std=-0.5)
bl = QuadraticBaseline(spec)
agent = NPG(None, pol, bl, normalized_step_size=0.05)
log = []
T0 = [0.0]
orig_stage = ingest.PathStager.stage
def stage(self, paths, keys=("observations","actions"), wait=True, hostcast=()):
    t = time.perf_counter()
    r = orig_stage(self, paths, keys, wait, hostcast)
    log.append((threading.current_thread().name[:12], keys, hostcast, round(1e3*(t-T0[0]),2), round(1e3*(time.perf_counter()-t),2)))
    return r
ingest.PathStager.stage = stage
orig_gather = None
for it in range(6):
    paths = [dict(observations=rng.randn(1000, 17), actions=rng.randn(1000, 6), rewards=rng.randn(1000), terminated=False) for _ in range(1000)]
    torch.cuda.synchronize(); log.clear(); T0[0] = t0 = time.perf_counter()
    with ingest.trusted_iteration():
        process_samples.compute_returns(paths, 0.995); t1 = time.perf_counter()
        process_samples.compute_advantages(paths, bl, 0.995, 0.97); t2 = time.perf_counter()
        agent.train_from_paths(paths); torch.cuda.synchronize(); t3 = time.perf_counter()
        bl.fit(paths); torch.cuda.synchronize(); t4 = time.perf_counter()
    ingest.drop_shared_batch()
    print(it, [round(1e3*x,2) for x in (t1-t0, t2-t1, t3-t2, t4-t3)], log)
