"""Host-side attribution of one post-sampling training iteration at 1M timesteps (quadratic baseline): cProfile over
returns -> advantages -> NPG update -> baseline fit, with a device synchronisation at the end of every phase."""
import cProfile, io, os, pstats, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.algos.npg_cg import NPG
from mjrl_amd.utils import ingest as _ingest; _ingest.tune_malloc()   # a training process (what train_step / dropin.install do)
from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
from mjrl_amd.baselines.mlp_baseline import MLPBaseline
from mjrl_amd.policies.gaussian_mlp import MLP
from mjrl_amd.utils import process_samples
kind = sys.argv[1] if len(sys.argv) > 1 else "quadratic"
spec = type("Spec", (), dict(observation_dim=17, action_dim=6, horizon=1000))
rng = np.random.RandomState(0)
pol = MLP(spec, hidden_sizes=(64, 64), seed=1, init_log_std=-0.5)
bl = QuadraticBaseline(spec) if kind == "quadratic" else MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3)
agent = NPG(None, pol, bl, normalized_step_size=0.05)

def make():
    """STREAM=1 (r06): the batch is handed to utils/ingest.StreamedBatch in 20 chunks as it is produced, like mjrl_amd.samplers does"""
    paths = [dict(observations=rng.randn(1000, 17), actions=rng.randn(1000, 6), rewards=rng.randn(1000), terminated=False) for _ in range(1000)]
    if os.environ.get("STREAM") == "1":
        _ingest.drop_shared_batch()
        sb = _ingest.StreamedBatch.for_current_device()
        sb.begin(len(paths))
        for lo in range(0, len(paths), 50):
            sb.add(paths[lo:lo + 50], 1000)
        assert sb.finish(paths), sb.why
    return paths

def iteration(paths, ts):
    from mjrl_amd.utils import ingest
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with ingest.trusted_iteration():               # as train_step does (batch_reinforce.py)
        process_samples.compute_returns(paths, 0.995); torch.cuda.synchronize(); t1 = time.perf_counter()
        process_samples.compute_advantages(paths, bl, 0.995, 0.97); torch.cuda.synchronize(); t2 = time.perf_counter()
        agent.train_from_paths(paths); torch.cuda.synchronize(); t3 = time.perf_counter()
        (bl.fit_async(paths) if os.environ.get('ASYNC_FIT', '1') == '1' and hasattr(bl, 'fit_async') else bl.fit(paths)); t4 = time.perf_counter()
    ingest.drop_shared_batch()
    ts.append([round(1e3 * x, 2) for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)])

ts = []
for _ in range(3):
    iteration(make(), ts)
pr = cProfile.Profile()
for _ in range(5):
    p = make()                     # (made right before its iteration: a streamed batch is the one registered batch)
    pr.enable()
    iteration(p, ts)
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
print(json.dumps({"kind": kind, "phases_ms [returns, advantages, update, fit, total]": ts}))
