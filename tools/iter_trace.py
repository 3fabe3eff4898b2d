"""Per-iteration, per-function wall times (with a device synchronisation after each wrapped call) of the post-sampling
iteration at 1M timesteps: where do the slow iterations lose their time?"""
import functools, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.algos.npg_cg import NPG
from mjrl_amd.utils import ingest as _ingest; _ingest.tune_malloc()   # a training process (what train_step / dropin.install do)
from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
from mjrl_amd.baselines import _features
from mjrl_amd.policies.gaussian_mlp import MLP
from mjrl_amd.utils import process_samples, ingest

LOG = []
def timed(obj, name, tag=None):
    f = getattr(obj, name)
    @functools.wraps(f)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); LOG.append((tag or name, 1e3 * (time.perf_counter() - t0)))
        return r
    setattr(obj, name, g)

timed(ingest.PathStager, "stage", "stager.stage")
timed(ingest.PathStager, "_add_paths_native", "stager.gather+send")
timed(_features.DeviceBlock, "__init__", "DeviceBlock")
timed(_features.DeviceBlock, "predict_linear_dev", "predict_dev")
timed(_features.DeviceBlock, "gram", "gram")
timed(process_samples, "_hand_out", "hand_out")
timed(process_samples, "_rewards_block", "rewards_block")
timed(_features, "_time_index", "time_index")
timed(ingest, "upload", "upload")
import numpy as _np
_orig_empty = torch.empty
def _empty(*a, **k):
    t0 = time.perf_counter(); r = _orig_empty(*a, **k); dt = 1e3 * (time.perf_counter() - t0)
    if dt > 1.0: LOG.append(("torch.empty%s%s" % (tuple(a[0]) if a and not isinstance(a[0], int) else a, "pin" if k.get("pin_memory") else ""), dt))
    return r
torch.empty = _empty
_orig_asc = _np.ascontiguousarray
def _asc(*a, **k):
    t0 = time.perf_counter(); r = _orig_asc(*a, **k); dt = 1e3 * (time.perf_counter() - t0)
    if dt > 1.0: LOG.append(("ascontiguousarray", dt))
    return r
_np.ascontiguousarray = _asc
timed(ingest, "derived", "derived")
timed(ingest, "stage_shared", "stage_shared")
timed(_features, "torch_dev", "torch_dev")

spec = type("Spec", (), dict(observation_dim=17, action_dim=6, horizon=1000))
rng = np.random.RandomState(0)
pol = MLP(spec, hidden_sizes=(64, 64), seed=1, init_log_std=-0.5)
bl = QuadraticBaseline(spec)
agent = NPG(None, pol, bl, normalized_step_size=0.05)
def make():
    return [dict(observations=rng.randn(1000, 17), actions=rng.randn(1000, 6), rewards=rng.randn(1000), terminated=False) for _ in range(1000)]
KEEP = []           # every batch stays referenced: no host arrays are freed inside the timed regions
for it in range(10):
    paths = make()
    if "--keep" in sys.argv:
        KEEP.append(paths)
    LOG.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    process_samples.compute_returns(paths, 0.995); torch.cuda.synchronize(); t1 = time.perf_counter()
    process_samples.compute_advantages(paths, bl, 0.995, 0.97); torch.cuda.synchronize(); t2 = time.perf_counter()
    agent.train_from_paths(paths); torch.cuda.synchronize(); t3 = time.perf_counter()
    bl.fit(paths); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(json.dumps({"it": it, "returns": round(1e3 * (t1 - t0), 1), "adv": round(1e3 * (t2 - t1), 1), "update": round(1e3 * (t3 - t2), 1),
                      "fit": round(1e3 * (t4 - t3), 1), "calls": [(k, round(v, 1)) for k, v in LOG]}))
