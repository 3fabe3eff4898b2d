"""Where in the real iteration does a small upload become slow?  Probes (4 MB upload + sync) between the real calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd import _lib
from mjrl_amd.algos.npg_cg import NPG
from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
from mjrl_amd.baselines import _features
from mjrl_amd.policies.gaussian_mlp import MLP
from mjrl_amd.utils import process_samples, ingest
dev = torch.device("cuda", 0)
h = ingest.DeviceHandle(torch, dev, _lib.load())
small = np.arange(1_000_000, dtype=np.int32)
def ms(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return round(1e3 * (time.perf_counter() - t0), 2)
def probe():
    return (ms(lambda: ingest.upload(h, small)), ms(lambda: ingest.upload(h, small)), ms(lambda: torch.empty(1000, device=dev).fill_(1.0)))
if "--mallopt" in sys.argv:
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    M_TRIM_THRESHOLD, M_MMAP_THRESHOLD = -1, -3
    print("mallopt", libc.mallopt(M_MMAP_THRESHOLD, 1 << 30), libc.mallopt(M_TRIM_THRESHOLD, 1 << 31 - 1))
HOLD = []
if "--hold" in sys.argv:                       # keep _time_index's temporaries alive: no munmap inside the probe window
    _orig = _features._time_index
    def _ti(paths):
        lens = np.fromiter((len(p["rewards"]) for p in paths), dtype=np.int64, count=len(paths))
        starts = np.zeros(len(paths), np.int64); np.cumsum(lens[:-1], out=starts[1:])
        a = np.arange(int(lens.sum()), dtype=np.int64); b = np.repeat(starts, lens); c = a - b; d = c.astype(np.int32)
        HOLD.extend([a, b, c, d])
        return d
    _features._time_index = _ti
spec = type("Spec", (), dict(observation_dim=17, action_dim=6, horizon=1000))
rng = np.random.RandomState(0)
pol = MLP(spec, hidden_sizes=(64, 64), seed=1, init_log_std=-0.5)
bl = QuadraticBaseline(spec)
agent = NPG(None, pol, bl, normalized_step_size=0.05)
def make():
    return [dict(observations=rng.randn(1000, 17), actions=rng.randn(1000, 6), rewards=rng.randn(1000), terminated=False) for _ in range(1000)]
for it in range(6):
    paths = make()
    out = {"it": it}
    out["start"] = probe()
    out["returns_ms"] = ms(lambda: process_samples.compute_returns(paths, 0.995))
    out["after_returns"] = probe()
    out["stage_obs_ms"] = ms(lambda: ingest.stage_shared(h, paths, ("observations",)))
    out["after_stage_obs"] = probe()
    mode = [a for a in sys.argv if a.startswith("--mid=")]
    mode = mode[0][6:] if mode else "time_index"
    if mode == "time_index":
        t = ms(lambda: _features._time_index(paths))
    elif mode == "sleep":
        t = ms(lambda: time.sleep(0.002))
    elif mode == "numpy":
        t = ms(lambda: (np.arange(1_000_000, dtype=np.int64) - np.repeat(np.arange(1000) * 1000, 1000)).astype(np.int32))
    elif mode == "fromiter":
        t = ms(lambda: np.fromiter((len(p["rewards"]) for p in paths), dtype=np.int64, count=len(paths)))
    elif mode == "none":
        t = 0
    if "--fillfirst" in sys.argv:
        out["after_mid_fill_first(%s, %s ms)" % (mode, t)] = (ms(lambda: torch.empty(1000, device=dev).fill_(1.0)),) + probe()
    else:
        out["after_mid(%s, %s ms)" % (mode, t)] = probe()
    out["adv_ms"] = ms(lambda: process_samples.compute_advantages(paths, bl, 0.995, 0.97))
    out["after_adv"] = probe()
    out["update_ms"] = ms(lambda: agent.train_from_paths(paths))
    out["after_update"] = probe()
    out["fit_ms"] = ms(lambda: bl.fit(paths))
    out["after_fit"] = probe()
    print(out)
