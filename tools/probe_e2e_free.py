"""Does releasing the PREVIOUS batch inside train_from_paths cost time?  The staged-batch registry (utils/ingest.py) holds strong
references to the arrays it uploaded; a caller that has dropped its own references by the time it passes the next batch (the
normal training loop: `paths = sampler(...)`) leaves the registry as the last owner, and the 1 000 dicts / 4 000 arrays are freed
where the registry entry is replaced -- inside the call.  Three loops over fresh 1M-timestep batches:
  A  batches created up front and kept alive (tools/bench_e2e.py's loop): nothing is freed inside the call
  B  `b = fresh()` per iteration (bench.py's loop): the previous batch dies inside the call
  C  like B, the caller keeps the previous batch alive until the call has returned: freed outside
  D  like A, the device idle for 100 ms before every call (host asleep);  E  the same with the host busy
"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.algos.npg_cg import NPG
from mjrl_amd.utils import ingest as _ingest; _ingest.tune_malloc()   # a training process (what train_step / dropin.install do)
from mjrl_amd.policies.gaussian_mlp import MLP
spec = type("Spec", (), dict(observation_dim=17, action_dim=6, horizon=1000))
rng = np.random.RandomState(0)
paths = [dict(observations=rng.randn(1000, 17), actions=rng.randn(1000, 6), rewards=rng.randn(1000), advantages=rng.randn(1000),
              terminated=False) for _ in range(1000)]
pol = MLP(spec, hidden_sizes=(64, 64), seed=1, init_log_std=-0.5)
agent = NPG(None, pol, None, normalized_step_size=0.05)
def fresh():
    return [dict(observations=p["observations"].copy(), actions=p["actions"].copy(), rewards=p["rewards"], advantages=p["advantages"],
                 terminated=False) for p in paths]
def timed(b):
    torch.cuda.synchronize(); t0 = time.perf_counter(); agent.train_from_paths(b); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)
out = {}
for rep in range(2):
    batches = [fresh() for _ in range(9)]
    ts = [timed(b) for b in batches]
    out["A_kept_alive_%d" % rep] = sorted(round(t, 2) for t in ts[2:])
    ts = []
    for b in batches:                  # D: like A with the device left idle for 100 ms before every call (what fresh() takes in B / C)
        torch.cuda.synchronize(); time.sleep(0.1)
        ts.append(timed(b))
    out["D_kept_alive_after_100ms_idle_%d" % rep] = sorted(round(t, 2) for t in ts[2:])
    ts = []
    for b in batches:                  # E: D with the host busy instead of asleep (a NumPy copy loop of the same length)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1:
            paths[0]["observations"].copy()
        ts.append(timed(b))
    out["E_kept_alive_after_100ms_host_busy_%d" % rep] = sorted(round(t, 2) for t in ts[2:])
    side = torch.cuda.Stream()
    wa = torch.randn(4096, 4096, device="cuda"); wb = torch.randn(4096, 4096, device="cuda")
    for ms_label, reps_mm in (("1.2ms", 1), ("2.5ms", 2)):
        ts = []
        for b in batches:              # F: D with the compute units kept busy while the rollouts are staged (a matrix product on a side stream)
            torch.cuda.synchronize(); time.sleep(0.1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.cuda.stream(side):
                for _ in range(reps_mm):
                    torch.mm(wa, wb)
            agent.train_from_paths(b); torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        out["F_idle_then_busy_units_%s_%d" % (ms_label, rep)] = sorted(round(t, 2) for t in ts[2:])
    del batches
    ts = []
    b = None
    for it in range(9):
        b = fresh()
        ts.append(timed(b))
    out["B_previous_dies_inside_%d" % rep] = sorted(round(t, 2) for t in ts[2:])
    ts = []
    for it in range(9):
        old = b
        b = fresh()
        ts.append(timed(b))
        del old
    out["C_previous_freed_outside_%d" % rep] = sorted(round(t, 2) for t in ts[2:])
print(json.dumps(out))
