#!/usr/bin/env python
"""Where the one-in-five 15 ms end-to-end NPG.train_from_paths call comes from (bench.py `secondary.end_to_end`: 8.5-8.8 ms with
one outlier): per-call times in order, with the garbage collector's generation-2 passes and their durations logged next to them,
then the same loop with the collector frozen / disabled around the call."""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from mjrl_amd.algos.npg_cg import NPG
from mjrl_amd.utils import ingest as _ingest; _ingest.tune_malloc()   # a training process (what train_step / dropin.install do)
from mjrl_amd.policies.gaussian_mlp import MLP

spec = type("Spec", (), dict(observation_dim=bench.N_OBS, action_dim=bench.N_ACT, horizon=bench.T))
rng = np.random.RandomState(0)
base = bench._host_paths(rng, advantages=True)
pol = MLP(spec, hidden_sizes=bench.HIDDEN, seed=1, init_log_std=-0.5)
agent = NPG(None, pol, None, normalized_step_size=bench.STEP, FIM_invert_args={'iters': bench.CG_ITERS, 'damping': bench.DAMPING})


def fresh():
    return [dict(observations=p["observations"].copy(), actions=p["actions"].copy(), rewards=p["rewards"], advantages=p["advantages"],
                 terminated=False) for p in base]


events = []
t_gc = [0.0]


def cb(phase, info):
    if phase == "start":
        t_gc[0] = time.perf_counter()
    else:
        events.append((info["generation"], 1e3 * (time.perf_counter() - t_gc[0])))


gc.callbacks.append(cb)
for mode in ("default", "gc.disable around the call", "gc.freeze after warm-up"):
    if mode == "gc.freeze after warm-up":
        gc.collect(); gc.freeze()
    rows = []
    for it in range(14):
        b = fresh()
        torch.cuda.synchronize()
        events.clear()
        if mode.startswith("gc.disable"):
            gc.disable()
        t0 = time.perf_counter()
        agent.train_from_paths(b)
        torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0)
        if mode.startswith("gc.disable"):
            gc.enable()
        rows.append((round(dt, 2), [(g, round(ms, 2)) for g, ms in events if g == 2 or ms > 0.3]))
    print(mode, flush=True)
    for r in rows[2:]:
        print("   ", r, flush=True)
    ts = sorted(r[0] for r in rows[2:])
    print("    median %.2f  min %.2f  max %.2f  max/min %.2f" % (ts[len(ts) // 2], ts[0], ts[-1], ts[-1] / ts[0]), flush=True)

# ---- hypothesis: the slow calls pay for RELEASING the previous batch (the staging registry holds the last references to its
# 2 000 host arrays until the next batch replaces it -- inside the next call).  Release it outside the timed region instead.
from mjrl_amd.utils import ingest
gc.unfreeze()
rows = []
b = None
for it in range(14):
    t0 = time.perf_counter()
    ingest.drop_shared_batch()
    b = None
    rel_ms = 1e3 * (time.perf_counter() - t0)
    b = fresh()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agent.train_from_paths(b)
    torch.cuda.synchronize()
    rows.append((round(1e3 * (time.perf_counter() - t0), 2), round(rel_ms, 2)))
print("previous batch released before the call (call ms, release ms)")
for r in rows[2:]:
    print("   ", r)
ts = sorted(r[0] for r in rows[2:])
print("    median %.2f  min %.2f  max %.2f  max/min %.2f" % (ts[len(ts) // 2], ts[0], ts[-1], ts[-1] / ts[0]), flush=True)


# ---- the registry KEEPS the batch after the call (MJX_KEEP_BATCH=1: a baseline.fit(paths) that follows re-uses the upload) and lets
# go of it at the START of the next batch's staging, before any gather thread runs -- the caller's loop as in the first experiment
os.environ["MJX_KEEP_BATCH"] = "1"
rows = []
for it in range(14):
    b = fresh()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agent.train_from_paths(b)
    torch.cuda.synchronize()
    rows.append(round(1e3 * (time.perf_counter() - t0), 2))
print("registry keeps the batch, released at the start of the next staging:", rows[2:])
ts = sorted(rows[2:])
print("    median %.2f  min %.2f  max %.2f  max/min %.2f" % (ts[len(ts) // 2], ts[0], ts[-1], ts[-1] / ts[0]), flush=True)
