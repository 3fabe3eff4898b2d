"""Timeline of one NPG.train_from_paths on fresh fp64 host paths (1M timesteps): when each stage of the main thread and of the
staging helper starts and ends, relative to the call (ms).  python tools/e2e_timeline.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.algos.npg_cg import NPG
from mjrl_amd.utils import ingest as _ingest; _ingest.tune_malloc()   # a training process (what train_step / dropin.install do)
from mjrl_amd.algos import batch_reinforce as br
from mjrl_amd.policies.gaussian_mlp import MLP
from mjrl_amd import engine as eng_mod
from mjrl_amd.utils import ingest
spec = type("Spec", (), dict(observation_dim=17, action_dim=6, horizon=1000))
rng = np.random.RandomState(0)
paths = [dict(observations=rng.randn(1000, 17), actions=rng.randn(1000, 6), rewards=rng.randn(1000), advantages=rng.randn(1000), terminated=False) for _ in range(1000)]
pol = MLP(spec, hidden_sizes=(64, 64), seed=1, init_log_std=-0.5)
agent = NPG(None, pol, None, normalized_step_size=0.05)
def fresh():
    return [dict(observations=p["observations"].copy(), actions=p["actions"].copy(), rewards=p["rewards"], advantages=p["advantages"], terminated=False) for p in paths]
T0 = [0.0]; LOG = []
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); LOG.append((tag, 1e3 * (t - T0[0]), 1e3 * (time.perf_counter() - T0[0]))); return r
    setattr(obj, name, g)
wrap(eng_mod.UpdateEngine, "stage_paths", "helper: stage obs+act")
wrap(eng_mod.UpdateEngine, "whitened_advantages", "main: whiten advantages (2 syncs)")
wrap(eng_mod.UpdateEngine, "set_batch", "main: set_batch")
wrap(eng_mod.UpdateEngine, "set_policy", "main: set_policy")
wrap(eng_mod.UpdateEngine, "npg_update", "main: npg_update (enqueue + read-back)")
wrap(eng_mod.UpdateEngine, "to_host", "main: theta to host")
wrap(br.BatchREINFORCE, "_path_statistics", "main: path statistics")
wrap(br.BatchREINFORCE, "_process_and_bind", "main: _process_and_bind (all)")
wrap(pol, "set_param_values", "main: policy.set_param_values")
_ss = ingest.stage_shared
def ss(backend, paths_, keys, raw=None, defer=False):
    t = time.perf_counter(); r = _ss(backend, paths_, keys, raw, defer); LOG.append(("stage_shared%s" % (tuple(keys),), 1e3 * (t - T0[0]), 1e3 * (time.perf_counter() - T0[0]))); return r
ingest.stage_shared = ss
_rob = ingest._release_other_batches
def rob(dev, paths_, keys):
    t = time.perf_counter(); _rob(dev, paths_, keys); LOG.append(("  release of the previous batch's references", 1e3 * (t - T0[0]), 1e3 * (time.perf_counter() - T0[0])))
ingest._release_other_batches = rob
_settle = ingest.settle
def settle(backend):
    t = time.perf_counter(); _settle(backend); LOG.append(("main: settle (join staging jobs)", 1e3 * (t - T0[0]), 1e3 * (time.perf_counter() - T0[0])))
ingest.settle = settle
if os.environ.get("TL_ALTERNATE") == "1":
    # the caller's loop of tools/probe_e2e_outlier.py: a new batch allocated while the previous one is alive, the registry keeping
    # the batch (MJX_KEEP_BATCH=1) -- calls alternate between ~8 and ~20 ms; print the timelines of the last two
    os.environ["MJX_KEEP_BATCH"] = "1"
    runs, b = [], None
    for it in range(8):
        b = fresh()
        LOG.clear(); torch.cuda.synchronize(); T0[0] = time.perf_counter()
        agent.train_from_paths(b)
        torch.cuda.synchronize(); runs.append((1e3 * (time.perf_counter() - T0[0]), list(LOG)))
    for total, log in runs[-2:]:
        for tag, a, c in sorted(log, key=lambda x: x[1]):
            print("%-44s %7.2f -> %7.2f  (%.2f ms)" % (tag, a, c, c - a))
        print("total %.2f ms\n" % total)
    sys.exit(0)
if os.environ.get("TL_FINE") == "1":        # finer stamps inside the stager (r06: where do calls 2-3 of a process lose 7 ms?)
    for nm in ("begin", "_stage_async", "join", "_slot"):
        wrap(ingest.PathStager, nm, "    stager." + nm)
    _oa = ingest._order_after
    def oa(backend, ent):
        t = time.perf_counter(); _oa(backend, ent); LOG.append(("    _order_after", 1e3 * (t - T0[0]), 1e3 * (time.perf_counter() - T0[0])))
    ingest._order_after = oa
    _sb = ingest._same_batch
    def sb(ent, paths_, key):
        t = time.perf_counter(); r = _sb(ent, paths_, key); LOG.append(("    _same_batch(%s)" % key, 1e3 * (t - T0[0]), 1e3 * (time.perf_counter() - T0[0]))); return r
    ingest._same_batch = sb
if os.environ.get("TL_FIRST") == "1":
    # r06: the FIRST calls of a process, one fresh batch at a time like bench.py's end_to_end loop (calls 2-3 ran 24 ms against 8.5 later)
    if os.environ.get("TL_PREGROW") == "1":      # the host heap already holds three batches' worth of freed chunks when call 1 starts
        tmp = [fresh() for _ in range(3)]
        del tmp
    for it in range(6):
        b = fresh()
        LOG.clear(); torch.cuda.synchronize(); T0[0] = time.perf_counter()
        agent.train_from_paths(b)
        torch.cuda.synchronize(); total = 1e3 * (time.perf_counter() - T0[0])
        print("---- call %d: total %.2f ms" % (it + 1, total))
        for tag, a, c in sorted(LOG, key=lambda x: x[1]):
            print("%-44s %7.2f -> %7.2f  (%.2f ms)" % (tag, a, c, c - a))
    sys.exit(0)
batches = [fresh() for _ in range(6)]
for b in batches:
    LOG.clear(); torch.cuda.synchronize(); T0[0] = time.perf_counter()
    agent.train_from_paths(b)
    torch.cuda.synchronize(); total = 1e3 * (time.perf_counter() - T0[0])
for tag, a, b in sorted(LOG, key=lambda x: x[1]):
    print("%-44s %7.2f -> %7.2f  (%.2f ms)" % (tag, a, b, b - a))
print("total %.2f ms" % total)
