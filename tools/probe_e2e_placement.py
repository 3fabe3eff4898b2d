#!/usr/bin/env python
"""Which property of the HOST memory a rollout batch lives in decides whether NPG.train_from_paths takes 8 or 20 ms
(tools/probe_e2e_outlier.py: the calls alternate when a new batch is allocated while the previous one is alive -- two regions,
one fast, one slow).  Per call: the time, the NUMA node of the batch's pages (get_mempolicy MPOL_F_NODE | MPOL_F_ADDR), and the
transparent-huge-page share and size of the mappings its arrays sit in (/proc/self/smaps)."""
import ctypes
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

libc = ctypes.CDLL(None, use_errno=True)


def node_of(addr):
    mode = ctypes.c_int(-1)
    rc = libc.syscall(239, ctypes.byref(mode), None, ctypes.c_ulong(0), ctypes.c_void_p(addr), ctypes.c_ulong(3))   # get_mempolicy
    return mode.value if rc == 0 else -1


def smaps():
    out, cur = [], None
    for line in open("/proc/self/smaps"):
        if "-" in line.split()[0] and line[0] in "0123456789abcdef":
            lo, hi = (int(x, 16) for x in line.split()[0].split("-"))
            cur = dict(lo=lo, hi=hi, huge=0, rss=0)
            out.append(cur)
        elif line.startswith("AnonHugePages:"):
            cur["huge"] = int(line.split()[1])
        elif line.startswith("Rss:"):
            cur["rss"] = int(line.split()[1])
    return out


def where(paths):
    maps = smaps()
    nodes, huge, size, nmaps = {}, 0, 0, set()
    for p in paths[::50]:
        a = p["observations"].ctypes.data
        n = node_of(a)
        nodes[n] = nodes.get(n, 0) + 1
        for m in maps:
            if m["lo"] <= a < m["hi"]:
                nmaps.add(m["lo"]); huge += m["huge"]; size += (m["hi"] - m["lo"]) // 1024
                break
    return dict(nodes=nodes, mappings=len(nmaps), mapping_kb_avg=size // max(1, len(paths[::50])), huge_kb_avg=huge // max(1, len(paths[::50])))


spec = type("Spec", (), dict(observation_dim=bench.N_OBS, action_dim=bench.N_ACT, horizon=bench.T))
rng = np.random.RandomState(0)
base = bench._host_paths(rng, advantages=True)
pol = MLP(spec, hidden_sizes=bench.HIDDEN, seed=1, init_log_std=-0.5)
agent = NPG(None, pol, None, normalized_step_size=bench.STEP, FIM_invert_args={'iters': bench.CG_ITERS, 'damping': bench.DAMPING})
os.environ["MJX_KEEP_BATCH"] = "1"
print("cpus allowed:", len(os.sched_getaffinity(0)), "| THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
b = None
for it in range(12):
    b = [dict(observations=p["observations"].copy(), actions=p["actions"].copy(), rewards=p["rewards"], advantages=p["advantages"], terminated=False) for p in base]
    w = where(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agent.train_from_paths(b)
    torch.cuda.synchronize()
    print("%6.2f ms  cpu %3d  %s" % (1e3 * (time.perf_counter() - t0), libc.sched_getcpu(), w), flush=True)
