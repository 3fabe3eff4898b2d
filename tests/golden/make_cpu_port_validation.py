#!/usr/bin/env python
"""Is bench.py's cpu_baseline (oracle/torch_port.py, kind "port") a fair stand-in for the reference's wall time?
Times the port's npg_update and the UNMODIFIED reference's NPG.train_from_paths (imported from /root/reference) on the
same 100 000-timestep cfg2 batch (obs 17, act 6, 64x64, 10 CG iterations), interleaved, best of 3, and stores the ratio
(tests/golden/cpu_port_vs_reference.json; bench.py quotes it in cpu_baseline.kind).  Build container only."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import _ref_import  # noqa: E402

_ref_import.install()
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mjrl.algos.npg_cg import NPG  # noqa: E402
from mjrl.policies.gaussian_mlp import MLP  # noqa: E402
from mjrl.utils.gym_env import EnvSpec  # noqa: E402

import bench  # noqa: E402
from oracle import torch_port  # noqa: E402


def measure(n_traj=100, reps=3, threads=None):
    if threads:
        torch.set_num_threads(threads)
    theta0 = bench.initial_params()
    obs, act, adv = bench.synth_shard(0, bench.N_TRAJ // n_traj)
    obs, act = obs.astype(np.float64), act.astype(np.float64)
    paths = [dict(observations=obs[i * bench.T:(i + 1) * bench.T], actions=act[i * bench.T:(i + 1) * bench.T],
                  advantages=adv[i * bench.T:(i + 1) * bench.T], rewards=np.zeros(bench.T)) for i in range(n_traj)]
    adv_w = (adv - adv.mean()) / (adv.std() + 1e-6)
    spec = EnvSpec(bench.N_OBS, bench.N_ACT, bench.T)
    tp, tr = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = torch_port.npg_update(theta0, obs, act, adv_w, bench.N_OBS, bench.N_ACT, bench.HIDDEN, cg_iters=bench.CG_ITERS,
                                  damping=bench.DAMPING, delta=bench.STEP)
        tp.append(time.perf_counter() - t0)
        pol = MLP(spec, hidden_sizes=bench.HIDDEN, seed=1, init_log_std=-0.5)
        pol.set_param_values(theta0.copy())
        agent = NPG(None, pol, None, normalized_step_size=bench.STEP, FIM_invert_args={'iters': bench.CG_ITERS, 'damping': bench.DAMPING})
        t0 = time.perf_counter()
        agent.train_from_paths(paths)
        tr.append(time.perf_counter() - t0)
        same = float(np.linalg.norm(pol.get_param_values() - r["new_params"]) / np.linalg.norm(r["new_params"] - theta0))
    return dict(timesteps=int(obs.shape[0]), port_seconds=min(tp), reference_seconds=min(tr), port_over_reference=min(tp) / min(tr),
                port_seconds_all=tp, reference_seconds_all=tr, step_rel_difference=same, threads=torch.get_num_threads(),
                nproc=os.cpu_count())


if __name__ == "__main__":
    out = measure(threads=8)
    json.dump(out, open(os.path.join(HERE, "cpu_port_vs_reference.json"), "w"), indent=1)
    print(json.dumps(out))
