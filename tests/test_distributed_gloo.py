"""N > 1 path on CPU: two gloo ranks, each with its own trajectory shard, must produce the update a
single process produces on the whole batch (sharding, all-reduce placement, global-N means,
cross-rank advantage whitening and return statistics, lock-step CG)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, algo, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mjrl_amd.engine import UpdateEngine
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.algos.trpo import TRPO
    from mjrl_amd.algos.dapg import DAPG
    from mjrl_amd.policies.gaussian_mlp import MLP
    from tests._cpu_backend import OracleBackend
    from tests.test_distributed_gloo import make_problem
    spec, theta0, paths, demos = make_problem()
    pol = MLP(spec, hidden_sizes=(16, 16), seed=1, init_log_std=-0.5)
    pol.set_param_values(theta0)
    kw = dict(FIM_invert_args={'iters': 6, 'damping': 1e-4})
    if algo == "npg":
        agent = NPG(None, pol, None, normalized_step_size=0.05, **kw)
    elif algo == "trpo":
        agent = TRPO(None, pol, None, kl_dist=0.01, **kw)
    else:
        agent = DAPG(None, pol, None, demo_paths=demos, kl_dist=0.02, lam_0=1e-2, lam_1=0.95, **kw)
    agent._engine_obj = UpdateEngine(spec.observation_dim, spec.action_dim, (16, 16), backend=OracleBackend(5, 2, (16, 16)))
    mine = paths[rank::world] if world > 1 else paths          # this rank's trajectories
    stats = agent.train_from_paths(mine)
    np.savez(os.path.join(outdir, "%s_w%d_r%d.npz" % (algo, world, rank)), theta=pol.get_param_values(),
             stats=np.array(stats), alpha=agent.last_update["alpha"], kl=agent.last_update["kl_dist"],
             running=agent.running_score)
    dist.barrier()
    dist.destroy_process_group()


def make_problem():
    from oracle import synth
    spec = type("Spec", (), dict(observation_dim=5, action_dim=2, horizon=50))
    theta0 = synth.perturbed_params(synth.init_params(5, 2, (16, 16)))
    paths = synth.make_paths(12, 50, 5, 2, seed=0, ragged=True)
    rng = np.random.RandomState(5)
    for p in paths:
        p["advantages"] = rng.randn(len(p["rewards"])) * 2.0 + 0.3
    demos = synth.make_paths(4, 20, 5, 2, seed=7)
    return spec, theta0, paths, demos


@pytest.mark.parametrize("algo", ["npg", "trpo", "dapg"])
def test_two_ranks_match_one(algo, tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path)
    mp.start_processes(_worker, args=(1, _free_port(), algo, out), nprocs=1, join=True, start_method="spawn")
    mp.start_processes(_worker, args=(2, _free_port(), algo, out), nprocs=2, join=True, start_method="spawn")
    one = np.load(os.path.join(out, "%s_w1_r0.npz" % algo))
    r0 = np.load(os.path.join(out, "%s_w2_r0.npz" % algo))
    r1 = np.load(os.path.join(out, "%s_w2_r1.npz" % algo))
    assert np.array_equal(r0["theta"], r1["theta"])                       # ranks stay in lock-step, bit for bit
    step1, step2 = one["theta"].astype(np.float64), r0["theta"].astype(np.float64)
    from tests.test_distributed_gloo import make_problem as mk
    theta0 = mk()[1].astype(np.float64)
    err = np.linalg.norm((step2 - theta0) - (step1 - theta0)) / np.linalg.norm(step1 - theta0)
    assert err < 1e-5, err                                                # the north-star bar on the step direction
    assert abs(float(r0["alpha"]) - float(one["alpha"])) < 1e-5 * float(one["alpha"])
    assert abs(float(r0["kl"]) - float(one["kl"])) < 1e-4 * abs(float(one["kl"])) + 1e-8
    np.testing.assert_allclose(r0["stats"], one["stats"], rtol=1e-12)      # return statistics over ALL ranks' paths
    np.testing.assert_allclose(r1["stats"], one["stats"], rtol=1e-12)
