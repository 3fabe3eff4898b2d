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


def _chain_worker(rank, world, port, outdir):
    """transport chain of engine._native_comm on two gloo ranks with a stand-in backend whose in-library RCCL cannot produce a
    communicator id on rank 0 (what MJX_RCCL_DISABLE=1 / a missing librccl does)"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      MJX_TRANSPORT_ORDER="rccl,hook")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mjrl_amd.engine import UpdateEngine
    from tests._cpu_backend import OracleBackend

    class Stub(OracleBackend):
        hooked = False

        def comm_unique_id(self):
            raise RuntimeError("librccl.so could not be loaded (stand-in)")

        def comm_init(self, rank, world, uid):
            raise AssertionError("no rank may get here: there is no communicator id")

        def comm_set_callback(self, d, world):
            self.hooked = True

        def allreduce(self, t):
            dist.all_reduce(t)
    eng = UpdateEngine(5, 2, (16, 16), backend=Stub(5, 2, (16, 16)))
    import warnings
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        ok = eng._native_comm()
    np.savez(os.path.join(outdir, "chain_r%d.npz" % rank), ok=np.array([ok]), kind=np.array([str(eng.comm_kind)]),
             hooked=np.array([eng.backend.hooked]), warned=np.array([sum("rccl" in str(w.message) for w in caught)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_rccl_id_failure_on_rank0_does_not_strand_the_other_ranks(tmp_path):
    """rank 0 failing to create the RCCL communicator id used to raise BEFORE the broadcast the other ranks were
    blocked in, then went on to the agreement all-reduce -- mismatched collectives, a hang.  Now the failure travels through the
    broadcast as None, every rank drops the attempt, and the chain continues with the hook (known-answer sum included)."""
    import torch.multiprocessing as mp
    out = str(tmp_path)
    mp.start_processes(_chain_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    for r in (0, 1):
        g = np.load(os.path.join(out, "chain_r%d.npz" % r))
        assert bool(g["ok"][0]) and str(g["kind"][0]) == "hook" and bool(g["hooked"][0]), dict(g)
        assert int(g["warned"][0]) == 1
