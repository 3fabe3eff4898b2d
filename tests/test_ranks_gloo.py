"""Rank-level host logic outside the update engine on two gloo ranks (CPU): mjrl_amd/utils/ranks.py and the value baselines'
ONE-fit-over-all-ranks rule (SURVEY 8e "Collectives": the (F+1)^2 fp64 Gram sum of K6c).  The per-rank device arithmetic is
replaced by an oracle-backed stand-in (test infrastructure); what runs for real is the product's host code: which quantities
are summed, in which order the collectives are issued, what every rank ends up holding."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _problem():
    from oracle import synth
    paths = synth.make_paths(10, 40, 5, 2, seed=3, ragged=True)
    rng = np.random.RandomState(11)
    for p in paths:
        p["returns"] = np.cumsum(p["rewards"][::-1])[::-1] + 0.1 * rng.randn(len(p["rewards"]))
    return paths


class _OracleBlock:
    """stand-in for mjrl_amd.baselines._features.DeviceBlock: this rank's feature arithmetic from oracle/npg_oracle.py"""

    def __init__(self, paths, inp, shared=True):
        from oracle import npg_oracle as O
        self.paths = paths
        self.A = O.quadratic_baseline_features([p["observations"] for p in paths])

    def returns_dev(self):
        return np.concatenate([p["returns"] for p in self.paths])

    def gram(self, kind, y):
        Ay = np.concatenate([self.A, y[:, None]], axis=1)
        return Ay.T @ Ay

    def predict_linear(self, kind, coef):
        return self.A @ coef


def _worker(rank, world, port, outdir, empty_rank):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mjrl_amd.baselines import quadratic_baseline as qb
    from mjrl_amd.utils import ranks
    from tests.test_ranks_gloo import _OracleBlock, _problem
    paths = _problem()
    if world == 1:
        mine = paths
    elif empty_rank:
        mine = paths if rank == 1 else []
    else:
        mine = paths[:4] if rank == 0 else paths[4:]                    # contiguous, ragged shards
    res = {}
    # ---- helpers
    v = np.arange(6, dtype=np.float64) * (rank + 1)
    res["sum"] = ranks.sum_host(v)
    t = torch.arange(3 * (rank + 2), dtype=torch.float32).reshape(rank + 2, 3) + 100 * rank
    res["gather"] = ranks.gather_rows(t).numpy()
    res["counts"] = np.array(ranks.counts(rank + 2))
    res["bcast_last"] = ranks.broadcast_host(np.full(4, rank, np.int32), src=-1)
    x = np.concatenate([p["returns"] for p in mine]) if mine else np.zeros(0)
    res["mean_std"] = np.array(ranks.mean_std(x))
    res["all_true"] = np.array([ranks.all_true(True), ranks.all_true(rank == 0)])
    # ---- ONE ridge fit over all ranks' paths (quadratic_baseline.py:44-69)
    qb.DeviceBlock = _OracleBlock
    qb.num_features = lambda kind, n: n + n * (n + 1) // 2 + 5
    spec = type("Spec", (), dict(observation_dim=5, action_dim=2))
    bl = qb.QuadraticBaseline(spec)
    e0 = bl.fit(mine, return_errors=True)
    e1 = bl.fit(mine, return_errors=True)                                # second fit: error_before uses the fitted coefficients
    res["coeffs"], res["errors"] = bl._coeffs, np.array(list(e0) + list(e1))
    np.savez(os.path.join(outdir, "w%d_r%d.npz" % (world, rank)), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("empty_rank", [False, True])
def test_rank_helpers_and_one_ridge_fit_over_two_ranks(tmp_path, empty_rank):
    import torch.multiprocessing as mp
    out = str(tmp_path)
    mp.start_processes(_worker, args=(1, _free_port(), out, empty_rank), nprocs=1, join=True, start_method="spawn")
    mp.start_processes(_worker, args=(2, _free_port(), out, empty_rank), nprocs=2, join=True, start_method="spawn")
    one = np.load(os.path.join(out, "w1_r0.npz"))
    r0, r1 = np.load(os.path.join(out, "w2_r0.npz")), np.load(os.path.join(out, "w2_r1.npz"))
    np.testing.assert_array_equal(r0["sum"], np.arange(6) * 3.0)
    np.testing.assert_array_equal(r0["counts"], [2, 3])
    exp = np.concatenate([np.arange(6, dtype=np.float32).reshape(2, 3), np.arange(9, dtype=np.float32).reshape(3, 3) + 100])
    np.testing.assert_array_equal(r0["gather"], exp)
    np.testing.assert_array_equal(r1["gather"], exp)
    np.testing.assert_array_equal(r0["bcast_last"], np.ones(4, np.int32))
    np.testing.assert_array_equal(r0["all_true"], [True, False])
    np.testing.assert_array_equal(r1["all_true"], [True, False])
    np.testing.assert_allclose(r0["mean_std"], one["mean_std"], rtol=1e-13)
    # the fit: identical bits on both ranks, the one-rank fit to 1e-9, the same logged errors
    np.testing.assert_array_equal(r0["coeffs"], r1["coeffs"])
    np.testing.assert_array_equal(r0["errors"], r1["errors"])
    assert np.linalg.norm(r0["coeffs"] - one["coeffs"]) <= 1e-9 * np.linalg.norm(one["coeffs"])
    np.testing.assert_allclose(r0["errors"], one["errors"], rtol=1e-9)
    assert one["errors"][1] < one["errors"][0] and abs(one["errors"][2] - one["errors"][1]) < 1e-12
