"""Worker of tests/test_gpu_multirank.py: `train_step` (sampling -> returns -> GAE -> NPG update -> baseline fit) on one rank
or as one of two ranks sharing the GPU (gloo process group for set-up traffic; the update's rank sums run inside libmjx over
its peer exchange, the baseline / statistics sums over utils/ranks.py).  argv: out.npz baseline_kind; RANK / WORLD_SIZE from
torch.distributed.run when there are two."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NTRAJ = 2001


class NoiseEnv:
    """observations are fresh N(0, 1) draws every step, the reward prefers actions near a fixed linear map of them: a
    well-conditioned Fisher matrix (like the synthetic batches of the engine-level tests), so that a one-rank / two-rank
    difference measures the multi-rank arithmetic.  (On a physical toy task -- a point mass: constant targets, slowly varying
    positions -- the 10-iteration CG solve amplifies last-bit differences of the products to per cent of the step, in ANY two
    correct implementations; the reference itself moves by 5e-5 between 8 and 16 BLAS threads at configs[1].)"""
    horizon = 50

    def __init__(self):
        self.rng = np.random.RandomState(0)
        self.W = np.random.RandomState(7).randn(2, 6) * 0.3

    def set_seed(self, s):
        self.rng = np.random.RandomState(s)

    def reset(self):
        self.o = self.rng.randn(6)
        return self.o

    def step(self, a):
        r = -float(np.sum((a - self.W @ self.o) ** 2))
        self.o = self.rng.randn(6)
        return self.o, r, False, {}


def main():
    import torch
    import torch.distributed as dist
    out_path, kind = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from mjrl_amd.algos.npg_cg import NPG
    from mjrl_amd.baselines.mlp_baseline import MLPBaseline
    from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
    from mjrl_amd.policies.gaussian_mlp import MLP
    spec = type("Spec", (), dict(observation_dim=6, action_dim=2, horizon=50))
    pol = MLP(spec, hidden_sizes=(32, 32), seed=2, init_log_std=-0.5)
    bl = QuadraticBaseline(spec) if kind == "quadratic" else MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3)
    agent = NPG(NoiseEnv(), pol, bl, normalized_step_size=0.05, seed=2, save_logs=True,
                FIM_invert_args={'iters': 5, 'damping': 1e-4})         # BASELINE configs[0]: NPG with 5 CG iterations
    prng = np.random.RandomState(4)
    probe = dict(observations=prng.randn(25, 6), rewards=np.zeros(25))
    res = {"theta0": pol.get_param_values()}
    keys = ("alpha", "kl_dist", "surr_improvement", "running_score", "num_samples", "VF_error_before", "VF_error_after",
            "stoc_pol_mean", "stoc_pol_std", "stoc_pol_max", "stoc_pol_min")
    for it in range(2):
        # 2 001 trajectories (shares of 1 000 / 1 001) x 50 steps = 100 050 samples for d = 1 348 parameters: a well-conditioned Fisher, so that
        # the comparison measures the multi-rank arithmetic and not CG's amplification of summation-order noise
        stats = agent.train_step(N=NTRAJ, sample_mode='trajectories', gamma=0.95, gae_lambda=0.97, num_cpu=1)
        lg = agent.logger.get_current_log()
        res["theta%d" % (it + 1)] = pol.get_param_values()
        res["grad%d" % (it + 1)] = agent.engine.grad.cpu().numpy()
        res["x%d" % (it + 1)] = agent.engine.x.cpu().numpy()
        res["stats%d" % (it + 1)] = np.array(stats, np.float64)
        res["log%d" % (it + 1)] = np.array([float(lg[k]) for k in keys])
        res["bl%d" % (it + 1)] = np.asarray(bl._coeffs if kind == "quadratic" else bl.params).copy()
        res["pred%d" % (it + 1)] = np.asarray(bl.predict(probe), np.float64)
    res["seed"] = np.array([agent.seed])
    res["comm_kind"] = np.array([str(agent.engine.comm_kind)])
    if world > 1:
        t = torch.from_numpy(np.concatenate([res[k].astype(np.float64).ravel() for k in sorted(res) if k != "comm_kind"]))
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        res["ranks_identical"] = np.array([bool(torch.equal(lo, hi))])
    if rank == 0:
        np.savez(out_path, **res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
