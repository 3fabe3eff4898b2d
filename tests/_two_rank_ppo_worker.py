"""Worker of tests/test_gpu_multirank.py::test_two_rank_ppo_equals_one_rank: PPO.train_from_paths (mjrl/algos/ppo_clip.py:59-110) on
one rank, or as one of two ranks that share the GPU and each hold a shard of the trajectories -- every rank gathers all rows and
runs the identical minibatch-Adam chain from the last rank's index draws (mjrl_amd/algos/ppo_clip.py).  argv: out.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from oracle import synth
    out_path = sys.argv[1]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from mjrl_amd.algos.ppo_clip import PPO
    from mjrl_amd.policies.gaussian_mlp import MLP
    n, m = 11, 3
    spec = type("Spec", (), dict(observation_dim=n, action_dim=m, horizon=60))
    pol = MLP(spec, hidden_sizes=(32, 32), seed=3, init_log_std=-0.5)
    paths = synth.make_paths(40, 60, n, m, seed=2, ragged=True)
    rng = np.random.RandomState(9)
    for p in paths:
        p["advantages"] = rng.randn(len(p["rewards"]))
    mine = paths if world == 1 else (paths[:17] if rank == 0 else paths[17:])          # contiguous shards: rank order = the one-process batch
    agent = PPO(None, pol, None, clip_coef=0.2, epochs=2, mb_size=64, learn_rate=3e-4, save_logs=False)
    thetas = []
    for it in range(2):                                                                  # the second iteration runs with Adam state and the aliasing quirk
        np.random.seed(100 + it)
        stats = agent.train_from_paths(mine)
        thetas.append(pol.get_param_values().copy())
    res = dict(theta1=thetas[0], theta2=thetas[1], stats=np.array(stats), kl=np.array([agent.last_update["kl_dist"]]),
               surr=np.array([agent.last_update["surr_after"] - agent.last_update["surr_before"]]))
    if world > 1:
        every = [None] * world
        dist.all_gather_object(every, thetas[1].tobytes())
        res["ranks_identical"] = np.array([all(e == every[0] for e in every)])
    if rank == 0:
        np.savez(out_path, **res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    agent.engine.close()


if __name__ == "__main__":
    main()
