"""Worker of test_gpu_multirank.py::test_a_corrupted_peer_transport_is_rejected_before_the_first_update: W ranks share the GPU
(gloo for set-up traffic), libmjx's peer exchange is the first transport in the chain.  With MJX_PEER_FAULT=slot|flag one rank's
producer misbehaves (vectors land in the wrong slot at the peers / the arrival flags are never raised); the known-answer sum of
engine._transport_self_test must reject the transport on EVERY rank, the chain must fall through to the hook, and the update that
follows must be the clean run's."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from oracle import synth
    from mjrl_amd.engine import UpdateEngine
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, m, hid, N = 17, 6, (64, 64), 30000
    rng = np.random.RandomState(5)
    obs, act, adv = rng.randn(N, n).astype(np.float32), rng.randn(N, m).astype(np.float32), rng.randn(N).astype(np.float32)
    lo, hi = rank * N // world, (rank + 1) * N // world
    th = synth.perturbed_params(synth.init_params(n, m, hid))
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    eng = UpdateEngine(n, m, hid)
    eng.set_policy(th, th, ident, ident)
    eng.set_batch(obs[lo:hi], act[lo:hi], adv[lo:hi])
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        res = eng.npg_update(10, 1e-4, 0.05, -3.0)
    theta = eng.theta_new.cpu().numpy()
    every = [None] * world
    dist.all_gather_object(every, (theta.tobytes(), str(eng.comm_kind)))
    if rank == 0:
        np.savez(out_path, theta=theta, res=np.array(res), comm_kind=np.array([str(eng.comm_kind)]),
                 ranks_identical=np.array([all(e == every[0] for e in every)]),
                 warned=np.array([any("known-answer" in str(w.message) for w in caught)]))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
