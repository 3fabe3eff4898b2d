"""Worker of test_peer_timeout_surfaces_as_an_error: two ranks share the GPU over libmjx's peer exchange; after one good update
rank 1 stops taking part.  Rank 0's next update must come back as an MjxError naming the timeout (the consumer kernels give up
after MJX_PEER_TIMEOUT_MS and poison their result with NaN, csrc/vecops.h peer_arrived) -- not as NaN parameters."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from oracle import synth
    from mjrl_amd import _lib
    from mjrl_amd.engine import UpdateEngine
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, m, hid, N = 17, 6, (64, 64), 20000
    rng = np.random.RandomState(5)
    obs, act, adv = rng.randn(N, n).astype(np.float32), rng.randn(N, m).astype(np.float32), rng.randn(N).astype(np.float32)
    lo, hi = (0, N // 2) if rank == 0 else (N // 2, N)
    th = synth.perturbed_params(synth.init_params(n, m, hid))
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    eng = UpdateEngine(n, m, hid)
    eng.set_policy(th, th, ident, ident)
    eng.set_batch(obs[lo:hi], act[lo:hi], adv[lo:hi])
    good = eng.npg_update(10, 1e-4, 0.05, -3.0)                  # both ranks: the transport works
    res = dict(good=np.array(good), comm_kind=np.array([str(eng.comm_kind)]))
    dist.barrier()
    if rank == 0:
        eng.set_policy(th, th, ident, ident)
        t0 = time.time()
        try:
            eng.npg_update(10, 1e-4, 0.05, -3.0)                 # rank 1 never sends: every wait of the loop times out
            res["raised"] = np.array(["nothing"])
        except _lib.MjxError as e:
            res["raised"] = np.array([str(e)])
        res["seconds"] = np.array([time.time() - t0])
        res["comm_kind_after"] = np.array([str(eng.comm_kind)])
        np.savez(out_path, **res)
    else:
        time.sleep(1.0)                                          # (stays alive -- its buffer stays mapped -- but sends nothing)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
