"""Worker of test_two_ranks_on_one_gpu_equal_one_rank: one of two ranks (torch.distributed.run; the process group is gloo
-- RCCL refuses two ranks on one device -- and only carries set-up traffic) that share the GPU.  Each rank binds its
trajectory shard; the rank sums run inside libmjx over its peer exchange (HIP IPC buffers + stream-ordered waits; with
MJX_PEER_COMM=0 over the transport hook); rank 0 writes the update's results."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from oracle import synth
    from mjrl_amd.engine import UpdateEngine
    out_path = sys.argv[1]
    # (no RANK in the environment: the ONE-rank reference of the same batch, in a process of its own -- the eight-rank test keeps the
    #  pytest process off the GPU, see tests/test_a_eight_ranks_gpu.py; MJX_TEST_REF_CUTS = the shard bounds of the multi-rank run)
    solo = "RANK" not in os.environ
    rank, world = (0, 1) if solo else (int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]))
    torch.cuda.set_device(0)
    if not solo:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n, m, hid, N = 17, 6, (64, 64), 60000
    if os.environ.get("MJX_TEST_SHAPE"):                 # "n,m,h1,h2": another fused instance (e.g. one whose d is not a multiple of 4)
        n, m, h1, h2 = (int(v) for v in os.environ["MJX_TEST_SHAPE"].split(","))
        hid = (h1, h2)
    rng = np.random.RandomState(5)                       # identical on all ranks
    obs, act, adv = rng.randn(N, n).astype(np.float32), rng.randn(N, m).astype(np.float32), rng.randn(N).astype(np.float32)
    cut = int(os.environ.get("MJX_TEST_CUT", "23456"))   # ragged shards (0: rank 0 holds NO trajectories)
    cuts = [0, N] if solo else [0, cut, N] if world == 2 else [0] + [int(c) for c in os.environ["MJX_TEST_CUTS"].split(",")] + [N]   # world > 2: explicit cuts
    lo, hi = cuts[rank], cuts[rank + 1]
    th = synth.perturbed_params(synth.init_params(n, m, hid))
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    eng = UpdateEngine(n, m, hid)
    eng.set_policy(th, th, ident, ident)
    eng.set_batch(obs[lo:hi], act[lo:hi], adv[lo:hi])    # N_global through the process group
    assert eng.N_global == N
    g, _ = eng.surr_vpg(sync=False)
    eng.cg_solve(g, 10, 1e-4, sync=False)
    eng.apply_npg_step(0.05, -3.0)
    surr_after, kl = eng.eval_surr_kl()
    late = eng.deferred()
    res = dict(grad=g.cpu().numpy(), x=eng.x.cpu().numpy(), theta=eng.theta_new.cpu().numpy(),
               scal=np.array([late["surr_before"], late["gdotx"], late["alpha"], surr_after, kl]))
    # the rank sums ran inside libmjx's C loops (peer exchange / transport hook here; RCCL on a real multi-GPU node) ...
    res["native_comm"] = np.array([True, True] if solo else [bool(eng._native_comm()), eng.backend.comm_world() == world])
    res["comm_kind"] = np.array([eng.comm_kind])
    # ... and the whole update as ONE call (mjx_npg_update) gives the same bits as the call-by-call sequence
    eng.set_policy(th, th, ident, ident)
    sa2, kl2 = eng.npg_update(10, 1e-4, 0.05, -3.0)
    late2 = eng.deferred()
    res["one_call_equal"] = np.array([np.array_equal(eng.x.cpu().numpy(), res["x"]) and np.array_equal(eng.theta_new.cpu().numpy(), res["theta"])
                                      and sa2 == surr_after and kl2 == kl and late2 == late])
    # the TRPO update with its line search decided on the device (mjx_trpo_update): the rank sums of K1, the solve and every
    # trial's K3 run inside the C loop; a KL bound that forces several shrinks
    eng.set_policy(th, th, ident, ident)
    tr = eng.trpo_update(10, 1e-4, 0.02, 0.002, -3.0)
    res["trpo"] = np.array([tr["alpha"], tr["trials"], tr["kl"], tr["surr_after"], float(tr["accepted"])])
    res["trpo_theta"] = eng.theta_new.cpu().numpy()
    # DAPG as one call (mjx_dapg_update): K1 over [on-policy ; demonstrations] of this rank, the Fisher / surrogate on the on-policy
    # prefix; every rank appends its share of the demonstrations (here: the last 500 rows of its shard play that part)
    if solo:
        # the ranks' blocks are [on-policy ; demonstrations] each; one rank sees the same rows as [all on-policy ; all demonstrations]
        bounds = [0] + [int(c) for c in os.environ["MJX_TEST_REF_CUTS"].split(",")] + [N]
        on_idx, demo_idx = [], []
        for b0, b1 in zip(bounds[:-1], bounds[1:]):
            nd = min(500, b1 - b0)
            on_idx += list(range(b0, b1 - nd)); demo_idx += list(range(b1 - nd, b1))
        idx = np.array(on_idx + demo_idx)
        n_on = len(on_idx)
        eng.set_policy(th, th, ident, ident)
        eng.set_batch(obs[idx], act[idx], np.concatenate([adv[on_idx], 0.01 * np.ones(len(demo_idx), np.float32)]))
        n_on_global, n_all_global = n_on, eng.N_global
        dres = eng.dapg_update(10, 1e-4, 0.05, -3.0, n_on, adv[on_idx])
    else:
        n_demo = min(500, hi - lo)
        n_on = (hi - lo) - n_demo
        eng.set_policy(th, th, ident, ident)
        adv_all = np.concatenate([adv[lo:lo + n_on], 0.01 * np.ones(n_demo, np.float32)])
        eng.set_batch(obs[lo:hi], act[lo:hi], adv_all)
        n_on_global, n_all_global = eng.global_count(n_on), eng.N_global
        dres = eng.dapg_update(10, 1e-4, 0.05, -3.0, n_on, adv[lo:lo + n_on], N_on_global=n_on_global)
    res["dapg"] = np.array([np.nan, np.nan] if dres is None else list(dres))
    res["dapg_theta"] = eng.theta_new.cpu().numpy()
    res["dapg_counts"] = np.array([n_on_global, n_all_global])
    # every rank must hold identical results (the CG scalars are recomputed redundantly from the reduced vectors)
    t = torch.from_numpy(np.concatenate([res["x"], res["theta"], res["scal"].astype(np.float32), res["trpo_theta"],
                                         res["trpo"].astype(np.float32), res["dapg_theta"], res["dapg"].astype(np.float32)])).cuda()
    if solo:
        res["ranks_identical"] = np.array([True])
        np.savez(out_path, **res)
        return
    lo_t, hi_t = t.clone(), t.clone()
    dist.all_reduce(lo_t, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi_t, op=dist.ReduceOp.MAX)
    res["ranks_identical"] = np.array([bool(torch.equal(lo_t, hi_t))])
    if rank == 0:
        np.savez(out_path, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
