#!/usr/bin/env python
"""bench.py -- NPG updates/sec on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): HalfCheetah shapes obs=17 / act=6, 64x64 tanh MLP,
NPG with 10 CG iterations (damping 1e-4, normalized step 0.05) over a 1M-timestep batch
(1000 synthetic trajectories x 1000 steps), STRONG scaling: the 1M batch is sharded by
trajectory over the N ranks, one RCCL all-reduce of the flat gradient and one per CG
iteration.

One "step" = one device-resident NPG.train_from_paths (mjrl/algos/npg_cg.py:108-142) with the
whitened batch already in HBM:  K1 surrogate+VPG -> CG(10 x [K2 FVP + vector update]) ->
alpha from g.x (host scalar) -> theta += alpha x -> K3 surrogate+KL (host scalars) -> old := new.
Prints ONE JSON line (rank 0) with the `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

N_OBS, N_ACT, HIDDEN = 17, 6, (64, 64)
N_TRAJ, T = 1000, 1000
CG_ITERS, DAMPING, STEP = 10, 1e-4, 0.05
FP32_MFMA_PEAK_TF = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBS = 8000.0


def synth_shard(rank, world, n_traj=N_TRAJ, T_=T):
    """this rank's trajectories of the global synthetic batch (seeded per trajectory)."""
    lo, hi = rank * n_traj // world, (rank + 1) * n_traj // world
    obs = np.empty(((hi - lo) * T_, N_OBS), np.float32)
    act = np.empty(((hi - lo) * T_, N_ACT), np.float32)
    adv = np.empty((hi - lo) * T_, np.float64)
    for i, tr in enumerate(range(lo, hi)):
        rng = np.random.RandomState(1000 + tr)
        obs[i * T_:(i + 1) * T_] = rng.randn(T_, N_OBS)
        act[i * T_:(i + 1) * T_] = rng.randn(T_, N_ACT)
        adv[i * T_:(i + 1) * T_] = rng.randn(T_)
    return obs, act, adv


def initial_params():
    """random-init weights of the benchmark architecture (nn.Linear-style U(-1/sqrt(fan_in), 1/sqrt(fan_in)), last layer
    x 1e-2, log_std -0.5) + the 0.1 N(0,1) perturbation of SURVEY 8d.  (No checkpoints, no network: "data": "synthetic".)"""
    rng = np.random.RandomState(1)
    sizes = (N_OBS,) + tuple(HIDDEN) + (N_ACT,)
    flat = []
    for i in range(len(sizes) - 1):
        k = 1.0 / np.sqrt(sizes[i])
        W, b = rng.uniform(-k, k, (sizes[i + 1], sizes[i])), rng.uniform(-k, k, sizes[i + 1])
        if i == len(sizes) - 2:
            W, b = 1e-2 * W, 1e-2 * b
        flat += [W.ravel(), b]
    flat.append(np.full(N_ACT, -0.5))
    th = np.concatenate(flat).astype(np.float32)
    return (th + 0.1 * np.random.RandomState(1).randn(th.size)).astype(np.float32)


def cpu_baseline(theta0, sample_traj):
    """The reference's CPU algorithm (torch-autograd port, oracle/torch_port.py) timed on this
    box's host cores on a bounded slice of the same workload.  torch's intra-op thread count is
    calibrated first on a small slice (many-core hosts are slower at their default of one thread per
    core on these skinny matrices), so the baseline is the CPU's best showing."""
    import torch
    from oracle import torch_port
    obs, act, adv = synth_shard(0, N_TRAJ // sample_traj)        # first `sample_traj` trajectories
    obs, act = obs.astype(np.float64), act.astype(np.float64)    # the reference holds fp64 rollouts
    adv = (adv - adv.mean()) / (adv.std() + 1e-6)
    kw = dict(cg_iters=CG_ITERS, damping=DAMPING, delta=STEP)
    default_threads = torch.get_num_threads()
    cal = {}
    ncal = 40000
    for k in sorted({8, 16, 32, 64, default_threads}):
        if k > default_threads:
            continue
        torch.set_num_threads(k)
        torch_port.npg_update(theta0, obs[:10000], act[:10000], adv[:10000], N_OBS, N_ACT, HIDDEN, **kw)   # warm-up
        t0 = time.time()
        torch_port.npg_update(theta0, obs[:ncal], act[:ncal], adv[:ncal], N_OBS, N_ACT, HIDDEN, **kw)
        cal[k] = time.time() - t0
    best = min(cal, key=cal.get)
    torch.set_num_threads(best)
    t0 = time.time()
    torch_port.npg_update(theta0, obs, act, adv, N_OBS, N_ACT, HIDDEN, **kw)
    dt = time.time() - t0
    torch.set_num_threads(default_threads)
    n = obs.shape[0]
    ups = (n / float(N_TRAJ * T)) / dt                            # linear-in-N extrapolation to 1M
    return dict(value=ups, unit="updates/s", cores=int(best), kind="port",
                sample="%d-timestep slice (%d traj) of the 1M batch, one NPG update in %.2f s on torch-CPU "
                       "(autograd double-backward HVP, as the reference) with %d intra-op threads (best of %s on a "
                       "%d-sample calibration), scaled linearly to 1M" % (n, sample_traj, dt, best,
                                                                          {k: round(v, 2) for k, v in cal.items()}, ncal),
                seconds=dt, nproc=os.cpu_count())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fvp-event-stride", type=int, default=11,
                    help="bracket every k-th Fisher-vector-product launch with HIP events (k coprime to the CG iteration count: every CG position is sampled equally); 0: none")
    ap.add_argument("--cpu-sample-traj", type=int, default=200)
    ap.add_argument("--rehearse-world", type=int, default=0,
                    help="diagnostic, 1 GPU: run rank 0's share of an R-rank job (1/R of the batch, global N, "
                         "RCCL collectives on a 1-rank group); the line is tagged 'rehearsal' and is not the metric")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    # (test hooks: MJX_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and MJX_BENCH_BACKEND=gloo replaces RCCL, which refuses
    #  two ranks on one device -- tests/test_gpu_parity.py runs the N = 2 path of this script on a 1-GPU box that way)
    dev_index = 0 if os.environ.get("MJX_BENCH_SHARE_GPU") == "1" else local_rank
    backend = os.environ.get("MJX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    shards = world
    if args.rehearse_world > 1:
        assert world == 1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ["MJX_COLLECTIVES_AT_WORLD1"] = "1"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        shards = args.rehearse_world

    from mjrl_amd._lib import check
    from mjrl_amd.engine import UpdateEngine

    theta0 = initial_params()
    obs, act, adv = synth_shard(rank, shards)
    # advantage whitening over the global batch (batch_reinforce.py:185); identical on all ranks
    s = torch.tensor([adv.sum(), float(adv.size)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(s)
    mean = float(s[0] / s[1])
    q = torch.tensor([((adv - mean) ** 2).sum()], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(q)
    std = float(np.sqrt(q.item() / s[1].item()))
    adv = (adv - mean) / (std + 1e-6)

    eng = UpdateEngine(N_OBS, N_ACT, HIDDEN)
    ident = np.concatenate([np.zeros(N_OBS), np.ones(N_OBS), np.zeros(N_ACT), np.ones(N_ACT)]).astype(np.float32)
    eng.set_policy(theta0, theta0, ident, ident)
    eng.set_batch(obs, act, adv, N_global=N_TRAJ * T if args.rehearse_world > 1 else None)   # resident in HBM from here on
    assert eng.N_global == N_TRAJ * T, eng.N_global
    theta0_dev = torch.from_numpy(theta0).to(eng.device)
    last = {}

    def one_update():
        # the call sequence of NPG.train_from_paths (mjrl_amd/algos/npg_cg.py): everything is enqueued, one read-back
        # ONE call into libmjx (mjx_npg_update): K1, rank sum, CG (10 x [K2, rank sum, vector update]), step length on the
        # device, step, K3, rank sum -- then one read-back
        surr_after, kl = eng.npg_update(CG_ITERS, DAMPING, STEP, -3.0)
        late = eng.deferred()
        last.update(alpha=late["alpha"], kl=kl, surr_improvement=surr_after - late["surr_before"])
        # old := new happens here in training; the bench restores theta0 so every step does identical work
        eng.theta_new.copy_(theta0_dev)
        eng.old_is_new = True
        eng._bind_policy()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_update()
    fence()
    check(eng.lib.mjx_profile_enable(eng.ctx, args.fvp_event_stride))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_update()
    fence()
    dt = time.perf_counter() - t0
    prof = (ctypes.c_double * 2)()
    check(eng.lib.mjx_profile_read(eng.ctx, prof))
    check(eng.lib.mjx_profile_enable(eng.ctx, 0))
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        fvp_ms = prof[0] / prof[1] if prof[1] > 0 else float("nan")      # (--fvp-event-stride 0: no roofline figures)
        P = N_OBS * 64 + 64 * 64 + 64 * N_ACT
        # The CG loop runs the cached-forward FVP instance: K1 stores h1 / h2 and the normalised observation image once
        # per update (theta is fixed during CG), each product then costs the tangent + backward passes:
        # 2(4P - 2 n h1) FLOP and 4(h1 + h2 + NP) B per sample, NP = n + 1 padded to 4 (SURVEY 8d "cached-activation
        # variant").  The recompute instance (2(5P - 2 n h1) = 51 328 FLOP, 68 B per sample) is what mjx_fvp runs
        # without a preceding K1.
        flop_per_sample = 2 * (4 * P - 2 * N_OBS * 64)               # 40 192 @cfg2
        bytes_per_sample = 4 * (64 + 64 + ((N_OBS + 1 + 3) & ~3))    # 592 @cfg2
        flop_recompute = 2 * (5 * P - 2 * N_OBS * 64)                # 51 328 @cfg2
        n_loc = eng.N_local
        achieved_tf = flop_per_sample * n_loc / (fvp_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_fvp_pmc.json")
        if os.path.exists(pmc) and world == 1:
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "NPG updates/sec (1M-timestep batch, 64x64 MLP)",
            "value": args.steps / dt,
            "unit": "updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: HalfCheetah-v2 shapes (obs=17, act=6), 64x64 tanh MLP, "
                                   "NPG 10 CG iters, 1M timesteps/batch (1000 traj x 1000), device-resident update",
                       "global_batch": N_TRAJ * T, "parallelism": "dp%d (trajectory shards, RCCL all-reduce per CG iter)" % world,
                       "cg_iters": CG_ITERS, "damping": DAMPING},
            "roofline": {"bound": "mfma", "kernel": "k_fused<64,64,1,8,MODE_FVP,NP=20,CACHED>",
                         "achieved": achieved_tf, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP32_MFMA_PEAK_TF, "traffic": traffic,
                         "avg_launch_ms": fvp_ms, "launches": int(prof[1]),
                         "launches_timed": "every %d-th of %d (HIP events on the launch stream, inside the timed region)" % (max(args.fvp_event_stride, 1), args.steps * CG_ITERS),
                         "flop_per_launch": flop_per_sample * n_loc,
                         "algorithmic_bytes_per_launch": bytes_per_sample * n_loc,
                         "hbm_GBps_algorithmic": bytes_per_sample * n_loc / (fvp_ms * 1e-3) / 1e9,
                         "hbm_frac_algorithmic": bytes_per_sample * n_loc / (fvp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "fvp_equivalent_TFLOPs_if_recomputed": flop_recompute * n_loc / (fvp_ms * 1e-3) / 1e12},
            "check": last,
        }
        if args.rehearse_world > 1:
            out["rehearsal"] = "rank 0 of %d on one GPU, 1-rank RCCL group: NOT the metric" % args.rehearse_world
            out["roofline"]["traffic"] = None
        if world == 1 and not args.no_cpu_baseline and args.rehearse_world <= 1:
            out["cpu_baseline"] = cpu_baseline(theta0, args.cpu_sample_traj)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
