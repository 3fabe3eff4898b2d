"""8-rank rehearsal of the configs that ARE 8-GPU (BASELINE configs[3] Humanoid 376-256-256-17, 4M timesteps = 500 k per GPU, 25 CG
iterations; configs[4] Adroit 39-512-512-28, 8M = 1M per GPU, 10 iterations) on ONE GPU: rank 0's shard, first with no rank sums
at all, then with libmjx's peer exchange in loop-back at world 8 -- every exchange stores the d-float vector (667 KB / 1.19 MB)
into 8 slots, raises 7 flags, waits, and sums 8 slots, all onto this rank's own buffer -- i.e. the device work of an 8-rank
exchange without the xGMI hop.  Weak scaling: the per-GPU shard is the configs' own, so efficiency = t(no sums) / t(with sums).

    python tools/rehearse_lw.py            # prints one JSON object
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mjrl_amd.engine import UpdateEngine  # noqa: E402

CFGS = (("configs3_humanoid_256x256", 376, 17, (256, 256), 500000, 25), ("configs4_adroit_512x512", 39, 28, (512, 512), 1000000, 10))
WORLD = 8


def params(n, m, hid):
    rng = np.random.RandomState(1)
    sizes = (n,) + tuple(hid) + (m,)
    flat = []
    for i in range(len(sizes) - 1):
        k = 1.0 / np.sqrt(sizes[i])
        flat += [rng.uniform(-k, k, (sizes[i + 1], sizes[i])).ravel() * (1e-2 if i == len(sizes) - 2 else 1.0), rng.uniform(-k, k, sizes[i + 1])]
    flat.append(np.full(m, -0.5))
    th = np.concatenate(flat).astype(np.float32)
    return (th + 0.02 * np.random.RandomState(1).randn(th.size)).astype(np.float32)


def time_update(n, m, hid, N, iters, n_global):
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    e = UpdateEngine(n, m, hid)
    th = params(n, m, hid)
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    e.set_policy(th, th, ident, ident)
    e.set_batch(torch.randn((N, n), generator=gen, device="cuda"), torch.randn((N, m), generator=gen, device="cuda"),
                torch.randn((N,), generator=gen, device="cuda"), N_global=n_global)
    ts = []
    for rep in range(3):
        e.set_policy(th, th, ident, ident)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.npg_update(iters, 1e-4, 0.05, -3.0)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    kind, d = e.comm_kind, e.d
    e.close()
    del e
    torch.cuda.empty_cache()
    return min(ts[1:]), kind, d


def main():
    out = {"world": WORLD, "what": __doc__.split("\n\n")[0]}
    plain = {}
    for name, n, m, hid, N, iters in CFGS:
        plain[name] = time_update(n, m, hid, N, iters, N)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", MJX_COLLECTIVES_AT_WORLD1="1", MJX_PEER_COMM="1",
                      MJX_PEER_LOOPBACK_WORLD=str(WORLD))
    import torch.distributed as dist
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    for name, n, m, hid, N, iters in CFGS:
        ms, kind, d = time_update(n, m, hid, N, iters, WORLD * N)
        p = plain[name][0]
        out[name] = {"rows_per_rank": N, "d": d, "cg_iters": iters, "exchange_bytes": 4 * d, "npg_update_ms_no_rank_sums": p,
                     "npg_update_ms_peer_loopback_world8": ms, "transport": kind, "exchanges_per_update": iters + 3,
                     "ms_per_exchange": (ms - p) / (iters + 3), "projected_weak_scaling_efficiency": p / ms}
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
