"""Rank-level plumbing of the host classes that live outside the update engine (value baselines, sample processing).

One process per GPU, every rank holds a shard of whole trajectories.  The reference has no notion of ranks: its
baseline fit, advantage normalisation and return statistics see ALL paths of an iteration
(mjrl/algos/batch_reinforce.py:94-110, baselines/quadratic_baseline.py:44-69, baselines/mlp_baseline.py:61-95).  The helpers
below restore that: sums over the ranks for everything that is a sum over samples, a rank-ordered concatenation for the one
consumer that is not (the sequential minibatch-Adam chain of the MLP baseline), ONE rank's draw for host random numbers.  All of
them are no-ops (and cost nothing) in a single process.

torch.distributed is the transport -- set-up-class traffic, a handful of calls per iteration; the per-CG-iteration sums of the
policy update run inside libmjx (engine.py).  Tensors travel on the device for the RCCL backend ("nccl") and through host
memory for any other (gloo: the tests put two ranks on one GPU).
"""
import numpy as np


def group():
    """torch.distributed when this process is one of several ranks, else None"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


def _on_device(d):
    return d.get_backend() == "nccl"


def _work_device(d, torch):
    return torch.device("cuda", torch.cuda.current_device()) if _on_device(d) else torch.device("cpu")


def sum_host(a):
    """element-wise sum over the ranks of a host array (any float / int dtype) -> ndarray of the same shape, fp64 sums for
    floats; every rank receives the same bits (an all-reduce delivers one reduction result to all)"""
    d = group()
    a = np.asarray(a)
    if d is None:
        return a
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64 if a.dtype.kind == "f" else np.int64)).to(_work_device(d, torch))
    d.all_reduce(t)
    return t.cpu().numpy().reshape(a.shape)


def sum_tensor_(t):
    """in-place sum of a device (or host) tensor over the ranks; returns it"""
    d = group()
    if d is None:
        return t
    if t.is_cuda and not _on_device(d):          # gloo: through host memory
        h = t.cpu()
        d.all_reduce(h)
        t.copy_(h)
    else:
        d.all_reduce(t)
    return t


def all_true(flag):
    """logical AND of a per-rank predicate: every rank must take the same branch wherever a branch decides which collectives
    are issued next"""
    d = group()
    if d is None:
        return bool(flag)
    return bool(sum_host(np.array([0 if flag else 1], np.int64))[0] == 0)


def counts(n_local):
    """-> list of every rank's count (rank order)"""
    d = group()
    if d is None:
        return [int(n_local)]
    v = np.zeros(d.get_world_size(), np.int64)
    v[d.get_rank()] = int(n_local)
    return [int(x) for x in sum_host(v)]


def gather_rows(t):
    """rank-ordered concatenation (dim 0) of a per-rank tensor whose row counts may differ -> tensor on t's device.  The
    ranks' blocks are padded to the longest, gathered, and trimmed: one collective."""
    d = group()
    if d is None:
        return t
    import torch
    cnt = counts(t.shape[0])
    mx = max(cnt)
    wdev = _work_device(d, torch)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=wdev)
    pad[:t.shape[0]].copy_(t)
    outs = [torch.empty_like(pad) for _ in cnt]
    d.all_gather(outs, pad)
    return torch.cat([o[:c] for o, c in zip(outs, cnt)]).to(t.device)


def broadcast_host(a, src=0):
    """rank `src`'s host array on every rank (src < 0: counted from the last rank; the shapes / dtypes must agree: callers draw
    the same-sized array everywhere, which also keeps the ranks' random streams advancing alike)"""
    d = group()
    a = np.ascontiguousarray(a)
    if d is None:
        return a
    import torch
    src = src % d.get_world_size()
    t = torch.from_numpy(a.copy()).to(_work_device(d, torch))
    d.broadcast(t, src=src)
    return t.cpu().numpy()


def mean_std(x):
    """population mean / std of a sample vector that is sharded over the ranks (two passes: mean, then squared deviations
    about it -- NumPy's algorithm, so one rank reproduces np.mean / np.std to the last bits)"""
    x = np.asarray(x, np.float64)
    d = group()
    if d is None:
        return float(np.mean(x)), float(np.std(x))
    s = sum_host(np.array([x.sum(), float(x.size)]))
    mean = float(s[0] / s[1])
    q = sum_host(np.array([((x - mean) ** 2).sum()]))
    return mean, float(np.sqrt(q[0] / s[1]))
