"""Returns and (GAE) advantages over batched trajectories on the GPU.

Same entry points and in-place path mutation as the reference
(mjrl/utils/process_samples.py:3-44); the per-timestep Python loops (``discount_sum``) and
the per-path baseline forward passes become one segmented reverse scan over the concatenated
fp64 reward block (``mjx_discount_scan`` / ``mjx_gae``, csrc/vecops.h k_traj_scan) plus one
batched baseline prediction when the baseline offers ``predict_batch_device`` / ``predict_batch``.

The chain stays on the device: the rewards are uploaded once (page-locked stager, shared with the
rest of the iteration; ``compute_returns`` also starts the upload of the batch's observations and actions in the
background, utils/ingest.prefetch), returns, baseline values and advantages are computed there and REGISTERED
(utils/ingest.publish), so that the advantage whitening of ``process_paths`` and the baseline fit
later in the same ``train_step`` read the device blocks instead of concatenating and uploading the
host arrays again.  The paths still receive ``returns`` / ``baseline`` / ``advantages`` as NumPy arrays
(one read-back per block; per-path views of it), as every consumer of the reference's path format
expects.
"""
import ctypes

import numpy as np

from .. import _lib
from .._lib import check, ptr
from . import ingest, ranks


def _handle():
    import torch
    if not torch.cuda.is_available():
        raise _lib.MjxError("mjrl_amd.utils.process_samples needs a GPU (no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device())
    return ingest.DeviceHandle(torch, dev, _lib.load())


def _offsets(paths):
    lens = np.fromiter((len(p["rewards"]) for p in paths), dtype=np.int64, count=len(paths))
    off = np.zeros(len(paths) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    return off


def _stream(h):
    return ctypes.c_void_p(h.torch.cuda.current_stream(h.device).cuda_stream)


def _rewards_block(h, paths):
    """the concatenated fp64 reward block on the device: one upload per batch (shared registry)"""
    first = paths[0]["rewards"]
    if isinstance(first, np.ndarray) and first.dtype == np.float64 and first.ndim == 1:
        return ingest.stage_shared(h, paths, ("rewards",))["rewards"]["raw"].view(-1)
    r = np.concatenate([np.asarray(p["rewards"], np.float64).ravel() for p in paths])
    return ingest.upload(h, r)


def _offsets_dev(h, paths, off):
    return ingest.derived(h, paths, "rewards", "offsets_dev", lambda: h.torch.from_numpy(off).to(h.device))


def _hand_out(h, paths, key, block, off):
    """one read-back of a device block; the paths get per-path views of the host copy -> the list of views"""
    host = ingest.download_owned(h, block)   # (page-locked memory that lives as long as the paths' views of it)
    host.setflags(write=False)          # the device block stays registered for this batch: an in-place edit of a view must not go unnoticed (utils/ingest.py)
    pw = ingest._pathwalk
    if pw is not None and hasattr(pw, "hand_out") and type(paths) is list and isinstance(off, np.ndarray) and off.dtype == np.int64 and off.flags.c_contiguous:
        return pw.hand_out(paths, key, host, off)               # the same slices and dict stores in one C loop (csrc/pathwalk.c)
    views = [host[off[i]:off[i + 1]] for i in range(len(paths))]
    for p, v in zip(paths, views):
        p[key] = v
    return views


def discount_sum(x, gamma, terminal=0.0):
    """Single-sequence form (process_samples.py:37-44) on the device scan; `terminal` folds in
    as an extra trailing element."""
    h = _handle()
    torch, dev = h.torch, h.device
    xs = np.append(np.asarray(x, np.float64), float(terminal)) if terminal != 0.0 else np.asarray(x, np.float64)
    xt = torch.from_numpy(np.ascontiguousarray(xs)).to(dev)
    off = torch.tensor([0, xs.shape[0]], dtype=torch.int64, device=dev)
    y = torch.empty_like(xt)
    check(h.lib.mjx_discount_scan(ptr(xt), ptr(off), 1, float(gamma), ptr(y), _stream(h)))
    out = y.cpu().numpy()
    return out[:len(x)]


def compute_returns(paths, gamma):
    """process_samples.py:3-5"""
    if not paths:
        return
    h = _handle()
    # the first touch of a rollout batch in an iteration.  The rewards (8 bytes per timestep) go up first -- their copies are
    # queued ahead of everything else and the scan below can start at once --, then the batch's observations / actions start
    # their way to the device as native asynchronous staging jobs (utils/ingest.prefetch -> mjx_stage_async; 160 MB at 1M
    # timesteps) under the returns / advantage work: the baseline prediction and the policy update find them staged.
    # (r03 staged them before the rewards: with a Python helper thread per block that was 2-3 ms better; with native jobs the
    # rewards' copies then sit behind 2.6 ms of observation transfers in the DMA queue.)
    off = _offsets(paths)
    if np.ndim(paths[0]["rewards"]) == 2:
        return _returns_by_column(h, paths, off, gamma)
    r = _rewards_block(h, paths)
    ingest.prefetch(h, paths, ("observations", "actions"))
    y = h.torch.empty_like(r)
    check(h.lib.mjx_discount_scan(ptr(r), ptr(_offsets_dev(h, paths, off)), len(paths), float(gamma), ptr(y), _stream(h)))
    ingest.publish(h, paths, "returns", y, _hand_out(h, paths, "returns", y, off))


# ---- vector-valued rewards / baselines (process_samples.py:26-27: b.ndim == 2).  The reference's expressions are row-wise array
# arithmetic, so K reward / value components are K independent scans: the same kernels, column by column (a rare path: one small
# upload, one launch and one read-back per column; nothing is registered for the rest of the iteration).
def _dev_offsets_plain(h, off):
    return h.torch.from_numpy(off).to(h.device)


def _column_blocks(h, paths, key):
    """-> (N, K) host block of paths[.][key] (1-D arrays count as K = 1)"""
    return np.concatenate([np.asarray(p[key], np.float64).reshape(len(p["rewards"]), -1) for p in paths])


def _returns_by_column(h, paths, off, gamma):
    R = _column_blocks(h, paths, "rewards")
    offd = _dev_offsets_plain(h, off)
    out = np.empty_like(R)
    for k in range(R.shape[1]):
        x = ingest.upload(h, np.ascontiguousarray(R[:, k]))
        y = h.torch.empty_like(x)
        check(h.lib.mjx_discount_scan(ptr(x), ptr(offd), len(paths), float(gamma), ptr(y), _stream(h)))
        out[:, k] = y.cpu().numpy()
    for i, p in enumerate(paths):
        p["returns"] = out[off[i]:off[i + 1]]


def _advantages_by_column(h, paths, B, off, gamma, gae_lambda, use_gae, normalize):
    """B: (N, K) baseline values.  GAE: td = rewards + gamma b1[1:] - b1[:-1] with NumPy's broadcasting of `rewards` against the
    (T, K) block (process_samples.py:26-29: (T, K) or (T, 1) rewards; a 1-D reward vector only broadcasts when T == K, as there);
    otherwise returns - baseline (:10-13)."""
    K = B.shape[1]
    X = _column_blocks(h, paths, "rewards" if use_gae else "returns")
    if X.shape[1] not in (1, K) or (X.shape[1] == 1 and K > 1 and np.ndim(paths[0]["rewards" if use_gae else "returns"]) == 1):
        raise ValueError("operands could not be broadcast together: %s of shape (T%s) against a (T, %d) baseline"
                         % ("rewards" if use_gae else "returns", "" if X.shape[1] == 1 else ", %d" % X.shape[1], K))
    offd = _dev_offsets_plain(h, off)
    term = h.torch.from_numpy(np.fromiter((1 if p.get("terminated", False) else 0 for p in paths), dtype=np.uint8, count=len(paths))).to(h.device)
    A = np.empty_like(B)
    lam = float(gae_lambda) if use_gae else -1.0
    for k in range(K):
        x = ingest.upload(h, np.ascontiguousarray(X[:, min(k, X.shape[1] - 1)]))
        b = ingest.upload(h, np.ascontiguousarray(B[:, k]))
        a = h.torch.empty_like(x)
        check(h.lib.mjx_gae(ptr(x), ptr(b), ptr(offd), ptr(term), len(paths), float(gamma), lam, ptr(a), _stream(h)))
        A[:, k] = a.cpu().numpy()
    if normalize:
        mean, std = ranks.mean_std(A.reshape(-1))                 # (the reference: over ALL entries, :30-35)
        A = (A - mean) / (std + 1e-8)
    for i, p in enumerate(paths):
        p["advantages"] = A[off[i]:off[i + 1]]


def _baseline_block(h, paths, baseline, off):
    """baseline values of every timestep as an fp64 device block (N,), paths[i]["baseline"] set"""
    torch = h.torch
    if hasattr(baseline, "predict_batch_device"):
        b = baseline.predict_batch_device(paths)
        if b is not None:
            if b.dtype != torch.float64:
                b = b.to(torch.float64)                 # (the reference's fp32 predictions are promoted by NumPy the same way)
            ingest.publish(h, paths, "baseline", b, _hand_out(h, paths, "baseline", b, off))
            return b
    if hasattr(baseline, "predict_batch"):
        flat = np.asarray(baseline.predict_batch(paths), np.float64)
        for i, p in enumerate(paths):
            p["baseline"] = flat[off[i]:off[i + 1]].copy()
    else:
        for p in paths:
            p["baseline"] = baseline.predict(p)
        flat = np.concatenate([np.asarray(p["baseline"], np.float64) for p in paths])
    if flat.ndim != 1:
        return flat                                      # vector-valued: compute_advantages goes column by column
    return ingest.upload(h, flat)


def compute_advantages(paths, baseline, gamma, gae_lambda=None, normalize=False):
    """process_samples.py:7-35 (vector-valued baselines, :26-27: column by column).  GAE when 0 <= gae_lambda <= 1, else
    advantages = returns - baseline."""
    if not paths:
        if normalize:
            ranks.mean_std(np.zeros(0))                  # a rank without trajectories still takes part in the statistics
        return
    h = _handle()
    torch, dev = h.torch, h.device
    off = _offsets(paths)
    b = _baseline_block(h, paths, baseline, off)
    use_gae = not (gae_lambda is None or gae_lambda < 0.0 or gae_lambda > 1.0)
    if isinstance(b, np.ndarray):                        # (N, K) baseline values, process_samples.py:26-27
        return _advantages_by_column(h, paths, b, off, gamma, gae_lambda, use_gae, normalize)
    if use_gae:
        x = _rewards_block(h, paths)
    else:
        x = ingest.lookup(h, paths, "returns")           # what compute_returns left on the device
        if x is None:
            x = ingest.upload(h, np.concatenate([np.asarray(p["returns"], np.float64) for p in paths]))
    term = ingest.derived(h, paths, "rewards", "terminated_dev", lambda: torch.from_numpy(
        np.fromiter((1 if p.get("terminated", False) else 0 for p in paths), dtype=np.uint8, count=len(paths))).to(dev))
    adv = torch.empty_like(x)
    lam = float(gae_lambda) if use_gae else -1.0
    check(h.lib.mjx_gae(ptr(x), ptr(b), ptr(_offsets_dev(h, paths, off)), ptr(term), len(paths), float(gamma), lam, ptr(adv), _stream(h)))
    if normalize:                                        # process_samples.py:14-19 / 30-35: over the whole batch -- of ALL ranks
        out = ingest.download(h, adv)
        mean, std = ranks.mean_std(out)
        out = (out - mean) / (std + 1e-8)
        for i, p in enumerate(paths):
            p["advantages"] = out[off[i]:off[i + 1]]
        return
    ingest.publish(h, paths, "advantages", adv, _hand_out(h, paths, "advantages", adv, off))
