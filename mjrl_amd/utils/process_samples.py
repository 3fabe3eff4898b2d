"""Returns and (GAE) advantages over batched trajectories on the GPU.

Same entry points and in-place path mutation as the reference
(mjrl/utils/process_samples.py:3-44); the per-timestep Python loops (``discount_sum``) and
the per-path baseline forward passes become one segmented reverse scan over the concatenated
fp64 reward block (``mjx_discount_scan`` / ``mjx_gae``, csrc/vecops.h k_traj_scan) plus one
batched baseline prediction when the baseline offers ``predict_batch``.
"""
import ctypes

import numpy as np

from .. import _lib
from .._lib import check, ptr


def _torch_dev():
    import torch
    if not torch.cuda.is_available():
        raise _lib.MjxError("mjrl_amd.utils.process_samples needs a GPU (no CPU fallback)")
    return torch, torch.device("cuda", torch.cuda.current_device())


def _offsets(paths):
    lens = np.array([len(p["rewards"]) for p in paths], dtype=np.int64)
    off = np.zeros(len(paths) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    return off


def _stream(torch, dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def discount_sum(x, gamma, terminal=0.0):
    """Single-sequence form (process_samples.py:37-44) on the device scan; `terminal` folds in
    as an extra trailing element."""
    torch, dev = _torch_dev()
    lib = _lib.load()
    xs = np.append(np.asarray(x, np.float64), float(terminal)) if terminal != 0.0 else np.asarray(x, np.float64)
    xt = torch.from_numpy(np.ascontiguousarray(xs)).to(dev)
    off = torch.tensor([0, xs.shape[0]], dtype=torch.int64, device=dev)
    y = torch.empty_like(xt)
    check(lib.mjx_discount_scan(ptr(xt), ptr(off), 1, float(gamma), ptr(y), _stream(torch, dev)))
    out = y.cpu().numpy()
    return out[:len(x)]


def compute_returns(paths, gamma):
    """process_samples.py:3-5"""
    if not paths:
        return
    torch, dev = _torch_dev()
    lib = _lib.load()
    off = _offsets(paths)
    r = torch.from_numpy(np.concatenate([np.asarray(p["rewards"], np.float64) for p in paths])).to(dev)
    offt = torch.from_numpy(off).to(dev)
    y = torch.empty_like(r)
    check(lib.mjx_discount_scan(ptr(r), ptr(offt), len(paths), float(gamma), ptr(y), _stream(torch, dev)))
    out = y.cpu().numpy()
    for i, p in enumerate(paths):
        p["returns"] = out[off[i]:off[i + 1]].copy()


def _predict_all(paths, baseline):
    if hasattr(baseline, "predict_batch"):
        flat = np.asarray(baseline.predict_batch(paths), np.float64)
        off = _offsets(paths)
        for i, p in enumerate(paths):
            p["baseline"] = flat[off[i]:off[i + 1]].copy()
        return flat
    for p in paths:
        p["baseline"] = baseline.predict(p)
    return np.concatenate([np.asarray(p["baseline"], np.float64) for p in paths])


def compute_advantages(paths, baseline, gamma, gae_lambda=None, normalize=False):
    """process_samples.py:7-35 (1-D baselines).  GAE when 0 <= gae_lambda <= 1, else
    advantages = returns - baseline."""
    if not paths:
        return
    torch, dev = _torch_dev()
    lib = _lib.load()
    off = _offsets(paths)
    b = _predict_all(paths, baseline)
    if b.ndim != 1:
        raise NotImplementedError("vector-valued baselines (process_samples.py:26-27) are not supported on the device path")
    use_gae = not (gae_lambda is None or gae_lambda < 0.0 or gae_lambda > 1.0)
    src = "rewards" if use_gae else "returns"
    x = torch.from_numpy(np.concatenate([np.asarray(p[src], np.float64) for p in paths])).to(dev)
    bt = torch.from_numpy(np.ascontiguousarray(b)).to(dev)
    offt = torch.from_numpy(off).to(dev)
    term = torch.from_numpy(np.array([1 if p.get("terminated", False) else 0 for p in paths], dtype=np.uint8)).to(dev)
    adv = torch.empty_like(x)
    lam = float(gae_lambda) if use_gae else -1.0
    check(lib.mjx_gae(ptr(x), ptr(bt), ptr(offt), ptr(term), len(paths), float(gamma), lam, ptr(adv), _stream(torch, dev)))
    out = adv.cpu().numpy()
    if normalize:
        out = (out - out.mean()) / (out.std() + 1e-8)
    for i, p in enumerate(paths):
        p["advantages"] = out[off[i]:off[i + 1]].copy()
