"""Shared device plumbing of the value baselines: upload the concatenated fp64 observation block and
the within-trajectory time index once, then run the feature / Gram / prediction kernels of
csrc/baseline.h through the C ABI (include/mjx.h, K6)."""
import ctypes

import numpy as np

from .. import _lib
from .._lib import check, ptr
from ..utils import ingest

FEAT_MLP, FEAT_LINEAR, FEAT_QUADRATIC = 0, 1, 2


def torch_dev():
    import torch
    if not torch.cuda.is_available():
        raise _lib.MjxError("mjrl_amd baselines need a GPU (no CPU fallback)")
    return torch, torch.device("cuda", torch.cuda.current_device())


def num_features(kind, n):
    """F of the ridge baselines' feature map (quadratic_baseline.py:20 / linear_baseline.py:20)"""
    return int(_lib.load().mjx_bl_num_features(int(kind), int(n)))


def stream(torch, dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def path_inputs(paths, inp):
    """the observation block and time index the reference's _features builds
    (mlp_baseline.py:36-58 / quadratic_baseline.py:11-41 / linear_baseline.py:11-35)."""
    if inp == 'env_features':
        o = np.concatenate([p["env_infos"]["env_features"][0] for p in paths])
    else:
        o = np.concatenate([p["observations"] for p in paths])
    if o.ndim > 2:
        o = o.reshape(o.shape[0], -1)
    tpos = np.concatenate([np.arange(len(p["rewards"]), dtype=np.int32) for p in paths])
    return np.ascontiguousarray(o, dtype=np.float64), tpos


def _time_index(paths):
    """tpos[s] = position of sample s inside its trajectory (the reference's np.arange(l) per path), vectorised"""
    lens = np.fromiter((len(p["rewards"]) for p in paths), dtype=np.int64, count=len(paths))
    starts = np.zeros(len(paths), np.int64)
    np.cumsum(lens[:-1], out=starts[1:])
    return (np.arange(int(lens.sum()), dtype=np.int64) - np.repeat(starts, lens)).astype(np.int32)


class DeviceOnly:
    """torch, the current device, libmjx and the registry handle -- what a kernel call needs besides its tensors"""

    def __init__(self):
        self.torch, self.dev = torch_dev()
        self.lib = _lib.load()
        self.handle = ingest.DeviceHandle(self.torch, self.dev, self.lib)

    def st(self):
        return stream(self.torch, self.dev)


class DeviceBlock:
    def __init__(self, paths, inp, shared=True):
        """shared=False (single-path predictions, throw-away path lists): a plain upload that leaves the iteration's
        shared batch registered."""
        self.torch, self.dev = torch_dev()
        self.lib = _lib.load()
        self.handle = ingest.DeviceHandle(self.torch, self.dev, self.lib)
        self.paths, self.shared = paths, False
        first = paths[0]["observations"] if inp != 'env_features' else None
        if shared and first is not None and first.ndim == 2 and first.dtype == np.float64:
            # the fp64 observation block of this batch: uploaded once per process (page-locked staging), shared with
            # the policy update and the other baseline call of the iteration (utils/ingest.stage_shared); the time index
            # is built once per batch as well
            self.shared = True
            self.obs = ingest.stage_shared(self.handle, paths, ("observations",))["observations"]["raw"]
            self.N, self.n = int(self.obs.shape[0]), int(self.obs.shape[1])
            self.tpos = ingest.derived(self.handle, paths, "observations", "tpos", lambda: self._time_index_dev(paths))
        else:
            o, tpos = path_inputs(paths, inp)
            self.N, self.n = o.shape
            self.obs = ingest.upload(self.handle, o)
            self.tpos = ingest.upload(self.handle, tpos)

    def st(self):
        return stream(self.torch, self.dev)

    def _time_index_dev(self, paths):
        """the time index formed on the device from the trajectory offsets (8 bytes per trajectory cross the bus instead of
        4 per timestep, and the host does not build three N-sized temporaries)"""
        got = ingest.collect_arrays(paths, "rewards")           # (the reference counts a path's steps on its rewards)
        if got is None or self.dev.type != "cuda":
            return ingest.upload(self.handle, _time_index(paths))
        off = np.zeros(len(paths) + 1, np.int64)
        np.cumsum(got[1], out=off[1:])
        offd = self.torch.from_numpy(off).to(self.dev)
        tpos = self.torch.empty(int(off[-1]), dtype=self.torch.int32, device=self.dev)
        check(self.lib.mjx_time_index(ptr(offd), len(paths), ptr(tpos), self.st()))
        return tpos

    def returns_dev(self):
        """the concatenated fp64 returns on the device: the block compute_returns left there (utils/process_samples.py)
        when these are the paths it handed them to, else one upload"""
        y = ingest.lookup(self.handle, self.paths, "returns") if self.shared else None
        if y is None:
            y = ingest.upload(self.handle, np.concatenate([np.asarray(p["returns"], np.float64) for p in self.paths]))
        return y

    def gram_dev(self, kind, y):
        """[A y]^T [A y] as an (F + 1) x (F + 1) fp64 device tensor (enqueued on the current stream, nothing waited for)"""
        F = self.lib.mjx_bl_num_features(kind, self.n)
        yt = y if hasattr(y, "data_ptr") else ingest.upload(self.handle, y, np.float64)
        G = self.torch.empty((F + 1, F + 1), dtype=self.torch.float64, device=self.dev)
        check(self.lib.mjx_bl_gram(kind, ptr(self.obs), ptr(self.tpos), ptr(yt), self.N, self.n, ptr(G), self.st()))
        return G

    def gram(self, kind, y):
        return ingest.download(self.handle, self.gram_dev(kind, y))

    def predict_linear_dev(self, kind, coef):
        ct = self.torch.from_numpy(np.ascontiguousarray(coef, dtype=np.float64)).to(self.dev)
        out = self.torch.empty(self.N, dtype=self.torch.float64, device=self.dev)
        check(self.lib.mjx_bl_predict(kind, ptr(self.obs), ptr(self.tpos), self.N, self.n, ptr(ct), ptr(out), self.st()))
        return out

    def predict_linear(self, kind, coef):
        return ingest.download(self.handle, self.predict_linear_dev(kind, coef))

    def mlp_features(self):
        out = self.torch.empty((self.N, self.n + 4), dtype=self.torch.float32, device=self.dev)
        check(self.lib.mjx_bl_features_f32(ptr(self.obs), ptr(self.tpos), self.N, self.n, ptr(out), self.st()))
        return out
