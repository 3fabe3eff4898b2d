"""Shared device plumbing of the value baselines: upload the concatenated fp64 observation block and
the within-trajectory time index once, then run the feature / Gram / prediction kernels of
csrc/baseline.h through the C ABI (include/mjx.h, K6)."""
import ctypes

import numpy as np

from .. import _lib
from .._lib import check, ptr

FEAT_MLP, FEAT_LINEAR, FEAT_QUADRATIC = 0, 1, 2


def torch_dev():
    import torch
    if not torch.cuda.is_available():
        raise _lib.MjxError("mjrl_amd baselines need a GPU (no CPU fallback)")
    return torch, torch.device("cuda", torch.cuda.current_device())


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


class _StagerBackend:
    """what utils/ingest.PathStager needs from a backend"""

    def __init__(self, torch, device, lib):
        self.torch, self.device, self.lib = torch, device, lib


class DeviceBlock:
    def __init__(self, paths, inp, shared=True):
        """shared=False (single-path predictions, throw-away path lists): a plain upload that leaves the iteration's
        shared batch registered."""
        self.torch, self.dev = torch_dev()
        self.lib = _lib.load()
        first = paths[0]["observations"] if inp != 'env_features' else None
        if shared and first is not None and first.ndim == 2 and first.dtype == np.float64:
            # the fp64 observation block of this batch: uploaded once per process (page-locked staging), shared with
            # the policy update and the other baseline call of the iteration (utils/ingest.stage_shared)
            from ..utils.ingest import stage_shared
            self.obs = stage_shared(_StagerBackend(self.torch, self.dev, self.lib), paths, ("observations",))["observations"]["raw"]
            self.N, self.n = int(self.obs.shape[0]), int(self.obs.shape[1])
            tpos = np.concatenate([np.arange(len(p["rewards"]), dtype=np.int32) for p in paths])
        else:
            o, tpos = path_inputs(paths, inp)
            self.N, self.n = o.shape
            self.obs = self.torch.from_numpy(o).to(self.dev)
        self.tpos = self.torch.from_numpy(tpos).to(self.dev)

    def st(self):
        return stream(self.torch, self.dev)

    def gram(self, kind, y):
        F = self.lib.mjx_bl_num_features(kind, self.n)
        yt = self.torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64)).to(self.dev)
        G = self.torch.empty((F + 1, F + 1), dtype=self.torch.float64, device=self.dev)
        check(self.lib.mjx_bl_gram(kind, ptr(self.obs), ptr(self.tpos), ptr(yt), self.N, self.n, ptr(G), self.st()))
        return G.cpu().numpy()

    def predict_linear(self, kind, coef):
        ct = self.torch.from_numpy(np.ascontiguousarray(coef, dtype=np.float64)).to(self.dev)
        out = self.torch.empty(self.N, dtype=self.torch.float64, device=self.dev)
        check(self.lib.mjx_bl_predict(kind, ptr(self.obs), ptr(self.tpos), self.N, self.n, ptr(ct), ptr(out), self.st()))
        return out.cpu().numpy()

    def mlp_features(self):
        out = self.torch.empty((self.N, self.n + 4), dtype=self.torch.float32, device=self.dev)
        check(self.lib.mjx_bl_features_f32(ptr(self.obs), ptr(self.tpos), self.N, self.n, ptr(out), self.st()))
        return out
