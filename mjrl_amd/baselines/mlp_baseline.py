"""MLP value baseline trained with minibatch Adam on the GPU.

Mirrors mjrl/baselines/mlp_baseline.py:10-105 (+ utils/optimize_model.py:7-36): same constructor,
``fit(paths, return_errors)``, ``predict(path)``; the ReLU network (n+4) -> hidden -> 1 is initialised
through torch's nn.Linear so a given seed reproduces the reference's initial weights, and every epoch
draws its row permutation from NumPy's global RNG like ``fit_data`` does.  State (weights, Adam
moments, step count) is plain NumPy, so the object pickles / deep-copies like the reference's.
"""
import ctypes

import numpy as np

from .._lib import check, ptr
from ..utils import ingest, ranks
from ._features import DeviceBlock, DeviceOnly


class MLPBaseline:
    def __init__(self, env_spec, inp_dim=None, inp='obs', learn_rate=1e-3, reg_coef=0.0, batch_size=64, epochs=1,
                 use_gpu=False, hidden_sizes=(128, 128)):
        self.n = inp_dim if inp_dim is not None else env_spec.observation_dim
        self.batch_size = batch_size
        self.epochs = epochs
        self.reg_coef = reg_coef
        self.learn_rate = learn_rate
        self.use_gpu = use_gpu          # accepted for compatibility; the kernels always run on the GPU
        self.inp = inp
        self.hidden_sizes = tuple(int(h) for h in hidden_sizes)
        import torch
        sizes = (self.n + 4,) + self.hidden_sizes + (1,)
        flat = []
        for i in range(len(sizes) - 1):               # same construction order as mlp_baseline.py:21-28
            lin = torch.nn.Linear(sizes[i], sizes[i + 1])
            flat += [lin.weight.detach().numpy().ravel(), lin.bias.detach().numpy().ravel()]
        self.params = np.concatenate(flat).astype(np.float32)
        self.adam_m = np.zeros_like(self.params)
        self.adam_v = np.zeros_like(self.params)
        self.adam_steps = 0
        self.epoch_losses = []

    def _hid(self):
        return (ctypes.c_int * max(1, len(self.hidden_sizes)))(*self.hidden_sizes)

    def _forward(self, blk, feat, params_t):
        N = int(feat.shape[0])
        out = blk.torch.empty(N, dtype=blk.torch.float32, device=blk.dev)
        check(blk.lib.mjx_mlp_predict(ptr(feat), N, self.n + 4, self._hid(), len(self.hidden_sizes), ptr(params_t), ptr(out), blk.st()))
        return out

    def fit(self, paths, return_errors=False):
        """mlp_baseline.py:61-95 + optimize_model.py:7-36.  With torch.distributed initialised `paths` is this rank's trajectory
        shard; the reference fits ONE network on all paths of the iteration (batch_reinforce.py:94-110), and minibatch Adam is a
        sequential chain, not a sum over samples -- so the ranks' fp32 feature blocks and returns are concatenated in rank order
        (one all-gather of N x (n + 5) floats) and EVERY rank runs the identical persistent trainer on the whole block from the
        same permutations (the LAST rank's draw from NumPy's global stream -- it sampled the batch's last episodes, so under the
        samplers' per-episode seeding its stream stands where a single process's would; every rank draws, so the streams advance
        alike).  The
        trainer is one workgroup whatever the batch size: running it redundantly costs no wall time over running it once and
        broadcasting, and the ranks' baselines stay bit-identical -- equal to a one-rank fit on the rank-ordered paths."""
        if paths:
            blk = DeviceBlock(paths, self.inp)
            torch = blk.torch
            feat = blk.mlp_features()
            y64 = blk.returns_dev()                            # fp64, left on the device by compute_returns when possible
            y = torch.empty(int(y64.shape[0]), dtype=torch.float32, device=blk.dev)
            check(blk.lib.mjx_cast_f64_f32(ptr(y64), int(y64.shape[0]), ptr(y), blk.st()))   # == astype('float32') (mlp_baseline.py:66)
        else:                                                  # a rank without trajectories still takes part in the gather
            blk = DeviceOnly()
            torch = blk.torch
            feat = torch.zeros((0, self.n + 4), dtype=torch.float32, device=blk.dev)
            y = torch.zeros(0, dtype=torch.float32, device=blk.dev)
        if ranks.group() is not None:
            feat, y = ranks.gather_rows(feat).contiguous(), ranks.gather_rows(y).contiguous()
        num_samples = int(y.shape[0])
        if return_errors:
            returns = ingest.download(blk.handle, y)
        p = torch.from_numpy(self.params).to(blk.dev)
        if return_errors:
            errors = returns - ingest.download(blk.handle, self._forward(blk, feat, p))
            error_before = np.sum(errors ** 2) / (np.sum(returns ** 2) + 1e-8)
        m, v = torch.from_numpy(self.adam_m).to(blk.dev), torch.from_numpy(self.adam_v).to(blk.dev)
        perm = np.concatenate([np.random.permutation(num_samples) for _ in range(self.epochs)]).astype(np.int32) \
            if self.epochs > 0 else np.zeros(1, np.int32)
        perm = ranks.broadcast_host(perm, src=-1)               # (one process: itself)
        perm_t = ingest.upload(blk.handle, perm)
        losses = torch.zeros(max(self.epochs, 1), dtype=torch.float64, device=blk.dev)
        check(blk.lib.mjx_mlp_fit_adam(ptr(feat), ptr(y), num_samples, self.n + 4, self._hid(), len(self.hidden_sizes), ptr(p), ptr(m),
                                       ptr(v), self.adam_steps, ptr(perm_t), int(self.epochs), int(self.batch_size),
                                       float(self.learn_rate), float(self.reg_coef), ptr(losses), blk.st()))
        steps = max(int(num_samples / self.batch_size) - 1, 0)
        self.adam_steps += steps * self.epochs
        self.params, self.adam_m, self.adam_v = (ingest.download(blk.handle, t) for t in (p, m, v))
        self.epoch_losses = list(losses.cpu().numpy()[:self.epochs] / max(steps, 1))
        if return_errors:
            errors = returns - ingest.download(blk.handle, self._forward(blk, feat, p))
            error_after = np.sum(errors ** 2) / (np.sum(returns ** 2) + 1e-8)
            return error_before, error_after

    def predict_batch_device(self, paths, shared=True):
        """fp32 predictions of all timesteps as a device block (the GAE chain of utils/process_samples stays there)"""
        blk = DeviceBlock(paths, self.inp, shared)
        p = blk.torch.from_numpy(self.params).to(blk.dev)
        return self._forward(blk, blk.mlp_features(), p)

    def predict_batch(self, paths, shared=True):
        out = self.predict_batch_device(paths, shared)
        import torch
        return ingest.download(ingest.DeviceHandle(torch, out.device, None), out)

    def predict(self, path):
        return self.predict_batch([path], shared=False)
