"""Behaviour cloning with the minibatch-Adam loop on the GPU.

Mirror of the reference class (mjrl/algos/behavior_cloning.py:15-143): same constructor
arguments, `compute_transformations / set_transformations / set_variance_with_data / loss /
fit / train`.  The expert data is uploaded once; every epoch's minibatches (`np.random.choice`
draws, the reference's random stream) run through `mjx_policy_minibatch_adam`: gather,
forward, loss head, backward and the torch-Adam update per step, all on the device.
"""
import time as timer

import numpy as np

from .._lib import check, ptr
from ..engine import UpdateEngine
from ..utils.ingest import upload
from ..utils.logger import DataLog

LOSS_IDS = {"MSE": 0, "MLE": 1}


class BC:
    def __init__(self, expert_paths, policy, epochs=5, batch_size=64, lr=1e-3, optimizer=None, loss_type='MSE',
                 save_logs=True, set_transforms=False, **kwargs):
        if optimizer is not None:
            raise NotImplementedError("a caller-supplied torch optimizer cannot drive the device loop; use the reference BC for that")
        if loss_type not in LOSS_IDS:
            raise ValueError("Please use valid loss type ('MSE' or 'MLE')")
        self.policy = policy
        self.expert_paths = expert_paths
        self.epochs = epochs
        self.mb_size = batch_size
        self.lr = lr
        self.logger = DataLog()
        self.loss_type = loss_type
        self.save_logs = save_logs
        if set_transforms:
            in_shift, in_scale, out_shift, out_scale = self.compute_transformations()
            self.set_transformations(in_shift, in_scale, out_shift, out_scale)
            self.set_variance_with_data(out_scale)
        self._adam = None           # (m, v) device tensors + steps taken: the torch.optim.Adam state of the reference

    # ------------------------------------------------------------------ data-driven transforms (behavior_cloning.py:51-75)
    def _stacked(self, key):
        return np.concatenate([p[key] for p in self.expert_paths])

    def compute_transformations(self):
        """(in_shift, in_scale, out_shift, out_scale) = per-column mean / std of the expert observations and actions;
        four Nones without demonstrations."""
        if not self.expert_paths:
            return (None,) * 4
        o, a = self._stacked("observations"), self._stacked("actions")
        return o.mean(axis=0), o.std(axis=0), a.mean(axis=0), a.std(axis=0)

    def set_transformations(self, in_shift=None, in_scale=None, out_shift=None, out_scale=None):
        for net in (self.policy.model, self.policy.old_model):
            net.set_transformations(in_shift, in_scale, out_shift, out_scale)

    def set_variance_with_data(self, out_scale):
        """log_std <- log(std of the expert actions) (+1e-12 inside the log, as the reference)"""
        theta = self.policy.get_param_values()
        theta[-self.policy.m:] = np.log(out_scale + 1e-12)
        self.policy.set_param_values(theta)

    # ------------------------------------------------------------------ losses (host values, for logging)
    def loss(self, data, idx=None):
        """loss on (a subset of) the data, evaluated like the reference does (behavior_cloning.py:77-105); NumPy fp32"""
        idx = np.arange(data['observations'].shape[0]) if idx is None else np.asarray(idx)
        obs, act = np.float32(data['observations'][idx]), np.float32(data['expert_actions'][idx])
        mu = self.policy.model.forward(obs)
        if self.loss_type == 'MSE':
            return float(np.mean((mu - act) ** 2))
        ls = np.float32(self.policy.get_param_values()[-self.policy.m:])
        z = (act - mu) / np.exp(ls)
        LL = -0.5 * np.sum(z ** 2, axis=1) - np.sum(ls) - 0.5 * self.policy.m * np.log(2 * np.pi)
        return float(-np.mean(LL))

    # ------------------------------------------------------------------ behavior_cloning.py:107-143
    def fit(self, data, suppress_fit_tqdm=False, **kwargs):
        assert all(k in data.keys() for k in ["observations", "expert_actions"])
        ts = timer.time()
        num_samples = data["observations"].shape[0]
        if self.save_logs:
            self.logger.log_kv('loss_before', self.loss(data))
        steps_per_epoch = int(num_samples / self.mb_size)
        steps = self.epochs * steps_per_epoch
        if steps > 0:
            # the reference draws np.random.choice(num_samples, size=mb_size) once per step, epoch after epoch
            idx = np.stack([np.random.choice(num_samples, size=self.mb_size) for _ in range(steps)]).astype(np.int32)
            eng = self._engine()
            torch = eng.torch
            p = self.policy
            theta = torch.from_numpy(np.float32(p.get_param_values())).to(eng.device)
            tr = torch.from_numpy(np.float32(p.model.packed_transforms())).to(eng.device)
            obs = eng.to_device_f32(data["observations"])
            act = eng.to_device_f32(data["expert_actions"])
            if self._adam is None:
                self._adam = [torch.zeros_like(theta), torch.zeros_like(theta), 0]
            didx = upload(eng.backend, idx)
            check(eng.lib.mjx_policy_minibatch_adam(eng.ctx, LOSS_IDS[self.loss_type], ptr(obs), ptr(act), None, ptr(didx), steps,
                                                    self.mb_size, ptr(theta), ptr(tr), None, None, 0, ptr(self._adam[0]),
                                                    ptr(self._adam[1]), self._adam[2], self.lr, 0.0, None, eng.stream()))
            self._adam[2] += steps
            p.set_param_values(eng.to_host(theta), set_new=True, set_old=True)
        else:
            p = self.policy
            p.set_param_values(p.get_param_values(), set_new=True, set_old=True)
        if self.save_logs:
            self.logger.log_kv('epoch', self.epochs)
            self.logger.log_kv('loss_after', self.loss(data))
            self.logger.log_kv('time', (timer.time() - ts))

    def train(self, **kwargs):
        self.fit(dict(observations=self._stacked("observations"), expert_actions=self._stacked("actions")), **kwargs)

    # ------------------------------------------------------------------ device plumbing
    _engine_obj = None

    def _engine(self):
        if self._engine_obj is None:
            self._engine_obj = UpdateEngine(self.policy.n, self.policy.m, self.policy.hidden_sizes)
        return self._engine_obj

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_engine_obj", None); state["_adam"] = None
        return state
