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
from ..utils.ingest import minibatch_indices, upload
from ..utils.logger import DataLog

LOSS_IDS = {"MSE": 0, "MLE": 1}


class BC:
    def __init__(self, expert_paths, policy, epochs=5, batch_size=64, lr=1e-3, optimizer=None, loss_type='MSE',
                 save_logs=True, set_transforms=False, **kwargs):
        if loss_type not in LOSS_IDS:
            raise ValueError("Please use valid loss type ('MSE' or 'MLE')")
        self.policy = policy
        self.expert_paths = expert_paths
        self.epochs = epochs
        self.mb_size = batch_size
        self.lr = lr
        # behavior_cloning.py:42: `optimizer` defaults to torch.optim.Adam(policy.trainable_params, lr=lr).  A caller-supplied one
        # is honoured when it IS that optimizer with another learning rate / an existing state -- what the device loop implements
        # (Adam, betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad, one group over the policy's parameters): its lr and
        # moments are taken over and written back after every fit, so the caller's object stays the optimizer's state.  Anything
        # else cannot drive the device loop and is refused (no CPU fallback).
        self.optimizer = optimizer
        if optimizer is not None:
            self.lr = self._adopt_optimizer(optimizer)
        self.logger = DataLog()
        self.loss_type = loss_type
        self.save_logs = save_logs
        if set_transforms:
            in_shift, in_scale, out_shift, out_scale = self.compute_transformations()
            self.set_transformations(in_shift, in_scale, out_shift, out_scale)
            self.set_variance_with_data(out_scale)
        self._adam = None           # (m, v) device tensors + steps taken: the torch.optim.Adam state of the reference

    def _adopt_optimizer(self, opt):
        import torch
        why = None
        if type(opt) is not torch.optim.Adam:
            why = "it is a %s, the device loop implements torch.optim.Adam" % type(opt).__name__
        elif len(opt.param_groups) != 1:
            why = "it has %d parameter groups" % len(opt.param_groups)
        else:
            g = opt.param_groups[0]
            tp = self.policy.trainable_params
            if len(g["params"]) != len(tp) or any(a is not b for a, b in zip(g["params"], tp)):
                why = "its parameters are not policy.trainable_params"
            elif tuple(g["betas"]) != (0.9, 0.999) or g["eps"] != 1e-8 or g["weight_decay"] != 0 or g.get("amsgrad", False) or g.get("maximize", False):
                why = "betas / eps / weight_decay / amsgrad differ from Adam's defaults (%r)" % ({k: g[k] for k in ("betas", "eps", "weight_decay")},)
        if why is not None:
            raise NotImplementedError("BC(optimizer=...): %s -- such an optimizer cannot drive the device loop (behavior_cloning.py:42)" % why)
        return float(opt.param_groups[0]["lr"])

    def _optimizer_state_in(self, torch, like):
        """the caller's optimizer state (exp_avg / exp_avg_sq / step per parameter) as flat device vectors, or None when it has none"""
        opt = self.optimizer
        tp = self.policy.trainable_params
        if opt is None or not all(p in opt.state and "exp_avg" in opt.state[p] for p in tp):
            return None
        m = torch.cat([opt.state[p]["exp_avg"].reshape(-1).float() for p in tp]).to(like.device)
        v = torch.cat([opt.state[p]["exp_avg_sq"].reshape(-1).float() for p in tp]).to(like.device)
        return [m, v, int(float(opt.state[tp[0]]["step"]))]

    def _optimizer_state_out(self, torch):
        opt = self.optimizer
        if opt is None or self._adam is None:
            return
        m, v, k = self._adam[0].cpu(), self._adam[1].cpu(), 0
        for p in self.policy.trainable_params:
            n = p.numel()
            st = opt.state[p]
            st["exp_avg"], st["exp_avg_sq"] = m[k:k + n].reshape(p.shape).clone(), v[k:k + n].reshape(p.shape).clone()
            st["step"] = torch.tensor(float(self._adam[2]))
            k += n

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
            idx = minibatch_indices(self._engine().lib, num_samples, steps, self.mb_size)
            eng = self._engine()
            torch = eng.torch
            p = self.policy
            theta = torch.from_numpy(np.float32(p.get_param_values())).to(eng.device)
            tr = torch.from_numpy(np.float32(p.model.packed_transforms())).to(eng.device)
            obs = eng.to_device_f32(data["observations"])
            act = eng.to_device_f32(data["expert_actions"])
            if self.optimizer is not None:
                self.lr = float(self.optimizer.param_groups[0]["lr"])          # (a scheduler may have moved it)
                self._adam = self._optimizer_state_in(torch, theta) or self._adam
            if self._adam is None:
                self._adam = [torch.zeros_like(theta), torch.zeros_like(theta), 0]
            didx = upload(eng.backend, idx)
            check(eng.lib.mjx_policy_minibatch_adam(eng.ctx, LOSS_IDS[self.loss_type], ptr(obs), ptr(act), None, ptr(didx), steps,
                                                    self.mb_size, ptr(theta), ptr(tr), None, None, 0, ptr(self._adam[0]),
                                                    ptr(self._adam[1]), self._adam[2], self.lr, 0.0, None, eng.stream()))
            self._adam[2] += steps
            p.set_param_values(eng.to_host(theta), set_new=True, set_old=True)
            self._optimizer_state_out(torch)
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
