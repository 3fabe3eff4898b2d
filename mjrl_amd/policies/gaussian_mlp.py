"""Diagonal-Gaussian tanh-MLP policy with mjrl's operator surface.

Mirrors ``mjrl.policies.gaussian_mlp.MLP`` (reference mjrl/policies/gaussian_mlp.py:7-145):
same constructor, same flat parameter order, same ``get_action`` RNG stream, same
``set_param_values`` clamping.  Differences by design:

* state is plain NumPy (flat fp32 vectors for the new and the old parameter copy plus the
  affine transforms), so the object pickles / deep-copies / forks freely
  (mjrl/utils/train_agent.py:83,102,129-131; mjrl/samplers/core.py:196) and ``get_action``
  never touches HIP;
* the batch operators (likelihoods, surrogate, gradient, Fisher-vector products) of the NPG /
  TRPO / DAPG agents are not evaluated here -- the agents drive them on the GPU through
  ``mjrl_amd.engine``.  A torch-CPU mirror (``model`` / ``trainable_params`` / ``new_dist_info``
  ...) aliasing the same memory exists for callers that optimise the policy with torch
  optimisers (behavior_cloning.py:42, ppo_clip.py:46).
"""
import numpy as np

LOG_2PI = float(np.log(2.0 * np.pi))


def _layer_sizes(n, m, hidden):
    return (int(n),) + tuple(int(h) for h in hidden) + (int(m),)


class _NetView:
    """Stand-in for ``policy.model`` / ``policy.old_model`` (reference FCNetwork,
    mjrl/utils/fc_network.py:6-52): owns the four affine transforms and evaluates the
    network on small host batches with NumPy."""

    def __init__(self, policy, which):
        self._p, self._which = policy, which
        n, m = policy.n, policy.m
        self.obs_dim, self.act_dim = n, m
        self.layer_sizes = _layer_sizes(n, m, policy.hidden_sizes)
        self.set_transformations()

    # fc_network.py:27-37
    def set_transformations(self, in_shift=None, in_scale=None, out_shift=None, out_scale=None):
        n, m = self.obs_dim, self.act_dim
        self.transformations = dict(in_shift=in_shift, in_scale=in_scale, out_shift=out_shift, out_scale=out_scale)
        f = lambda v, d, k: np.full(k, d, np.float32) if v is None else np.float32(v).reshape(k).copy()
        self.in_shift, self.in_scale = f(in_shift, 0.0, n), f(in_scale, 1.0, n)
        self.out_shift, self.out_scale = f(out_shift, 0.0, m), f(out_scale, 1.0, m)

    def packed_transforms(self):
        return np.concatenate([np.asarray(self.in_shift, np.float32).ravel(), np.asarray(self.in_scale, np.float32).ravel(),
                               np.asarray(self.out_shift, np.float32).ravel(), np.asarray(self.out_scale, np.float32).ravel()])

    def _params(self):
        return self._p._new if self._which == "new" else self._p._old

    # fc_network.py:39-52
    def forward(self, x):
        """NumPy in -> NumPy out (fp32; what get_action uses: no torch, fork-safe).
        torch in -> torch out, differentiable w.r.t. ``policy.trainable_params`` for the new model
        (what BC / PPO optimise through, behavior_cloning.py:104, ppo_clip.py:58-102)."""
        if hasattr(x, "detach") and x.is_cuda:
            return self._p._device_forward(x, self)           # batched inference on the GPU, no autograd
        if hasattr(x, "detach"):
            import torch
            ps = self._p.trainable_params if self._which == "new" else self._p.old_params
            out = (x.float() - torch.from_numpy(self.in_shift)) / (torch.from_numpy(self.in_scale) + 1e-8)
            nl = (len(ps) - 1) // 2
            for i in range(nl):
                out = torch.nn.functional.linear(out, ps[2 * i], ps[2 * i + 1])
                if i < nl - 1:
                    out = torch.tanh(out)
            return out * torch.from_numpy(self.out_scale) + torch.from_numpy(self.out_shift)
        xin = np.asarray(x, np.float32)
        out = (xin - self.in_shift) / (self.in_scale + np.float32(1e-8))
        Ws, bs = self._p._unflatten(self._params())
        for W, b in zip(Ws[:-1], bs[:-1]):
            out = np.tanh(out @ W.T + b)
        return (out @ Ws[-1].T + bs[-1]) * self.out_scale + self.out_shift

    __call__ = forward


class MLP:
    def __init__(self, env_spec, hidden_sizes=(64, 64), min_log_std=-3, init_log_std=0, seed=None):
        """Same arguments as the reference (gaussian_mlp.py:8-12)."""
        self.n = env_spec.observation_dim
        self.m = env_spec.action_dim
        self.min_log_std = min_log_std
        self.hidden_sizes = tuple(int(h) for h in hidden_sizes)
        sizes = _layer_sizes(self.n, self.m, self.hidden_sizes)

        # Initialise through torch's own nn.Linear so that a given seed yields the reference's
        # initial policy and leaves the global RNG streams where the reference leaves them
        # (gaussian_mlp.py:26-56: model, old_model and obs_var all draw from torch's RNG).
        import torch
        if seed is not None:
            torch.manual_seed(seed)
            np.random.seed(seed)
        layers = [torch.nn.Linear(sizes[i], sizes[i + 1]) for i in range(len(sizes) - 1)]
        with torch.no_grad():
            layers[-1].weight.mul_(1e-2)
            layers[-1].bias.mul_(1e-2)
        flat = []
        for l in layers:
            flat += [l.weight.detach().numpy().ravel(), l.bias.detach().numpy().ravel()]
        flat.append(np.full(self.m, init_log_std, np.float32))
        _ = [torch.nn.Linear(sizes[i], sizes[i + 1]) for i in range(len(sizes) - 1)]     # old_model's draws
        _ = torch.randn(self.n)                                                          # obs_var's draw

        self._new = np.ascontiguousarray(np.concatenate(flat), dtype=np.float32)
        self._old = self._new.copy()
        self._tp = self._op = None
        self.param_shapes = [s for i in range(len(sizes) - 1) for s in ((sizes[i + 1], sizes[i]), (sizes[i + 1],))] + [(self.m,)]
        self.param_sizes = [int(np.prod(s)) for s in self.param_shapes]
        self.d = int(np.sum(self.param_sizes))
        self.log_std_val = np.float64(self._new[-self.m:].copy())
        self.model = _NetView(self, "new")
        self.old_model = _NetView(self, "old")

    # ------------------------------------------------------------------ utilities
    def _unflatten(self, theta):
        Ws, bs, k = [], [], 0
        for i in range(0, len(self.param_shapes) - 1, 2):
            sw, sb = self.param_sizes[i], self.param_sizes[i + 1]
            Ws.append(theta[k:k + sw].reshape(self.param_shapes[i])); k += sw
            bs.append(theta[k:k + sb]); k += sb
        return Ws, bs

    # ------------------------------------------------------------------ torch mirror (shared memory)
    def _torch_views(self, flat, grad):
        import torch
        out, k = [], 0
        for shape, size in zip(self.param_shapes, self.param_sizes):
            t = torch.from_numpy(flat[k:k + size].reshape(shape))      # a view: optimiser steps land in the NumPy store
            out.append(t.requires_grad_(grad)); k += size
        return out

    @property
    def trainable_params(self):
        """list of torch leaves [W1, b1, ..., log_std] aliasing the NumPy parameter store
        (gaussian_mlp.py:38; behavior_cloning.py:42 and ppo_clip.py:46 hand it to torch.optim.Adam)."""
        if self.__dict__.get("_tp") is None:
            self._tp = self._torch_views(self._new, True)
        return self._tp

    @property
    def old_params(self):
        if self.__dict__.get("_op") is None:
            self._op = self._torch_views(self._old, False)
        return self._op

    @property
    def log_std(self):
        return self.trainable_params[-1]

    @property
    def old_log_std(self):
        return self.old_params[-1]

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_tp", None); state.pop("_op", None)       # rebuilt lazily; the NumPy store is the state
        state.pop("_dev", None)                                # device context: training process only
        return state

    def get_param_values(self):
        return self._new.copy()

    def get_old_param_values(self):
        return self._old.copy()

    def set_param_values(self, new_params, set_new=True, set_old=True):
        """gaussian_mlp.py:65-87 (float32 cast, log_std clamped at min_log_std); written in place so
        the torch views handed to optimisers stay valid."""
        # Reference quirk worth knowing (gaussian_mlp.py:65-87): `torch.from_numpy(vals).float()` is a no-copy view
        # when `new_params` is float32, so after a call with set_new and set_old the reference's new and old NETWORK
        # tensors share memory (log_std does not: torch.clamp makes a fresh tensor).  Nothing in NPG / TRPO / DAPG
        # mutates parameters in place, so it is invisible there; the in-place torch optimizers (PPO) do see it --
        # `ppo_clip.PPO` reads this flag to reproduce the reference's numbers.
        self.reference_new_old_alias = bool(set_new and set_old and isinstance(new_params, np.ndarray)
                                            and new_params.dtype == np.float32) if (set_new or set_old) else \
            getattr(self, "reference_new_old_alias", False)
        vals = np.asarray(new_params, dtype=np.float32).ravel()
        assert vals.size == self.d
        if set_new:
            self._new[:] = vals
            np.maximum(self._new[-self.m:], np.float32(self.min_log_std), out=self._new[-self.m:])
            self.log_std_val = np.float64(self._new[-self.m:].copy())
        if set_old:
            self._old[:] = vals
            np.maximum(self._old[-self.m:], np.float32(self.min_log_std), out=self._old[-self.m:])

    def old_equals_new(self):
        """True when both parameter copies and both transform sets describe the same function
        (the state at entry to every train_from_paths, SURVEY 8a note)."""
        return bool(np.array_equal(self._new, self._old) and
                    np.array_equal(self.model.packed_transforms(), self.old_model.packed_transforms()))

    # ------------------------------------------------------------------ acting (host, fork-safe)
    def _device_forward(self, x, net):
        """(N, n) CUDA tensor -> (N, m) CUDA tensor of action means through libmjx (mjx_policy_forward): what
        learned-model rollouts / evaluation sweeps call as policy.model.forward on a batch
        (model_accel/sampling.py:66-89).  The parameters are read from the NumPy store at call time."""
        import torch
        from .._lib import check, ptr
        dev = getattr(self, "_dev", None)
        if dev is None or dev["device"] != x.device:
            from ..engine import HipBackend
            be = HipBackend(self.n, self.m, self.hidden_sizes, device=x.device)
            dev = self._dev = dict(device=x.device, backend=be,
                                   theta=torch.empty(be.d, dtype=torch.float32, device=x.device),
                                   tr=torch.empty(2 * (self.n + self.m), dtype=torch.float32, device=x.device))
        be = dev["backend"]
        dev["theta"].copy_(torch.from_numpy(net._params()))
        dev["tr"].copy_(torch.from_numpy(net.packed_transforms()))
        xin = x.to(torch.float32).contiguous().reshape(-1, self.n)
        out = torch.empty((xin.shape[0], self.m), dtype=torch.float32, device=x.device)
        check(be.lib.mjx_policy_forward(be.ctx, ptr(xin), xin.shape[0], ptr(dev["theta"]), ptr(dev["tr"]), ptr(out), be.stream()))
        return out

    def get_action(self, observation):
        """gaussian_mlp.py:91-97: fp32 mean + exp(log_std) * np.random.randn(m).  NumPy only."""
        o = np.float32(observation.reshape(1, -1))
        mean = self.model.forward(o).ravel()
        noise = np.exp(self.log_std_val) * np.random.randn(self.m)
        action = mean + noise
        return [action, {'mean': mean, 'log_std': self.log_std_val, 'evaluation': mean}]

    # ------------------------------------------------------------------ torch-valued operators (reference API)
    # These return torch tensors like the reference's (gaussian_mlp.py:99-145) and are differentiable
    # w.r.t. trainable_params: they serve BC / PPO and any caller written against mjrl.  The NPG / TRPO /
    # DAPG agents of this package never call them -- their batch math runs in libmjx on the GPU.
    def mean_LL(self, observations, actions, model=None, log_std=None):
        import torch
        model = self.model if model is None else model
        log_std = self.log_std if log_std is None else log_std
        obs_var = observations if isinstance(observations, torch.Tensor) else torch.from_numpy(np.asarray(observations)).float()
        act_var = actions if isinstance(actions, torch.Tensor) else torch.from_numpy(np.asarray(actions)).float()
        mean = model(obs_var)
        zs = (act_var - mean) / torch.exp(log_std)
        LL = -0.5 * torch.sum(zs ** 2, dim=1) - torch.sum(log_std) - 0.5 * self.m * LOG_2PI
        return mean, LL

    def log_likelihood(self, observations, actions, model=None, log_std=None):
        return self.mean_LL(observations, actions, model, log_std)[1].data.numpy()

    def old_dist_info(self, observations, actions):
        mean, LL = self.mean_LL(observations, actions, self.old_model, self.old_log_std)
        return [LL, mean, self.old_log_std]

    def new_dist_info(self, observations, actions):
        mean, LL = self.mean_LL(observations, actions, self.model, self.log_std)
        return [LL, mean, self.log_std]

    def likelihood_ratio(self, new_dist_info, old_dist_info):
        import torch
        return torch.exp(new_dist_info[0] - old_dist_info[0])

    def mean_kl(self, new_dist_info, old_dist_info):
        """gaussian_mlp.py:135-145"""
        import torch
        old_std, new_std = torch.exp(old_dist_info[2]), torch.exp(new_dist_info[2])
        Nr = (old_dist_info[1] - new_dist_info[1]) ** 2 + old_std ** 2 - new_std ** 2
        Dr = 2 * new_std ** 2 + 1e-8
        return torch.mean(torch.sum(Nr / Dr + new_dist_info[2] - old_dist_info[2], dim=1))


class LinearPolicy(MLP):
    """mjrl/policies/gaussian_linear.py:9-139 == the MLP with no hidden layer."""

    def __init__(self, env_spec, min_log_std=-3, init_log_std=0, seed=None):
        super().__init__(env_spec, hidden_sizes=(), min_log_std=min_log_std, init_log_std=init_log_std, seed=seed)
