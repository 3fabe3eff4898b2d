"""Diagonal-Gaussian tanh-MLP policy with mjrl's operator surface.

Mirrors ``mjrl.policies.gaussian_mlp.MLP`` (reference mjrl/policies/gaussian_mlp.py:7-145):
same constructor, same flat parameter order, same ``get_action`` RNG stream, same
``set_param_values`` clamping.  Differences by design:

* state is plain NumPy (flat fp32 vectors for the new and the old parameter copy plus the
  affine transforms), so the object pickles / deep-copies / forks freely
  (mjrl/utils/train_agent.py:83,102,129-131; mjrl/samplers/core.py:196) and ``get_action``
  never touches HIP;
* the batch operators (likelihoods, surrogate, gradient, Fisher-vector products) are not
  evaluated here -- agents drive them on the GPU through ``mjrl_amd.engine``.  A torch-CPU
  mirror (``model`` / ``trainable_params`` / ``new_dist_info`` ...) exists for the callers
  that optimise the policy with torch optimisers (behavior_cloning.py:42, ppo_clip.py:46).
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

    # fc_network.py:39-52 (fp32, NumPy)
    def forward(self, x):
        is_torch = hasattr(x, "detach")
        xin = np.asarray(x.detach().cpu().numpy() if is_torch else x, np.float32)
        out = (xin - np.asarray(self.in_shift, np.float32)) / (np.asarray(self.in_scale, np.float32) + np.float32(1e-8))
        Ws, bs = self._p._unflatten(self._params())
        for W, b in zip(Ws[:-1], bs[:-1]):
            out = np.tanh(out @ W.T + b)
        out = (out @ Ws[-1].T + bs[-1]) * np.asarray(self.out_scale, np.float32) + np.asarray(self.out_shift, np.float32)
        if is_torch:
            import torch
            return torch.from_numpy(np.ascontiguousarray(out))
        return out

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

        self._new = np.concatenate(flat).astype(np.float32)
        self._old = self._new.copy()
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

    @property
    def log_std(self):
        return self._new[-self.m:]

    @property
    def old_log_std(self):
        return self._old[-self.m:]

    def get_param_values(self):
        return self._new.copy()

    def get_old_param_values(self):
        return self._old.copy()

    def set_param_values(self, new_params, set_new=True, set_old=True):
        """gaussian_mlp.py:65-87 (float32 cast, log_std clamped at min_log_std)."""
        vals = np.asarray(new_params, dtype=np.float32).ravel()
        assert vals.size == self.d
        if set_new:
            self._new = vals.copy()
            self._new[-self.m:] = np.maximum(self._new[-self.m:], np.float32(self.min_log_std))
            self.log_std_val = np.float64(self._new[-self.m:].copy())
        if set_old:
            self._old = vals.copy()
            self._old[-self.m:] = np.maximum(self._old[-self.m:], np.float32(self.min_log_std))

    def old_equals_new(self):
        """True when both parameter copies and both transform sets describe the same function
        (the state at entry to every train_from_paths, SURVEY 8a note)."""
        return bool(np.array_equal(self._new, self._old) and
                    np.array_equal(self.model.packed_transforms(), self.old_model.packed_transforms()))

    # ------------------------------------------------------------------ acting (host, fork-safe)
    def get_action(self, observation):
        """gaussian_mlp.py:91-97: fp32 mean + exp(log_std) * np.random.randn(m)."""
        o = np.float32(observation.reshape(1, -1))
        mean = self.model.forward(o).ravel()
        noise = np.exp(self.log_std_val) * np.random.randn(self.m)
        action = mean + noise
        return [action, {'mean': mean, 'log_std': self.log_std_val, 'evaluation': mean}]

    # ------------------------------------------------------------------ small-batch host operators
    def mean_LL(self, observations, actions, model=None, log_std=None):
        """gaussian_mlp.py:99-115 on the host (NumPy fp32) -- for small batches / tests;
        the training path evaluates these on the GPU."""
        model = self.model if model is None else model
        log_std = self.log_std if log_std is None else log_std
        mean = model.forward(np.asarray(observations, np.float32))
        ls = np.asarray(log_std, np.float32)
        zs = (np.asarray(actions, np.float32) - mean) / np.exp(ls)
        LL = -0.5 * np.sum(zs ** 2, axis=1) - np.sum(ls) - np.float32(0.5 * self.m * LOG_2PI)
        return mean, LL

    def log_likelihood(self, observations, actions, model=None, log_std=None):
        return self.mean_LL(observations, actions, model, log_std)[1]

    def old_dist_info(self, observations, actions):
        mean, LL = self.mean_LL(observations, actions, self.old_model, self.old_log_std)
        return [LL, mean, self.old_log_std]

    def new_dist_info(self, observations, actions):
        mean, LL = self.mean_LL(observations, actions, self.model, self.log_std)
        return [LL, mean, self.log_std]

    def likelihood_ratio(self, new_dist_info, old_dist_info):
        return np.exp(new_dist_info[0] - old_dist_info[0])

    def mean_kl(self, new_dist_info, old_dist_info):
        """gaussian_mlp.py:135-145"""
        old_std, new_std = np.exp(old_dist_info[2]), np.exp(new_dist_info[2])
        Nr = (old_dist_info[1] - new_dist_info[1]) ** 2 + old_std ** 2 - new_std ** 2
        Dr = 2 * new_std ** 2 + 1e-8
        return np.mean(np.sum(Nr / Dr + new_dist_info[2] - old_dist_info[2], axis=1))


class LinearPolicy(MLP):
    """mjrl/policies/gaussian_linear.py:9-139 == the MLP with no hidden layer."""

    def __init__(self, env_spec, min_log_std=-3, init_log_std=0, seed=None):
        super().__init__(env_spec, hidden_sizes=(), min_log_std=min_log_std, init_log_std=init_log_std, seed=seed)
