"""NumPy restatement (analytic, no autograd) of the reference update path.

Every function cites the reference lines (relative to /root/reference) whose
arithmetic it restates.  ``dtype`` selects the working precision: float64 is
the "truth" used for tolerance studies, float32 mirrors the reference's
precision (the reference casts everything to fp32 torch tensors).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
import numpy as np

LOG_2PI = np.log(2.0 * np.pi)


# ---------------------------------------------------------------------------
# flat parameter layout  (mjrl/policies/gaussian_mlp.py:37,50-52,60-63;
# torch Linear.weight is (out,in) row-major)
# ---------------------------------------------------------------------------
def layer_sizes(n, m, hidden):
    return (n,) + tuple(hidden) + (m,)


def num_params(n, m, hidden):
    ls = layer_sizes(n, m, hidden)
    return sum(ls[i] * ls[i + 1] + ls[i + 1] for i in range(len(ls) - 1)) + m


def unflatten(theta, n, m, hidden):
    """-> (Ws, bs, log_std) views into theta."""
    ls = layer_sizes(n, m, hidden)
    Ws, bs, k = [], [], 0
    for i in range(len(ls) - 1):
        Ws.append(theta[k:k + ls[i] * ls[i + 1]].reshape(ls[i + 1], ls[i])); k += ls[i] * ls[i + 1]
        bs.append(theta[k:k + ls[i + 1]]); k += ls[i + 1]
    log_std = theta[k:k + m]
    assert k + m == theta.size
    return Ws, bs, log_std


def flatten(Ws, bs, log_std):
    out = []
    for W, b in zip(Ws, bs):
        out += [W.ravel(), b.ravel()]
    out.append(np.ravel(log_std))
    return np.concatenate(out)


class Transforms:
    """in/out affine transforms of FCNetwork (mjrl/utils/fc_network.py:27-37)."""

    def __init__(self, n, m, in_shift=None, in_scale=None, out_shift=None, out_scale=None, dtype=np.float64):
        f = lambda v, d, k: (np.full(k, d, dtype) if v is None else np.asarray(v, np.float32).astype(dtype))
        self.in_shift, self.in_scale = f(in_shift, 0.0, n), f(in_scale, 1.0, n)
        self.out_shift, self.out_scale = f(out_shift, 0.0, m), f(out_scale, 1.0, m)


# ---------------------------------------------------------------------------
# forward   (mjrl/utils/fc_network.py:39-52)
# ---------------------------------------------------------------------------
def forward(theta, obs, n, m, hidden, tr=None, keep=False):
    dt = theta.dtype
    tr = tr or Transforms(n, m, dtype=dt)
    Ws, bs, _ = unflatten(theta, n, m, hidden)
    x = (obs.astype(dt) - tr.in_shift.astype(dt)) / (tr.in_scale.astype(dt) + dt.type(1e-8))
    acts = [x]
    for W, b in zip(Ws[:-1], bs[:-1]):
        acts.append(np.tanh(acts[-1] @ W.T + b))
    mu = (acts[-1] @ Ws[-1].T + bs[-1]) * tr.out_scale.astype(dt) + tr.out_shift.astype(dt)
    return (mu, acts) if keep else mu


# mean_LL  (mjrl/policies/gaussian_mlp.py:99-115)
def log_likelihood(theta, obs, act, n, m, hidden, tr=None):
    dt = theta.dtype
    mu = forward(theta, obs, n, m, hidden, tr)
    log_std = unflatten(theta, n, m, hidden)[2]
    z = (act.astype(dt) - mu) / np.exp(log_std)
    return -0.5 * np.sum(z ** 2, axis=1) - np.sum(log_std) - 0.5 * m * dt.type(LOG_2PI), mu


# CPI surrogate  (mjrl/algos/batch_reinforce.py:40-46, gaussian_mlp.py:129-133)
def surrogate(theta_new, theta_old, obs, act, adv, n, m, hidden, tr_new=None, tr_old=None):
    ll_n, _ = log_likelihood(theta_new, obs, act, n, m, hidden, tr_new)
    ll_o, _ = log_likelihood(theta_old, obs, act, n, m, hidden, tr_old)
    return np.mean(np.exp(ll_n - ll_o) * adv.astype(theta_new.dtype))


# mean_kl  (mjrl/policies/gaussian_mlp.py:135-145) -- argument order (new, old)
def mean_kl(theta_new, theta_old, obs, n, m, hidden, tr_new=None, tr_old=None):
    dt = theta_new.dtype
    mu_n = forward(theta_new, obs, n, m, hidden, tr_new)
    mu_o = forward(theta_old, obs, n, m, hidden, tr_old)
    s_n = unflatten(theta_new, n, m, hidden)[2]
    s_o = unflatten(theta_old, n, m, hidden)[2]
    Nr = (mu_o - mu_n) ** 2 + np.exp(s_o) ** 2 - np.exp(s_n) ** 2
    Dr = 2 * np.exp(s_n) ** 2 + dt.type(1e-8)
    return np.mean(np.sum(Nr / Dr + s_n - s_o, axis=1))


# ---------------------------------------------------------------------------
# backprop of an output-space cotangent through the tanh MLP
# ---------------------------------------------------------------------------
def _backprop(Ws, acts, dmu, out_scale):
    """dmu: (N,m) cotangent on mu (already including 1/N); returns list of (gW, gb)."""
    grads = []
    delta = dmu * out_scale
    for l in range(len(Ws) - 1, -1, -1):
        grads.append((delta.T @ acts[l], delta.sum(axis=0)))
        if l > 0:
            delta = (delta @ Ws[l]) * (1.0 - acts[l] ** 2)
    return grads[::-1]


# flat_vpg  (mjrl/algos/batch_reinforce.py:54-58): gradient of mean(LR*adv) wrt theta_new
def vpg(theta_new, theta_old, obs, act, adv, n, m, hidden, tr_new=None, tr_old=None):
    dt = theta_new.dtype
    tr_new = tr_new or Transforms(n, m, dtype=dt)
    N = obs.shape[0]
    Ws, bs, s = unflatten(theta_new, n, m, hidden)
    mu, acts = forward(theta_new, obs, n, m, hidden, tr_new, keep=True)
    sig = np.exp(s)
    z = (act.astype(dt) - mu) / sig
    ll_n = -0.5 * np.sum(z ** 2, axis=1) - np.sum(s) - 0.5 * m * dt.type(LOG_2PI)
    ll_o, _ = log_likelihood(theta_old, obs, act, n, m, hidden, tr_old)
    w = np.exp(ll_n - ll_o) * adv.astype(dt) / N          # d surr / d LL_i
    dmu = w[:, None] * z / sig                             # dLL/dmu = (a-mu)/sigma^2
    g_s = (w[:, None] * (z ** 2 - 1.0)).sum(axis=0)       # dLL/ds  = z^2 - 1
    grads = _backprop(Ws, acts, dmu, tr_new.out_scale.astype(dt))
    return flatten([g[0] for g in grads], [g[1] for g in grads], g_s)


# NPG.HVP at theta_new == theta_old  (mjrl/algos/npg_cg.py:62-81):
# Hessian of mean_kl(new, old) wrt new = Gauss-Newton term J^T D J plus a
# log_std diagonal; the 1e-8 in Dr is kept.
def fvp(theta, obs, v, n, m, hidden, tr=None, damping=0.0):
    dt = theta.dtype
    tr = tr or Transforms(n, m, dtype=dt)
    N = obs.shape[0]
    Ws, bs, s = unflatten(theta, n, m, hidden)
    Vs, cs, vs = unflatten(v.astype(dt), n, m, hidden)
    mu, acts = forward(theta, obs, n, m, hidden, tr, keep=True)
    osc = tr.out_scale.astype(dt)
    # tangent (R-op) pass
    t = np.zeros_like(acts[0])
    for l in range(len(Ws) - 1):
        t = (acts[l] @ Vs[l].T + t @ Ws[l].T + cs[l]) * (1.0 - acts[l + 1] ** 2)
    mu_dot = (acts[-1] @ Vs[-1].T + t @ Ws[-1].T + cs[-1]) * osc
    u = np.exp(s) ** 2
    eps = dt.type(1e-8)
    D = 2.0 / (2.0 * u + eps)
    grads = _backprop(Ws, acts, D * mu_dot / N, osc)
    c = 16.0 * u * u / (2.0 * u + eps) ** 2 - 4.0 * u / (2.0 * u + eps)
    out = flatten([g[0] for g in grads], [g[1] for g in grads], c * vs)
    return out + dt.type(damping) * v.astype(dt)


# cg_solve  (mjrl/utils/cg_solve.py:3-22): x0 = 0 regardless of the x_0 argument
def cg_solve(f_Ax, b, cg_iters=10, residual_tol=1e-10):
    x = np.zeros_like(b)
    r = b.copy()
    p = r.copy()
    rdotr = r.dot(r)
    for _ in range(cg_iters):
        z = f_Ax(p)
        a = rdotr / p.dot(z)
        x += a * p
        r -= a * z
        new = r.dot(r)
        p = r + (new / rdotr) * p
        rdotr = new
        if rdotr < residual_tol:
            break
    return x


# NPG.train_from_paths core  (mjrl/algos/npg_cg.py:108-142)
def npg_update(theta, obs, act, adv, n, m, hidden, tr=None, cg_iters=10, damping=1e-4,
               delta=0.05, const_alpha=None, min_log_std=-3.0):
    """theta_new == theta_old == theta at entry.  adv is already whitened."""
    dt = theta.dtype
    out = {}
    out["surr_before"] = surrogate(theta, theta, obs, act, adv, n, m, hidden, tr, tr)
    g = vpg(theta, theta, obs, act, adv, n, m, hidden, tr, tr)
    x = cg_solve(lambda p: fvp(theta, obs, p, n, m, hidden, tr, damping), g, cg_iters)
    if const_alpha is not None:
        alpha = const_alpha
    else:
        alpha = np.sqrt(np.abs(delta / (np.dot(g, x) + 1e-20)))
    new = theta + dt.type(alpha) * x
    new[-m:] = np.maximum(new[-m:], dt.type(min_log_std))      # gaussian_mlp.py:73-75
    out.update(vpg=g, npg=x, alpha=float(alpha), new_params=new)
    out["surr_after"] = surrogate(new, theta, obs, act, adv, n, m, hidden, tr, tr)
    out["kl"] = mean_kl(new, theta, obs, n, m, hidden, tr, tr)
    return out


# DAPG.train_from_paths core  (mjrl/algos/dapg.py:58-121): gradient over [on-policy ; demonstrations] with the demo
# "advantages" lam_0 * lam_1^iter and the 1e-2 / std rescaling (:62-70), rescaled by N_all / N (:97-98); Fisher,
# surrogate and KL on the on-policy block only (:103-121); delta = 2 kl_dist (:111)
def dapg_update(theta, obs, act, adv, demo_obs, demo_act, n, m, hidden, tr=None, cg_iters=10, damping=1e-4,
                kl_dist=0.025, lam_0=1.0, lam_1=0.95, iter_count=0.0, min_log_std=-3.0):
    """adv: whitened on-policy advantages (dapg.py:61)."""
    dt = theta.dtype
    demo_adv = lam_0 * (lam_1 ** iter_count) * np.ones(demo_obs.shape[0])
    all_obs, all_act = np.concatenate([obs, demo_obs]), np.concatenate([act, demo_act])
    all_adv = 1e-2 * np.concatenate([adv / (np.std(adv) + 1e-8), demo_adv])
    coef = all_adv.shape[0] / adv.shape[0]
    g = dt.type(coef) * vpg(theta, theta, all_obs, all_act, all_adv, n, m, hidden, tr, tr)
    x = cg_solve(lambda p: fvp(theta, obs, p, n, m, hidden, tr, damping), g, cg_iters)
    alpha = np.sqrt(np.abs(2.0 * kl_dist / (np.dot(g, x) + 1e-20)))
    new = theta + dt.type(alpha) * x
    new[-m:] = np.maximum(new[-m:], dt.type(min_log_std))
    return dict(vpg=g, npg=x, alpha=float(alpha), new_params=new,
                surr_before=surrogate(theta, theta, obs, act, adv, n, m, hidden, tr, tr),
                surr_after=surrogate(new, theta, obs, act, adv, n, m, hidden, tr, tr),
                kl=mean_kl(new, theta, obs, n, m, hidden, tr, tr))


# TRPO line search  (mjrl/algos/trpo.py:100-126)
def trpo_update(theta, obs, act, adv, n, m, hidden, tr=None, cg_iters=10, damping=1e-4,
                kl_dist=0.01, min_log_std=-3.0):
    dt = theta.dtype
    g = vpg(theta, theta, obs, act, adv, n, m, hidden, tr, tr)
    x = cg_solve(lambda p: fvp(theta, obs, p, n, m, hidden, tr, damping), g, cg_iters)
    alpha = np.sqrt(np.abs(2.0 * kl_dist / (np.dot(g, x) + 1e-20)))
    tries = 0
    for k in range(100):
        new = theta + dt.type(alpha) * x
        new[-m:] = np.maximum(new[-m:], dt.type(min_log_std))
        kl = mean_kl(new, theta, obs, n, m, hidden, tr, tr)
        tries += 1
        if kl < kl_dist:
            break
        alpha = 0.9 * alpha
        if k == 99:
            alpha = 0.0
    new = theta + dt.type(alpha) * x
    new[-m:] = np.maximum(new[-m:], dt.type(min_log_std))
    return dict(vpg=g, npg=x, alpha=float(alpha), new_params=new, tries=tries,
                kl=mean_kl(new, theta, obs, n, m, hidden, tr, tr),
                surr_after=surrogate(new, theta, obs, act, adv, n, m, hidden, tr, tr))


# ---------------------------------------------------------------------------
# sample processing  (mjrl/utils/process_samples.py)
# ---------------------------------------------------------------------------
def discount_sum(x, gamma, terminal=0.0):
    """process_samples.py:37-44 -- sequential reverse recurrence, fp64."""
    y = np.empty(len(x), dtype=np.float64)
    run = terminal
    for t in range(len(x) - 1, -1, -1):
        run = x[t] + gamma * run
        y[t] = run
    return y


def gae_path(rewards, baseline, terminated, gamma, lam):
    """process_samples.py:21-29 (1-D baseline branch)."""
    b1 = np.append(baseline, 0.0 if terminated else baseline[-1])
    td = rewards + gamma * b1[1:] - b1[:-1]
    return discount_sum(td, gamma * lam)


def whiten(adv):
    """batch_reinforce.py:185"""
    return (adv - np.mean(adv)) / (np.std(adv) + 1e-6)


# ---------------------------------------------------------------------------
# baselines  (mjrl/baselines/*.py)
# ---------------------------------------------------------------------------
def time_features(lengths):
    cols = []
    for l in lengths:
        al = np.arange(l) / 1000.0
        cols.append(np.stack([al ** (j + 1) for j in range(4)], axis=1))
    return np.concatenate(cols)


def mlp_baseline_features(obs_list):
    """mlp_baseline.py:36-58 -> (N, n+4)"""
    o = np.clip(np.concatenate(obs_list), -10, 10) / 10.0
    return np.concatenate([o, time_features([len(x) for x in obs_list])], axis=1)


def linear_baseline_features(obs_list):
    """linear_baseline.py:11-35 -> (N, n+5): obs, 1, t..t^4"""
    o = np.clip(np.concatenate(obs_list), -10, 10) / 10.0
    return np.concatenate([o, np.ones((o.shape[0], 1)), time_features([len(x) for x in obs_list])], axis=1)


def quadratic_baseline_features(obs_list):
    """quadratic_baseline.py:11-41 -> (N, n + n(n+1)/2 + 5)"""
    o = np.clip(np.concatenate(obs_list), -10, 10) / 10.0
    n = o.shape[1]
    iu = np.triu_indices(n)
    quad = o[:, iu[0]] * o[:, iu[1]]
    return np.concatenate([o, quad, np.ones((o.shape[0], 1)), time_features([len(x) for x in obs_list])], axis=1)


def ridge_fit(F, y, reg):
    """quadratic_baseline.py:54-63 / linear_baseline.py:45-54"""
    for _ in range(10):
        coef = np.linalg.lstsq(F.T.dot(F) + reg * np.identity(F.shape[1]), F.T.dot(y), rcond=-1)[0]
        if not np.any(np.isnan(coef)):
            break
        reg *= 10
    return coef


def mlp_baseline_forward(Ws, bs, feat):
    """mlp_baseline.py:21-28: ReLU MLP (n+4)->h->h->1"""
    a = feat
    for W, b in zip(Ws[:-1], bs[:-1]):
        a = np.maximum(a @ W.T + b, 0.0)
    return (a @ Ws[-1].T + bs[-1]).ravel()
