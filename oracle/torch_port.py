"""torch-CPU autograd restatement of the reference update path (same op sequence
as the reference: two forwards per surrogate, double-backward HVP), used
 (a) as the general old!=new Hessian-vector oracle, and
 (b) as bench.py's ``cpu_baseline`` (kind "port"): it executes the same torch
     CPU kernels the reference executes, so its wall time is the reference's.

Follows: mjrl/utils/fc_network.py:39-52, mjrl/policies/gaussian_mlp.py:99-145,
mjrl/algos/batch_reinforce.py:40-58, mjrl/algos/npg_cg.py:62-81,108-142,
mjrl/utils/cg_solve.py:3-22.            TEST INFRASTRUCTURE ONLY.
"""
import numpy as np
import torch

from . import npg_oracle as O


class TorchPolicy:
    def __init__(self, theta, n, m, hidden, theta_old=None, tr_new=None, tr_old=None, dtype=np.float32):
        """dtype float32: the reference's precision (it casts everything with .float()); float64: the same op
        sequence in double -- the "truth" for tolerance studies of the general (old != new) Hessian."""
        self.n, self.m, self.hidden = n, m, tuple(hidden)
        self.np_dt = np.dtype(dtype).type
        self.t_dt = torch.float64 if self.np_dt == np.float64 else torch.float32
        self.new = self._split(np.asarray(theta, self.np_dt), True)
        self.old = self._split(np.asarray(theta if theta_old is None else theta_old, self.np_dt), False)
        f = lambda tr: [torch.from_numpy(np.asarray(np.asarray(a, np.float32), self.np_dt)) for a in
                        (tr.in_shift, tr.in_scale, tr.out_shift, tr.out_scale)]
        self.tr_new = f(tr_new or O.Transforms(n, m))
        self.tr_old = f(tr_old or O.Transforms(n, m))

    def _split(self, theta, grad):
        Ws, bs, s = O.unflatten(theta, self.n, self.m, self.hidden)
        ps = []
        for W, b in zip(Ws, bs):
            ps += [torch.tensor(W), torch.tensor(b)]
        ps.append(torch.tensor(s))
        for p in ps:
            p.requires_grad_(grad)
        return ps

    def flat(self):
        return np.concatenate([p.detach().numpy().ravel() for p in self.new])

    def _net(self, ps, tr, x):
        out = (x - tr[0]) / (tr[1] + 1e-8)
        nl = (len(ps) - 1) // 2
        for i in range(nl):
            out = torch.nn.functional.linear(out, ps[2 * i], ps[2 * i + 1])
            if i < nl - 1:
                out = torch.tanh(out)
        return out * tr[3] + tr[2]

    def dist(self, obs, act, which):
        ps, tr = (self.new, self.tr_new) if which == "new" else (self.old, self.tr_old)
        x = torch.from_numpy(obs).to(self.t_dt)    # the reference re-casts fp64->fp32 on every call
        a = torch.from_numpy(act).to(self.t_dt)
        mean = self._net(ps, tr, x)
        s = ps[-1]
        z = (a - mean) / torch.exp(s)
        ll = -0.5 * torch.sum(z ** 2, dim=1) - torch.sum(s) - 0.5 * self.m * np.log(2 * np.pi)
        return ll, mean, s

    def surrogate(self, obs, act, adv):
        ll_o, _, _ = self.dist(obs, act, "old")
        ll_n, _, _ = self.dist(obs, act, "new")
        return torch.mean(torch.exp(ll_n - ll_o) * torch.from_numpy(adv).to(self.t_dt))

    def kl(self, obs, act):
        _, mo, so = self.dist(obs, act, "old")
        _, mn, sn = self.dist(obs, act, "new")
        Nr = (mo - mn) ** 2 + torch.exp(so) ** 2 - torch.exp(sn) ** 2
        Dr = 2 * torch.exp(sn) ** 2 + 1e-8
        return torch.mean(torch.sum(Nr / Dr + sn - so, dim=1))

    def vpg(self, obs, act, adv):
        g = torch.autograd.grad(self.surrogate(obs, act, adv), self.new)
        return np.concatenate([x.contiguous().view(-1).numpy() for x in g])

    def hvp(self, obs, act, v, damping):
        vec = torch.from_numpy(np.asarray(v)).to(self.t_dt)
        g = torch.autograd.grad(self.kl(obs, act), self.new, create_graph=True)
        flat = torch.cat([x.contiguous().view(-1) for x in g])
        h = torch.autograd.grad(torch.sum(flat * vec), self.new)
        return np.concatenate([x.contiguous().view(-1).numpy() for x in h]) + damping * v

    def set_new(self, theta):
        for p, q in zip(self.new, self._split(np.asarray(theta, self.np_dt), True)):
            p.data = q.data


def npg_update(theta, obs, act, adv, n, m, hidden, cg_iters=10, damping=1e-4, delta=0.05, min_log_std=-3.0):
    """One NPG.train_from_paths worth of torch-CPU work (npg_cg.py:108-142)."""
    pol = TorchPolicy(theta, n, m, hidden)
    surr_before = float(pol.surrogate(obs, act, adv))
    g = pol.vpg(obs, act, adv)
    x = O.cg_solve(lambda p: pol.hvp(obs, act, p, damping), g, cg_iters)
    alpha = np.sqrt(np.abs(delta / (np.dot(g, x) + 1e-20)))
    new = (np.asarray(theta, np.float32) + alpha * x).astype(np.float32)
    new[-m:] = np.maximum(new[-m:], min_log_std)
    pol.set_new(new)
    surr_after = float(pol.surrogate(obs, act, adv))
    kl = float(pol.kl(obs, act))
    return dict(vpg=g, npg=x, alpha=float(alpha), new_params=new, surr_before=surr_before,
                surr_after=surr_after, kl=kl)
