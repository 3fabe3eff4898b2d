"""Oracle-backed stand-in for mjrl_amd.engine.HipBackend (TEST INFRASTRUCTURE).

It lets the multi-rank orchestration of UpdateEngine -- trajectory sharding, the placement of the
all-reduces, global-N normalisation, redundant CG scalar updates -- run on CPU tensors under the
gloo backend, with the per-rank arithmetic done by oracle/npg_oracle.py.  It is never shipped:
nothing under mjrl_amd/ imports it."""
import numpy as np
import torch

from oracle import npg_oracle as O


class OracleBackend:
    torch = torch
    device = torch.device("cpu")
    fused = False
    lib = ctx = None

    def __init__(self, n, m, hidden):
        self.n, self.m, self.hidden = n, m, tuple(hidden)
        self.d = O.num_params(n, m, hidden)

    def close(self):
        pass

    def stream(self):
        return None

    def upload_f32(self, a):
        return torch.as_tensor(np.asarray(a)).to(torch.float32).contiguous()

    def _tr(self, t):
        a = t.numpy().astype(np.float64)
        n, m = self.n, self.m
        return O.Transforms(n, m, a[:n], a[n:2 * n], a[2 * n:2 * n + m], a[2 * n + m:])

    def bind_policy(self, theta_new, theta_old, tr_new, tr_old, old_is_new):
        self.tn, self.to, self.trn, self.tro, self.same = theta_new, theta_old, tr_new, tr_old, old_is_new

    def bind_rows(self, rows, N_global, adv=None):
        obs, act, adv0 = self._full
        self.bind_batch(obs, act, adv0 if adv is None else adv, rows, N_global, keep=True)

    def bind_batch(self, obs, act, adv, rows, N_global, keep=False):
        if not keep:
            self._full = (obs, act, adv)
        self.obs = obs[:rows].numpy().astype(np.float64)
        self.act = None if act is None else act[:rows].numpy().astype(np.float64)
        self.adv = None if adv is None else adv[:rows].numpy().astype(np.float64)
        self.Ng = N_global

    def _args(self):
        return self.n, self.m, self.hidden

    def surr_vpg(self, grad_out, scal_out):
        tn, to = self.tn.numpy().astype(np.float64), self.to.numpy().astype(np.float64)
        Nl = self.obs.shape[0]
        if Nl == 0:
            grad_out.zero_(); scal_out.zero_(); return
        g = O.vpg(tn, to, self.obs, self.act, self.adv, *self._args(), self._tr(self.trn), self._tr(self.tro)) * (Nl / self.Ng)
        s = O.surrogate(tn, to, self.obs, self.act, self.adv, *self._args(), self._tr(self.trn), self._tr(self.tro)) * Nl
        grad_out.copy_(torch.from_numpy(g.astype(np.float32)))
        scal_out.copy_(torch.tensor([s, float(Nl), float(Nl), 0.0], dtype=torch.float64))

    def fvp(self, v, out):
        Nl = self.obs.shape[0]
        h = O.fvp(self.tn.numpy().astype(np.float64), self.obs, v.numpy().astype(np.float64), *self._args(), self._tr(self.trn)) * (Nl / self.Ng)
        out.copy_(torch.from_numpy(h.astype(np.float32)))

    def eval_surr_kl(self, scal_out):
        tn, to = self.tn.numpy().astype(np.float64), self.to.numpy().astype(np.float64)
        Nl = self.obs.shape[0]
        s = O.surrogate(tn, to, self.obs, self.act, self.adv, *self._args(), self._tr(self.trn), self._tr(self.tro)) * Nl
        k = O.mean_kl(tn, to, self.obs, *self._args(), self._tr(self.trn), self._tr(self.tro)) * Nl
        scal_out.copy_(torch.tensor([s, k, float(Nl), 0.0], dtype=torch.float64))

    # CG bookkeeping (cg_solve.py:3-22), vectors fp32 like the device kernels
    def cg_init(self, b):
        self.cx = torch.zeros_like(b); self.cr = b.clone(); self.cp = b.clone()
        self.rr = float(torch.dot(b.double(), b.double())); self.done = False

    def fvp_of_cg_direction(self, out):
        self.fvp(self.cp, out)

    def cg_step(self, Ap, damping, tol):
        if self.done:
            return
        z = Ap + np.float32(damping) * self.cp
        a = np.float32(self.rr / float(torch.dot(self.cp.double(), z.double())))
        self.cx += a * self.cp
        self.cr -= a * z
        nrr = float(torch.dot(self.cr.double(), self.cr.double()))
        self.cp = self.cr + np.float32(nrr / self.rr) * self.cp
        self.rr = nrr
        self.done = nrr < tol

    def cg_finish(self, b, x_out, bdotx_out):
        x_out.copy_(self.cx)
        bdotx_out[0] = float(torch.dot(b.double(), self.cx.double()))

    def cg_solve_local(self, b, iters, damping, tol, x_out, bdotx_out):
        self.cg_init(b)
        tmp = torch.zeros_like(b)
        for _ in range(iters):
            self.fvp_of_cg_direction(tmp)
            self.cg_step(tmp, damping, tol)
        self.cg_finish(b, x_out, bdotx_out)

    def apply_npg_step(self, base, x, gdotx, step_size, min_log_std, out, alpha_out):
        alpha = np.sqrt(np.abs(float(step_size) / (float(gdotx[0]) + 1e-20)))
        alpha_out[0] = alpha
        self.apply_step(base, x, alpha, min_log_std, out)

    def apply_step(self, base, x, alpha, min_log_std, out):
        v = base + np.float32(alpha) * x
        v[-self.m:] = torch.clamp(v[-self.m:], min=float(min_log_std))
        out.copy_(v)
