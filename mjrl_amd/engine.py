"""Device-side state of the policy-update hot path.

``UpdateEngine`` owns one ``mjx_ctx`` (include/mjx.h) plus the torch tensors whose device
pointers the C ABI reads and writes.  torch is plumbing only: allocation, the current HIP
stream and ``torch.distributed`` (backend "nccl" == RCCL) for the one all-reduce per
gradient / per CG iteration.  No torch.autograd, no torch math on the hot path.

Multi-GPU: one process per GPU; every rank binds its own trajectory shard and the
quantities that are sums over samples (gradient, Fisher-vector product, surrogate / KL
sums) are all-reduced; CG scalars are recomputed redundantly on every rank from the reduced
vectors, so ranks stay in lock-step without further communication.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, ptr


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


class UpdateEngine:
    def __init__(self, n, m, hidden_sizes, device=None):
        import torch
        self.torch = torch
        self.lib = _lib.load()
        if self.lib.mjx_device_count() < 1 or not torch.cuda.is_available():
            raise _lib.MjxError("mjrl_amd needs an AMD GPU (gfx950): no HIP device visible and there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.n, self.m, self.hidden = int(n), int(m), tuple(int(h) for h in hidden_sizes)
        ctx = ctypes.c_void_p()
        harr = (ctypes.c_int * max(1, len(self.hidden)))(*self.hidden)
        check(self.lib.mjx_create(ctypes.byref(ctx), self.device.index or 0, self.n, self.m, harr, len(self.hidden)))
        self.ctx = ctx
        self.d = int(self.lib.mjx_num_params(ctx))
        self.fused = bool(self.lib.mjx_uses_fused_path(ctx))
        f32 = dict(dtype=torch.float32, device=self.device)
        self.theta_new = torch.zeros(self.d, **f32)
        self.theta_old = torch.zeros(self.d, **f32)
        self.tr_new = torch.zeros(2 * self.n + 2 * self.m, **f32)
        self.tr_old = torch.zeros(2 * self.n + 2 * self.m, **f32)
        self.grad = torch.zeros(self.d, **f32)
        self.x = torch.zeros(self.d, **f32)
        self.Ap = torch.zeros(self.d, **f32)
        self.scal = torch.zeros(4, dtype=torch.float64, device=self.device)
        self.bdotx = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.obs = self.act = self.adv = None
        self.N_local = self.N_global = 0
        self.old_is_new = True
        self._dbg = None

    def close(self):
        if getattr(self, "ctx", None) is not None:
            self.torch.cuda.synchronize(self.device)
            self.lib.mjx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def to_device_f32(self, a):
        """host ndarray (fp64 or fp32) -> fp32 device tensor.  fp64 input is cast on the device
        (one PCIe pass of the raw fp64 rollouts; the reference re-casts on the CPU at every
        call, gaussian_mlp.py:102-109)."""
        torch = self.torch
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.float32).contiguous()
        a = np.ascontiguousarray(a)
        t = torch.from_numpy(a).to(self.device, non_blocking=False)
        if t.dtype == torch.float64:
            out = torch.empty(t.shape, dtype=torch.float32, device=self.device)
            check(self.lib.mjx_cast_f64_f32(ptr(t), t.numel(), ptr(out), self.stream()))
            return out
        return t.to(torch.float32)

    # ------------------------------------------------------------------ binding
    def set_policy(self, theta_new, theta_old, tr_new, tr_old):
        """flat fp32 parameter vectors + packed transforms (host ndarrays)."""
        torch = self.torch
        tn, to = np.asarray(theta_new, np.float32), np.asarray(theta_old, np.float32)
        trn, tro = np.asarray(tr_new, np.float32), np.asarray(tr_old, np.float32)
        self.theta_new.copy_(torch.from_numpy(tn))
        self.theta_old.copy_(torch.from_numpy(to))
        self.tr_new.copy_(torch.from_numpy(trn))
        self.tr_old.copy_(torch.from_numpy(tro))
        self.old_is_new = bool(np.array_equal(tn, to) and np.array_equal(trn, tro))
        self._bind_policy()

    def _bind_policy(self):
        check(self.lib.mjx_bind_policy(self.ctx, ptr(self.theta_new), ptr(self.theta_old), ptr(self.tr_new),
                                       ptr(self.tr_old), int(self.old_is_new)))

    def set_batch(self, obs, act=None, adv=None, N_global=None):
        """obs (N,n), act (N,m), adv (N,) of THIS rank's shard; host arrays or device tensors."""
        self.obs = self.to_device_f32(obs)
        self.act = None if act is None else self.to_device_f32(act)
        self.adv = None if adv is None else self.to_device_f32(adv)
        self.N_local = int(self.obs.shape[0])
        if N_global is None:
            N_global = self.N_local
            d = _dist()
            if d is not None:
                t = self.torch.tensor([float(self.N_local)], dtype=self.torch.float64, device=self.device)
                d.all_reduce(t)
                N_global = int(t.item())
        self.N_global = int(N_global)
        self.bind_rows(self.N_local)

    def bind_rows(self, rows, adv=None):
        """(re)bind the first `rows` samples of the uploaded block (DAPG runs the Fisher on the
        on-policy prefix of the demo-augmented block, dapg.py:97-103)."""
        if adv is not None:
            self.adv = self.to_device_f32(adv)
        check(self.lib.mjx_bind_batch(self.ctx, ptr(self.obs), ptr(self.act), ptr(self.adv), int(rows), int(self.N_global)))

    def set_N_global(self, N_global, rows=None):
        self.N_global = int(N_global)
        self.bind_rows(self.N_local if rows is None else rows)

    # ------------------------------------------------------------------ kernels
    def surr_vpg(self):
        """K1 -> (grad device tensor, surrogate float).  flat_vpg + CPI_surrogate
        (batch_reinforce.py:40-58)."""
        check(self.lib.mjx_surr_vpg(self.ctx, ptr(self.grad), ptr(self.scal), self.stream()))
        d = _dist()
        if d is not None:
            d.all_reduce(self.grad)
            d.all_reduce(self.scal)
        s = self.scal.cpu().numpy()
        return self.grad, float(s[0] / self.N_global)

    def fvp(self, v, out=None):
        """K2: (H v) without damping, reduced over ranks (npg_cg.py:62-81)."""
        out = self.Ap if out is None else out
        check(self.lib.mjx_fvp(self.ctx, ptr(v), ptr(out), self.stream()))
        d = _dist()
        if d is not None:
            d.all_reduce(out)
        return out

    def cg_solve(self, b, iters, damping, tol=1e-10):
        """K4: x = CG(H + damping I, b), x0 = 0 (cg_solve.py:3-22) -> (x device tensor, b.x)."""
        d = _dist()
        st = self.stream()
        if d is None:
            check(self.lib.mjx_cg_solve(self.ctx, ptr(b), int(iters), float(damping), float(tol), ptr(self.x),
                                        ptr(self.bdotx), None, None, st))
        else:
            check(self.lib.mjx_cg_init(self.ctx, ptr(b), st))
            p = ctypes.c_void_p(self.lib.mjx_cg_p(self.ctx))
            for _ in range(int(iters)):
                check(self.lib.mjx_fvp(self.ctx, p, ptr(self.Ap), st))
                d.all_reduce(self.Ap)
                check(self.lib.mjx_cg_step(self.ctx, ptr(self.Ap), float(damping), float(tol), st))
            check(self.lib.mjx_cg_finish(self.ctx, ptr(b), ptr(self.x), ptr(self.bdotx), st))
        return self.x, float(self.bdotx.item())

    def apply_step(self, alpha, min_log_std, base=None):
        """theta_new <- base + alpha * x with the log_std clamp (npg_cg.py:137-139)."""
        base = self.theta_old if base is None else base
        check(self.lib.mjx_apply_step(self.ctx, ptr(base), ptr(self.x), float(alpha), float(min_log_std),
                                      ptr(self.theta_new), self.stream()))
        self.old_is_new = False
        self._bind_policy()

    def eval_surr_kl(self):
        """K3 -> (surrogate, mean KL) (batch_reinforce.py:40-52)."""
        check(self.lib.mjx_eval_surr_kl(self.ctx, ptr(self.scal), self.stream()))
        d = _dist()
        if d is not None:
            d.all_reduce(self.scal)
        s = self.scal.cpu().numpy()
        return float(s[0] / self.N_global), float(s[1] / self.N_global)

    def enable_debug(self):
        self._dbg = self.torch.zeros(2048 * 8, dtype=self.torch.float32, device=self.device)
        check(self.lib.mjx_set_debug_buffer(self.ctx, ptr(self._dbg), self._dbg.numel()))
        return self._dbg
