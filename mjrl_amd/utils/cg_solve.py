"""Conjugate gradient with mjrl's ``cg_solve(f_Ax, b, x_0, cg_iters, residual_tol)`` signature
(reference mjrl/utils/cg_solve.py:3-22).

Two back ends:
* ``f_Ax`` is a :class:`DeviceFisher` (what the agents in this package build): the whole
  solve runs on the GPU inside ``mjx_cg_solve`` -- Fisher-vector products, vector updates
  and dot products never leave the device and there is no per-iteration host sync;
* ``f_Ax`` is any other callable on host vectors: a plain host loop with the reference's
  semantics (x0 = 0 whatever ``x_0`` says, stop once r.r < residual_tol).
"""
import numpy as np


class DeviceFisher:
    """A (H + damping I) operator bound to an UpdateEngine's current batch / policy."""

    def __init__(self, engine, damping):
        self.engine, self.damping = engine, float(damping)

    def __call__(self, v):
        torch = self.engine.torch
        vt = self.engine.to_device_f32(np.asarray(v, np.float32))
        out = self.engine.to_host(self.engine.fvp(vt))
        return out + np.float32(self.damping) * np.asarray(v, np.float32)


def cg_solve(f_Ax, b, x_0=None, cg_iters=10, residual_tol=1e-10):
    if isinstance(f_Ax, DeviceFisher):
        eng = f_Ax.engine
        bt = eng.to_device_f32(np.asarray(b, np.float32))
        x, _ = eng.cg_solve(bt, cg_iters, f_Ax.damping, residual_tol)
        return eng.to_host(x)
    sol = np.zeros_like(b)
    resid = np.array(b, copy=True)
    direction = np.array(b, copy=True)
    rr = float(resid.dot(resid)) if b.dtype == np.float64 else resid.dot(resid)
    it = 0
    while it < cg_iters:
        Ad = f_Ax(direction)
        step = rr / direction.dot(Ad)
        sol += step * direction
        resid -= step * Ad
        rr_next = resid.dot(resid)
        direction *= rr_next / rr
        direction += resid
        rr = rr_next
        it += 1
        if rr < residual_tol:
            break
    return sol
