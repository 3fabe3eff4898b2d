"""First-contact diagnostic on a GPU box: per-stage errors of the HIP path vs the oracle
(prints instead of asserting, dumps fused-kernel intermediates)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from mjrl_amd.engine import UpdateEngine
from oracle import npg_oracle as O
from tests._cases import NpgCase


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def run_case(name, dbg=False, layerwise=False):
    c = NpgCase(name)
    if layerwise:
        os.environ["MJX_FORCE_LAYERWISE"] = "1"
    else:
        os.environ.pop("MJX_FORCE_LAYERWISE", None)
    eng = UpdateEngine(c.n, c.m, c.hidden)
    print("== %s  N=%d d=%d fused=%s" % (name, c.obs.shape[0], eng.d, eng.fused), flush=True)
    th = c.theta0
    trp = np.concatenate([np.zeros(c.n), np.ones(c.n), np.zeros(c.m), np.ones(c.m)]).astype(np.float32) if c.tr is None \
        else np.concatenate([np.float32(x).ravel() for x in c.tr])
    eng.set_policy(th, th, trp, trp)
    eng.set_batch(c.obs, c.act, c.adv_w)
    if dbg:
        d = eng.enable_debug()
    th64 = th.astype(np.float64)
    tr = c.transforms()
    a = (c.n, c.m, c.hidden)
    g, surr = eng.surr_vpg()
    g = g.cpu().numpy()
    g_or = O.vpg(th64, th64, c.obs, c.act, c.adv_w, *a, tr, tr)
    print("  vpg  rel vs oracle %.3e   surr %.6e vs %.6e" % (rel(g, g_or), surr, float(c.g["surr_before"])))
    if not c.big:
        print("  vpg  rel vs golden %.3e" % rel(g, c.g["vpg"]))
    fo_ = None
    if dbg:
        dd = d.cpu().numpy()
        mu, acts = O.forward(th64, c.obs[:32], *a, tr, keep=True)
        h1 = dd[:2048].reshape(64, 32).T; h2 = dd[2048:4096].reshape(64, 32).T
        print("   dbg(vpg) mu err %.3e" % np.abs(dd[2048 * 4:2048 * 4 + 8 * 32].reshape(8, 32)[:c.m].T - mu).max())
    # per-block errors of the gradient
    Ws, bs, s = O.unflatten(g_or, *a)
    k = 0
    for i, (W, b) in enumerate(zip(Ws, bs)):
        print("   block W%d rel %.3e  b%d rel %.3e" % (i, rel(g[k:k + W.size], W.ravel()), i, rel(g[k + W.size:k + W.size + b.size], b)))
        k += W.size + b.size
    print("   block log_std rel %.3e" % rel(g[k:], s))
    v = g_or.astype(np.float32)
    vt = torch.from_numpy(v).to(eng.device)
    hv = eng.fvp(vt).cpu().numpy()
    hv_or = O.fvp(th64, c.obs, v.astype(np.float64), *a, tr, damping=0.0)
    print("  fvp  rel vs oracle %.3e" % rel(hv, hv_or))
    Ws, bs, s = O.unflatten(hv_or, *a)
    k = 0
    for i, (W, b) in enumerate(zip(Ws, bs)):
        print("   block W%d rel %.3e  b%d rel %.3e" % (i, rel(hv[k:k + W.size], W.ravel()), i, rel(hv[k + W.size:k + W.size + b.size], b)))
        k += W.size + b.size
    print("   block log_std rel %.3e" % rel(hv[k:], s))
    if dbg:
        dd = d.cpu().numpy()
        mu, acts = O.forward(th64, c.obs[:32], *a, tr, keep=True)
        Vs, cs, vs = O.unflatten(v.astype(np.float64), *a)
        Wt, bt, _ = O.unflatten(th64, *a)
        t1 = (acts[0] @ Vs[0].T + cs[0]) * (1 - acts[1] ** 2)
        t2 = (acts[1] @ Vs[1].T + t1 @ Wt[1].T + cs[1]) * (1 - acts[2] ** 2)
        mudot = acts[2] @ Vs[2].T + t2 @ Wt[2].T + cs[2]
        H = c.hidden[0]
        for nm, off, ref in (("h1", 0, acts[1]), ("h2", 2048, acts[2]), ("t1", 4096, t1), ("t2", 6144, t2)):
            got = dd[off:off + H * 32].reshape(H, 32).T
            print("   dbg %s maxabs err %.3e (scale %.3e)" % (nm, np.abs(got - ref).max(), np.abs(ref).max()))
        got = dd[2048 * 4:2048 * 4 + 8 * 32].reshape(8, 32)[:c.m].T
        print("   dbg mudot maxabs err %.3e (scale %.3e)" % (np.abs(got - mudot).max(), np.abs(mudot).max()))
    t0 = time.time()
    x, gx = eng.cg_solve(torch.from_numpy(g_or.astype(np.float32)).to(eng.device), c.cg_iters, 1e-4)
    torch.cuda.synchronize()
    x = x.cpu().numpy()
    x_or = O.cg_solve(lambda p: O.fvp(th64, c.obs, p, *a, tr, damping=1e-4), g_or, c.cg_iters)
    print("  cg   rel vs oracle(fp64) %.3e   g.x %.6e vs %.6e   (%.1f ms)" % (rel(x, x_or), gx, g_or.dot(x_or), 1e3 * (time.time() - t0)))
    if not c.big:
        print("  cg   rel vs golden %.3e ; golden vs oracle %.3e" % (rel(x, c.g["cg_x"]), rel(c.g["cg_x"], x_or)))
    alpha = np.sqrt(abs(float(c.g["step"]) / (gx + 1e-20)))
    eng.apply_step(alpha, -3.0)
    sa, kl = eng.eval_surr_kl()
    print("  alpha %.6f (golden %.6f)  kl %.6e (golden %.6e)  surr_imp %.6e (golden %.6e)"
          % (alpha, float(c.g["alpha"]), kl, float(c.g["kl"]), sa - surr, float(c.g["surr_improvement"])))
    if not c.big:
        newp = eng.theta_new.cpu().numpy()
        print("  step rel vs golden %.3e" % rel(newp - th, c.g["new_params"] - th))
    eng.close()


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    run_case("npg_cfg2_small", dbg=True)
    run_case("npg_pointmass_32x32")
    run_case("npg_cfg2_ragged_tr")
    run_case("npg_cfg2_small", layerwise=True)
    run_case("npg_cfg1_linear")
    run_case("npg_cfg4_small")
