"""Layer-wise Fisher-vector product by its POSITION after K1 (configs[3] / [4] shard): HIP events around each of 14 products of
6 K1 + products sequences.  python tools/lw_fvp_position.py --cfg cfg4"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import _synth as synth
from mjrl_amd.engine import UpdateEngine
CFG = {"cfg4": (376, 17, (256, 256), 500000), "cfg5": (39, 28, (512, 512), 1000000)}
ap = argparse.ArgumentParser(); ap.add_argument("--cfg", default="cfg4"); a = ap.parse_args()
n, m, hid, N = CFG[a.cfg]
g = torch.Generator(device="cuda"); g.manual_seed(0)
obs = torch.randn((N, n), generator=g, device="cuda"); act = torch.randn((N, m), generator=g, device="cuda"); adv = torch.randn((N,), generator=g, device="cuda")
th = synth.perturbed_params(synth.init_params(n, m, hid), scale=0.02)
ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
e = UpdateEngine(n, m, hid)
e.set_policy(th, th, ident, ident); e.set_batch(obs, act, adv)
v = e.surr_vpg()[0].clone()
for _ in range(3): e.fvp(v)
torch.cuda.synchronize()
R, P = 6, 14
ts = np.zeros((R, P)); k1 = np.zeros(R)
for r in range(R):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(P + 2)]
    ev[0].record(); e.backend.surr_vpg(e.grad, e.scal_vpg); ev[1].record()
    for p in range(P):
        e.backend.fvp(v, e.Ap); ev[p + 2].record()
    torch.cuda.synchronize()
    k1[r] = ev[0].elapsed_time(ev[1])
    for p in range(P): ts[r, p] = ev[p + 1].elapsed_time(ev[p + 2])
print(a.cfg, "K1 median %.3f ms" % np.median(k1))
print("product by position after K1 (median of %d, ms):" % R, " ".join("%.3f" % x for x in np.median(ts, axis=0)))
