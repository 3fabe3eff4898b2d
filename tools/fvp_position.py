"""Launch time of the cached Fisher-vector product by its POSITION in a solve (1st .. 10th product after K1) on bench.py's
1M-timestep batch: HIP events around every product of 20 K1 + 10-product sequences.  MJX_FVP_SWEEP=0/1 A/B of the sweep order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd import _lib
_lib.LIB_PATH = os.environ.get("MJX_LIB", _lib.LIB_PATH)
from mjrl_amd.engine import UpdateEngine
import bench
theta0 = bench.initial_params()
obs, act, adv = bench.synth_shard(0, 1)
adv = (adv - adv.mean()) / (adv.std() + 1e-6)
eng = UpdateEngine(bench.N_OBS, bench.N_ACT, bench.HIDDEN)
ident = np.concatenate([np.zeros(bench.N_OBS), np.ones(bench.N_OBS), np.zeros(bench.N_ACT), np.ones(bench.N_ACT)]).astype(np.float32)
eng.set_policy(theta0, theta0, ident, ident)
eng.set_batch(obs, act, adv)
g = eng.surr_vpg()[0].clone()
for _ in range(5):
    eng.fvp(g)
torch.cuda.synchronize()
R, P = 20, 10
ts = np.zeros((R, P)); k1 = np.zeros(R)
for r in range(R):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(P + 2)]
    ev[0].record()
    eng.backend.surr_vpg(eng.grad, eng.scal_vpg)
    ev[1].record()
    for p in range(P):
        eng.backend.fvp(g, eng.Ap)
        ev[p + 2].record()
    torch.cuda.synchronize()
    k1[r] = ev[0].elapsed_time(ev[1])
    for p in range(P):
        ts[r, p] = ev[p + 1].elapsed_time(ev[p + 2])
print("K1 (+ reduce) median %.4f ms" % np.median(k1))
print("product (+ reduce) by position after K1, median over %d sequences (ms):" % R, " ".join("%.4f" % x for x in np.median(ts, axis=0)))
print("sum of the ten: %.4f ms" % np.median(ts, axis=0).sum())
