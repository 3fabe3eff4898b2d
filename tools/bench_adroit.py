"""One NPG update at Adroit door-v0 sizes with the hand_dapg policy (39 obs, 28 actions, 32x32, 200 x 200 timesteps): fused 32-action variant vs MJX_FORCE_LAYERWISE=1."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import _synth as synth
from mjrl_amd.engine import UpdateEngine
n, m, hid, N = 39, 28, (32, 32), 40000
rng = np.random.RandomState(0)
th = synth.perturbed_params(synth.init_params(n, m, hid), scale=0.05)
ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
e = UpdateEngine(n, m, hid)
e.set_policy(th, th, ident, ident)
e.set_batch(rng.randn(N, n).astype(np.float32), rng.randn(N, m).astype(np.float32), rng.randn(N).astype(np.float32))
def upd():
    g, _ = e.surr_vpg(sync=False); e.cg_solve(g, 10, 1e-4, sync=False); e.apply_npg_step(0.05, -3.0); e.eval_surr_kl(); e.deferred()
    e.set_policy(th, th, ident, ident)
for _ in range(3): upd()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): upd()
torch.cuda.synchronize()
print(json.dumps({"fused": bool(e.fused), "ms_per_update": 1e3 * (time.perf_counter() - t0) / 50}))
