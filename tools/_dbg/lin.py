import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from oracle import synth
from mjrl_amd.engine import UpdateEngine
def rel(a, b): return float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b) + 1e-300))
n, m, hid, N = 17, 6, (64, 64), 1000 * 1000
rng = np.random.RandomState(0)
obs = rng.randn(N, n).astype(np.float32)
th = synth.perturbed_params(synth.init_params(n, m, hid))
tr = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
eng = UpdateEngine(n, m, hid)
eng.set_policy(th, th, tr, tr)
eng.set_batch(obs)
v1 = torch.from_numpy(rng.randn(th.size).astype(np.float32)).to(eng.device)
v2 = torch.from_numpy(rng.randn(th.size).astype(np.float32)).to(eng.device)
for k in range(3):
    h1, h2 = eng.fvp(v1).clone(), eng.fvp(v2).clone()
    h12 = eng.fvp(2.0 * v1 - 0.5 * v2).clone()
    print(os.environ.get("MJX_LIB"), "linearity", rel(h12.cpu().numpy(), (2.0 * h1 - 0.5 * h2).cpu().numpy().astype(np.float64)), "repeat", rel(eng.fvp(v1).cpu().numpy(), h1.cpu().numpy().astype(np.float64)))
np.save("/tmp/h1_%s.npy" % os.path.basename(os.environ.get("MJX_LIB", "x")), h1.cpu().numpy())
