"""Non-cached Fisher-vector product at the bench size: linearity, and bit-exact repeatability of the same product (an inline-asm\nVALU instruction placed directly behind an MFMA gets no wait states from the compiler: the product then comes out 2 % wrong and\ndifferent from call to call -- this is the check that caught it).  MJX_LIB selects the build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _synth as synth          # (tools keep their own copy of the initialiser: nothing outside tests / smoke / bench imports oracle/)
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

