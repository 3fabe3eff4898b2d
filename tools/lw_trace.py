"""one layer-wise FVP at cfg4 shapes (for rocprofv3 --kernel-trace): prints nothing; read the per-launch durations from <out>_kernel_trace.csv (the last ~25 dispatches are one FVP)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.engine import UpdateEngine
import _synth as synth
rng = np.random.RandomState(0)
n, m, hid, N = 376, 17, (256, 256), 200000
th = synth.perturbed_params(synth.init_params(n, m, hid), scale=0.02)
ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
e = UpdateEngine(n, m, hid)
e.set_policy(th, th, ident, ident)
e.set_batch(rng.randn(N, n).astype(np.float32), rng.randn(N, m).astype(np.float32), rng.randn(N).astype(np.float32))
g = e.surr_vpg()[0].clone()
e.fvp(g); e.fvp(g)
torch.cuda.synchronize()
