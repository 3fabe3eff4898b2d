"""Per-phase cycle stamps of the fused FVP kernel (debug build with -DMJX_PHASE_CLOCK):
hipcc ... -DMJX_PHASE_CLOCK -o /tmp/libmjx_clock.so ; MJX_LIB=/tmp/... python tools/phase_clock.py"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd import _lib
_lib.LIB_PATH = os.environ.get("MJX_LIB", _lib.LIB_PATH)
from mjrl_amd.engine import UpdateEngine
from oracle import synth
n, m, hid, N = 17, 6, (64, 64), 1000000
rng = np.random.RandomState(0)
th = synth.perturbed_params(synth.init_params(n, m, hid))
ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
eng = UpdateEngine(n, m, hid)
eng.set_policy(th, th, ident, ident)
eng.set_batch(rng.randn(N, n).astype(np.float32))
dbg = eng.enable_debug()
v = torch.from_numpy(rng.randn(th.size).astype(np.float32)).to(eng.device)
for _ in range(3):
    eng.fvp(v)
torch.cuda.synchronize()
st = dbg.cpu().numpy().view(np.int64)[:14]
names = ["stage x", "L1 fwd+tan MFMA", "tanh z1 + t1 scale", "bias init", "pass A", "pass B", "tanh z2 + t2 scale", "out_small x2",
         "out_finish + d3", "transposes + delta2", "gW3 + factor + bias sums", "delta1u", "gW2 + factor", "gW1"]
d = np.diff(st)
tot = st[13] - st[0]
for i, x in enumerate(d):
    print("%-28s %7d cycles  %5.1f%%" % (names[i] if i < len(names) else i, x, 100.0 * x / tot))
print("tile total", tot)
