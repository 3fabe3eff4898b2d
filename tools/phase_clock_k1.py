"""Per-phase cycle stamps of one K1 (MODE_VPG) tile (debug build with -DMJX_PHASE_CLOCK, see tools/build_dbg.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.engine import UpdateEngine
import _synth as synth
n, m, hid, N = 17, 6, (64, 64), 1000000
rng = np.random.RandomState(0)
th = synth.perturbed_params(synth.init_params(n, m, hid))
ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
eng = UpdateEngine(n, m, hid)
eng.set_policy(th, th, ident, ident)
eng.set_batch(rng.randn(N, n).astype(np.float32), rng.randn(N, m).astype(np.float32), rng.randn(N).astype(np.float32))
dbg = eng.enable_debug()
for _ in range(3):
    eng.surr_vpg()
torch.cuda.synchronize()
st = dbg.cpu().numpy().view(np.int64)[:14]
names = ["0 stage x", "1 (lambdas)", "2 L1 MFMAs + xnorm", "3 tanh z1 + pass A", "4 tanh z2", "5 -", "6 cache stores", "7 out_small + LL head", "8 delta2", "9 gW3", "10 delta1u", "11 gW2", "12 gW1"]
print([(i, int(x)) for i, x in enumerate(st - st[0])])
g = dbg.cpu().numpy().view(np.int64)[16:21]
print("kernel: prologue %d, tile loop %d, wait %d, reduce+write %d, total %d cycles" % (g[1] - g[0], g[2] - g[1], g[3] - g[2], g[4] - g[3], g[4] - g[0]))
