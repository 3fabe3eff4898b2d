"""Per-phase cycle stamps of one K3 (MODE_EVAL: surrogate + KL at the new parameters) tile; debug build with
-DMJX_PHASE_CLOCK (tools/build_dbg.sh), MJX_LIB=tools/_dbg/libmjx_clock.so."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.engine import UpdateEngine
import _synth as synth
n, m, hid, N = 17, 6, (64, 64), int(os.environ.get("N", "1000000"))
rng = np.random.RandomState(0)
th = synth.perturbed_params(synth.init_params(n, m, hid))
ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
eng = UpdateEngine(n, m, hid)
eng.set_policy(th, th, ident, ident)
eng.set_batch(rng.randn(N, n).astype(np.float32), rng.randn(N, m).astype(np.float32), rng.randn(N).astype(np.float32))
eng.surr_vpg()                      # fills the old-policy cache K3 reads
th2 = (th + 0.01 * rng.randn(th.size)).astype(np.float32)
eng.set_policy(th2, th, ident, ident)
dbg = eng.enable_debug()
for _ in range(3):
    eng.eval_surr_kl()
torch.cuda.synchronize()
st = dbg.cpu().numpy().view(np.int64)[:14]
print([(i, int(x)) for i, x in enumerate(st - st[0])])
g = dbg.cpu().numpy().view(np.int64)[16:21]
print("kernel: prologue %d, tile loop %d, wait %d, reduce+write %d, total %d cycles" % (g[1] - g[0], g[2] - g[1], g[3] - g[2], g[4] - g[3], g[4] - g[0]))
