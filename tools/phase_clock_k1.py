"""Per-phase cycle stamps of one K1 (MODE_VPG: surrogate + vanilla gradient + the activation caches) tile; debug build with
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
dbg = eng.enable_debug()
for _ in range(3):
    eng.surr_vpg()
torch.cuda.synchronize()
st = dbg.cpu().numpy().view(np.int64)[:14]
names = ["0 stage x", "1 layer 1 (20 MFMA = 1280)", "2 (-)", "3 tanh z1 + cache stores + bias init", "4 layer 2 (64 = 4096)", "5 tanh z2 (+ transposes)", "6 output layer (64 x 4x4x1 = 512)",
         "7 likelihood head + d3 + cache stores", "8 delta2 (16 = 1024)", "9 gW3 (4x4x1) + factor", "10 delta1 (64 = 4096)", "11 gW2 (64 = 4096) + factor", "12 gW1 (160 x 4x4x1 = 1280)"]
d = np.diff(st)
for i, x in enumerate(d):
    print("%-46s %7d cycles" % (names[i] if i < len(names) else i, x))
print("tile total", st[-1] - st[0])
g = dbg.cpu().numpy().view(np.int64)[16:21]
print("kernel: prologue %d, tile loop %d, wait %d, reduce+write %d, total %d cycles" % (g[1] - g[0], g[2] - g[1], g[3] - g[2], g[4] - g[3], g[4] - g[0]))
