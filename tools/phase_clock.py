"""Per-phase cycle stamps of the fused FVP kernel (debug build with -DMJX_PHASE_CLOCK):
hipcc ... -DMJX_PHASE_CLOCK -o /tmp/libmjx_clock.so ; MJX_LIB=/tmp/... python tools/phase_clock.py"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd import _lib
_lib.LIB_PATH = os.environ.get("MJX_LIB", _lib.LIB_PATH)
from mjrl_amd.engine import UpdateEngine
import _synth as synth
n, m, hid, N = 17, 6, (64, 64), int(os.environ.get("N", "1000000"))
rng = np.random.RandomState(0)
th = synth.perturbed_params(synth.init_params(n, m, hid))
ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
eng = UpdateEngine(n, m, hid)
eng.set_policy(th, th, ident, ident)
eng.set_batch(rng.randn(N, n).astype(np.float32), rng.randn(N, m).astype(np.float32), rng.randn(N).astype(np.float32))
if os.environ.get('CACHED', '1') == '1':
    eng.surr_vpg()          # fills the forward-activation cache -> the cached FVP instance runs
dbg = eng.enable_debug()
v = torch.from_numpy(rng.randn(th.size).astype(np.float32)).to(eng.device)
for _ in range(3):
    eng.fvp(v)
torch.cuda.synchronize()
st = dbg.cpu().numpy().view(np.int64)[:14]
if os.environ.get('CACHED', '1') == '1':
    # the cached instance's own schedule (fused_policy.h, "cached-forward Fisher-vector product")
    names = ["R1 t1 = V1a x~ (20 MFMA = 1280)", "R2 t2 = c2 + V2 h1 (64 = 4096)", "R3 t2 += W2 t1 (64 = 4096)", "R4 output layer (128 x 4x4x1 = 1024)",
             "R5 d3 epilogue", "R6 delta2 (8 = 512)", "R7 gW3 (64 x 4x4x1 = 512)", "R8 delta1 + delta2^T trip (64 = 4096)", "R9 gW2 (64 = 4096)",
             "R10 gW1 (160 x 4x4x1 = 1280)"]
    st = np.concatenate([st[:10], st[13:14]])
else:
    names = ["0 stage x", "1 L1 fwd+tan MFMAs (40)", "2 (tanh z1, sunk)", "3 tanh z1 + bias init + pass A (128)", "4 pass B (64) + tanh z2 + transposes", "5 -", "6 out_small (128 x 4x4)",
             "7 out_finish + d3", "8 delta2 (16)", "9 gW3 (32) + factor", "10 delta1u (64)", "11 gW2 (64) + factor", "12 gW1 (32)"]
    st[5] = st[4]  # stamp 5 no longer exists
d = np.diff(st)
tot = st[-1] - st[0]
for i, x in enumerate(d):
    print("%-28s %7d cycles  %5.1f%%" % (names[i] if i < len(names) else i, x, 100.0 * x / tot))
print("tile total", tot)

g = dbg.cpu().numpy().view(np.int64)[16:21]
print("kernel phases (block 0, thread 0): prologue %d, tile loop %d (wave 0), wait for other waves %d, reduce+write %d, total %d cycles"
      % (g[1] - g[0], g[2] - g[1], g[3] - g[2], g[4] - g[3], g[4] - g[0]))

rt = dbg.cpu().numpy().view(np.int64)[24:29]
us = (rt[4] - rt[0]) / 100.0          # s_memrealtime ticks at 100 MHz
print("block 0 wall time %.1f us -> shader clock %.3f GHz" % (us, (g[4] - g[0]) / us / 1e3))
