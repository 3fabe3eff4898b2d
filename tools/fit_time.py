"""Seconds per minibatch step of the persistent MLP-baseline trainer (k_mlp_fit), d_in = 21 (locomotion observations + 4 time
features) by default: python tools/fit_time.py [d_in] [steps]; MJX_LIB selects the build."""
import os, sys, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd import _lib
_lib.LIB_PATH = os.environ.get("MJX_LIB", _lib.LIB_PATH)
from mjrl_amd._lib import check, ptr
lib = _lib.load()
d_in = int(sys.argv[1]) if len(sys.argv) > 1 else 21
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
N = 64 * (steps + 1)
dev = torch.device("cuda", 0)
rng = np.random.RandomState(0)
feat = torch.from_numpy(rng.randn(N, d_in).astype(np.float32)).to(dev)
y = torch.from_numpy(rng.randn(N).astype(np.float32)).to(dev)
P = 128 * d_in + 128 + 128 * 128 + 128 + 128 + 1
params = torch.from_numpy((0.1 * rng.randn(P)).astype(np.float32)).to(dev)
m = torch.zeros(P, device=dev); v = torch.zeros(P, device=dev)
perm = torch.from_numpy(rng.permutation(N).astype(np.int32)).to(dev)
loss = torch.zeros(32, dtype=torch.float64, device=dev)
hid = (ctypes.c_int * 2)(128, 128)
ts = []
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    check(lib.mjx_mlp_fit_adam(ptr(feat), ptr(y), N, d_in, hid, 2, ptr(params), ptr(m), ptr(v), 0, ptr(perm), 1, 64, 1e-3, 0.0, ptr(loss), None))
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("%s REGMOM=%s  d_in %d: %.2f us / step (best of 3 x %d steps); loss %.6f" % (os.environ.get("MJX_LIB", "product"), os.environ.get("MJX_FIT_REGMOM", "1"), d_in, 1e6 * min(ts) / steps, steps, float(loss[0]) / steps))
