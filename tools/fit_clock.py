"""phase stamps of the persistent MLP-fit kernel (debug build -DMJX_PHASE_CLOCK)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd import _lib
_lib.LIB_PATH = os.environ.get("MJX_LIB", _lib.LIB_PATH)
from mjrl_amd._lib import check, ptr
lib = _lib.load()
N, d_in = 64 * 400, 21
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
check(lib.mjx_mlp_fit_adam(ptr(feat), ptr(y), N, d_in, hid, 2, ptr(params), ptr(m), ptr(v), 0, ptr(perm), 1, 64, 1e-3, 0.0, ptr(loss), None))
torch.cuda.synchronize()
st = loss.cpu().numpy().view(np.int64)[8:19]
names = ["gather store+sync", "L1 + h1T + sync", "L2 (64 MFMA) + relu + h2T + sync", "yhat/dy + d2T + sync", "gW3/gb + d2u", "gW2 (64 MFMA)", "delta1u (64 MFMA) + mask", "gW1 (16)+sync",
         "(second half)", "Adam"]
d = np.diff(st)
if st[10] == 0 or os.environ.get("MJX_FIT_ONEPASS", "1") != "0":        # the one-pass kernel (k_mlp_fit1p) leaves 10 stamps
    names = ["gather store+sync", "L1 (2 chains) + h1T + sync", "L2 (128 MFMA) + relu + partials + sync", "yhat/dy + d2T + gW3 (DPP) + sync",
             "d2u + gb2", "gW2 (128 MFMA)", "delta1u (128 MFMA) + mask", "gW1 (32) + sync", "Adam + sync"]
    for i, x in enumerate(np.diff(st[:10])):
        print("%-44s %8d cycles" % (names[i], x))
    print("step total", st[9] - st[0])
    sys.exit(0)
for i, x in enumerate(d):
    print("%-36s %8d cycles" % (names[i], x))
print("half total", st[8] - st[0], " step total ~", st[10] - st[0])
