"""MLP-baseline fit at an Adroit-sized input (39 observations + 4 time features): persistent trainer vs the per-step launches."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd import _lib
from mjrl_amd._lib import check, ptr
lib = _lib.load()
N, d_in = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 2000, int(sys.argv[2]) if len(sys.argv) > 2 else 43
dev = torch.device("cuda", 0)
rng = np.random.RandomState(int(os.environ.get("SEED", "0")))
feat = torch.from_numpy(rng.randn(N, d_in).astype(np.float32)).to(dev)
y = torch.from_numpy(rng.randn(N).astype(np.float32)).to(dev)
P = 128 * d_in + 128 + 128 * 128 + 128 + 128 + 1
p0 = (0.1 * rng.randn(P)).astype(np.float32)
EPOCHS = int(sys.argv[3]) if len(sys.argv) > 3 else 1
perm = torch.from_numpy(np.concatenate([rng.permutation(N) for _ in range(EPOCHS)]).astype(np.int32)).to(dev)
hid = (ctypes.c_int * 2)(128, 128)
res = {}
for mode in ("persistent", "launches"):
    os.environ["MJX_MLP_FIT_LAUNCHES"] = "1" if mode == "launches" else "0"
    params = torch.from_numpy(p0.copy()).to(dev); m = torch.zeros(P, device=dev); v = torch.zeros(P, device=dev)
    loss = torch.zeros(32, dtype=torch.float64, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    check(lib.mjx_mlp_fit_adam(ptr(feat), ptr(y), N, d_in, hid, 2, ptr(params), ptr(m), ptr(v), 0, ptr(perm), EPOCHS, 64, 1e-3, 1e-3, ptr(loss), None))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[mode] = (params.cpu().numpy(), float(loss[0].item()), dt)
    print(mode, "%.3f s for %d steps = %.1f us/step, epoch loss %.6f" % (dt, N // 64 - 1, 1e6 * dt / (N // 64 - 1), res[mode][1]))
a, b = res["persistent"][0], res["launches"][0]
print("param rel diff persistent vs launches:", float(np.linalg.norm(a - b) / np.linalg.norm(b - p0)), "(relative to the parameter movement)")
# torch reference of the same chain (fit_data of mjrl/utils/optimize_model.py: minibatches of 64 in permutation order, N // 64 - 1 steps)
W1 = torch.from_numpy(p0[:128 * d_in].reshape(128, d_in).copy()).to(dev).requires_grad_()
o = 128 * d_in
b1 = torch.from_numpy(p0[o:o + 128].copy()).to(dev).requires_grad_(); o += 128
W2 = torch.from_numpy(p0[o:o + 128 * 128].reshape(128, 128).copy()).to(dev).requires_grad_(); o += 128 * 128
b2 = torch.from_numpy(p0[o:o + 128].copy()).to(dev).requires_grad_(); o += 128
W3 = torch.from_numpy(p0[o:o + 128].reshape(1, 128).copy()).to(dev).requires_grad_(); o += 128
b3 = torch.from_numpy(p0[o:o + 1].copy()).to(dev).requires_grad_()
opt = torch.optim.Adam([W1, b1, W2, b2, W3, b3], lr=1e-3, weight_decay=1e-3)
pl = perm.long()
for mb in range(N // 64 - 1):
    idx = pl[mb * 64:(mb + 1) * 64]
    x, t = feat[idx], y[idx]
    h = torch.relu(torch.relu(x @ W1.T + b1) @ W2.T + b2) @ W3.T + b3
    loss_t = ((h[:, 0] - t) ** 2).mean()
    opt.zero_grad(); loss_t.backward(); opt.step()
ref = torch.cat([W1.reshape(-1), b1, W2.reshape(-1), b2, W3.reshape(-1), b3]).detach().cpu().numpy()
for mode in res:
    print(mode, "vs torch:", float(np.linalg.norm(res[mode][0] - ref) / np.linalg.norm(ref - p0)))
