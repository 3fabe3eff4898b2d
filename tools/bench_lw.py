"""Layer-wise path at cfg4 / cfg5 shapes (per-kernel view: run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.engine import UpdateEngine
import _synth as synth
out = {}
rng = np.random.RandomState(0)
def timeit(fn, reps):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for name, n, m, hid, N in (("cfg4", 376, 17, (256, 256), 200000), ("cfg5", 39, 28, (512, 512), 200000)):
    th = synth.perturbed_params(synth.init_params(n, m, hid), scale=0.02)
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    e = UpdateEngine(n, m, hid)
    e.set_policy(th, th, ident, ident)
    e.set_batch(rng.randn(N, n).astype(np.float32), rng.randn(N, m).astype(np.float32), rng.randn(N).astype(np.float32))
    out[name + "_K1_ms"] = 1e3 * timeit(lambda: e.surr_vpg(), 3)
    g = e.surr_vpg()[0].clone()
    out[name + "_fvp_ms"] = 1e3 * timeit(lambda: e.fvp(g), 5)
    P = n * hid[0] + hid[0] * hid[1] + hid[1] * m
    out[name + "_fvp_TFLOPs_cached_fwd"] = 2 * (4 * P - 2 * n * hid[0]) * N / (out[name + "_fvp_ms"] * 1e-3) / 1e12
    out[name + "_K3_ms"] = 1e3 * timeit(lambda: e.eval_surr_kl(), 3)
    e.close()
print(json.dumps(out))
