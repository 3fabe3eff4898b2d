"""Average launch time of the cached Fisher-vector-product kernel on bench.py's 1M-timestep batch (HIP events around blocks of
back-to-back products): the quick A/B probe for kernel variants -- MJX_LIB=<build> python tools/fvp_time.py [reps]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd import _lib
_lib.LIB_PATH = os.environ.get("MJX_LIB", _lib.LIB_PATH)
from mjrl_amd.engine import UpdateEngine
import bench
json_out = None
if "--json" in sys.argv:
    k = sys.argv.index("--json"); json_out = sys.argv[k + 1]; del sys.argv[k:k + 2]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
theta0 = bench.initial_params()
obs, act, adv = bench.synth_shard(0, 1)
adv = (adv - adv.mean()) / (adv.std() + 1e-6)
eng = UpdateEngine(bench.N_OBS, bench.N_ACT, bench.HIDDEN)
ident = np.concatenate([np.zeros(bench.N_OBS), np.ones(bench.N_OBS), np.zeros(bench.N_ACT), np.ones(bench.N_ACT)]).astype(np.float32)
eng.set_policy(theta0, theta0, ident, ident)
eng.set_batch(obs, act, adv)
g = eng.surr_vpg()[0].clone()
for _ in range(5):
    eng.fvp(g)
torch.cuda.synchronize()
best = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.backend.fvp(g, eng.Ap)
    e1.record(); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 20)
def timed(fn, n=10):
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return min(ts)
k1 = timed(lambda: eng.backend.surr_vpg(eng.grad, eng.scal_vpg))
eng.theta_new.add_(1e-3); eng.old_is_new = False; eng._bind_policy()        # K3 at theta_new != theta_old (as after a step)
k3 = timed(lambda: eng.backend.eval_surr_kl(eng.scal))
eng.theta_new.copy_(eng.theta_old); eng.old_is_new = True; eng._bind_policy()
eng.surr_vpg()
print("%s  K1 (+ reduce) %.4f ms   K3 (+ reduce) %.4f ms" % (os.environ.get("MJX_LIB", "product"), k1, k3))
clk = torch.zeros(4, dtype=torch.int64, device=eng.device)
import ctypes
_lib.check(eng.lib.mjx_set_clock_buffer(eng.ctx, ctypes.c_void_p(clk.data_ptr())))
cyc, ghz = [], []
for _ in range(10):
    eng.backend.fvp(g, eng.Ap)
    torch.cuda.synchronize()
    c = clk.cpu().numpy()
    cyc.append(int(c[2] - c[0])); ghz.append((c[2] - c[0]) / ((c[3] - c[1]) * 10.0))
_lib.check(eng.lib.mjx_set_clock_buffer(eng.ctx, None))
print("%s  fvp+reduce ms: min %.4f median %.4f | workgroup 0: %d cycles, %.3f GHz" % (os.environ.get("MJX_LIB", "product"), min(best), sorted(best)[len(best) // 2],
                                                                             int(np.median(cyc)), float(np.median(ghz))))
if json_out:
    import json
    json.dump({"what": "cached Fisher-vector-product kernel (+ its reduction launch) on bench.py's 1M-timestep batch: HIP-event time of "
                       "blocks of 20 back-to-back products, and workgroup 0's own clock (mjx_set_clock_buffer: s_memtime cycles over "
                       "s_memrealtime 100 MHz ticks -> sustained shader clock) for 10 single launches",
               "fvp_plus_reduce_ms_blocks_of_20": best, "fvp_plus_reduce_ms_min": min(best), "workgroup0_cycles": cyc, "sustained_GHz": [float(x) for x in ghz],
               "workgroup0_ms_from_cycles": [c / (g * 1e6) for c, g in zip(cyc, ghz)], "K1_plus_reduce_ms": k1, "K3_plus_reduce_ms": k3},
              open(json_out, "w"), indent=1)
