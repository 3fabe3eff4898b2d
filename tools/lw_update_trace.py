"""One whole NPG update (mjx_npg_update: K1, the CG solve, step, K3) on the layer-wise path at the configs[3] / [4] shard, for
rocprofv3 --kernel-trace --stats: which kernels an UPDATE spends its time in (tools/lw_profile.py looks at the products only).
    python tools/lw_update_trace.py --cfg cfg4 [--updates 2]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import _synth as synth
from mjrl_amd.engine import UpdateEngine
CFG = {"cfg4": (376, 17, (256, 256), 500000, 25), "cfg5": (39, 28, (512, 512), 1000000, 10)}
ap = argparse.ArgumentParser(); ap.add_argument("--cfg", default="cfg4"); ap.add_argument("--updates", type=int, default=2)
a = ap.parse_args()
n, m, hid, N, cg = CFG[a.cfg]
g = torch.Generator(device="cuda"); g.manual_seed(0)
obs = torch.randn((N, n), generator=g, device="cuda"); act = torch.randn((N, m), generator=g, device="cuda"); adv = torch.randn((N,), generator=g, device="cuda")
th = synth.perturbed_params(synth.init_params(n, m, hid), scale=0.02)
ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
e = UpdateEngine(n, m, hid)
e.set_policy(th, th, ident, ident); e.set_batch(obs, act, adv)
e.npg_update(cg, 1e-4, 0.05, -3.0); torch.cuda.synchronize()
ts = []
for _ in range(a.updates):
    e.set_policy(th, th, ident, ident)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.npg_update(cg, 1e-4, 0.05, -3.0)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(json.dumps({"cfg": a.cfg, "rows": N, "cg_iters": cg, "update_ms": ts}))
if os.environ.get("DAPG") == "1":           # ... and one DAPG update: [N on-policy ; 5 000 demonstration] rows (dapg.py:92-121)
    nd = 5000
    obs2 = torch.cat([obs, obs[:nd]]); act2 = torch.cat([act, act[:nd]]); adv2 = torch.cat([adv, torch.full((nd,), 0.01, device="cuda")])
    e.set_policy(th, th, ident, ident); e.set_batch(obs2, act2, adv2)
    e.dapg_update(cg, 1e-4, 0.05, -3.0, N, adv); torch.cuda.synchronize()
    td = []
    for _ in range(a.updates):
        e.set_policy(th, th, ident, ident)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.dapg_update(cg, 1e-4, 0.05, -3.0, N, adv)
        torch.cuda.synchronize(); td.append(1e3 * (time.perf_counter() - t0))
    print(json.dumps({"cfg": a.cfg, "dapg_update_ms": td}))
