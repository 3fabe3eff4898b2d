"""Layer-wise path (BASELINE configs[3] / [4] shapes) at the per-GPU shard size of the 8-GPU configs: K1 once, then
`--fvps` Fisher-vector products and one K3, for rocprofv3 (--kernel-trace --stats / --pmc passes, tools/profile_lw.sh).
Prints one JSON line with the HIP-event time of the whole FVP chain (mjx_profile_*) and its algorithmic rates.

    python tools/lw_profile.py --cfg cfg4 [--rows 500000] [--fvps 5]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import _synth as synth  # noqa: E402
from mjrl_amd._lib import check  # noqa: E402
from mjrl_amd.engine import UpdateEngine  # noqa: E402

CFG = {"cfg4": (376, 17, (256, 256), 500000), "cfg5": (39, 28, (512, 512), 1000000)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="cfg4")
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--fvps", type=int, default=5)
    a = ap.parse_args()
    n, m, hid, N = CFG[a.cfg]
    N = a.rows or N
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    obs = torch.randn((N, n), generator=g, device="cuda")
    act = torch.randn((N, m), generator=g, device="cuda")
    adv = torch.randn((N,), generator=g, device="cuda")
    th = synth.perturbed_params(synth.init_params(n, m, hid), scale=0.02)
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    e = UpdateEngine(n, m, hid)
    assert not e.fused
    e.set_policy(th, th, ident, ident)
    e.set_batch(obs, act, adv)
    grad = e.surr_vpg()[0].clone()
    e.fvp(grad)                                     # warm-up (allocations, code objects)
    torch.cuda.synchronize()
    check(e.lib.mjx_profile_enable(e.ctx, 1))
    for _ in range(a.fvps):
        e.fvp(grad)
    prof = (ctypes.c_double * 2)()
    check(e.lib.mjx_profile_read(e.ctx, prof))
    check(e.lib.mjx_profile_enable(e.ctx, 0))
    e.eval_surr_kl()
    torch.cuda.synchronize()
    P = n * hid[0] + hid[0] * hid[1] + hid[1] * m
    flop = 2 * (4 * P - 2 * n * hid[0]) * N         # cached-forward product: tangent + transpose passes
    ms = prof[0] / prof[1]
    print(json.dumps({"cfg": a.cfg, "rows": N, "fvp_ms": ms, "launches": int(prof[1]), "flop_per_fvp": flop,
                      "TFLOPs": flop / (ms * 1e-3) / 1e12, "frac_fp32_mfma_peak": flop / (ms * 1e-3) / 1e12 / 157.3}))
    e.close()


if __name__ == "__main__":
    main()
