#!/usr/bin/env python
"""How the length of the fp32 accumulation chains of the layer-wise weight gradients (MJX_LW_CHAIN samples per workgroup,
MJX_LW_WG_CAP workgroups per launch) moves the distance to the reference at a shard size: gradient and whole update of
bench.LW_SHARDS[key] against tests/golden/<fixture>.npz, plus the Fisher-vector-product time.   python tools/probe_chain_error.py [key]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(key):
    import ctypes
    import numpy as np
    import torch
    import bench
    from mjrl_amd._lib import check
    from mjrl_amd.engine import UpdateEngine
    cfg = bench.LW_SHARDS[key]
    g = np.load(os.path.join(ROOT, "tests", "golden", cfg["fixture"] + ".npz"))
    inp = bench.lw_shard_inputs(key)
    n, m, hid = cfg["n"], cfg["m"], cfg["hidden"]
    N = inp["obs"].shape[0]
    th = inp["theta"]
    adv_w = (inp["adv"] - inp["adv"].mean()) / (inp["adv"].std() + 1e-6)
    ident = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
    e = UpdateEngine(n, m, hid)
    e.set_policy(th, th, ident, ident)
    S = int(g["stride"])
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    out = {}
    if cfg["algo"] == "dapg":
        Nd = cfg["demo_rows"]
        obs_all = np.concatenate([inp["obs"], inp["demo_obs"]]); act_all = np.concatenate([inp["act"], inp["demo_act"]])
        all_adv = 1e-2 * np.concatenate([adv_w / (np.std(adv_w) + 1e-8), cfg["lam_0"] * np.ones(Nd)])
        e.set_batch(obs_all, act_all, all_adv)
        gr = e.surr_vpg()[0].cpu().numpy().astype(np.float64) * (all_adv.shape[0] / N)
        out["vpg"] = rel(gr[::S], g["vpg_sub"].astype(np.float64))
        e.set_policy(th, th, ident, ident)
        e.set_batch(obs_all, act_all, all_adv)
        res = e.dapg_update(cfg["cg_iters"], 1e-4, 2.0 * cfg["kl_dist"], -3.0, N, adv_w, N_on_global=N)
    else:
        e.set_batch(inp["obs"], inp["act"], adv_w)
        gr = e.surr_vpg()[0].cpu().numpy().astype(np.float64)
        out["vpg"] = rel(gr[::S], g["vpg_sub"].astype(np.float64))
        e.npg_update(cfg["cg_iters"], 1e-4, 0.05, -3.0)
    step = e.theta_new.cpu().numpy().astype(np.float64) - th
    out["step"] = rel(step[::S], g["update_step_sub"].astype(np.float64))
    out["alpha"] = abs(e.deferred()["alpha"] - float(g["alpha"])) / float(g["alpha"])
    e.set_policy(th, th, ident, ident)
    e.set_batch(inp["obs"], inp["act"], adv_w)
    grad = e.surr_vpg()[0].clone()
    e.fvp(grad); torch.cuda.synchronize()
    check(e.lib.mjx_profile_enable(e.ctx, 1))
    for _ in range(4):
        e.fvp(grad)
    prof = (ctypes.c_double * 2)()
    check(e.lib.mjx_profile_read(e.ctx, prof))
    out["fvp_ms"] = prof[0] / prof[1]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        one(sys.argv[2])
    else:
        key = sys.argv[1] if len(sys.argv) > 1 else "configs4_adroit_512x512"
        for chain, cap in ((2048, 1024), (2048, 2048), (2048, 4096), (1024, 8192), (8192, 1024)):
            env = dict(os.environ, MJX_LW_CHAIN=str(chain), MJX_LW_WG_CAP=str(cap))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", key], env=env, capture_output=True, text=True)
            print(key, "chain", chain, "wg cap", cap, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:], flush=True)
