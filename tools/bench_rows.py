"""Timings of the other hot-path rows at BASELINE sizes (K1/K3, TRPO, K5, K6, layer-wise path).
Run on the GPU box: python tools/bench_rows.py > gpurun_out/rows.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from mjrl_amd.engine import UpdateEngine
from mjrl_amd.utils import process_samples
import _synth as synth


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = {}
spec = type("Spec", (), dict(observation_dim=17, action_dim=6, horizon=1000))
rng = np.random.RandomState(0)
paths = [dict(observations=rng.randn(1000, 17), actions=rng.randn(1000, 6), rewards=rng.randn(1000), terminated=False) for _ in range(1000)]

# K5
out["compute_returns_1M_s"] = timeit(lambda: process_samples.compute_returns(paths, 0.995))
from mjrl_amd.baselines.mlp_baseline import MLPBaseline
from mjrl_amd.baselines.quadratic_baseline import QuadraticBaseline
blm = MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3)
out["gae_plus_mlp_predict_1M_s"] = timeit(lambda: process_samples.compute_advantages(paths, blm, 0.995, 0.97))
blq = QuadraticBaseline(spec)
out["quadratic_fit_1M_s"] = timeit(lambda: blq.fit(paths, return_errors=True), reps=2)
out["gae_plus_quadratic_predict_1M_s"] = timeit(lambda: process_samples.compute_advantages(paths, blq, 0.995, 0.97))
t0 = time.perf_counter(); e = blm.fit(paths, return_errors=True); torch.cuda.synchronize()
out["mlp_baseline_fit_1M_2epochs_s"] = time.perf_counter() - t0
out["mlp_baseline_fit_errors"] = [float(e[0]), float(e[1])]

# K1 / K3 / TRPO at cfg2 (device resident)
th = synth.perturbed_params(synth.init_params(17, 6, (64, 64)))
ident = np.concatenate([np.zeros(17), np.ones(17), np.zeros(6), np.ones(6)]).astype(np.float32)
obs = np.concatenate([p["observations"] for p in paths]).astype(np.float32)
act = np.concatenate([p["actions"] for p in paths]).astype(np.float32)
adv = rng.randn(obs.shape[0]); adv = (adv - adv.mean()) / adv.std()
eng = UpdateEngine(17, 6, (64, 64))
eng.set_policy(th, th, ident, ident); eng.set_batch(obs, act, adv)
out["K1_surr_vpg_ms"] = 1e3 * timeit(lambda: eng.surr_vpg(), 10)
g = eng.surr_vpg()[0].clone()
out["K2_fvp_ms"] = 1e3 * timeit(lambda: eng.fvp(g), 20)
x, gx = eng.cg_solve(g, 10, 1e-4)
eng.apply_step(np.sqrt(0.05 / gx), -3.0)
out["K3_eval_ms"] = 1e3 * timeit(lambda: eng.eval_surr_kl(), 10)
t0 = time.perf_counter()
out["upload_1M_fp64_to_f32_s"] = timeit(lambda: eng.set_batch(obs.astype(np.float64), act.astype(np.float64), adv), 2)

# end-to-end NPG.train_from_paths on fp64 host paths (process_paths + upload + update + read-back)
from mjrl_amd.algos.npg_cg import NPG
from mjrl_amd.algos.trpo import TRPO
from mjrl_amd.policies.gaussian_mlp import MLP
for p_, a_ in zip(paths, np.split(adv, 1000)):
    p_["advantages"] = a_
pol = MLP(spec, hidden_sizes=(64, 64), seed=1, init_log_std=-0.5)
agent = NPG(None, pol, None, normalized_step_size=0.05)
agent.train_from_paths(paths)
t0 = time.perf_counter(); agent.train_from_paths(paths); torch.cuda.synchronize()
out["npg_train_from_paths_end_to_end_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); agent.process_paths(paths); out["process_paths_host_s"] = time.perf_counter() - t0
tr = TRPO(None, pol, None, kl_dist=0.01)
tr.train_from_paths(paths)
t0 = time.perf_counter(); tr.train_from_paths(paths); torch.cuda.synchronize()
out["trpo_train_from_paths_end_to_end_s"] = time.perf_counter() - t0
out["trpo_trials"] = tr.last_update["trials"]

# layer-wise path: cfg4 shapes, 200k samples on one GPU
n, m, hid, N = 376, 17, (256, 256), 200000
th4 = synth.perturbed_params(synth.init_params(n, m, hid), scale=0.02)
id4 = np.concatenate([np.zeros(n), np.ones(n), np.zeros(m), np.ones(m)]).astype(np.float32)
e4 = UpdateEngine(n, m, hid)
e4.set_policy(th4, th4, id4, id4)
e4.set_batch(rng.randn(N, n).astype(np.float32), rng.randn(N, m).astype(np.float32), rng.randn(N).astype(np.float32))
out["cfg4_200k_K1_ms"] = 1e3 * timeit(lambda: e4.surr_vpg(), 3)
g4 = e4.surr_vpg()[0].clone()
out["cfg4_200k_fvp_ms"] = 1e3 * timeit(lambda: e4.fvp(g4), 5)
P = n * 256 + 256 * 256 + 256 * m
out["cfg4_fvp_TFLOPs_cached_fwd"] = 2 * (4 * P - 2 * n * 256 + 0) * N / (out["cfg4_200k_fvp_ms"] * 1e-3) / 1e12
print(json.dumps(out, indent=1))
