import sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from mjrl_amd.baselines.mlp_baseline import MLPBaseline
from mjrl_amd.utils import process_samples, ingest
ingest.tune_malloc()
spec = type("Spec", (), dict(observation_dim=17, action_dim=6, horizon=1000))
rng = np.random.RandomState(0)
bl = MLPBaseline(spec, reg_coef=1e-3, batch_size=64, epochs=2, learn_rate=1e-3)
for it in range(4):
    paths = bench._host_paths(rng)
    with ingest.trusted_iteration():
        process_samples.compute_returns(paths, 0.995)
        pre = bl.predraw(1000 * 1000)
        process_samples.compute_advantages(paths, bl, 0.995, 0.97)
        torch.cuda.synchronize(); time.sleep(0.02)
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable(); pend = bl.fit_async(paths, predrawn=pre); pr.disable()
        t1 = time.perf_counter(); torch.cuda.current_stream().synchronize(); t2 = time.perf_counter()
    ingest.drop_shared_batch()
    print("fit_async returned after %.2f ms, main stream drained after %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    pend.result()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
