"""Host gather + upload of a 1M x 17 fp64 observation batch through the PathStager for several native thread counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd.utils import ingest
from mjrl_amd.utils.process_samples import _handle
h = _handle()
rng = np.random.RandomState(0)
paths = [dict(observations=rng.randn(1000, 17), actions=rng.randn(1000, 6)) for _ in range(1000)]
for nt in (4, 8, 16, 32, 64):
    st = ingest.PathStager(h, threads=nt)
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st.stage(paths, ("observations", "actions"))
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        if rep == 3:
            print("threads %2d: host %.2f ms, with transfers %.2f ms" % (nt, 1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    st.close()
