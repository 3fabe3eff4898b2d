"""The ~20 ms stall after staging: replay the real PathStager + a small upload, toggling its ingredients."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mjrl_amd import _lib
from mjrl_amd.utils import ingest
dev = torch.device("cuda", 0)
h = ingest.DeviceHandle(torch, dev, _lib.load())
rng = np.random.RandomState(0)
def make():
    return [dict(observations=rng.randn(1000, 17), rewards=rng.randn(1000)) for _ in range(1000)]
small = np.arange(1_000_000, dtype=np.int32)
def ms(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return round(1e3 * (time.perf_counter() - t0), 2)
for label, kw in (("native gather + cast (product)", {}), ("numpy gather (no native threads)", dict(native=False, threads=1)),
                  ("float32 paths (no cast kernel)", dict(f32=True)), ("one group (no chunking)", dict(group_rows=1 << 30))):
    f32 = kw.pop("f32", False)
    st = ingest.PathStager(h, **kw)
    res = []
    for it in range(6):
        paths = make()
        if f32:
            for p in paths: p["observations"] = p["observations"].astype(np.float32)
        a = ms(lambda: st.stage(paths, ("observations",)))
        if "--pageable" in sys.argv:              # one pageable read-back + one pageable upload per iteration, like the real flow
            keep = torch.zeros(1_000_000, dtype=torch.float64, device=dev).cpu().numpy()
            torch.from_numpy(np.arange(1001, dtype=np.int64)).to(dev)
        host = ms(lambda: (np.arange(1_000_000, dtype=np.int64) - np.repeat(np.arange(1000) * 1000, 1000)).astype(np.int32))
        b = ms(lambda: ingest.upload(h, small))
        c = ms(lambda: ingest.upload(h, small))
        res.append((a, host, b, c))
    print(label, "[stage, host work, upload, upload again] ms:", res[2:])
