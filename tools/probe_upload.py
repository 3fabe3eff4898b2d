"""Micro-timings of host -> device copies on this stack (pageable .to(), page-locked bounce, its pieces)."""
import time, numpy as np, torch
dev = torch.device("cuda", 0)
torch.cuda.synchronize()
def t(f, reps=5):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); out.append(round(1e3 * (time.perf_counter() - t0), 2))
    return out
for name, a in (("int32 4MB", np.arange(1_000_000, dtype=np.int32)), ("f64 8MB", np.random.randn(1_000_000)), ("f64 136MB", np.random.randn(1_000_000, 17))):
    print(name, "pageable .to():", t(lambda: torch.from_numpy(a).to(dev)))
    pin = torch.empty(a.nbytes, dtype=torch.uint8, pin_memory=True)
    pnp = pin.numpy()
    print(name, "numpy copy into pinned:", t(lambda: pnp.__setitem__(slice(None), a.reshape(-1).view(np.uint8))))
    tmp = np.empty(a.nbytes, np.uint8)
    print(name, "numpy copy into pageable:", t(lambda: tmp.__setitem__(slice(None), a.reshape(-1).view(np.uint8))))
    d = torch.empty(a.nbytes, dtype=torch.uint8, device=dev)
    print(name, "pinned -> device copy_:", t(lambda: d.copy_(pin, non_blocking=True)))
    print(name, "torch.empty device:", t(lambda: torch.empty(a.nbytes, dtype=torch.uint8, device=dev)))
    print(name, "torch.empty pinned:", t(lambda: torch.empty(a.nbytes, dtype=torch.uint8, pin_memory=True), reps=3))
    print(name, "event record+query:", t(lambda: (lambda e: (e.record(), e.query()))(torch.cuda.Event())))
    print(name, "fresh arange + astype:", t(lambda: (np.arange(1_000_000, dtype=np.int64) - np.repeat(np.arange(1000) * 1000, 1000)).astype(np.int32)))
