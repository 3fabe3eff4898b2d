"""What makes the first small default-stream operation after a side-stream staging pass take ~20 ms?"""
import time, numpy as np, torch
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev)
big_pin = torch.empty((1_000_000, 17), dtype=torch.float64, pin_memory=True)
big_dev = torch.empty((1_000_000, 17), dtype=torch.float64, device=dev)
small_pin = torch.empty(4_000_000, dtype=torch.uint8, pin_memory=True)
small_pageable = np.arange(1_000_000, dtype=np.int32)
def ms(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return round(1e3 * (time.perf_counter() - t0), 2)
def stage(wait=True, chunks=4):
    with torch.cuda.stream(side):
        n = big_pin.shape[0] // chunks
        for c in range(chunks):
            big_dev[c * n:(c + 1) * n].copy_(big_pin[c * n:(c + 1) * n], non_blocking=True)
    if wait:
        torch.cuda.current_stream(dev).wait_stream(side)
def small_copy():
    d = torch.empty(4_000_000, dtype=torch.uint8, device=dev)
    d.copy_(small_pin, non_blocking=True)
    return d
for trial in range(3):
    print("trial", trial)
    print("  stage (side stream, wait_stream):", ms(stage))
    print("  small pinned copy on default stream right after:", ms(small_copy))
    print("  again:", ms(small_copy))
    print("  stage without wait_stream:", ms(lambda: stage(False)))
    print("  small pinned copy after that:", ms(small_copy))
    print("  stage:", ms(stage))
    print("  pageable .to() after stage:", ms(lambda: torch.from_numpy(small_pageable).to(dev)))
    print("  stage:", ms(stage))
    print("  small kernel (fill) after stage:", ms(lambda: torch.empty(1000, device=dev).fill_(1.0)))
    print("  small copy after kernel:", ms(small_copy))
    print("  stage:", ms(stage))
    def fresh_then_copy():
        a = (np.arange(1_000_000, dtype=np.int64) - np.repeat(np.arange(1000) * 1000, 1000)).astype(np.int32)
        small_pin.numpy()[:] = a.view(np.uint8)
        return small_copy()
    print("  fresh host array -> pinned -> device after stage:", ms(fresh_then_copy))
