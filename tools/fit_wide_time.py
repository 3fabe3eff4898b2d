#!/usr/bin/env python
"""us per Adam step of the MLP-baseline trainer at a given input width (default 380 = Humanoid's 376 observations + 4 time
features, BASELINE configs[3]): the several-workgroup persistent trainer (csrc/mlp_fit.h, MULTI) against the per-step launches
(MJX_MLP_FIT_LAUNCHES=1).   python tools/fit_wide_time.py [d_in ...]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mjrl_amd import _lib
from mjrl_amd._lib import check, ptr

lib = _lib.load()
dev = torch.device("cuda", 0)
for d_in in ([int(x) for x in sys.argv[1:]] or [380]):
    res = {}
    for mode, steps in (("persistent", 4000), ("launches", 300)):
        os.environ["MJX_MLP_FIT_LAUNCHES"] = "1" if mode == "launches" else "0"
        N = 64 * (steps + 1)
        r = np.random.RandomState(0)
        feat = torch.from_numpy(r.randn(N, d_in).astype(np.float32)).to(dev)
        y = torch.from_numpy(r.randn(N).astype(np.float32)).to(dev)
        P = 128 * d_in + 128 + 128 * 128 + 128 + 128 + 1
        params = torch.from_numpy((0.1 * r.randn(P)).astype(np.float32)).to(dev)
        m_, v_ = torch.zeros(P, device=dev), torch.zeros(P, device=dev)
        perm = torch.from_numpy(r.permutation(N).astype(np.int32)).to(dev)
        loss = torch.zeros(32, dtype=torch.float64, device=dev)
        hid = (ctypes.c_int * 2)(128, 128)
        tt = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            check(lib.mjx_mlp_fit_adam(ptr(feat), ptr(y), N, d_in, hid, 2, ptr(params), ptr(m_), ptr(v_), 0, ptr(perm), 1, 64, 1e-3, 0.0, ptr(loss), None))
            torch.cuda.synchronize(); tt.append(time.perf_counter() - t0)
        res[mode] = 1e6 * min(tt) / steps
        assert np.isfinite(loss.cpu().numpy()[0])
    print("d_in %d: persistent %.1f us / step, per-step launches %.1f us / step" % (d_in, res["persistent"], res["launches"]), flush=True)
