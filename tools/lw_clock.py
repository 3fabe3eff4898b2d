"""Per-workgroup timeline of the k_gemm launches of ONE layer-wise Fisher-vector product (timing build of the library:
hipcc ... -DMJX_PHASE_CLOCK -o mjrl_amd/csrc/libmjx_clock.so; MJX_LIB=mjrl_amd/csrc/libmjx_clock.so python tools/lw_clock.py --cfg cfg4).
For every launch: lifetime of a workgroup split into prologue (entry -> first MFMA), k-loop, epilogue, the idle gap
between consecutive workgroups on the same CU (launch + dispatch overhead) and the shader clock over the k-loop.  For the
persistent kernel (BN printed as 1000 + 256) a "workgroup" is one output tile: `prologue_us` is its k-loop, `epilogue_us`
its epilogue up to the barrier before the next tile."""
import argparse
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import _synth as synth  # noqa: E402
from mjrl_amd._lib import check, ptr  # noqa: E402
from mjrl_amd.engine import UpdateEngine  # noqa: E402

CFG = {"cfg4": (376, 17, (256, 256), 500000), "cfg5": (39, 28, (512, 512), 1000000)}
SLOT, SLOTS = 8 + 8 * 16384, 24
EPI = ["STORE", "BIAS_TANH", "BIAS_AFFINE", "TANGENT", "BACK", "BIAS", "BIAS_RELU", "BACK_RELU", "RBACK", "FVP_HEAD"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="cfg4")
    ap.add_argument("--rows", type=int, default=0)
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
    e.set_policy(th, th, ident, ident)
    e.set_batch(obs, act, adv)
    grad = e.surr_vpg()[0].clone()
    e.fvp(grad); e.fvp(grad)
    torch.cuda.synchronize()
    buf = torch.zeros(SLOT * SLOTS, dtype=torch.int64, device="cuda")
    check(e.lib.mjx_set_debug_buffer(e.ctx, ptr(buf), buf.numel() * 2))
    e.fvp(grad)
    torch.cuda.synchronize()
    check(e.lib.mjx_set_debug_buffer(e.ctx, None, 0))
    h = buf.cpu().numpy().reshape(SLOTS, SLOT)
    out = []
    for s in range(SLOTS):
        hdr = h[s, :8]
        nb = int(hdr[6])
        if nb == 0:
            continue
        st = h[s, 8:8 + 8 * nb].reshape(nb, 8)
        t = st[:, :4].astype(np.float64) / 100.0                       # us
        life, pro, loop, epi = t[:, 3] - t[:, 0], t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
        cu = (st[:, 5] & 0xf) * 4096 + (st[:, 4] & 0xff00) // 256 * 1   # XCC, then HW_ID bits 8..15 (cu / sh / se)
        gaps, per_cu = [], collections.defaultdict(list)
        for i in np.argsort(t[:, 0]):
            per_cu[int(cu[i])].append(i)
        for lst in per_cu.values():
            for x, y in zip(lst[:-1], lst[1:]):
                gaps.append(t[y, 0] - t[x, 3])
        # shader clock over the k-loop (cycle counter against the 100 MHz real-time counter; persistent kernel: stamps 0..1)
        pers = int(hdr[5]) >= 1000
        dt_us = (t[:, 1] - t[:, 0]) if pers else (t[:, 2] - t[:, 1])
        cyc = (st[:, 7] - st[:, 6]).astype(np.float64)
        okc = (dt_us > 1.0) & (cyc > 0)
        ghz = float(np.median(cyc[okc] / dt_us[okc]) / 1e3) if okc.any() else None
        span = t[:, 3].max() - t[:, 0].min()
        M, Nn, K0, K1 = int(hdr[0]), int(hdr[1]), int(hdr[2]), int(hdr[3])
        flop = 2.0 * M * Nn * (K0 + K1)
        rec = dict(slot=s, M=M, N=Nn, K=[K0, K1], epi=EPI[int(hdr[4])], BN=int(hdr[5]), blocks=nb, splits=int(hdr[7]), cus=len(per_cu),
                   span_us=round(span, 1), TFLOPs=round(flop / span / 1e6, 1) if int(hdr[7]) == 1 else None,
                   wg_life_us=round(float(np.median(life)), 2), prologue_us=round(float(np.median(pro)), 2), loop_us=round(float(np.median(loop)), 2),
                   epilogue_us=round(float(np.median(epi)), 2), gap_us=round(float(np.median(gaps)), 2) if gaps else None,
                   gap_p90_us=round(float(np.percentile(gaps, 90)), 2) if gaps else None,
                   wgs_per_cu=round(nb / max(1, len(per_cu)), 2), kloop_clock_GHz=round(ghz, 3) if ghz else None)
        out.append(rec)
        print(json.dumps(rec))
    e.close()


if __name__ == "__main__":
    main()
