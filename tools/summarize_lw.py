#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/profile_lw.sh into profiles/<subdir>/ (tracked):
kernel_stats.csv (per-kernel time), fvp_chain.json (per-FVP time split by kernel from the kernel trace, HBM bytes per
FVP from FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, MFMA busy fraction) and the HIP-event line of the plain run.

usage: python tools/summarize_lw.py <tag> <cfg> <profiles-subdir>
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kname(s):
    s = s.replace("mjx::", "")
    return s.split("(")[0].strip()


def main():
    tag, cfg, sub = sys.argv[1:4]
    src = os.path.join(ROOT, "gpurun_out", "proflw_%s_%s" % (tag, cfg))
    dst = os.path.join(ROOT, "profiles", sub)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "lw_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    plain = json.loads(open(os.path.join(src, "plain.json")).read().strip().splitlines()[-1])
    fvps = plain["launches"] + 1                          # + the warm-up product
    # per-kernel time of one product: the kernels between consecutive k_fvp_logstd launches are one product's worth
    # (the backward chain of product k followed by the tangent chain of product k + 1; all products are identical)
    rows = list(csv.DictReader(open(os.path.join(src, "trace", "lw_kernel_trace.csv"))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "k_fvp_logstd" in r["Kernel_Name"]]
    per = collections.defaultdict(lambda: [0.0, 0])
    tot, wall, nseg = 0.0, 0.0, 0
    for k in range(1, len(marks) - 1):                  # skip the warm-up product
        for j in range(marks[k], marks[k + 1]):
            dur = (int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"])) / 1e3
            key = kname(rows[j]["Kernel_Name"])
            per[key][0] += dur; per[key][1] += 1
            tot += dur
        wall += (int(rows[marks[k + 1]]["Start_Timestamp"]) - int(rows[marks[k]]["Start_Timestamp"])) / 1e3
        nseg += 1
    chain = {k: {"us_per_fvp": v[0] / nseg, "launches_per_fvp": v[1] / nseg} for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])} if nseg else {}
    kernel_us = tot / nseg if nseg else None
    wall_us = wall / nseg if nseg else None
    pmc = collections.defaultdict(float)
    for p in ("pmc_fetch", "pmc_write", "pmc_sq"):
        f = os.path.join(src, p, "lw_counter_collection.csv")
        if os.path.isfile(f):
            for r in csv.DictReader(open(f)):
                pmc[r["Counter_Name"]] += float(r["Counter_Value"])
    out = {"cfg": cfg, "rows": plain["rows"], "hip_event": plain, "kernel_time_us_per_fvp": kernel_us, "wall_us_per_fvp_in_trace": wall_us, "chain": chain,
           "whole_run_counters": dict(pmc),
           "note": "counters are sums over the whole run (K1 + %d FVP + K3); FETCH_SIZE / WRITE_SIZE in KB (FETCH_SIZE x 2 on gfx950)" % fvps}
    if "FETCH_SIZE" in pmc:
        out["hbm_GB_whole_run"] = (2 * pmc["FETCH_SIZE"] + pmc.get("WRITE_SIZE", 0.0)) * 1024 / 1e9
    if pmc.get("SQ_BUSY_CYCLES"):
        out["mfma_busy_over_wave_cycles"] = pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / max(pmc["SQ_WAVE_CYCLES"], 1.0)
    json.dump(out, open(os.path.join(dst, "fvp_chain.json"), "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
