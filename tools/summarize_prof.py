#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/profile_bench.sh into the files kept under profiles/.

usage: python tools/summarize_prof.py <tag> <profiles-subdir>
  reads  gpurun_out/prof_<tag>/{trace,pmc_*}/bench_*.csv
  writes profiles/<subdir>/kernel_stats.csv, profiles/<subdir>/pmc_per_launch_avg.json and
         profiles/<round>_fvp_pmc.json (HBM bytes per FVP launch; bench.py reads the newest one for roofline.traffic)
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SAMPLES, N_OBS, H1, H2 = 1_000_000, 17, 64, 64


def short(name):
    if "k_fused" not in name:
        return None
    args = name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",")
    mode = {"0": "MODE_VPG", "1": "MODE_FVP", "2": "MODE_EVAL"}[args[4]]
    cached = len(args) > 7 and args[7] == "true"
    return "mjx::k_fused<%s,%s,%s,%s,%s%s>" % (args[0], args[1], args[2], args[3], mode, ",CACHED" if cached else "")


def main():
    tag, sub = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles", sub)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in sorted(os.listdir(src)):
        f = os.path.join(src, p, "bench_counter_collection.csv")
        if not os.path.isfile(f):
            continue
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} | {"launches": len(next(iter(cs.values())))}
           for k, cs in sorted(acc.items())}
    json.dump(out, open(os.path.join(dst, "pmc_per_launch_avg.json"), "w"), indent=1)
    fvp = next(k for k in out if "MODE_FVP" in k)
    cached = "CACHED" in fvp
    algo = 4 * (((N_OBS + 4) & ~3) + H1 + H2) * N_SAMPLES if cached else 4 * N_OBS * N_SAMPLES
    fetch, write = out[fvp]["FETCH_SIZE"], out[fvp]["WRITE_SIZE"]
    json.dump({
        "command": "python bench.py --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, "
                   "separate passes; tools/profile_bench.sh %s)" % tag,
        "kernel": fvp,
        "FETCH_SIZE_KB_raw": fetch,
        "WRITE_SIZE_KB_raw": write,
        "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section); "
                      "the kernel must read >= %.1f MB per launch" % (algo / 1e6),
        "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
        "algorithmic_bytes_per_launch": algo,
        "round": tag,
    }, open(os.path.join(ROOT, "profiles", tag[:3] + "_fvp_pmc.json"), "w"), indent=1)
    print(json.dumps(out[fvp], indent=1))
    print(open(os.path.join(ROOT, "profiles", tag[:3] + "_fvp_pmc.json")).read())


if __name__ == "__main__":
    main()
