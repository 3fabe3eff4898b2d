#!/usr/bin/env python3
"""Condense tools/profile_rows.sh output into profiles/<subdir>/ (tracked): kernel_stats.csv of bench_rows.py (K5 scans,
K6 Gram / prediction / MLP fit, K1 / K2 / K3) and of bench_ppo.py (BC / PPO trainer), plus rows.json with the HBM rate of
the scan / Gram kernels (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE over the kernel's duration).
usage: python tools/summarize_rows.py <tag> <profiles-subdir>"""
import collections, csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, sub = sys.argv[1:3]
    src = os.path.join(ROOT, "gpurun_out", "profrows_" + tag)
    dst = os.path.join(ROOT, "profiles", sub)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "rows", "rows_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    shutil.copy(os.path.join(src, "ppo", "ppo_kernel_stats.csv"), os.path.join(dst, "ppo_kernel_stats.csv"))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(src, "rows", "rows_kernel_trace.csv"))):
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in ("rows_fetch", "rows_write"):
        f = os.path.join(src, p, "rows_counter_collection.csv")
        if os.path.isfile(f):
            for r in csv.DictReader(open(f)):
                pmc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    txt = open(os.path.join(src, "rows.json")).read()
    out = {"bench_rows": json.loads(txt[txt.index("{\n"):]),
           "bench_ppo": json.loads(open(os.path.join(src, "ppo.json")).read().strip().splitlines()[-1]), "kernels": {}}
    for k, ds in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if not any(t in k for t in ("k_traj_scan", "k_bl_", "k_mlp_fit", "k_sum_stats", "k_whiten", "k_cast", "k_fused", "k_cg", "k_reduce")):
            continue
        e = {"launches": len(ds), "avg_us": sum(ds) / len(ds), "min_us": min(ds)}
        c = pmc.get(k)
        if c and c.get("FETCH_SIZE") and c.get("WRITE_SIZE"):
            hbm = (2 * sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) + sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])) * 1024
            e["hbm_bytes_per_launch"] = hbm
            e["hbm_GBps"] = hbm / (e["avg_us"] * 1e-6) / 1e9
        out["kernels"][k.replace("mjx::", "")] = e
    json.dump(out, open(os.path.join(dst, "rows.json"), "w"), indent=1)
    for k, e in out["kernels"].items():
        print("%-70s x%-5d avg %9.1f us  %s" % (k[:70], e["launches"], e["avg_us"], ("%.0f GB/s" % e["hbm_GBps"]) if "hbm_GBps" in e else ""))


if __name__ == "__main__":
    main()
