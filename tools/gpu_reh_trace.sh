#!/bin/bash
# kernel timeline of one update of `bench.py --rehearse-world 8` (rank 0's share of an 8-rank job on one GPU)
# usage: tools/gpu_reh_trace.sh [rccl|peer]
T=${1:-rccl}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/reh8_trace_$T; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o reh -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --rehearse-world 8 --rehearse-transport $T > $OUT/line.json 2> $OUT/err.txt
OUT=$OUT python - <<'PY'
import csv, os
rows=list(csv.DictReader(open(os.environ["OUT"]+"/reh_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last update: from the last k_fused MODE 0 (VPG) to the end
idx=[i for i,r in enumerate(rows) if "k_fused" in r["Kernel_Name"] and ", 0, false" in r["Kernel_Name"]]
s=idx[-1]
t0=int(rows[s]["Start_Timestamp"])
prev_end=t0
for r in rows[s:s+60]:
    st,en=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%-58s start %8.1f  dur %7.1f  gap %6.1f"%(r["Kernel_Name"].replace("mjx::","")[:58],(st-t0)/1e3,(en-st)/1e3,(st-prev_end)/1e3))
    prev_end=en
PY
