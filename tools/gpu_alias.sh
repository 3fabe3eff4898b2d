#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/alias; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MJX_LIB=$GRAFT_REPO_ROOT/tools/_dbg/libmjx_exp.so
for A in 0 1; do
  MJX_LW_DEBUG_ALIAS_A=$A rocprofv3 --kernel-trace --output-format csv -d $OUT/a$A -o lw -- python $GRAFT_REPO_ROOT/tools/lw_profile.py --cfg cfg4 > $OUT/a$A.json 2> $OUT/a$A.err
done
python - <<'PY'
import csv, os
for A in (0,1):
    rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/alias/a%d/lw_kernel_trace.csv"%A)))
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    marks=[i for i,r in enumerate(rows) if "k_fvp_logstd" in r["Kernel_Name"]]
    print("alias_a =",A)
    for r in rows[marks[2]:marks[3]]:
        if "k_gemm" in r["Kernel_Name"]:
            print("   %-34s grid %s/%s/%s  %8.1f us"%(r["Kernel_Name"].replace("mjx::","")[:34], r["Grid_Size_X"],r["Grid_Size_Y"],r["Grid_Size_Z"],(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
