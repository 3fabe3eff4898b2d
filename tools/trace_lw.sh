#!/bin/bash
# kernel trace of the layer-wise chain only (quick look at per-kernel time): tools/trace_lw.sh <tag> [cfg4|cfg5 ...]
# prints the kernel statistics and the per-launch durations of the last Fisher-vector product
TAG=${1:-x}; shift
CFGS=${@:-cfg4 cfg5}
cd /tmp && export TMPDIR=/tmp
for C in $CFGS; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/tracelw_${TAG}_$C
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o lw -- python $GRAFT_REPO_ROOT/tools/lw_profile.py --cfg $C > $OUT/trace.log 2>&1
  python - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/trace/lw_kernel_stats.csv')))
for r in rows[:8]:
    print('$C', r['Name'][:60], r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1), 'pct', r['Percentage'])
rows=list(csv.DictReader(open('$OUT/trace/lw_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_fvp_logstd' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
tot=0
for r in rows[a:b]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3; tot+=d
    if d>30: print('   %-42s %8.1f us  grid %sx%sx%s'%(r['Kernel_Name'][:42], d, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z']))
print('   kernel time per product', round(tot,1), 'us; wall', round((int(rows[b]['Start_Timestamp'])-int(rows[a]['Start_Timestamp']))/1e3,1))
PY
done
