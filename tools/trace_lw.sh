#!/bin/bash
# kernel trace of the layer-wise chain only (quick look at per-kernel time): tools/trace_lw.sh <tag> [cfg4|cfg5 ...]
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
for r in rows[:12]:
    print('$C', r['Name'][:60], r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1), 'pct', r['Percentage'])
PY
done
