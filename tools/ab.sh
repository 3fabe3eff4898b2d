#!/bin/bash
# A/B two builds of libmjx.so on the same box: tools/ab.sh <libA> <libB> [rounds]
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do
  for L in $A $B; do
    MJX_LIB=$L python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | grep metric | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', round(d['value'],2), 'upd/s', round(d['ms_per_step'],3), 'ms  fvp', round(d['roofline']['avg_launch_ms'],4))"
  done
done
