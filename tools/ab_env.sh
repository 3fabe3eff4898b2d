#!/bin/bash
# A/B an environment switch of libmjx on the headline update, same box, alternating: tools/ab_env.sh <VAR> [rounds]
V=$1; R=${2:-3}
for i in $(seq $R); do
  for X in 0 1; do
    env $V=$X python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>&1 | grep metric | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$V=$X', round(d['value'],2), 'upd/s', round(d['ms_per_step'],4), 'ms  fvp', round(d['roofline']['avg_launch_ms'],4))"
  done
done
