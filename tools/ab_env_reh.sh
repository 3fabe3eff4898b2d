#!/bin/bash
# A/B an environment switch of libmjx on rank 0's 1/8 share of the headline update (bench.py --rehearse-world 8, peer loop-back), same
# box, alternating: tools/ab_env_reh.sh <VAR> [rounds]          (r06: MJX_RAW_SLAB -- accumulator-order workgroup partials)
V=$1; R=${2:-3}
for i in $(seq $R); do
  for X in 0 1; do
    env $V=$X python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --rehearse-world 8 --rehearse-transport peer 2>&1 | grep metric | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['rehearsal_detail']; print('$V=$X share', round(d['ms_per_step'],4), 'ms  fvp', round(r['fvp_us'],2), 'us  iteration', round(r['cg_iteration_us'],2), 'us  outside', round(r['outside_the_cg_loop_us'],1))"
  done
done
