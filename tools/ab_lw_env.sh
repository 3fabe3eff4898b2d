#!/bin/bash
# A/B an environment switch of libmjx on the layer-wise shards, same box, alternating: tools/ab_lw_env.sh <VAR> [rounds]
V=$1; R=${2:-3}
for i in $(seq $R); do
  for X in 0 1; do
    for C in cfg4 cfg5; do
      env $V=$X python tools/lw_profile.py --cfg $C 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$V=$X', '$C', round(d['fvp_ms'],3), 'ms', round(100*d['frac_fp32_mfma_peak'],1), '%')"
    done
  done
done
