#!/bin/bash
# A/B builds of libmjx.so on the layer-wise shards, same box: tools/ab_lw.sh <rounds> <lib> [<lib> ...]
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    for C in cfg4 cfg5; do
      MJX_LIB=$L python tools/lw_profile.py --cfg $C 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', '$C', round(d['fvp_ms'],3), 'ms', round(100*d['frac_fp32_mfma_peak'],1), '%')"
    done
  done
done
