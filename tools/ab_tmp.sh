for i in 1 2; do for T in 0 1; do for C in cfg4 cfg5; do
MJX_LW_SPLITS=$T python tools/lw_profile.py --cfg $C 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('splits=$T', '$C', round(d['fvp_ms'],3), 'ms', round(100*d['frac_fp32_mfma_peak'],1), '%')"
done; done; done
