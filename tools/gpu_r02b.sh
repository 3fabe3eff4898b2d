#!/bin/bash
# round-2 GPU call B: -m gpu suite (C-level update, hooked two-rank path, RCCL binding), bench line, 8-rank rehearsal,
# layer-wise FVP time after the bias-gradient reduction fix
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/r02b_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err
for R in 2 4 8; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --rehearse-world $R > gpurun_out/r02b_rehearse$R.json 2> gpurun_out/r02b_rehearse$R.err; done
python tools/lw_profile.py --cfg cfg4 > gpurun_out/r02b_lw_cfg4.json 2>&1
python tools/lw_profile.py --cfg cfg5 > gpurun_out/r02b_lw_cfg5.json 2>&1
tail -5 gpurun_out/r02b_tests.log
for f in gpurun_out/r02b_bench.json gpurun_out/r02b_rehearse*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], round(j["value"],1), "updates/s", round(j["ms_per_step"],3), "ms", "fvp", round(j["roofline"]["avg_launch_ms"],4), j["check"])
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
tail -n 1 gpurun_out/r02b_lw_cfg4.json; tail -n 1 gpurun_out/r02b_lw_cfg5.json
