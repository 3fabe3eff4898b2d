#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/profile_lw.sh ${1:-r02k} cfg4 cfg5 > gpurun_out/${1:-r02k}_profile_lw.log 2>&1
bash tools/profile_bench.sh ${1:-r02k} > gpurun_out/${1:-r02k}_profile_bench.log 2>&1
python bench.py > gpurun_out/${1:-r02k}_bench.json 2> gpurun_out/${1:-r02k}_bench.err
tail -2 gpurun_out/${1:-r02k}_profile_lw.log; python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/${1:-r02k}_bench.json").read().splitlines() if l.startswith("{")][-1])
print(j["value"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["secondary"]["roofline_lw"]["configs3_humanoid_256x256"]["frac_of_fp32_mfma_peak"], j["secondary"]["roofline_lw"]["configs4_adroit_512x512"]["frac_of_fp32_mfma_peak"], j["secondary"]["trpo_configs2"]["ms_per_update"], j["cpu_baseline"])
PY
