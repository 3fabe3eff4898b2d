#!/bin/bash
# the round-end measurement in one gpurun call: tools/gpu_round_end.sh <tag>
export TAG=${1:-r02k}
cd $GRAFT_REPO_ROOT
bash tools/profile_lw.sh $TAG cfg4 cfg5 > gpurun_out/${TAG}_profile_lw.log 2>&1
bash tools/profile_bench.sh $TAG > gpurun_out/${TAG}_profile_bench.log 2>&1
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -2 gpurun_out/${TAG}_profile_lw.log
python - <<'PY'
import json, os
tag = os.environ["TAG"]
j = json.loads([l for l in open("gpurun_out/%s_bench.json" % tag).read().splitlines() if l.startswith("{")][-1])
lw = j["secondary"]["roofline_lw"]
print(j["value"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], [round(v["frac_of_fp32_mfma_peak"], 4) for v in lw.values()],
      j["secondary"]["trpo_configs2"]["ms_per_update"], j["cpu_baseline"]["value"])
PY
