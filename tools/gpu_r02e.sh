#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/r02e_tests.log
python bench.py > gpurun_out/r02e_bench.json 2> gpurun_out/r02e_bench.err; echo "rc=$?" >> gpurun_out/r02e_bench.err
python tools/bench_iter.py > gpurun_out/r02e_bench_iter.json 2> gpurun_out/r02e_bench_iter.err
grep -E "passed|failed|Error" gpurun_out/r02e_tests.log | tail -5; tail -3 gpurun_out/r02e_bench.err; python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r02e_bench.json").read().splitlines() if l.startswith("{")][-1])
print({k: j[k] for k in ("value","ms_per_step","timing","check","check_vs_fp64_oracle")})
print(j["roofline"]["frac"], j["roofline"]["avg_launch_ms"], j["roofline"]["traffic_source"])
print(json.dumps(j["secondary"], indent=1))
print(j["cpu_baseline"])
PY
cat gpurun_out/r02e_bench_iter.json
