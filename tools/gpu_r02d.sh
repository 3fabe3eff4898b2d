#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r02d_tests.log
python tools/iter_trace.py > gpurun_out/r02d_iter_trace.txt 2>&1
python tools/bench_iter.py > gpurun_out/r02d_bench_iter.json 2> gpurun_out/r02d_bench_iter.err
python tools/bench_e2e.py > gpurun_out/r02d_bench_e2e.json 2> gpurun_out/r02d_bench_e2e.err
grep -E "passed|failed|Error" gpurun_out/r02d_tests.log | tail -5; tail -n 3 gpurun_out/r02d_iter_trace.txt; cat gpurun_out/r02d_bench_iter.json; cat gpurun_out/r02d_bench_e2e.json
