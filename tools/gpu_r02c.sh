#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r02c_tests.log
python tools/iter_profile.py quadratic > gpurun_out/r02c_iter_q.txt 2>&1
python tools/bench_iter.py > gpurun_out/r02c_bench_iter.json 2> gpurun_out/r02c_bench_iter.err
tail -25 gpurun_out/r02c_tests.log; tail -3 gpurun_out/r02c_iter_q.txt; cat gpurun_out/r02c_bench_iter.json; tail -5 gpurun_out/r02c_bench_iter.err
