#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r02f > gpurun_out/r02f_profile_bench.log 2>&1
bash tools/profile_lw.sh r02f cfg4 cfg5 > gpurun_out/r02f_profile_lw.log 2>&1
bash tools/profile_rows.sh r02f > gpurun_out/r02f_profile_rows.log 2>&1
tail -3 gpurun_out/r02f_profile_bench.log; tail -2 gpurun_out/r02f_profile_lw.log; tail -2 gpurun_out/r02f_profile_rows.log
