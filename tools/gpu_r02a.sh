#!/bin/bash
# round-2 GPU call A: the whole -m gpu suite with the wide layer-wise tiles (default) and with the round-1 tiles,
# rocprofv3 passes over the layer-wise path in both modes, one bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -s 2>&1 | tail -60 > gpurun_out/r02a_tests_wide.log
MJX_LW_TILES=0 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -s 2>&1 | tail -60 > gpurun_out/r02a_tests_old.log
MJX_LW_TILES=0 tools/profile_lw.sh r02a_base cfg4 cfg5 > gpurun_out/r02a_prof_base.log 2>&1
tools/profile_lw.sh r02a_wide cfg4 cfg5 > gpurun_out/r02a_prof_wide.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err
tail -5 gpurun_out/r02a_tests_wide.log; tail -3 gpurun_out/r02a_tests_old.log; cat gpurun_out/r02a_prof_base.log gpurun_out/r02a_prof_wide.log | grep fvp_ms; head -c 600 gpurun_out/r02a_bench.json
