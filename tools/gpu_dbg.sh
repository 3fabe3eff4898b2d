#!/bin/bash
cd $GRAFT_REPO_ROOT
python -X faulthandler -m pytest tests/test_gpu_operators.py -m gpu -q --no-header -p no:cacheprovider -x -k "rccl or one_call" 2>&1 | tail -40 > gpurun_out/dbg_tests.log
python -X faulthandler bench.py --steps 10 --warmup 2 --no-cpu-baseline --rehearse-world 8 > gpurun_out/dbg_reh8.json 2> gpurun_out/dbg_reh8.err; echo "rc=$?" >> gpurun_out/dbg_reh8.err
tail -30 gpurun_out/dbg_tests.log; tail -30 gpurun_out/dbg_reh8.err; cat gpurun_out/dbg_reh8.json | head -c 400
