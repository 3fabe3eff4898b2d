#!/bin/bash
# rocprofv3 kernel-trace + HBM-counter passes over the other hot-path rows (K5 scans, K6 Gram / MLP fit, BC / PPO trainer)
# usage: tools/profile_rows.sh <tag>      (GPU box, repo root)
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profrows_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rows -o rows -- python $GRAFT_REPO_ROOT/tools/bench_rows.py > $OUT/rows.json 2> $OUT/rows.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/rows_fetch -o rows -- python $GRAFT_REPO_ROOT/tools/bench_rows.py > /dev/null 2> $OUT/rows_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/rows_write -o rows -- python $GRAFT_REPO_ROOT/tools/bench_rows.py > /dev/null 2> $OUT/rows_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ppo -o ppo -- python $GRAFT_REPO_ROOT/tools/bench_ppo.py > $OUT/ppo.json 2> $OUT/ppo.err
tail -n 1 $OUT/rows.json; tail -n 1 $OUT/ppo.json
