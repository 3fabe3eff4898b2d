#!/bin/bash
# rocprofv3 passes over the layer-wise path at the per-GPU shard sizes of BASELINE configs[3] / [4]
# (run on the GPU box from the repo root).   usage: tools/profile_lw.sh <tag> [cfg4|cfg5 ...]
set -u
TAG=${1:-r02}; shift
CFGS=${@:-cfg4 cfg5}
cd /tmp && export TMPDIR=/tmp
for C in $CFGS; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/proflw_${TAG}_$C
  mkdir -p $OUT
  CMD="python $GRAFT_REPO_ROOT/tools/lw_profile.py --cfg $C"
  $CMD > $OUT/plain.json 2> $OUT/plain.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o lw -- $CMD > $OUT/trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o lw -- $CMD > $OUT/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o lw -- $CMD > $OUT/pmc_write.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -o lw -- $CMD > $OUT/pmc_sq.log 2>&1
  cat $OUT/plain.json
done
