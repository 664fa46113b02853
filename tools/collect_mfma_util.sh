#!/bin/bash
# Matrix-pipe busy cycles of every kernel of one bench step (ONE --pmc pass, SQ + GRBM counters, --kernel-trace only):
#   bash tools/collect_mfma_util.sh <out.json> <clips per launch>      (on the GPU box, from the repo root)
set -e
OUT=$1; BATCH=$2
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_mfma -o p -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-transfers --no-graph --batch $BATCH --streams 1 > $ROOT/gpurun_out/pmc_mfma.log 2>&1
cd $ROOT/tools
python3 mfma_util_summary.py $ROOT/gpurun_out/pmc_mfma/p_counter_collection.csv $BATCH $(cat $ROOT/.build_commit 2>/dev/null || echo unknown) > $ROOT/$OUT
echo wrote $OUT
