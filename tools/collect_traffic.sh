#!/bin/bash
# HBM traffic of every kernel of one bench run, from rocprofv3 PMC counters collected in SEPARATE passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage (on the GPU box, from the repo root): bash tools/collect_traffic.sh <out.json> <clips per launch>
# (one stream: every launch then has the shape of the bench default's sub-batches, --batch 640 --streams 2 = 320 per launch)
# (.build_commit, written before the gpurun call with `git rev-parse --short HEAD > .build_commit`, tags the result)
set -e
OUT=$1; BATCH=$2
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_traffic_$c -o p -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-transfers --no-graph --batch $BATCH --streams 1 > $ROOT/gpurun_out/pmc_traffic_$c.log 2>&1
done
cd $ROOT
python3 tools/traffic_summary.py gpurun_out/pmc_traffic_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_traffic_WRITE_SIZE/p_counter_collection.csv $BATCH $(cat .build_commit 2>/dev/null || echo unknown) > $OUT
echo wrote $OUT
