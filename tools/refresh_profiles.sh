#!/bin/bash
# The round's evidence in one gpurun call (from the repo root on the GPU box; copy the results into profiles/ afterwards):
#   bash tools/refresh_profiles.sh <round tag, e.g. r04> [notraffic]
# bench lines (default / --stage frontend / --mixed / --batch 1 --latency / --dtype bf16), rocprofv3 --kernel-trace --stats of the
# default workload and of the batch-1 latency run and, unless `notraffic`, the two PMC passes of tools/collect_traffic.sh and the
# matrix-pipe busy pass of tools/collect_mfma_util.sh
# (write `git rev-parse --short HEAD > .build_commit` before the call).
set -e
TAG=${1:-r04}
ROOT=$(pwd)
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 400 gpurun_out/${TAG}_bench.json; echo
python bench.py --stage frontend > gpurun_out/${TAG}_bench_frontend.json 2> gpurun_out/${TAG}_bench_frontend.err
python bench.py --mixed > gpurun_out/${TAG}_bench_mixed.json 2> gpurun_out/${TAG}_bench_mixed.err
python bench.py --batch 1 --latency > gpurun_out/${TAG}_bench_b1.json 2> gpurun_out/${TAG}_bench_b1.err
python bench.py --dtype bf16 --no-transfers --cpu-clips 4 --cpu-warm 1 > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/${TAG}_bench_bf16.err
echo benches done
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG} -o bench -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-transfers > $ROOT/gpurun_out/${TAG}_bench_under_rocprof.json 2>&1)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_b1 -o bench -- python3 $ROOT/bench.py --batch 1 --latency --requests 50 > $ROOT/gpurun_out/${TAG}_bench_b1_under_rocprof.json 2>&1)
echo rocprof done
if [ "$2" != "notraffic" ]; then
  bash tools/collect_traffic.sh gpurun_out/${TAG}_traffic.json 640
  echo traffic done
  bash tools/collect_mfma_util.sh gpurun_out/${TAG}_mfma_util.json 640
  echo mfma util done
fi
