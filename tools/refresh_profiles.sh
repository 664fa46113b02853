#!/bin/bash
# The round's evidence in one gpurun call (from the repo root on the GPU box; copy the results into profiles/ afterwards):
#   bash tools/refresh_profiles.sh [notraffic]
# bench lines (default / --stage frontend / --mixed), rocprofv3 --kernel-trace --stats of the default workload and, unless
# `notraffic`, the two PMC passes of tools/collect_traffic.sh (write `git rev-parse --short HEAD > .build_commit` before the call).
set -e
ROOT=$(pwd)
python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
tail -c 400 gpurun_out/r03_bench.json; echo
python bench.py --stage frontend > gpurun_out/r03_bench_frontend.json 2> gpurun_out/r03_bench_frontend.err
python bench.py --mixed > gpurun_out/r03_bench_mixed.json 2> gpurun_out/r03_bench_mixed.err
echo benches done
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r03 -o bench -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-transfers > $ROOT/gpurun_out/r03_bench_under_rocprof.json 2>&1)
echo rocprof done
if [ "$1" != "notraffic" ]; then
  bash tools/collect_traffic.sh gpurun_out/r03_traffic.json 640
  echo traffic done
fi
