set -e
ROOT=$(pwd)
python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
tail -c 400 gpurun_out/r03_bench.json; echo
python bench.py --stage frontend > gpurun_out/r03_bench_frontend.json 2> gpurun_out/r03_bench_frontend.err
python bench.py --mixed > gpurun_out/r03_bench_mixed.json 2> gpurun_out/r03_bench_mixed.err
echo benches done
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r03 -o bench -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-transfers > $ROOT/gpurun_out/r03_bench_under_rocprof.json 2>&1)
echo rocprof done
bash tools/collect_traffic.sh gpurun_out/r03_traffic.json 640
echo traffic done
