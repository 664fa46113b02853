run() { echo "== $1"; L2S_LIB_PATH=$2 python tools/resblock_bench.py 640 2>&1 | grep -E "^stage C16"; }
run product lip2speech_unit_amd/liblip2speech_hip.so
run noload build_ab/rsNL/liblip2speech_hip.so
run nostore build_ab/rsNS/liblip2speech_hip.so
run nosum build_ab/rsNN/liblip2speech_hip.so
run ldsbar build_ab/rsLB/liblip2speech_hip.so
run product lip2speech_unit_amd/liblip2speech_hip.so
L2S_LIB_PATH=build_ab/rsLB/liblip2speech_hip.so python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "resstage or resblock" 2>&1 | tail -2
