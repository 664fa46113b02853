run() { echo "== $1"; L2S_LIB_PATH=$2 python tools/resblock_bench.py 640 2>&1 | grep -E "^stage"; }
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "resstage or resblock" 2>&1 | tail -3
run base build_ab/rb_base/liblip2speech_hip.so
run T_trim build_ab/rsT/liblip2speech_hip.so
run RSUM4 lip2speech_unit_amd/liblip2speech_hip.so
run T_trim build_ab/rsT/liblip2speech_hip.so
run RSUM4 lip2speech_unit_amd/liblip2speech_hip.so
