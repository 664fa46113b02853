run() { echo "== $1"; L2S_LIB_PATH=$2 python tools/glu_bench.py 640 2>&1 | grep glu_dwconv; }
run product lip2speech_unit_amd/liblip2speech_hip.so
run nosig build_ab/gA/liblip2speech_hip.so
run noconv build_ab/gB/liblip2speech_hip.so
run noswish build_ab/gC/liblip2speech_hip.so
run none build_ab/gD/liblip2speech_hip.so
run product lip2speech_unit_amd/liblip2speech_hip.so
